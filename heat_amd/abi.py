"""ctypes view of the C ABI in include/heat_cf.h (lib/libheat_cf.so).

Used by tests/, bench.py and the multi-GPU driver; the reference-compatible surface is heat_amd.cf_c.
Loading fails loudly when the library has not been built — there is no fallback implementation.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HEAT_CF_LIB") or os.path.join(HERE, "lib", "libheat_cf.so")   # HEAT_CF_LIB: development builds

OK, EINVAL, EHIP, ENOMEM, EUNSUP = 0, -1, -2, -3, -4
FLAG_SERIAL, FLAG_LAZY_SYNC, FLAG_SAMPLING_CALL, FLAG_NULL_STREAM, FLAG_TILE_GLOBAL, FLAG_TILE_LDS = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
COHERENCE_DEFAULT, COHERENCE_PLAIN, COHERENCE_DEVICE = 0, 1, 2
UPDATE_DEFAULT, UPDATE_OVERWRITE, UPDATE_ATOMIC_W, UPDATE_ATOMIC_WG, UPDATE_ATOMIC_POS, UPDATE_AUTO, UPDATE_REREAD_POS = 0, 1, 2, 3, 4, 5, 6


class Config(C.Structure):
    """struct heat_cf_config (include/heat_cf.h) — replaces cf::modules::CFConfig (cf_config.hpp:12-35)."""
    _fields_ = [("emb_dim", C.c_uint64), ("num_negs", C.c_uint64), ("num_users", C.c_uint64),
                ("num_items", C.c_uint64), ("train_size", C.c_uint64), ("neg_sampler", C.c_uint64),
                ("tile_size", C.c_uint64), ("refresh_interval", C.c_uint64), ("num_subepochs", C.c_uint64),
                ("l2", C.c_float), ("clip_val", C.c_float), ("milestones", C.POINTER(C.c_uint64)),
                ("n_milestones", C.c_uint64), ("l_r", C.c_float),
                ("seed", C.c_uint64), ("sample_index_base", C.c_uint64), ("use_aggregator", C.c_uint32),
                ("flags", C.c_uint32), ("coherence", C.c_uint32), ("device", C.c_int32),
                ("num_streams", C.c_uint32), ("update_mode", C.c_uint32)]


class LightGCN(C.Structure):
    _fields_ = [("num_lines", C.c_uint64), ("n_interactions", C.c_uint64), ("max_user_id", C.c_uint64),
                ("max_item_id", C.c_uint64), ("clicks", C.POINTER(C.c_uint64)), ("line_user", C.POINTER(C.c_uint64)),
                ("line_start", C.POINTER(C.c_uint64))]


class DeviceView(C.Structure):
    _fields_ = [("user_w", C.c_void_p), ("item_w", C.c_void_p), ("user_g", C.c_void_p), ("item_g", C.c_void_p),
                ("w0", C.c_void_p), ("clicks", C.c_void_p), ("data_rows", C.c_uint64), ("stream", C.c_void_p)]


# every symbol include/heat_cf.h declares (tests/test_frontend_cpu.py::test_abi_exports_every_declared_symbol checks the header against this list)
SYMBOLS = {
    "heat_cf_abi_version": (C.c_int, []),
    "heat_cf_last_error": (C.c_char_p, []),
    "heat_cf_device_count": (C.c_int, []),
    "heat_cf_plan": (C.c_int, [C.POINTER(Config), C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]),
    "heat_cf_engine_create": (C.c_int, [C.POINTER(Config), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "heat_cf_engine_create_device": (C.c_int, [C.POINTER(Config), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_void_p)]),
    "heat_cf_engine_destroy": (None, [C.c_void_p]),
    "heat_cf_train_one_epoch": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "heat_cf_begin_epoch": (C.c_int, [C.c_void_p]),
    "heat_cf_train_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_double)]),
    "heat_cf_end_epoch": (C.c_int, [C.c_void_p]),
    "heat_cf_sample_negatives": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "heat_cf_evaluate0": (C.c_int, [C.c_void_p, C.c_void_p]),
    "heat_cf_topk": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "heat_cf_sync_to_host": (C.c_int, [C.c_void_p]),
    "heat_cf_sync_from_host": (C.c_int, [C.c_void_p]),
    "heat_cf_synchronize": (C.c_int, [C.c_void_p]),
    "heat_cf_sync_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "heat_cf_sync_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    "heat_cf_sync_apply_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    "heat_cf_sync_apply_snap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "heat_cf_sync_delta_from": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "heat_cf_sync_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "heat_cf_get_device_view": (C.c_int, [C.c_void_p, C.POINTER(DeviceView)]),
    "heat_cf_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "heat_cf_epoch": (C.c_uint64, [C.c_void_p]),
    "heat_cf_learning_rate": (C.c_float, [C.c_void_p]),
    "heat_cf_set_learning_rate": (C.c_int, [C.c_void_p, C.c_float]),
    "heat_cf_set_epoch": (C.c_int, [C.c_void_p, C.c_uint64]),
    "heat_cf_zero_grad": (C.c_int, [C.c_void_p]),
    "heat_cf_kernel_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "heat_cf_kernel_name": (C.c_char_p, [C.c_void_p]),
    "heat_cf_parse_lightgcn": (C.c_int, [C.c_char_p, C.c_char, C.POINTER(LightGCN)]),
    "heat_cf_free_lightgcn": (None, [C.POINTER(LightGCN)]),
}

_lib = None


def load():
    """dlopen lib/libheat_cf.so and type every entry point.  Raises if the HIP library is missing."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and libheat_cf.so depends on the same SONAME, so
        # whichever is loaded first serves both.  Loading torch's first keeps torch usable next to the engine (device
        # tensors, streams, torch.distributed/RCCL); the other order leaves torch without visible GPUs.
        try:
            import torch  # noqa: F401
            torch.cuda.is_available()
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m heat_amd.build` "
                              "(hipcc --offload-arch=gfx950); heat_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.heat_cf_abi_version() != 1:
            raise ImportError("libheat_cf.so ABI version mismatch")
        _lib = lib
    return _lib


class HeatError(RuntimeError):
    pass


def _check(rc):
    if rc != OK:
        msg = load().heat_cf_last_error().decode()
        if rc == EINVAL:
            raise ValueError(msg)
        if rc == ENOMEM:
            raise MemoryError(msg)
        raise HeatError(f"[{rc}] {msg}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_config(*, emb_dim, num_negs, num_users, num_items, train_size, neg_sampler=0, tile_size=512,
                refresh_interval=8192, num_subepochs=2, l2=1e-7, clip_val=1.0, milestones=(10,), l_r=0.01, seed=2022,
                sample_index_base=0, use_aggregator=False, flags=0, coherence=COHERENCE_DEFAULT, device=-1,
                num_streams=0, update_mode=UPDATE_DEFAULT):
    ms = np.ascontiguousarray(np.asarray(list(milestones), dtype=np.uint64))
    cfg = Config(emb_dim, num_negs, num_users, num_items, train_size, neg_sampler, tile_size, refresh_interval,
                 num_subepochs, l2, clip_val, ms.ctypes.data_as(C.POINTER(C.c_uint64)), len(ms), l_r, seed,
                 sample_index_base, int(use_aggregator), flags, coherence, device, num_streams, update_mode)
    cfg._keepalive = ms
    return cfg


def _require(a, dtype, ndim, name):
    if not isinstance(a, np.ndarray) or a.dtype != dtype or a.ndim != ndim or not a.flags.c_contiguous:
        raise ValueError(f"{name} must be a C-contiguous {ndim}-D numpy array of {np.dtype(dtype).name}")


def plan(resident_workgroups=0, data_rows=None, **cfg_kwargs):
    """The launch plan (kernel variant, streams, update policy) for a configuration, as a dict.  No GPU needed."""
    import json
    cfg = make_config(**cfg_kwargs)
    buf = C.create_string_buffer(1024)
    _check(load().heat_cf_plan(C.byref(cfg), cfg.train_size if data_rows is None else data_rows, resident_workgroups, buf, 1024))
    return json.loads(buf.value.decode())


def parse_lightgcn(path, separator=" "):
    """(clicks [n,2] u64 in file order, line_user [lines] u64, line_start [lines+1] u64) of a LightGCN text file, parsed
    natively (heat_cf_parse_lightgcn).  No GPU needed."""
    g = LightGCN()
    _check(load().heat_cf_parse_lightgcn(os.fsencode(path), separator.encode()[:1] or b" ", C.byref(g)))
    try:
        n, lines = int(g.n_interactions), int(g.num_lines)
        clicks = np.ctypeslib.as_array(g.clicks, shape=(max(n, 1) * 2,))[:n * 2].reshape(n, 2).copy()
        line_user = np.ctypeslib.as_array(g.line_user, shape=(max(lines, 1),))[:lines].copy()
        line_start = np.ctypeslib.as_array(g.line_start, shape=(lines + 1,)).copy()
    finally:
        load().heat_cf_free_lightgcn(C.byref(g))
    return clicks, line_user, line_start


class Engine:
    """Host-mode engine: numpy buffers are borrowed and trained in place (like the reference's pybind layer).
    Engine.from_device(...) builds the device-mode twin on caller-owned device memory (e.g. torch tensors)."""

    @classmethod
    def from_device(cls, clicks_ptr, data_rows, user_w_ptr, item_w_ptr, *, num_users, num_items, emb_dim, num_negs,
                    stream=None, keep=None, his_ptr=None, max_his=0, masks_ptr=None, w0_ptr=None, **cfg_kwargs):
        """Device pointers (ints): clicks [data_rows,2] u64, user_w [num_users,emb_dim] f32, item_w [num_items,emb_dim]
        f32, all owned by the caller; `stream` is a hipStream_t handle (int) or None; `keep` holds references alive.
        With use_aggregator=1: his [num_users,max_his] u64, masks [num_users] u64, w0 [emb_dim,emb_dim] f32 (trained
        in place) are device pointers too."""
        self = cls.__new__(cls)
        self._keep = keep
        self.num_negs = num_negs
        self.data_rows = data_rows
        self.cfg = make_config(emb_dim=emb_dim, num_negs=num_negs, num_users=num_users, num_items=num_items,
                               train_size=data_rows, **cfg_kwargs)
        self._h = C.c_void_p()
        _check(load().heat_cf_engine_create_device(C.byref(self.cfg), C.c_void_p(clicks_ptr), data_rows,
                                                   C.c_void_p(his_ptr) if his_ptr else None, max_his,
                                                   C.c_void_p(masks_ptr) if masks_ptr else None,
                                                   C.c_void_p(user_w_ptr), C.c_void_p(item_w_ptr),
                                                   C.c_void_p(w0_ptr) if w0_ptr else None,
                                                   C.c_void_p(stream) if stream else None, C.byref(self._h)))
        return self

    def __init__(self, clicks, user_w, item_w, *, num_negs, his=None, masks=None, w0=None, **cfg_kwargs):
        _require(clicks, np.uint64, 2, "clicks")
        _require(user_w, np.float32, 2, "user_w")
        _require(item_w, np.float32, 2, "item_w")
        if clicks.shape[1] != 2 or user_w.shape[1] != item_w.shape[1]:
            raise ValueError("clicks must be [n,2]; user_w and item_w must share emb_dim")
        self._keep = (clicks, user_w, item_w, his, masks, w0)
        self.num_negs = num_negs
        self.data_rows = clicks.shape[0]
        self.cfg = make_config(emb_dim=user_w.shape[1], num_negs=num_negs, num_users=user_w.shape[0],
                               num_items=item_w.shape[0], train_size=clicks.shape[0], **cfg_kwargs)
        self._h = C.c_void_p()
        max_his = his.shape[1] if his is not None else 0
        _check(load().heat_cf_engine_create(C.byref(self.cfg), _ptr(clicks), clicks.shape[0], _ptr(his), max_his,
                                            _ptr(masks), _ptr(user_w), _ptr(item_w), _ptr(w0), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h and load is not None:   # load is None at interpreter teardown
            load().heat_cf_engine_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def train_one_epoch(self):
        loss = C.c_float()
        _check(load().heat_cf_train_one_epoch(self._h, C.byref(loss)))
        return float(loss.value)

    def begin_epoch(self):
        _check(load().heat_cf_begin_epoch(self._h))

    def end_epoch(self):
        _check(load().heat_cf_end_epoch(self._h))

    def train_range(self, begin, end, neg_ids=None, want_loss=True):
        if neg_ids is not None:
            neg_ids = np.ascontiguousarray(neg_ids, dtype=np.uint64)
            if neg_ids.shape != (end - begin, self.num_negs):
                raise ValueError("neg_ids must be [end-begin, num_negs]")
        loss = C.c_double()
        _check(load().heat_cf_train_range(self._h, begin, end, _ptr(neg_ids), C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def sample_negatives(self, begin, end):
        out = np.empty((end - begin, self.num_negs), dtype=np.uint64)
        _check(load().heat_cf_sample_negatives(self._h, begin, end, _ptr(out)))
        return out

    def evaluate0(self):
        sim = np.empty((self.cfg.num_users, self.cfg.num_items), dtype=np.float32)
        _check(load().heat_cf_evaluate0(self._h, _ptr(sim)))
        return sim

    def topk(self, k, u_begin=0, u_end=None, mask_indptr=None, mask_items=None):
        u_end = self.cfg.num_users if u_end is None else u_end
        out = np.empty((u_end - u_begin, k), dtype=np.uint32)
        if mask_indptr is not None:
            mask_indptr = np.ascontiguousarray(mask_indptr, dtype=np.uint64)
            mask_items = np.ascontiguousarray(mask_items, dtype=np.uint32)
        _check(load().heat_cf_topk(self._h, u_begin, u_end, k, _ptr(mask_indptr), _ptr(mask_items), _ptr(out)))
        return out

    def sync_to_host(self):
        _check(load().heat_cf_sync_to_host(self._h))

    def sync_from_host(self):
        _check(load().heat_cf_sync_from_host(self._h))

    def synchronize(self):
        _check(load().heat_cf_synchronize(self._h))

    def sync_delta(self, ref_ptr, mine_ptr, sum_ptr):
        """mine = sum = W_item - ref (device pointers; asynchronous on the engine's stream)."""
        _check(load().heat_cf_sync_delta(self._h, C.c_void_p(ref_ptr), C.c_void_p(mine_ptr) if mine_ptr else None,
                                         C.c_void_p(sum_ptr)))

    def sync_apply(self, ref_ptr, sum_ptr, mine_ptr, scale=1.0):
        """W_item += scale * sum - mine; ref += scale * sum; mine_ptr=None: W_item = ref = ref + scale * sum
        (device pointers; asynchronous on the engine's stream)."""
        _check(load().heat_cf_sync_apply(self._h, C.c_void_p(ref_ptr), C.c_void_p(sum_ptr),
                                         C.c_void_p(mine_ptr) if mine_ptr else None, scale))

    def sync_apply_delta(self, ref_ptr, sum_ptr, mine_ptr, scale=1.0):
        """W_item += scale * sum - mine; ref += scale * sum; mine = sum = W_item - ref — apply of one exchange and delta of
        the next in one pass (device pointers; asynchronous on the engine's stream)."""
        _check(load().heat_cf_sync_apply_delta(self._h, C.c_void_p(ref_ptr), C.c_void_p(sum_ptr), C.c_void_p(mine_ptr), scale))

    def sync_apply_snap(self, x_ptr, snap_ptr):
        """W_item += x (x_ptr 0: nothing to add); snap = W_item — the one pass of the pipelined exchange that runs on the
        engine's stream (device pointers)."""
        _check(load().heat_cf_sync_apply_snap(self._h, C.c_void_p(x_ptr) if x_ptr else None, C.c_void_p(snap_ptr)))

    def sync_delta_from(self, snap_ptr, ref_ptr, mine_ptr, sum_ptr, stream):
        """mine = sum = snap - ref on `stream` (a hipStream_t handle of the caller)."""
        _check(load().heat_cf_sync_delta_from(self._h, C.c_void_p(snap_ptr), C.c_void_p(ref_ptr), C.c_void_p(mine_ptr) if mine_ptr else None,
                                              C.c_void_p(sum_ptr), C.c_void_p(stream) if stream else None))

    def sync_finish(self, ref_ptr, sum_ptr, mine_x_ptr, scale, stream):
        """s = scale * sum; x = s - mine (over mine); ref += s on `stream`."""
        _check(load().heat_cf_sync_finish(self._h, C.c_void_p(ref_ptr), C.c_void_p(sum_ptr), C.c_void_p(mine_x_ptr), scale,
                                          C.c_void_p(stream) if stream else None))

    def zero_grad(self):
        _check(load().heat_cf_zero_grad(self._h))

    def read_device(self, ptr, shape, dtype=np.float32):
        """Host copy of a device buffer (e.g. device_view().item_g)."""
        out = np.empty(shape, dtype=dtype)
        _check(load().heat_cf_copy_to_host(self._h, C.c_void_p(ptr), _ptr(out), out.nbytes))
        return out

    def device_view(self):
        v = DeviceView()
        _check(load().heat_cf_get_device_view(self._h, C.byref(v)))
        return v

    @property
    def epoch(self):
        return int(load().heat_cf_epoch(self._h))

    @epoch.setter
    def epoch(self, v):
        _check(load().heat_cf_set_epoch(self._h, v))

    @property
    def l_r(self):
        return float(load().heat_cf_learning_rate(self._h))

    @l_r.setter
    def l_r(self, v):
        _check(load().heat_cf_set_learning_rate(self._h, v))

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        _check(load().heat_cf_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return float(ms.value), int(n.value)

    @property
    def kernel_name(self):
        return load().heat_cf_kernel_name(self._h).decode()
