"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/cf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (heat_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcf_oracle.so")


def build(force=False):
    """Compile oracle/cf_oracle.c -> oracle/libcf_oracle.so (gcc, OpenMP)."""
    src = os.path.join(_HERE, "cf_oracle.c")
    hdr = os.path.join(_HERE, "cf_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _MT64(C.Structure):
    _fields_ = [("mt", C.c_uint64 * 312), ("idx", C.c_int)]


class _Uniform(C.Structure):
    _fields_ = [("rng", _MT64), ("max_idx", C.c_uint64)]


class _Config(C.Structure):
    _fields_ = [("emb_dim", C.c_uint64), ("num_negs", C.c_uint64), ("num_users", C.c_uint64),
                ("num_items", C.c_uint64), ("train_size", C.c_uint64), ("neg_sampler", C.c_uint64),
                ("tile_size", C.c_uint64), ("refresh_interval", C.c_uint64), ("num_subepochs", C.c_uint64),
                ("l2", C.c_float), ("clip_val", C.c_float), ("milestones", C.POINTER(C.c_uint64)),
                ("n_milestones", C.c_uint64), ("l_r", C.c_float)]


class _Sampler(C.Structure):
    _fields_ = [("num_negs", C.c_uint64), ("is_tile", C.c_int), ("neg_sampler", _Uniform),
                ("tile_sampler", _Uniform), ("tile_size", C.c_uint64), ("refresh_interval", C.c_uint64),
                ("iterations", C.c_uint64), ("neg_tile", C.POINTER(C.c_uint64))]


class _Engine(C.Structure):
    _fields_ = [("cfg", _Config), ("clicks", C.c_void_p), ("data_rows", C.c_uint64), ("his", C.c_void_p),
                ("masks", C.c_void_p), ("max_his", C.c_uint64), ("user_w", C.c_void_p), ("item_w", C.c_void_p),
                ("user_g", C.POINTER(C.c_float)), ("item_g", C.POINTER(C.c_float)), ("w0", C.c_void_p),
                ("use_aggregator", C.c_int), ("l_r", C.c_float), ("epoch", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_uniform_init.argtypes = [C.POINTER(_Uniform), C.c_uint64, C.c_uint64]
        L.orc_uniform_read.argtypes = [C.POINTER(_Uniform)]
        L.orc_uniform_read.restype = C.c_uint64
        L.orc_sampler_init.argtypes = [C.POINTER(_Sampler), C.POINTER(_Config), C.c_uint64, C.c_int]
        L.orc_sampler_free.argtypes = [C.POINTER(_Sampler)]
        L.orc_sampler_sampling.argtypes = [C.POINTER(_Sampler), C.c_void_p]
        L.orc_sampler_ignore_pos_sampling.argtypes = [C.POINTER(_Sampler), C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_clip_grad.argtypes = [C.c_float, C.c_float]
        L.orc_clip_grad.restype = C.c_float
        L.orc_sparse_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float]
        L.orc_scheduler_step_lr.argtypes = [C.c_float, C.c_uint64, C.c_uint64, C.c_float]
        L.orc_scheduler_step_lr.restype = C.c_float
        L.orc_scheduler_multi_step_lr.argtypes = [C.c_float, C.c_uint64, C.c_void_p, C.c_uint64, C.c_float]
        L.orc_scheduler_multi_step_lr.restype = C.c_float
        L.orc_engine_create.argtypes = [C.POINTER(_Config), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_engine_create.restype = C.POINTER(_Engine)
        L.orc_engine_destroy.argtypes = [C.POINTER(_Engine)]
        L.orc_engine_zero_grad.argtypes = [C.POINTER(_Engine)]
        L.orc_engine_lr_step.argtypes = [C.POINTER(_Engine)]
        L.orc_worker_create.argtypes = [C.POINTER(_Engine)]
        L.orc_worker_create.restype = C.c_void_p
        L.orc_worker_destroy.argtypes = [C.c_void_p]
        L.orc_forward_backward.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_forward_backward.restype = C.c_float
        L.orc_train_range.argtypes = [C.POINTER(_Engine), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_train_range.restype = C.c_double
        L.orc_train_one_epoch.argtypes = [C.POINTER(_Engine), C.c_int, C.c_int, C.c_void_p]
        L.orc_train_one_epoch.restype = C.c_float
        L.orc_evaluate0.argtypes = [C.POINTER(_Engine), C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Uniform:
    """random/uniform.hpp:16-30"""

    def __init__(self, max_idx, seed):
        self._u = _Uniform()
        lib().orc_uniform_init(C.byref(self._u), max_idx, seed)

    def read(self):
        return int(lib().orc_uniform_read(C.byref(self._u)))


def _make_config(emb_dim, num_negs, num_users, num_items, train_size, neg_sampler=0, tile_size=512,
                 refresh_interval=8192, num_subepochs=2, l2=1e-7, clip_val=1.0, milestones=(10,), l_r=0.01):
    ms = np.ascontiguousarray(np.asarray(list(milestones), dtype=np.uint64))
    cfg = _Config(emb_dim, num_negs, num_users, num_items, train_size, neg_sampler, tile_size, refresh_interval,
                  num_subepochs, l2, clip_val, ms.ctypes.data_as(C.POINTER(C.c_uint64)), len(ms), l_r)
    return cfg, ms


class Sampler:
    """negative_samplers/{uniform_random,random_tile}_negative_sampler.cpp"""

    def __init__(self, num_items, num_negs, seed, tile=False, tile_size=512, refresh_interval=8192):
        cfg, self._ms = _make_config(64, num_negs, 1, num_items, 1, tile_size=tile_size,
                                     refresh_interval=refresh_interval)
        self._s = _Sampler()
        self.num_negs = num_negs
        lib().orc_sampler_init(C.byref(self._s), C.byref(cfg), seed, int(tile))
        self.neg_ids = np.zeros(num_negs, dtype=np.uint64)  # zero-initialised like engine.cpp:298

    def sampling(self):
        lib().orc_sampler_sampling(C.byref(self._s), _ptr(self.neg_ids))
        return self.neg_ids.copy()

    def ignore_pos_sampling(self, user_id, pos_id):
        lib().orc_sampler_ignore_pos_sampling(C.byref(self._s), user_id, pos_id, _ptr(self.neg_ids))
        return self.neg_ids.copy()

    def tile(self):
        return np.array([self._s.neg_tile[i] for i in range(self._s.tile_size)], dtype=np.uint64)

    def __del__(self):
        try:
            lib().orc_sampler_free(C.byref(self._s))
        except Exception:
            pass


def clip_grad(g, clip):
    return float(lib().orc_clip_grad(g, clip))


def sparse_step(emb, grad, clip, lr):
    assert emb.dtype == np.float32 and grad.dtype == np.float32
    lib().orc_sparse_step(_ptr(emb), _ptr(grad), emb.size, clip, lr)


def scheduler_step_lr(lr, epoch, step, gamma=0.1):
    return float(lib().orc_scheduler_step_lr(lr, epoch, step, gamma))


def scheduler_multi_step_lr(lr, epoch, milestones, gamma=0.1):
    ms = np.asarray(milestones, dtype=np.uint64)
    return float(lib().orc_scheduler_multi_step_lr(lr, epoch, _ptr(ms), len(ms), gamma))


class Engine:
    """CPU oracle engine.  Arrays are BORROWED and trained in place, like the reference
    (pybind/init_modules.cpp:45-56,74-84)."""

    def __init__(self, clicks, user_w, item_w, *, num_negs, his=None, masks=None, w0=None, use_aggregator=False,
                 neg_sampler=0, tile_size=512, refresh_interval=8192, clip_val=1.0, milestones=(10,), l_r=0.01,
                 l2=1e-7):
        assert clicks.dtype == np.uint64 and clicks.ndim == 2 and clicks.shape[1] == 2 and clicks.flags.c_contiguous
        assert user_w.dtype == np.float32 and item_w.dtype == np.float32
        assert user_w.flags.c_contiguous and item_w.flags.c_contiguous
        self.clicks, self.user_w, self.item_w, self.his, self.masks, self.w0 = clicks, user_w, item_w, his, masks, w0
        d = user_w.shape[1]
        if use_aggregator:
            assert his is not None and masks is not None and w0 is not None
            assert his.dtype == np.uint64 and masks.dtype == np.uint64 and w0.dtype == np.float32
        self.num_negs = num_negs
        self.emb_dim = d
        cfg, self._ms = _make_config(d, num_negs, user_w.shape[0], item_w.shape[0], clicks.shape[0], neg_sampler,
                                     tile_size, refresh_interval, 2, l2, clip_val, milestones, l_r)
        max_his = his.shape[1] if his is not None else 0
        self._e = lib().orc_engine_create(C.byref(cfg), _ptr(clicks), clicks.shape[0], _ptr(his), max_his,
                                          _ptr(masks), _ptr(user_w), _ptr(item_w), _ptr(w0), int(use_aggregator))
        self._w = None

    # -- state ---------------------------------------------------------------------------
    @property
    def l_r(self):
        return float(self._e.contents.l_r)

    @l_r.setter
    def l_r(self, v):
        self._e.contents.l_r = v

    @property
    def epoch(self):
        return int(self._e.contents.epoch)

    @epoch.setter
    def epoch(self, v):
        self._e.contents.epoch = v

    def user_grads(self):
        n = self.user_w.size
        return np.ctypeslib.as_array(self._e.contents.user_g, shape=(n,)).reshape(self.user_w.shape)

    def item_grads(self):
        n = self.item_w.size
        return np.ctypeslib.as_array(self._e.contents.item_g, shape=(n,)).reshape(self.item_w.shape)

    # -- steps ----------------------------------------------------------------------------
    def worker(self):
        if self._w is None:
            self._w = lib().orc_worker_create(self._e)
        return self._w

    def reset_worker(self):
        if self._w is not None:
            lib().orc_worker_destroy(self._w)
            self._w = None

    def forward_backward(self, user_id, pos_id, neg_ids):
        neg_ids = np.ascontiguousarray(neg_ids, dtype=np.uint64)
        assert neg_ids.size == self.num_negs
        return float(lib().orc_forward_backward(self.worker(), int(user_id), int(pos_id), _ptr(neg_ids)))

    def train_range(self, begin, end, neg_ids):
        neg_ids = np.ascontiguousarray(neg_ids, dtype=np.uint64)
        assert neg_ids.shape == (end - begin, self.num_negs)
        return float(lib().orc_train_range(self._e, self.worker(), begin, end, _ptr(neg_ids)))

    def lr_step(self):
        lib().orc_engine_lr_step(self._e)

    def zero_grad(self):
        lib().orc_engine_zero_grad(self._e)

    def train_one_epoch(self, num_threads=0, sampler_call=0, record_negs=False):
        neg_out = np.zeros((self.clicks.shape[0], self.num_negs), dtype=np.uint64) if record_negs else None
        loss = float(lib().orc_train_one_epoch(self._e, num_threads, sampler_call, _ptr(neg_out)))
        return (loss, neg_out) if record_negs else loss

    def evaluate0(self):
        sim = np.empty((self.user_w.shape[0], self.item_w.shape[0]), dtype=np.float32)
        lib().orc_evaluate0(self._e, _ptr(sim))
        return sim

    def __del__(self):
        try:
            self.reset_worker()
            lib().orc_engine_destroy(self._e)
        except Exception:
            pass
