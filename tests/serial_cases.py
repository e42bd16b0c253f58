"""Test infrastructure: the random serial-walk configurations of tools/serial_sweep.py as a generator, and the four
trajectories the arbiter compares on one configuration — serial GPU walk, fp32 oracle, float64 model (tests/f64_model.py),
fp32 oracle started one ulp away.  Used by tests/test_gpu_parity.py and tools/serial_arbiter.py."""
import numpy as np

from heat_amd import abi
from oracle import cf_oracle as orc
from tests import f64_model


def sweep_cases(cases, seed):
    """The generator of tools/serial_sweep.py, draw for draw."""
    rng = np.random.default_rng(seed)
    modes = [abi.UPDATE_OVERWRITE, abi.UPDATE_ATOMIC_W, abi.UPDATE_ATOMIC_WG, abi.UPDATE_ATOMIC_POS, abi.UPDATE_AUTO, 16 + 0x1C, 16 + 0x10]
    for case in range(cases):
        d = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 96, 128, 160, 256]))
        N = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 50, 64, 100]))
        U = int(rng.integers(2, 12)); I = int(rng.integers(max(3, N // 4), 400)); T = int(rng.integers(1, 150))
        agg = case % 3 == 2
        mode = modes[case % len(modes)] if not agg else abi.UPDATE_AUTO
        clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
        uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32); iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
        his = masks = w0 = None
        if agg:
            H = int(rng.integers(1, 120))
            masks = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
            his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
            w0 = (rng.standard_normal((d, d)) * 0.05).astype(np.float32)
        yield case, dict(d=d, N=N, U=U, I=I, T=T, agg=agg, mode=mode, clicks=clicks, uw=uw, iw=iw, negs=negs, his=his, masks=masks, w0=w0)


def run_gpu(c, steps=None, want_loss=False):
    T = c["T"] if steps is None else steps
    ug, ig = c["uw"].copy(), c["iw"].copy()
    kw = dict(his=c["his"], masks=c["masks"], w0=c["w0"].copy(), use_aggregator=True) if c["agg"] else {}
    eng = abi.Engine(c["clicks"], ug, ig, num_negs=c["N"], flags=abi.FLAG_SERIAL, update_mode=c["mode"], clip_val=0.5, l_r=0.01, **kw)
    run_gpu.last_loss = eng.train_range(0, T, c["negs"][:T], want_loss=True)
    eng.sync_to_host()
    name = eng.kernel_name
    eng.close()
    return ug, ig, kw.get("w0"), name


def run_oracle(c, steps=None, nudge=False, want_loss=False):
    T = c["T"] if steps is None else steps
    uo, io = c["uw"].copy(), c["iw"].copy()
    if nudge:          # one ulp in one element of the first positive's row
        r = int(c["clicks"][0, 1])
        io[r, 0] = np.nextafter(io[r, 0], np.float32(1.0))
    kw = dict(his=c["his"], masks=c["masks"], w0=c["w0"].copy(), use_aggregator=True) if c["agg"] else {}
    run_oracle.last_loss = orc.Engine(c["clicks"], uo, io, num_negs=c["N"], clip_val=0.5, l_r=0.01, **kw).train_range(0, T, c["negs"][:T])
    return uo, io, kw.get("w0")


def run_f64(c, steps=None):
    T = c["T"] if steps is None else steps
    uw, iw = c["uw"].astype(np.float64), c["iw"].astype(np.float64)
    ug, ig = np.zeros_like(uw), np.zeros_like(iw)
    w0 = c["w0"].astype(np.float64) if c["agg"] else None
    states = {}
    for t in range(T):
        u, p = int(c["clicks"][t, 0]), int(c["clicks"][t, 1])
        agg = None
        if c["agg"]:
            h = int(c["masks"][u, 0])
            st = states.setdefault("s", dict(iteration=0, accu=np.zeros_like(w0)))      # ONE worker: serial walk
            agg = dict(his=c["his"][u, :h].astype(np.int64).tolist(), w0=w0, state=st, gamma=float(np.float32(0.4)), agg_lr=float(np.float32(0.01)))
        f64_model.step_f64(uw, iw, ug, ig, u, p, c["negs"][t].astype(np.int64).tolist(), clip=0.5, lr=float(np.float32(0.01)), agg=agg)
    return uw, iw, w0


def dist(a, b):
    """max relative table error in units of the suite's tolerance (3e-4 of the table's largest entry)."""
    out = 0.0
    for x, y in zip(a, b):
        if x is None or y is None:
            continue
        out = max(out, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max() / max(1e-30, np.abs(y).max()) / 3e-4))
    return out
