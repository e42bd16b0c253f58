"""Development aid: isolates where the serial GPU walk and the oracle diverge (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heat_amd import abi
from oracle import cf_oracle as orc


def run(name, clicks, negs, U, I, d, N, clip=1.0, lr=0.01, scale=0.01, coherence=0, seed=0):
    rng = np.random.default_rng(seed)
    uw = (rng.standard_normal((U, d)) * scale).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * scale).astype(np.float32)
    clicks = np.ascontiguousarray(clicks, dtype=np.uint64)
    negs = np.ascontiguousarray(negs, dtype=np.uint64)
    ug, ig, uo, io = uw.copy(), iw.copy(), uw.copy(), iw.copy()
    eng = abi.Engine(clicks, ug, ig, num_negs=N, clip_val=clip, l_r=lr, flags=abi.FLAG_SERIAL, coherence=coherence)
    ora = orc.Engine(clicks, uo, io, num_negs=N, clip_val=clip, l_r=lr)
    T = clicks.shape[0]
    lg = eng.train_range(0, T, negs)
    lo = ora.train_range(0, T, negs)
    eng.sync_to_host()
    eu = np.abs(ug - uo).max() / max(np.abs(uo).max(), 1e-30)
    ei = np.abs(ig - io).max() / max(np.abs(io).max(), 1e-30)
    bad_items = np.flatnonzero(np.abs(ig - io).max(axis=1) > 1e-4 * np.abs(io).max())
    print(f"{name:44s} loss gpu={lg:.6f} cpu={lo:.6f} rel={abs(lg-lo)/abs(lo):.2e}  user_err={eu:.2e} item_err={ei:.2e} bad_item_rows={bad_items[:8].tolist()}", flush=True)
    eng.close()


rng = np.random.default_rng(5)
for (d2, N2) in [(64, 100), (64, 64), (64, 32), (128, 64), (32, 4), (20, 3), (64, 16)]:
    I2 = 4000
    for T in (1, 2, 20, 300):
        cl = np.stack([np.sort(rng.integers(0, 10, T)), rng.integers(0, I2, T)], axis=1)
        ng = np.stack([rng.choice(I2, N2, replace=False) for _ in range(T)])
        run(f"T={T} d={d2} N={N2} no dups scale .1", cl, ng, 10, I2, d2, N2, scale=0.1)
    T = 300
    cl = np.stack([np.sort(rng.integers(0, 10, T)), rng.integers(0, I2, T)], axis=1)
    ng = rng.integers(0, 300, size=(T, N2))
    run(f"T={T} d={d2} N={N2} dups scale .1", cl, ng, 10, I2, d2, N2, scale=0.1)
