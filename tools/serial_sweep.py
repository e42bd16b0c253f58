"""Development aid (GPU box): the randomized serial parity sweep of tests/test_gpu_parity.py with other seeds and more cases,
with and without behaviour aggregation — serial GPU walk vs the oracle on caller-fed negatives.
usage: python tools/serial_sweep.py <cases> <seed>"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from oracle import cf_oracle as orc
cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
modes = [abi.UPDATE_OVERWRITE, abi.UPDATE_ATOMIC_W, abi.UPDATE_ATOMIC_WG, abi.UPDATE_ATOMIC_POS, abi.UPDATE_AUTO, 16 + 0x1C, 16 + 0x10]
bad, seen = 0, {}
for case in range(cases):
    d = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 96, 128, 160, 256]))
    N = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 50, 64, 100]))
    U = int(rng.integers(2, 12)); I = int(rng.integers(max(3, N // 4), 400)); T = int(rng.integers(1, 150))
    agg = case % 3 == 2
    mode = modes[case % len(modes)] if not agg else abi.UPDATE_AUTO
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32); iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
    kw_g, kw_o = {}, {}
    if agg:
        H = int(rng.integers(1, 120))
        masks = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
        his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
        w0 = (rng.standard_normal((d, d)) * 0.05).astype(np.float32)
        kw_g = dict(his=his, masks=masks, w0=w0.copy(), use_aggregator=True); kw_o = dict(his=his, masks=masks, w0=w0.copy(), use_aggregator=True)
    ug, ig, uo, io = uw.copy(), iw.copy(), uw.copy(), iw.copy()
    eng = abi.Engine(clicks, ug, ig, num_negs=N, flags=abi.FLAG_SERIAL, update_mode=mode, clip_val=0.5, l_r=0.01, **kw_g)
    lg = eng.train_range(0, T, negs); eng.sync_to_host(); name = eng.kernel_name; eng.close()
    lo = orc.Engine(clicks, uo, io, num_negs=N, clip_val=0.5, l_r=0.01, **kw_o).train_range(0, T, negs)
    e = max(abs(lg - lo) / max(1.0, abs(lo)) / 2e-5, np.abs(ug - uo).max() / np.abs(uo).max() / 3e-4, np.abs(ig - io).max() / np.abs(io).max() / 3e-4)
    if agg: e = max(e, np.abs(kw_g["w0"] - kw_o["w0"]).max() / max(1e-9, np.abs(kw_o["w0"]).max()) / 3e-4)
    key = name.split("/")[0] + ("+agg" if agg else "")
    seen[key] = max(seen.get(key, 0.0), float(e))
    if not e <= 1.0:
        bad += 1
        print(f"FAIL case {case}: d={d} N={N} U={U} I={I} T={T} mode={mode} agg={agg} {name} err/tol={e:.3g}", flush=True)
print(f"{cases} cases, {bad} failures; worst error / tolerance per kernel:")
for k in sorted(seen): print(f"  {k:40s} {seen[k]:.3f}")
