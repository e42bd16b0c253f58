"""Scratch timing of the training kernel on a named synthetic shape (development aid; bench.py is the contract)."""
import argparse
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="amazonbooks")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--streams", type=str, default="0")
ap.add_argument("--coherence", type=str, default="2")
ap.add_argument("--update", type=str, default="0")
ap.add_argument("--d", type=int, default=0)
ap.add_argument("--negs", type=int, default=0)
ap.add_argument("--agg", action="store_true", help="behaviour aggregation (ACCL): histories of up to 100 items, W0 d x d")
args = ap.parse_args()

(g, d, N) = synthetic.make_named(args.shape, scale=args.scale, with_test=False)
d = args.d or d
N = args.negs or N
B = 16 * d * (N + 2) + 16
agg_kw = {}
if args.agg:
    his, masks = synthetic.make_history(g, 100, seed=2022)
    agg_kw = dict(his=his, masks=masks, use_aggregator=True)
    B += int(round(float(masks[g.clicks[:, 0].astype(np.int64)].mean()) * 4 * d))      # + the history rows an interaction reads
print(f"shape={args.shape} users={g.num_users} items={g.num_items} n={g.clicks.shape[0]} d={d} N={N} B/sample={B}", flush=True)
for coh, upd in [(int(c), int(u)) for c in args.coherence.split(",") for u in args.update.split(",")]:
    for streams in [int(x) for x in args.streams.split(",")]:
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d)
        if args.agg:
            agg_kw["w0"] = (np.random.default_rng(2022).standard_normal((d, d)) * 0.01).astype(np.float32)
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, coherence=coh, num_streams=streams, flags=abi.FLAG_LAZY_SYNC,
                         update_mode=upd, **agg_kw)
        losses = []
        eng.train_one_epoch()  # warm-up
        eng.kernel_time(reset=True)
        t0 = time.time()
        for _ in range(args.epochs):
            losses.append(eng.train_one_epoch())
        wall = time.time() - t0
        ms, n = eng.kernel_time()
        per = ms / n
        sps = g.clicks.shape[0] / (per * 1e-3)
        print(f"coherence={coh} streams={streams} kernel={eng.kernel_name} {per:.3f} ms/epoch-kernel  {sps/1e6:.1f} M samples/s  "
              f"{sps*B/1e12:.2f} TB/s algorithmic  wall/epoch={wall/args.epochs*1e3:.2f} ms  losses={[round(x,4) for x in losses]}", flush=True)
        eng.close()
