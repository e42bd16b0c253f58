"""Development aid (GPU box): epoch time of ONE user shard of an N-GPU job (bench.py's strong-scaling leg sees exactly
this per rank) for kernel variants / stream counts — what bounds strong scaling before the collective does."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic
from heat_amd.cf.distributed import shard_clicks

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="amazonbooks")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--streams", type=str, default="0")
ap.add_argument("--epochs", type=int, default=10)
args = ap.parse_args()
g, d, N = synthetic.make_named(args.shape, with_test=False)
uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d)
shard, lo, hi = shard_clicks(g.clicks, g.num_users, args.world, 0)
print(f"shard 0 of {args.world}: users={hi - lo} interactions={shard.shape[0]} variant={os.environ.get('HEAT_CF_VARIANT', 'auto')}", flush=True)
for streams in [int(x) for x in args.streams.split(",")]:
    u, i = uw[lo:hi].copy(), iw.copy()
    eng = abi.Engine(shard, u, i, num_negs=N, num_streams=streams, flags=abi.FLAG_LAZY_SYNC)
    eng.train_one_epoch()
    eng.kernel_time(reset=True)
    for _ in range(args.epochs):
        eng.train_one_epoch()
    ms, n = eng.kernel_time()
    per = ms / n
    print(f"  {eng.kernel_name}: {per:.3f} ms/epoch-shard -> {shard.shape[0] / per / 1e3:.1f} M samples/s per GPU; "
          f"x{args.world} = {args.world * shard.shape[0] / per / 1e3:.0f} M/s if the exchange is hidden", flush=True)
    eng.close()
