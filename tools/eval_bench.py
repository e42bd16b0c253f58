"""Development aid: time top-k evaluation at a named shape, fused path against the materialised panel path
(run on the GPU box; under `rocprofv3 --kernel-trace --stats` for per-kernel times)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic

shape = sys.argv[1] if len(sys.argv) > 1 else "amazonbooks"
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20, 50]
paths = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fused", "panel"]
max_users = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # rank only the first max_users users (huge shapes)
g, d, N = synthetic.make_named(shape)
uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d)
eng = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
n_rank = min(g.num_users, max_users) if max_users else g.num_users
flop = 2.0 * n_rank * g.num_items * d
res = {}
for k in ks:
    for path in paths:
        os.environ["HEAT_CF_TOPK_PATH"] = path
        best = 1e9
        for rep in range(3):
            t0 = time.time()
            top = eng.topk(k, u_end=n_rank, mask_indptr=g.train_indptr, mask_items=g.train_items)
            best = min(best, time.time() - t0)
        res[(k, path)] = top
        print(f"{shape}: top-{k} [{path}] {n_rank} users x {g.num_items} items d={d}: {best * 1e3:.1f} ms "
              f"(host call, best of 3) = {flop / best / 1e12:.1f} TFLOP/s", flush=True)
    if len(paths) == 2:
        print(f"  paths agree id for id: {np.array_equal(res[(k, paths[0])], res[(k, paths[1])])}", flush=True)
