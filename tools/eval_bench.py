"""Development aid: time evaluate / top-k at a named shape (run on the GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic

shape = sys.argv[1] if len(sys.argv) > 1 else "amazonbooks"
g, d, N = synthetic.make_named(shape)
uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d)
eng = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
for k in (20, 50):
    t0 = time.time()
    top = eng.topk(k, mask_indptr=g.train_indptr, mask_items=g.train_items)
    print(f"{shape}: topk k={k} over {g.num_users} users x {g.num_items} items: {time.time() - t0:.3f} s", flush=True)
