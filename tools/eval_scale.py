"""Development aid: fused top-k at the item count and width of BASELINE.json configs[4] (1 M items, d=256) for a slice of
its 10 M users — the evaluation the reference cannot run at all there (its evaluate0 would materialise a
10 M x 1 M fp32 matrix, 40 TB).  Device-mode engine on torch tensors; run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi

users = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
I, d, k, per_user = 1_000_000, 256, 20, 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(2022)
user_w = torch.empty((users, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
item_w = torch.empty((I, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
clicks = torch.zeros((1, 2), dtype=torch.int64, device=dev)
eng = abi.Engine.from_device(clicks.data_ptr(), 1, user_w.data_ptr(), item_w.data_ptr(), num_users=users, num_items=I,
                             emb_dim=d, num_negs=100, keep=(clicks, user_w, item_w))
rng = np.random.default_rng(0)
indptr = (np.arange(users + 1, dtype=np.uint64) * per_user)
items = rng.integers(0, I, size=users * per_user, dtype=np.uint32)      # unsorted rows: the device sort is part of the call
for rep in range(2):
    t0 = time.time()
    top = eng.topk(k, mask_indptr=indptr, mask_items=items)
    dt = time.time() - t0
    print(f"top-{k} for {users} users x {I} items d={d}: {dt:.3f} s = {2.0 * users * I * d / dt / 1e12:.1f} TFLOP/s; "
          f"10 M users at this rate: {dt * 1e7 / users:.0f} s on one GPU", flush=True)
# spot check 3 users against a torch fp32 matmul (different summation order: compare the id sets, not bits)
for u in (0, users // 2, users - 1):
    s = (item_w @ user_w[u]).cpu().numpy()
    s[items[u * per_user:(u + 1) * per_user]] = -np.inf
    want = set(np.argsort(-s, kind="stable")[:k].tolist())
    print(f"user {u}: {len(want & set(top[u].tolist()))}/{k} ids agree with torch", flush=True)
