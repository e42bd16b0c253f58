"""Development aid (GPU box): does the asynchrony bound of the wide configs come from the FIRST epoch?  Epoch 0 with S0
streams, the remaining epochs with S1 (two device-mode engines on the same torch tensors), Recall@20 / NDCG@20 after the
yaml's 8 epochs on the clustered graph; compare with the oracle means in profiles/r02_yelp18_*.txt."""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heat_amd import abi
from heat_amd.cf import metrics, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="gowalla")
ap.add_argument("--plans", default="144:144,128:256,96:256,144:320")
ap.add_argument("--warm-epochs", type=int, default=1)
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--seeds", default="1,2")
args = ap.parse_args()
g, d, N = synthetic.make_named(args.shape, n_clusters=64)
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist() for u in range(g.num_users) if ep[u + 1] > ep[u]})
ms = ["Recall(k=20)", "NDCG(k=20)"]
dev = torch.device("cuda", 0)
abi.load()
side = torch.cuda.Stream(device=dev)
T = g.clicks.shape[0]
for plan in args.plans.split(","):
    s0, s1 = (int(x) for x in plan.split(":"))
    res = []
    for seed in [int(x) for x in args.seeds.split(",")]:
        uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        with torch.cuda.stream(side):
            tc = torch.from_numpy(g.clicks.view(np.int64)).to(dev)
            tu, ti = torch.from_numpy(uw0).to(dev), torch.from_numpy(iw0).to(dev)
            common = dict(num_users=g.num_users, num_items=g.num_items, emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=seed,
                          clip_val=0.1, l_r=0.01, keep=(tc, tu, ti))
            a = abi.Engine.from_device(tc.data_ptr(), T, tu.data_ptr(), ti.data_ptr(), num_streams=s0, **common)
            b = abi.Engine.from_device(tc.data_ptr(), T, tu.data_ptr(), ti.data_ptr(), num_streams=s1, **common)
            losses = [a.train_one_epoch() for _ in range(args.warm_epochs)]
            b.epoch = args.warm_epochs
            losses += [b.train_one_epoch() for _ in range(args.epochs - args.warm_epochs)]
            ms_b, n_b = b.kernel_time()
            side.synchronize()
            uw, iw = tu.cpu().numpy(), ti.cpu().numpy()
            a.close(); b.close()
        e = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = e.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        e.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        res.append((r[ms[0]], r[ms[1]], losses[-1]))
        print(f"  plan {s0}->{s1} seed {seed}: Recall@20={r[ms[0]]:.5f} NDCG@20={r[ms[1]]:.5f} final loss {losses[-1]:.4f}  ({ms_b / max(n_b, 1):.2f} ms/epoch at {s1} streams)", flush=True)
    m = np.array(res).mean(axis=0)
    print(f"{args.shape} plan {s0}->{s1} (first {args.warm_epochs} epoch(s) at {s0}): mean Recall {m[0]:.5f} NDCG {m[1]:.5f} loss {m[2]:.4f}", flush=True)
