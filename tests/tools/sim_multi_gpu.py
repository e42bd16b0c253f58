"""CPU simulation of the multi-GPU item-table synchronisation rules (evidence for DESIGN.md §5): P replicas, each trained
by the oracle on its user shard, merged every `window` interactions per replica with the `sum` or `mean` rule; Recall@20 /
NDCG@20 against single-process training on the whole graph.  Test infrastructure (uses the oracle)."""
import argparse
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heat_amd.cf import metrics, synthetic
from heat_amd.cf.distributed import shard_clicks
from oracle import cf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.1)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--clusters", type=int, default=0)
args = ap.parse_args()
g, d, N = synthetic.make_named("amazonbooks", scale=args.scale, n_clusters=args.clusters)
U, I, T = g.num_users, g.num_items, g.clicks.shape[0]
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist() for u in range(U) if ep[u + 1] > ep[u]})
tp = g.train_indptr.astype(np.int64)
train = types.SimpleNamespace(user_items_dic={u: g.train_items[tp[u]:tp[u + 1]].tolist() for u in range(U)})
ms = ["Recall(k=20)", "NDCG(k=20)"]
uw0, iw0 = synthetic.init_embeddings(U, I, d, seed=2022)


def score(uw, iw, tag):
    sim = uw @ iw.T
    r = metrics.evaluate_metrics(train, test, sim, ms, quiet=True)
    print(f"{tag:60s} Recall@20={r[ms[0]]:.5f} NDCG@20={r[ms[1]]:.5f}", flush=True)


uw, iw = uw0.copy(), iw0.copy()
e = orc.Engine(g.clicks, uw, iw, num_negs=N)
losses = [e.train_one_epoch(num_threads=1) for _ in range(args.epochs)]
print(f"users={U} items={I} train={T}  single process losses={[round(x, 4) for x in losses]}")
score(uw, iw, "single process (1 thread)")

for P in (2, 8):
    for mode in ("sum", "mean"):
        for window in (10 ** 9, 8192, 1024):
            shards = [shard_clicks(g.clicks, U, P, r) for r in range(P)]
            uws = [uw0[lo:hi].copy() for (_, lo, hi) in shards]
            iws = [iw0.copy() for _ in range(P)]
            engs = [orc.Engine(sh, uws[r], iws[r], num_negs=N) for r, (sh, _, _) in enumerate(shards)]
            samplers = [orc.Sampler(I, N, 1000 + r) for r in range(P)]
            ref = iw0.copy()
            nmax = max(sh.shape[0] for sh, _, _ in shards)
            w = min(window, nmax)
            last = []
            for epoch in range(args.epochs):
                tot = 0.0
                for en in engs:
                    en.lr_step()
                for k in range(-(-nmax // w)):
                    for r, (sh, _, _) in enumerate(shards):
                        lo, hi = min(sh.shape[0], k * w), min(sh.shape[0], (k + 1) * w)
                        if hi > lo:
                            negs = np.stack([samplers[r].ignore_pos_sampling(0, int(sh[i, 1])) for i in range(lo, hi)])
                            tot += engs[r].train_range(lo, hi, negs)
                    if mode == "mean":
                        merged = sum(iws) / P
                    else:
                        merged = ref + sum(x - ref for x in iws)
                    for x in iws:
                        x[:] = merged
                    ref = merged.copy()
                for en in engs:
                    en.zero_grad()
                    en.epoch = en.epoch + 1
                last.append(tot / T)
            score(np.concatenate(uws), iws[0], f"P={P} mode={mode} window={'epoch' if window > T else window} losses={[round(x, 3) for x in last]}")
