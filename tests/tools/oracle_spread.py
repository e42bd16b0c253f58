"""Development aid (CPU only): run-to-run spread of the 8-thread oracle's Recall@20 / NDCG@20 on a synthetic graph.
A parity criterion of +-1e-3 against the oracle only means something where the oracle agrees with itself to better
than that; this prints the metrics of several oracle runs (different table seeds, dynamic scheduling) at checkpoints."""
import argparse
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heat_amd.cf import metrics, synthetic
from oracle import cf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="yelp18")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--clusters", type=int, default=0)
ap.add_argument("--in-cluster", type=float, default=0.8)
ap.add_argument("--zipf", type=float, default=1.0)
ap.add_argument("--clip", type=float, default=0.1)
ap.add_argument("--lr", type=float, default=0.01)
ap.add_argument("--threads", type=str, default="8")
ap.add_argument("--seeds", type=str, default="1,2,3")
ap.add_argument("--checkpoints", type=str, default="8")
ap.add_argument("--same-init", action="store_true", help="every run starts from the seed-2022 tables (spread = scheduling only)")
args = ap.parse_args()

g, d, N = synthetic.make_named(args.shape, scale=args.scale, n_clusters=args.clusters, in_cluster=args.in_cluster,
                               zipf_s=args.zipf)
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                             for u in range(g.num_users) if ep[u + 1] > ep[u]})
ms = ["Recall(k=20)", "NDCG(k=20)"]
tp = g.train_indptr.astype(np.int64)
ti = g.train_items.astype(np.int64)
print(f"shape={args.shape} users={g.num_users} items={g.num_items} train={g.clicks.shape[0]} test={g.test_items.size} "
      f"d={d} N={N} clusters={args.clusters} clip={args.clip} lr={args.lr}", flush=True)
pop = np.bincount(ti, minlength=g.num_items)
print(f"hottest item share of positives: {pop.max() / ti.size:.4f}; top-20-popular Recall baseline:", end=" ")


def topk_cpu(uw, iw, k=20, chunk=2048):
    out = np.empty((g.num_users, k), dtype=np.uint32)
    for u0 in range(0, g.num_users, chunk):
        u1 = min(g.num_users, u0 + chunk)
        sim = uw[u0:u1] @ iw.T
        rows = np.repeat(np.arange(u1 - u0), np.diff(tp[u0:u1 + 1]))
        sim[rows, ti[tp[u0]:tp[u1]]] = -np.inf
        part = np.argpartition(-sim, k, axis=1)[:, :k]
        sc = np.take_along_axis(sim, part, axis=1)
        out[u0:u1] = np.take_along_axis(part, np.argsort(-sc, axis=1, kind="stable"), axis=1)
    return out


popular = np.argsort(-pop, kind="stable")[:200]
top_pop = np.empty((g.num_users, 20), dtype=np.uint32)
for u in range(g.num_users):
    seen = set(ti[tp[u]:tp[u + 1]].tolist())
    top_pop[u] = [i for i in popular if i not in seen][:20]
print(metrics.evaluate_topk(test, top_pop, ms, quiet=True, by_user_id=True), flush=True)

cps = [int(x) for x in args.checkpoints.split(",")]
for th in [int(x) for x in args.threads.split(",")]:
    rows = {c: [] for c in cps}
    for seed in [int(x) for x in args.seeds.split(",")]:
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=2022 if args.same_init else seed)
        ora = orc.Engine(g.clicks, uw, iw, num_negs=N, clip_val=args.clip, l_r=args.lr)
        t0 = time.time()
        losses = []
        for e in range(max(cps)):
            losses.append(ora.train_one_epoch(num_threads=th))
            if e + 1 in cps:
                r = metrics.evaluate_topk(test, topk_cpu(uw, iw), ms, quiet=True, by_user_id=True)
                rows[e + 1].append((r[ms[0]], r[ms[1]]))
                print(f"  threads={th} seed={seed} epoch={e + 1}: loss={losses[-1]:.4f} Recall@20={r[ms[0]]:.5f} "
                      f"NDCG@20={r[ms[1]]:.5f} ({time.time() - t0:.0f}s)", flush=True)
        print(f"  losses {[round(x, 4) for x in losses]}", flush=True)
    for c in cps:
        a = np.array(rows[c])
        print(f"threads={th} epoch={c}: Recall mean {a[:, 0].mean():.5f} spread {np.ptp(a[:, 0]):.5f} | "
              f"NDCG mean {a[:, 1].mean():.5f} spread {np.ptp(a[:, 1]):.5f}", flush=True)
