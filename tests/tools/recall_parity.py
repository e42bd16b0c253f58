"""Development aid / evidence for DESIGN.md: Recall@20 / NDCG@20 of the Hogwild GPU engine vs the CPU oracle on the
same synthetic graph and the same initial tables (run on the GPU box)."""
import argparse
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heat_amd import abi
from heat_amd.cf import metrics, synthetic
from oracle import cf_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="amazonbooks")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--streams", type=str, default="0")
ap.add_argument("--coherence", type=str, default="2")
ap.add_argument("--update", type=str, default="0")
ap.add_argument("--oracle-threads", type=str, default="8")
ap.add_argument("--clip", type=float, default=1.0)
ap.add_argument("--seeds", type=str, default="2022")
ap.add_argument("--clusters", type=int, default=0)
ap.add_argument("--lr", type=float, default=0.01)
ap.add_argument("--zipf", type=float, default=1.0, help="item popularity exponent of the synthetic graph")
ap.add_argument("--interactions", type=int, default=0, help="override the shape's number of train interactions")
ap.add_argument("--in-cluster", type=float, default=0.8)
ap.add_argument("--graph-seed", type=int, default=2022, help="seed of the synthetic graph (the tables are seeded by --seeds)")
ap.add_argument("--users", type=int, default=0, help="with --interactions: override the shape's number of users")
ap.add_argument("--items", type=int, default=0, help="with --interactions: override the shape's number of items")
ap.add_argument("--oracle-seeds", type=str, default="", help="seeds the oracle runs for (default: every seed)")
ap.add_argument("--agg", action="store_true", help="behaviour aggregation (ACCL) on both sides")
ap.add_argument("--tile", action="store_true", help="random-tile negative sampler (neg_sampler 1, tile 512, refresh 8192; the "
                "`sampling` call, random_tile_negative_sampler.cpp:31-45) on both sides")
args = ap.parse_args()

if args.interactions:
    _U, _I, _T, d, N = synthetic.SHAPES[args.shape]
    _U, _I = args.users or _U, args.items or _I
    g = synthetic.make_graph(_U, _I, args.interactions, seed=args.graph_seed, n_clusters=args.clusters, zipf_s=args.zipf, in_cluster=args.in_cluster)
else:
    g, d, N = synthetic.make_named(args.shape, seed=args.graph_seed, scale=args.scale, n_clusters=args.clusters, zipf_s=args.zipf, in_cluster=args.in_cluster)
_pop = np.bincount(g.train_items, minlength=g.num_items)
print(f"hottest item share of positives {_pop.max() / g.train_items.size:.4f}")
test_dic = {}
ep = g.test_indptr.astype(np.int64)
for u in range(g.num_users):
    if ep[u + 1] > ep[u]:
        test_dic[u] = g.test_items[ep[u]:ep[u + 1]].tolist()
test_data = types.SimpleNamespace(user_items_dic=test_dic)
mask_items = g.train_items
ms = ["Recall(k=20)", "NDCG(k=20)"]
print(f"shape={args.shape} scale={args.scale} users={g.num_users} items={g.num_items} train={g.clicks.shape[0]} "
      f"test={g.test_items.size} d={d} N={N} epochs={args.epochs}", flush=True)


def evaluate(uw, iw, tag):
    e = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
    top = e.topk(20, mask_indptr=g.train_indptr, mask_items=mask_items)
    e.close()
    r = metrics.evaluate_topk(test_data, top, ms, quiet=True, by_user_id=True)
    print(f"  [{tag}] Recall@20={r[ms[0]]:.5f} NDCG@20={r[ms[1]]:.5f}", flush=True)
    return r


his = masks = None
if args.agg:
    his, masks = synthetic.make_history(g, 100, seed=2022)
for seed in [int(x) for x in args.seeds.split(",")]:
    uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
    w00 = (np.random.default_rng(seed).standard_normal((d, d)) * 0.01).astype(np.float32)
    for coh, upd in [(int(c), int(u)) for c in args.coherence.split(",") for u in args.update.split(",")]:
        for streams in [int(x) for x in args.streams.split(",")]:
            uw, iw = uw0.copy(), iw0.copy()
            agg_kw = dict(his=his, masks=masks, w0=w00.copy(), use_aggregator=True) if args.agg else {}
            tile_kw = dict(neg_sampler=1, tile_size=512, refresh_interval=8192) if args.tile else {}
            eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, **agg_kw, **tile_kw, coherence=coh, num_streams=streams,
                             clip_val=args.clip, l_r=args.lr, update_mode=upd,
                             flags=abi.FLAG_LAZY_SYNC | (abi.FLAG_SAMPLING_CALL if args.tile else 0))
            t0 = time.time()
            losses = [eng.train_one_epoch() for _ in range(args.epochs)]
            dt = time.time() - t0
            eng.sync_to_host()
            kms, kn = eng.kernel_time()
            print(f"  kernel {kms / max(kn, 1):.3f} ms/epoch")
            print(f"GPU seed={seed} coherence={coh} update={upd} {eng.kernel_name} streams={streams}: losses={[round(x, 4) for x in losses]} ({dt:.2f}s)", flush=True)
            eng.close()
            evaluate(uw, iw, f"gpu coh={coh} upd={upd} streams={streams}")
    if args.oracle_seeds and seed not in [int(x) for x in args.oracle_seeds.split(",")]:
        continue
    for th in [int(x) for x in args.oracle_threads.split(",") if x]:
        uo, io = uw0.copy(), iw0.copy()
        agg_kw = dict(his=his, masks=masks, w0=w00.copy(), use_aggregator=True) if args.agg else {}
        tile_kw = dict(neg_sampler=1, tile_size=512, refresh_interval=8192) if args.tile else {}
        ora = orc.Engine(g.clicks, uo, io, num_negs=N, clip_val=args.clip, l_r=args.lr, **agg_kw, **tile_kw)
        t0 = time.time()
        losses = [ora.train_one_epoch(num_threads=th, sampler_call=1 if args.tile else 0) for _ in range(args.epochs)]
        dt = time.time() - t0
        print(f"ORACLE seed={seed} threads={th}: losses={[round(x, 4) for x in losses]} ({dt:.1f}s, "
              f"{g.clicks.shape[0] * args.epochs / dt / 1e3:.1f} k samples/s)", flush=True)
        evaluate(uo, io, f"oracle threads={th}")
