"""Evidence for DESIGN.md section 5 (run on the GPU box): Recall@20 / NDCG@20 of ONE graph trained as `world` user
shards (real HIP engines taking turns on one GPU, tests/shard_sim.py) against single-engine training and the oracle."""
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
from tests.shard_sim import train_sharded

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="amazonbooks")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--clusters", type=int, default=0)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--world", type=str, default="8")
ap.add_argument("--windows", type=str, default="1")
ap.add_argument("--overlap", type=str, default="0")
ap.add_argument("--streams", type=str, default="0", help="streams per shard engine (0 = the engine's own plan)")
ap.add_argument("--exchange-every", type=int, default=1, help="item-table exchange only every so many epochs")
ap.add_argument("--defer-final", type=int, default=0, help="1: the closing exchange of an epoch overlaps the next epoch (bench.py's steady state)")
ap.add_argument("--clip", type=float, default=1.0)
ap.add_argument("--seeds", type=str, default="2022")
ap.add_argument("--oracle-runs", type=int, default=1)
args = ap.parse_args()

g, d, N = synthetic.make_named(args.shape, scale=args.scale, n_clusters=args.clusters)
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                             for u in range(g.num_users) if ep[u + 1] > ep[u]})
ms = ["Recall(k=20)", "NDCG(k=20)"]
print(f"shape={args.shape} users={g.num_users} items={g.num_items} train={g.clicks.shape[0]} d={d} N={N} epochs={args.epochs}", flush=True)


def evaluate(uw, iw, tag):
    e = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
    top = e.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
    e.close()
    r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
    print(f"  [{tag}] Recall@20={r[ms[0]]:.5f} NDCG@20={r[ms[1]]:.5f}", flush=True)


for seed in [int(x) for x in args.seeds.split(",")]:
    uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
    uw, iw = uw0.copy(), iw0.copy()
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, clip_val=args.clip, flags=abi.FLAG_LAZY_SYNC)
    losses = [eng.train_one_epoch() for _ in range(args.epochs)]
    eng.sync_to_host()
    print(f"SINGLE seed={seed} {eng.kernel_name}: losses={[round(x, 4) for x in losses]}", flush=True)
    eng.close()
    evaluate(uw, iw, "single engine")
    for world in [int(x) for x in args.world.split(",")]:
        for streams in [int(x) for x in args.streams.split(",")]:
            for windows in [int(x) for x in args.windows.split(",")]:
                for overlap in [int(x) for x in args.overlap.split(",")]:
                    t0 = time.time()
                    su, si, sl, name = train_sharded(g, uw0, iw0, num_negs=N, world=world, epochs=args.epochs,
                                                     windows_per_epoch=windows, overlap=bool(overlap), seed=seed,
                                                     clip_val=args.clip, num_streams=streams, exchange_every=args.exchange_every,
                                                     defer_final=bool(args.defer_final))
                    print(f"SHARDED seed={seed} world={world} windows/epoch={windows} overlap={overlap} {name}: "
                          f"losses={[round(x, 4) for x in sl]} ({time.time() - t0:.1f}s)", flush=True)
                    evaluate(su, si, f"world={world} streams={streams} windows={windows} overlap={overlap}")
    for k in range(args.oracle_runs):
        uo, io = uw0.copy(), iw0.copy()
        ora = orc.Engine(g.clicks, uo, io, num_negs=N, clip_val=args.clip)
        losses = [ora.train_one_epoch(num_threads=8) for _ in range(args.epochs)]
        print(f"ORACLE seed={seed} run={k}: losses={[round(x, 4) for x in losses]}", flush=True)
        evaluate(uo, io, "oracle threads=8")
