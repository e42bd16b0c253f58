"""Analysis aid (CPU only, uses the oracle): loss curve and Recall@20 / NDCG@20 of the sequentially-consistent stream model
in stream_sim.c for a list of (streams, layout) pairs, each in its own process.
  python tests/tools/stream_sim.py --agg --configs 1:0,8:0,80:0,438:0,8:1,438:1 --epochs 5"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time
import types
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from heat_amd.cf import metrics, synthetic  # noqa: E402
from oracle import cf_oracle as orc  # noqa: E402

SO = "/tmp/libstream_sim.so"


def build():
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-std=c11", "-ffp-contract=off", "-shared",
                           "-o", SO, os.path.join(ROOT, "tests/tools/stream_sim.c"), "-L" + os.path.join(ROOT, "oracle"),
                           "-lcf_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])


def topk_cpu(uw, iw, g, k=20, block=1024):
    tp = g.train_indptr.astype(np.int64)
    ti = g.train_items.astype(np.int64)
    out = np.empty((g.num_users, k), dtype=np.uint32)
    for u0 in range(0, g.num_users, block):
        u1 = min(g.num_users, u0 + block)
        sim = uw[u0:u1] @ iw.T
        rows = np.repeat(np.arange(u1 - u0), np.diff(tp[u0:u1 + 1]))
        sim[rows, ti[tp[u0]:tp[u1]]] = -np.inf
        part = np.argpartition(-sim, k, axis=1)[:, :k]
        order = np.argsort(-np.take_along_axis(sim, part, axis=1), axis=1, kind="stable")
        out[u0:u1] = np.take_along_axis(part, order, axis=1)
    return out


def run(cfg):
    args, S, layout, mb = cfg
    os.environ["OMP_NUM_THREADS"] = "1"
    if args.interactions:
        _U, _I, _T, d, N = synthetic.SHAPES[args.shape]
        g = synthetic.make_graph(args.users or _U, args.items or _I, args.interactions, seed=2022, n_clusters=args.clusters)
    else:
        g, d, N = synthetic.make_named(args.shape, scale=args.scale, n_clusters=args.clusters)
    uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=args.seed)
    kw = {}
    if args.agg:
        his, masks = synthetic.make_history(g, 100, seed=2022)
        w0 = (np.random.default_rng(args.seed).standard_normal((d, d)) * 0.01).astype(np.float32)
        kw = dict(his=his, masks=masks, w0=w0, use_aggregator=True)
    if args.tile:
        kw.update(neg_sampler=1, tile_size=512, refresh_interval=8192)
    e = orc.Engine(g.clicks, uw, iw, num_negs=N, clip_val=args.clip, l_r=args.lr, **kw)
    lib = C.CDLL(SO)
    lib.sim_epoch.restype = C.c_double
    lib.sim_epoch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    t0 = time.time()
    losses = []
    for _ in range(args.epochs):
        if S < 0:   # the oracle's own OpenMP epoch with -S threads
            os.environ["OMP_NUM_THREADS"] = str(-S)
            losses.append(e.train_one_epoch(num_threads=-S, sampler_call=int(args.tile)))
        else:
            losses.append(lib.sim_epoch(C.cast(e._e, C.c_void_p), S, layout, args.chunk, args.seed, mb, (2 if args.unique_sampler_seeds else 1) if args.tile else 0))
    dt = time.time() - t0
    test_dic = {}
    ep = g.test_indptr.astype(np.int64)
    for u in range(g.num_users):
        if ep[u + 1] > ep[u]:
            test_dic[u] = g.test_items[ep[u]:ep[u + 1]].tolist()
    top = topk_cpu(uw, iw, g)
    ms = ["Recall(k=20)", "NDCG(k=20)"]
    r = metrics.evaluate_topk(types.SimpleNamespace(user_items_dic=test_dic), top, ms, quiet=True, by_user_id=True)
    if args.json:
        return dict(shape=args.shape, scale=args.scale, aggregator=bool(args.agg), tile_sampler=bool(args.tile), workers=S, layout="slices" if layout == 0 else "sweep",
                    w0_batch=mb, seed=args.seed, epochs=args.epochs, clip=args.clip, lr=args.lr,
                    losses=[float(x) for x in losses], recall20=float(r[ms[0]]), ndcg20=float(r[ms[1]]))
    return (f"streams={S} layout={'slices' if layout == 0 else 'sweep'} w0_batch={mb} losses={[round(x, 4) for x in losses]} "
            f"Recall@20={r[ms[0]]:.5f} NDCG@20={r[ms[1]]:.5f} ({dt:.0f}s)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="amazonbooks")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--clusters", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--users", type=int, default=0)
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--interactions", type=int, default=0, help="with --users / --items: a graph of another size with the shape's emb_dim / num_negs")
    ap.add_argument("--agg", action="store_true")
    ap.add_argument("--unique-sampler-seeds", action="store_true", help="with --tile: every (epoch, worker) its own sampler seed instead of "
                    "the reference's (epoch + 1) * worker id (train/engine.cpp:302), which repeats seeds across epochs and workers")
    ap.add_argument("--tile", action="store_true", help="random-tile sampler (tile 512, refresh 8192), one per worker, sampling() call")
    ap.add_argument("--clip", type=float, default=1.0)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--configs", default="1:0,8:0,80:0,438:0,8:1,438:1", help="streams:layout[:w0_batch], negative streams = oracle OpenMP threads")
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--json", default="", help="write the results as a JSON list (the committed fixture tests/golden/accl_stream_model.json)")
    ap.add_argument("--seeds", default="", help="comma-separated: every config once per seed (overrides --seed)")
    a = ap.parse_args()
    build()
    import copy
    import json
    cfgs = []
    for seed in ([int(x) for x in a.seeds.split(",")] if a.seeds else [a.seed]):
        b = copy.copy(a)
        b.seed = seed
        cfgs += [(b, int(c.split(":")[0]), int(c.split(":")[1]), int((c.split(":") + ["32"])[2])) for c in a.configs.split(",")]
    out = []
    with ProcessPoolExecutor(a.procs) as ex:
        for line in ex.map(run, cfgs):
            print(line, flush=True)
            out.append(line)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
