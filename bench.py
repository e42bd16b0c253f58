#!/usr/bin/env python3
"""bench.py — positive-samples/sec of the SimpleX/CCL training hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass (one training epoch: LR step, all interactions, zero_grad) over a seeded synthetic graph of the
AmazonBooks shape (52 643 users x 91 599 items, 2 380 730 interactions, d=64, 16 uniform negatives drawn on the
GPU): BASELINE.json configs[1].  With N GPUs every rank owns one such user shard (weak scaling: the job has
N x 52 643 users), the item table is replicated and synchronised with an RCCL all-reduce (heat_amd.cf.distributed).
Tables and the interaction list are resident in HBM before the timed region (torch tensors handed to the C ABI as raw
device pointers).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, further limited by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(graph, d, n_negs, threads=8, epochs=6):
    """The CPU oracle (a port of the reference's OpenMP path; the reference itself cannot be built here: Eigen is
    absent) timed on this box's host cores over a bounded sample: `epochs` passes of the same graph (~10-15 s of CPU
    work).  Checker code, measured — never shipped."""
    from heat_amd.cf import synthetic
    from oracle import cf_oracle as orc
    uw, iw = synthetic.init_embeddings(graph.num_users, graph.num_items, d, seed=2022)
    ora = orc.Engine(graph.clicks, uw, iw, num_negs=n_negs)
    n = graph.clicks.shape[0]
    cores = min(threads, os.cpu_count() or threads)
    t0 = time.perf_counter()
    for _ in range(epochs):
        ora.train_one_epoch(num_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n * epochs / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{epochs} epochs x {n} interactions of the same AmazonBooks-shaped graph, OpenMP "
                      f"schedule(dynamic,512), {cores} threads (README.md:94 ran max_threads 8), {dt:.1f} s; "
                      f"host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default="amazonbooks")
    ap.add_argument("--update-mode", type=int, default=0, help="HEAT_CF_UPDATE_* (0 = engine default: AUTO)")
    ap.add_argument("--sync-interactions", type=int, default=0,
                    help="interactions per GPU between item-table all-reduces (0 = streams x refresh_interval, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--interactions", type=int, default=0,
                    help="interactions per step for --shape synthetic_hbm (0 = 20 000 000: a sample of the 200 M list)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from heat_amd import abi
    from heat_amd.cf import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback)")
    abi.load()
    # HEAT_BENCH_BACKEND=gloo rehearses the N>1 code path with several ranks on ONE GPU (RCCL refuses two ranks on a
    # device; gloo stages the all-reduce through the host).  Its numbers mean nothing; it only has to run through.
    backend = os.environ.get("HEAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # one side stream carries everything: engine kernels, the torch element-wise ops of the item sync and (through
    # torch.distributed's stream hand-off) the RCCL all-reduce are ordered with respect to each other
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    U, I, T, d, N = synthetic.SHAPES[args.shape]
    graph = None
    if args.shape == "synthetic_hbm":
        # BASELINE.json configs[4]: 10 M users x 1 M items, d=256, negs=100 (tables 22.5 GB: the true HBM-resident run).
        # A step walks a 20 M-interaction sample of the 200 M list (a full pass is 83.6 TB of algorithmic traffic).
        T = args.interactions or 20_000_000
        clicks = synthetic.make_clicks_torch(U, I, T, dev, seed=2022 + rank)
        g = torch.Generator(device=dev)
        g.manual_seed(2022)
        user_w = torch.empty((U, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
        item_w = torch.empty((I, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
        args.no_cpu_baseline = True
    else:
        # every rank generates its own shard (different seed = different users), same item id space
        graph = synthetic.make_graph(U, I, T, seed=2022 + rank, with_test=False)
        uw_h, iw_h = synthetic.init_embeddings(U, I, d, seed=2022)
        clicks = torch.from_numpy(graph.clicks.view(np.int64)).to(dev)
        user_w = torch.from_numpy(uw_h).to(dev)
        item_w = torch.from_numpy(iw_h).to(dev)       # identical on every rank (replicated table)
    stream = torch.cuda.current_stream().cuda_stream
    eng = abi.Engine.from_device(clicks.data_ptr(), T, user_w.data_ptr(), item_w.data_ptr(), num_users=U, num_items=I,
                                 emb_dim=d, num_negs=N, stream=stream, keep=(clicks, user_w, item_w), seed=2022,
                                 sample_index_base=rank * T, update_mode=args.update_mode, device=local_rank)
    trainer = None
    if world > 1 or os.environ.get("HEAT_BENCH_FORCE_SYNC"):   # the env switch exercises the N>1 code path on one GPU
        from heat_amd.cf.distributed import ItemSync
        trainer = ItemSync(eng, item_w, world, refresh_interval=8192, sync_interactions=args.sync_interactions)

    def step():
        if trainer is None:
            eng.begin_epoch()
            eng.train_range(0, T, want_loss=False)
            eng.end_epoch()
        else:
            trainer.train_one_epoch()

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = eng.kernel_time()

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc_path) and args.shape == "amazonbooks" and args.update_mode == 0:
        # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
        # (tools/pmc_traffic.py; counters cannot be collected from inside the process being timed)
        with open(pmc_path) as f:
            pmc = json.load(f)
        if eng.kernel_name.startswith(pmc.get("kernel", "?").rstrip(">")):
            traffic = pmc["traffic_bytes_per_launch"] / 1e9

    if rank == 0:
        total = world * T * args.steps
        B = 16 * d * (N + 2) + 16                 # algorithmic bytes per interaction
        per_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        inter_per_launch = T * args.steps / max(launches, 1)
        achieved = B * inter_per_launch / per_launch_s / 1e9
        out = {
            "metric": "positive-samples/sec/node (AmazonBooks d=64, negs=16)" if args.shape == "amazonbooks"
            else f"positive-samples/sec/node ({args.shape} d={d}, negs={N})",
            "value": total / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.shape}-shaped synthetic graph per GPU: {U} users x {I} items, {T} interactions, "
                                   f"d={d}, negs={N}, uniform on-GPU Philox sampler; 1 step = 1 epoch",
                       "kernel": eng.kernel_name,
                       "item_sync": None if trainer is None else trainer.describe()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "GB per launch (PMC)",
                         "algorithmic_gb_per_launch": B * inter_per_launch / 1e9,
                         "bytes_per_interaction": B, "kernel_ms_per_launch": per_launch_s * 1e3,
                         "launches": launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(graph, d, N)
            ncpu = usable_cpus()
            if ncpu > 8:   # also at all the cores this process may use
                out["cpu_baseline_all_cores"] = cpu_baseline(graph, d, N, threads=ncpu, epochs=4)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
