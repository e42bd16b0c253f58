#!/usr/bin/env python3
"""bench.py — positive-samples/sec of the SimpleX/CCL training hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass (one training epoch: LR step, all interactions, zero_grad) over ONE seeded synthetic graph of the
AmazonBooks shape (52 643 users x 91 599 items, 2 380 730 interactions, d=64, 16 uniform negatives drawn on the GPU).

N = 1  : BASELINE.json configs[1].  Tables and the interaction list are resident in HBM before the timed region (torch
         tensors handed to the C ABI as raw device pointers).  The JSON line also carries the read-only roofline
         fraction, where the working set lives, a second roofline object from a short HBM-resident run (configs[4]
         shape, a sample of its interaction list), the same epoch through the reference's own boundary (`cf_c`, host
         buffers written back every epoch: `value_host_mode`) and the CPU oracle timed on this box's host cores.
N > 1  : BASELINE.json configs[3]: the SAME graph partitioned by user range (cf/main.py:51-57; ranges cut at equal
         interaction counts), one shard per rank, item table replicated, item-table deltas all-reduced over RCCL/xGMI
         (heat_amd.cf.distributed.ItemSync), one exchange per epoch (one every two epochs from 8 GPUs on: the exchange
         window stays at about half a million interactions per GPU; Recall-validated).  Two legs over the
         same K steps: every exchange completed before training goes on (`item_sync_blocking`), then the all-reduce
         overlapped with the next window — `value` is the second when it finishes (a watchdog falls back to the first).
         `value` = the graph's interactions x steps / time: strong scaling.  Extra keys: the direct exchange, the pipelined
         form, one exchange per epoch, the literal "all-reduce every 8192 steps" window, a short weak-scaling leg.
Rank 0 prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
INFINITY_CACHE_BYTES = 256 << 20
# bare random-row read-modify-write loop (tools/ceilings.hip, 256-B rows): profiles/r01_memory_ceilings.txt; re-run in round 2
# (profiles/r02_memory_ceilings.txt) the Infinity-Cache figure is 8.1-8.5 TB/s, the HBM-resident one moves 5.2-6.3 TB/s run to run
MEASURED_RMW_CEILING_GBS = {"infinity-cache": 8301.9, "hbm": 6185.2}


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


def roofline(B, B_rd, interactions, kernel_ms, launches, working_set_bytes, traffic=None, traffic_source=None):
    """Algorithmic bytes (B = 16 d (N+2) + 16 per interaction; B_rd = the gather half) over the training
    kernel's HIP-event time, against the HBM peak.  `residency` says where the working set actually lives: a set that
    fits the 256 MiB Infinity Cache is served from it, and the fraction is then a bandwidth figure quoted against the
    HBM peak, not DRAM traffic."""
    per_launch_s = kernel_ms * 1e-3 / max(launches, 1)
    per_launch = interactions / max(launches, 1)
    achieved = B * per_launch / per_launch_s / 1e9
    resident = working_set_bytes <= INFINITY_CACHE_BYTES
    # what a bare random-row read-modify-write loop reaches on this memory system (tools/ceilings.hip, measured in round 1,
    # profiles/r01_memory_ceilings.txt): context for `frac`, which stays quoted against the nominal HBM peak
    ceiling = MEASURED_RMW_CEILING_GBS["infinity-cache" if resident else "hbm"]
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "read_frac": B_rd * per_launch / per_launch_s / 1e9 / HBM_PEAK_GBS,
            "measured_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling,
            "measured_ceiling_source": "profiles/r02_memory_ceilings.txt (first measured in r01_memory_ceilings.txt): bare random-row "
                                       "read-modify-write loop, same residency; band across runs 8.1-8.5 TB/s (Infinity Cache), "
                                       "5.2-6.3 TB/s (HBM)",
            "traffic": traffic, "traffic_unit": "GB per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_source,
            "residency": "infinity-cache" if resident else "hbm",
            "working_set_mb": working_set_bytes / 1e6,
            "algorithmic_gb_per_launch": B * per_launch / 1e9, "bytes_per_interaction": B,
            "read_bytes_per_interaction": B_rd, "kernel_ms_per_launch": per_launch_s * 1e3, "launches": launches}


def replayed_traffic(name, kernel_name):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
    (tools/pmc_traffic.py: counters cannot be collected from inside the process being timed) — replayed, and labelled so."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        pmc = json.load(f)
    if not kernel_name.startswith(pmc.get("kernel", "?").rstrip(">")):
        return None, None
    return pmc["traffic_bytes_per_launch"] / 1e9, "replayed:profiles/" + name


def hbm_resident_leg(dev, stream, interactions, steps=2):
    """BASELINE.json configs[4] shape (10 M users x 1 M items, d=256, negs=100: 22.5 GB of tables, HBM-resident) on a
    sample of its 200 M-interaction list; returns a roofline object for the same kernel family."""
    import torch
    from heat_amd import abi
    from heat_amd.cf import synthetic
    U, I, _, d, N = synthetic.SHAPES["synthetic_hbm"]
    T = interactions
    clicks = synthetic.make_clicks_torch(U, I, T, dev, seed=2022, per_user=20)      # 200 M / 10 M interactions per user
    g = torch.Generator(device=dev)
    g.manual_seed(2022)
    user_w = torch.empty((U, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
    item_w = torch.empty((I, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
    eng = abi.Engine.from_device(clicks.data_ptr(), T, user_w.data_ptr(), item_w.data_ptr(), num_users=U, num_items=I,
                                 emb_dim=d, num_negs=N, stream=stream, keep=(clicks, user_w, item_w), seed=2022)
    for k in range(1 + steps):
        if k == 1:
            torch.cuda.synchronize()
            eng.kernel_time(reset=True)
            t0 = time.perf_counter()
        eng.begin_epoch()
        eng.train_range(0, T, want_loss=False)
        eng.end_epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_time()
    B, B_rd = 16 * d * (N + 2) + 16, 8 * d * (N + 2) + 16
    traffic, src = replayed_traffic("r03_pmc_traffic_hbm.json", eng.kernel_name)
    out = roofline(B, B_rd, T * steps, kernel_ms, launches, 2 * (U + I) * d * 4, traffic, src)
    out["workload"] = (f"synthetic 10M x 1M, d={d}, negs={N}: {T} interactions per launch = {T // 20} users spread over the whole "
                       f"table x 20 interactions each, a sample of the 200M list (a full pass is 83.6 TB of algorithmic traffic)")
    out["kernel"] = eng.kernel_name
    out["samples_per_s"] = T * steps / dt
    eng.close()
    del clicks, user_w, item_w
    torch.cuda.empty_cache()
    return out


class stdout_to_stderr:
    """cf_c mirrors the reference's constructor print (cf_config.hpp:19); rank 0's stdout must stay ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def tile_sampler_leg(clicks_t, n, uw_h, iw_h, dev, stream, U, I, d, N, steps):
    """the random-tile sampler (neg_sampler 1, tile 512, refresh 8192; its sampling() call) with the tile's
    weight deltas resident in LDS — the same graph and tables as the headline, another sampling distribution."""
    import torch
    from heat_amd import abi
    user_w = torch.from_numpy(uw_h).to(dev)
    item_w = torch.from_numpy(iw_h).to(dev)
    out = {}
    for name, extra in (("in_lds", abi.FLAG_TILE_LDS), ("in_global_memory", 0)):
        eng = abi.Engine.from_device(clicks_t.data_ptr(), n, user_w.data_ptr(), item_w.data_ptr(), num_users=U, num_items=I,
                                     emb_dim=d, num_negs=N, stream=stream, keep=(clicks_t, user_w, item_w), seed=2022,
                                     neg_sampler=1, tile_size=512, refresh_interval=8192, flags=abi.FLAG_SAMPLING_CALL | extra)
        for k in range(1 + steps):
            if k == 1:
                torch.cuda.synchronize()
                eng.kernel_time(reset=True)
                t0 = time.perf_counter()
            eng.begin_epoch()
            eng.train_range(0, n, want_loss=False)
            eng.end_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kernel_ms, launches = eng.kernel_time()
        out[name] = {"value": n * steps / dt, "unit": "samples/s", "kernel_ms_per_launch": kernel_ms / max(launches, 1),
                     "kernel": eng.kernel_name}
        eng.close()
    out["note"] = ("random-tile sampler (tile 512, refresh 8192, sampling() call): tile deltas resident in LDS (opt-in, 12 streams "
                   "per workgroup, flushed by float atomics at the end of the launch) vs every step written to the table (default)")
    return out


def accl_leg(graph, d, N, steps):
    """the same graph trained with behaviour aggregation (ACCL) on the GPU — histories of up to 100
    items per user, W0 d x d; algorithmic bytes add the history rows an interaction reads (W0 lives in LDS)."""
    from heat_amd import abi
    from heat_amd.cf import synthetic
    his, masks = synthetic.make_history(graph, 100, seed=2022)
    uw, iw = synthetic.init_embeddings(graph.num_users, graph.num_items, d, seed=2022)
    w0 = (np.random.default_rng(2022).standard_normal((d, d)) * 0.01).astype(np.float32)
    eng = abi.Engine(graph.clicks, uw, iw, num_negs=N, seed=2022, his=his, masks=masks, w0=w0, use_aggregator=True,
                     flags=abi.FLAG_LAZY_SYNC)
    n = graph.clicks.shape[0]
    eng.train_one_epoch()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_one_epoch()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_time()
    mean_h = float(masks[graph.clicks[:, 0].astype(np.int64)].mean())        # history rows read per interaction
    B = 16 * d * (N + 2) + 16 + int(round(mean_h * 4 * d))
    per_launch_s = kernel_ms * 1e-3 / max(launches, 1)
    out = {"value": n * steps / dt, "unit": "samples/s", "kernel": eng.kernel_name, "kernel_ms_per_launch": per_launch_s * 1e3,
           "mean_history_rows_per_interaction": mean_h, "bytes_per_interaction": B,
           "achieved": B * n / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "frac": B * n / per_launch_s / 1e9 / HBM_PEAK_GBS,
           "note": "behaviour aggregation (ACCL): stream count held by the Recall/NDCG bound of DESIGN.md section 3, not by the chip"}
    eng.close()
    return out


def host_mode_leg(graph, d, N, steps, pinned):
    """The same epoch through the reference's own boundary: `cf_c` objects on host numpy buffers, trained in place and
    written back after every epoch (init_modules.cpp:79-81 contract).  pinned=True: the weight arrays live in page-locked
    memory (what heat_amd.cf.frontend allocates), so the write-back is one DMA at PCIe rate."""
    import torch
    from heat_amd import cf_c
    from heat_amd.cf import synthetic
    U, I, T = graph.num_users, graph.num_items, graph.clicks.shape[0]
    uw, iw = synthetic.init_embeddings(U, I, d, seed=2022)
    if pinned:
        keep = (torch.from_numpy(uw).pin_memory(), torch.from_numpy(iw).pin_memory())
        uw, iw = keep[0].numpy(), keep[1].numpy()
    cfg = cf_c.modules.CFConfig(emb_dim=d, num_negs=N, num_users=U, num_items=I, train_size=T, neg_sampler=0, tile_size=512,
                                refresh_interval=8192, num_subepoches=2, l2=1e-7, clip_val=1.0, milestones=[10], l_r=0.01)
    his = np.zeros((U, 1), dtype=np.uint64)
    masks = np.zeros((U, 1), dtype=np.uint64)
    ds = cf_c.modules.datasets.ClickDataset(click_dataset=graph.clicks, historical_items=his, masks=masks)
    model = cf_c.modules.models.MatrixFactorization(cf_config=cfg, user_weights=uw, item_weights=iw)
    w0 = np.zeros((d, d), dtype=np.float32)
    agg = cf_c.modules.behavior_aggregators.AggregatorWeights(aggregator_weights0=w0)
    eng = cf_c.modules.train.Engine(dataset=ds, aggregator_weights=agg, model=model, cf_config=cfg)
    eng.train_one_epoch()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_one_epoch()
    dt = time.perf_counter() - t0
    return T * steps / dt


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv, runner=None):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1), relay the ONE
    JSON line rank 0 prints and return the children's exit code.  `runner` (tests) replaces subprocess.run."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    res = (runner or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in (res.stdout or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in (res.stdout or "").splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)                       # whatever else the ranks wrote to stdout is not the result
    if lines:
        print(lines[-1], flush=True)
    if res.returncode == 0 and not lines:
        print("bench.py launcher: the ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return res.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default="amazonbooks")
    ap.add_argument("--update-mode", type=int, default=0, help="HEAT_CF_UPDATE_* (0 = engine default: AUTO)")
    ap.add_argument("--num-streams", type=int, default=0)
    ap.add_argument("--windows", type=int, default=0, help="N>1: item-table exchanges per epoch (0 = default: 1, see DESIGN.md section 5)")
    ap.add_argument("--epochs-per-exchange", type=int, default=0, help="N>1: whole epochs between item-table exchanges (0 = default: 1, from 8 GPUs on 2)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: complete every exchange before the next window")
    ap.add_argument("--collective", default=os.environ.get("HEAT_BENCH_COLLECTIVE", "all_reduce"), choices=("all_reduce", "direct"),
                    help="N>1: how the item-table deltas are summed over the ranks (heat_amd.cf.distributed.ItemSync)")
    ap.add_argument("--balance", default="interactions", choices=("interactions", "users"),
                    help="N>1: cut the user ranges at equal interaction counts (default) or equal user counts (cf/main.py:51-57)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the HBM-resident, host-mode, literal-window and weak-scaling legs")
    ap.add_argument("--interactions", type=int, default=0,
                    help="interactions per step for --shape synthetic_hbm (0 = 20 000 000: a sample of the 200 M list)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` by itself: this process becomes the launcher (cf/main.py:47-70 is started by mpirun; here
        # one rank per GPU under torch.distributed.run).  Nothing in this process has touched torch or HIP yet, and it
        # never will: the ranks are CHILDREN, their rank 0's JSON line is relayed, their exit code is this one's.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from heat_amd import abi
    from heat_amd.cf import synthetic
    from heat_amd.cf.distributed import ItemSync, shard_bounds, shard_bounds_balanced, shard_clicks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` by itself or under "
                 f"torch.distributed.run --nproc-per-node N")
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
    # RCCL prints a version banner on stdout when its communicator comes up: keep stdout for the ONE JSON line
    with stdout_to_stderr():
        if world > 1 or "RANK" in os.environ:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
        elif os.environ.get("HEAT_BENCH_FORCE_SYNC"):      # one-rank group: the N>1 exchange path on a one-GPU box
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if dist.is_initialized():
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()

    # one side stream carries the engine kernels and the fused delta / apply passes of the item exchange; the RCCL
    # all-reduce runs on torch.distributed's own stream, ordered against this one by events (async collective)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream().cuda_stream
    U, I, T, d, N = synthetic.SHAPES[args.shape]
    B, B_rd = 16 * d * (N + 2) + 16, 8 * d * (N + 2) + 16          # algorithmic bytes per interaction
    force_sync = bool(os.environ.get("HEAT_BENCH_FORCE_SYNC"))     # exercises the N>1 exchange path on one GPU
    graph = None

    def build(clicks_np, n_users, user_rows, base, seed_graph_note):
        clicks = torch.from_numpy(clicks_np.view(np.int64)).to(dev)
        user_w = torch.from_numpy(np.ascontiguousarray(user_rows)).to(dev)
        item_w = torch.from_numpy(iw_h).to(dev)                   # identical on every rank (replicated table)
        eng = abi.Engine.from_device(clicks.data_ptr(), clicks_np.shape[0], user_w.data_ptr(), item_w.data_ptr(),
                                     num_users=n_users, num_items=I, emb_dim=d, num_negs=N, stream=stream,
                                     keep=(clicks, user_w, item_w), seed=2022, sample_index_base=base,
                                     update_mode=args.update_mode, num_streams=args.num_streams, device=local_rank)
        return eng, item_w

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, finish=None):
        for _ in range(warmup):
            step()
        if finish:
            finish()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish:
            finish()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    extra = {}
    if args.shape == "synthetic_hbm":
        # BASELINE.json configs[4] as the main workload (profiling runs): every rank its own sample of the list
        T = args.interactions or 20_000_000
        clicks = synthetic.make_clicks_torch(U, I, T, dev, seed=2022 + rank, per_user=20)
        g = torch.Generator(device=dev)
        g.manual_seed(2022)
        user_w = torch.empty((U, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
        item_w = torch.empty((I, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=g)
        eng = abi.Engine.from_device(clicks.data_ptr(), T, user_w.data_ptr(), item_w.data_ptr(), num_users=U, num_items=I,
                                     emb_dim=d, num_negs=N, stream=stream, keep=(clicks, user_w, item_w), seed=2022,
                                     sample_index_base=rank * T, update_mode=args.update_mode,
                                     num_streams=args.num_streams, device=local_rank)
        args.no_cpu_baseline = args.no_extra_legs = True
        my_T, total_T, scaling = T, world * T, "weak"
        workload = (f"synthetic 10M x 1M shape per GPU: {U} users x {I} items, d={d}, negs={N}, {T} interactions per step "
                    f"(a sample of the 200 M list), uniform on-GPU Philox sampler")
        working_set = 2 * (U + I) * d * 4
        trainer = None
    else:
        graph = synthetic.make_graph(U, I, T, seed=2022, with_test=False)       # ONE graph, the same on every rank
        uw_h, iw_h = synthetic.init_embeddings(U, I, d, seed=2022)
        by_users = [shard_bounds(U, world, r) for r in range(world)]            # cf/main.py:51-57 (rank 0 fixed)
        by_work = [shard_bounds_balanced(graph.train_indptr, world, r) for r in range(world)]
        # The slowest rank sets the epoch: ranges are cut at equal INTERACTION counts ("build may balance ranges
        # by interaction prefix-sum"), --balance users restores the reference's equal user counts
        all_bounds = by_users if args.balance == "users" else by_work
        lo, hi = all_bounds[rank]
        shard, lo, hi = shard_clicks(graph.clicks, U, world, rank, bounds=(lo, hi))
        base = int(np.searchsorted(graph.clicks[:, 0], lo, side="left"))
        eng, item_w = build(shard, hi - lo, uw_h[lo:hi], base, None)
        my_T, total_T, scaling = shard.shape[0], T, ("strong" if world > 1 else "n/a")
        workload = (f"{args.shape}-shaped synthetic graph: {U} users x {I} items, {T} interactions, d={d}, negs={N}, uniform "
                    f"on-GPU Philox sampler; 1 step = 1 epoch" +
                    (f"; user rows partitioned over {world} GPUs (cf/main.py:51-57; ranges cut at equal "
                     f"{'user' if args.balance == 'users' else 'interaction'} counts), item table replicated" if world > 1 else ""))
        working_set = 2 * ((hi - lo) + I) * d * 4
        if world > 1:
            tp = graph.train_indptr.astype(np.int64)
            extra["shards"] = {"balance": args.balance,
                               "interactions_per_rank": [int(tp[b] - tp[a]) for a, b in all_bounds],
                               "users_per_rank": [int(b - a) for a, b in all_bounds],
                               "interactions_per_rank_if_cut_by_user_count": [int(tp[b] - tp[a]) for a, b in by_users]}

    def step_plain():
        eng.begin_epoch()
        eng.train_range(0, my_T, want_loss=False)
        eng.end_epoch()

    # Exchange schedule (a measured choice): ONE exchange per epoch up to 4 GPUs, one every TWO epochs from
    # 8 GPUs on — the exchange window is held at about half a million interactions per GPU (AmazonBooks shape: 595 k at 4 GPUs,
    # 2 x 298 k at 8); every schedule bench.py can run holds Recall@20 / NDCG@20 within 1e-3 of single-engine training
    # (tests/test_gpu_parity.py::test_eight_user_shards_match_single_engine_recall_ndcg)
    epe = args.epochs_per_exchange or (2 if world >= 8 else 1)

    def sync_kwargs(overlap, **over):
        kw = dict(windows_per_epoch=args.windows or 1, mode="sum", force_collective=force_sync, overlap=overlap,
                  defer_final=overlap, collective=args.collective, epochs_per_exchange=epe, pipelined=False)
        kw.update(over)
        return kw

    def line(value_elapsed, steps, trainer, kernel_ms, launches, note=None):
        traffic, traffic_src = (None, None)
        if world == 1 and args.shape in ("amazonbooks", "yelp18") and args.update_mode == 0 and args.num_streams == 0:
            traffic, traffic_src = replayed_traffic("r03_pmc_traffic.json" if args.shape == "amazonbooks" else
                                                    "r03_pmc_traffic_yelp18.json", eng.kernel_name)
        out = {
            "metric": "positive-samples/sec/node (AmazonBooks d=64, negs=16)" if args.shape == "amazonbooks"
            else f"positive-samples/sec/node ({args.shape} d={d}, negs={N})",
            "value": total_T * steps / value_elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": value_elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "kernel": eng.kernel_name,
                       "item_sync": None if trainer is None else trainer.describe()},
            # rank 0's training kernel: its shard's interactions over its HIP-event time
            "roofline": roofline(B, B_rd, my_T * steps, kernel_ms, launches, working_set, traffic, traffic_src),
        }
        if note:
            out["note"] = note
        return out

    def guard(limit, fallback_line):
        """Watchdog for a leg built on collectives nobody has run on this node yet: if it is still going after `limit`
        seconds every rank leaves (same timer on every rank) and rank 0 prints what it already has."""
        import threading

        def bail():
            if rank == 0:
                out = fallback_line()
                out.update(extra)
                print(json.dumps(out), flush=True)
            os._exit(0)

        t = threading.Timer(limit, bail)
        t.daemon = True
        t.start()
        return t

    trainer = None
    if args.shape != "synthetic_hbm" and (world > 1 or force_sync):
        # N > 1, two legs over the same K steps.  First the schedule that completes every exchange before the next window
        # starts (blocking collective: nothing of it has ever been in doubt).  Then the default schedule — the all-reduce of
        # a window overlapped with the next window, also across the epoch boundary, the pipeline drained inside the timed
        # region — under a watchdog: should the overlapped collectives not finish on this node, every rank leaves and rank 0
        # reports the blocking leg instead of nothing.
        blocking = ItemSync(eng, item_w, world, **sync_kwargs(False))
        for _ in range(args.warmup):
            blocking.train_one_epoch()
        fence()
        eng.kernel_time(reset=True)
        el_b = timed(blocking.train_one_epoch, args.steps, 0, blocking.finalize if epe > 1 else None)
        kms_b, ln_b = eng.kernel_time()
        extra["item_sync_blocking"] = {"value": total_T * args.steps / el_b, "unit": "samples/s", "ms_per_step": el_b / args.steps * 1e3,
                                       "steps": args.steps, "item_sync": blocking.describe()}
        trainer, elapsed, kernel_ms, launches = blocking, el_b, kms_b, ln_b
        if not args.no_overlap:
            limit = float(os.environ.get("HEAT_BENCH_WATCHDOG_S", "0")) or max(90.0, 30.0 * el_b + 60.0)
            dog = guard(limit, lambda: line(el_b, args.steps, blocking, kms_b, ln_b,
                                            note=f"the overlapped schedule did not finish within {limit:.0f} s: `value` is "
                                                 f"the blocking schedule"))
            over = ItemSync(eng, item_w, world, **sync_kwargs(True))
            for _ in range(min(args.warmup, 2)):
                over.train_one_epoch()
            over.finalize()
            fence()
            eng.kernel_time(reset=True)
            el_o = timed(over.train_one_epoch, args.steps, 0, over.finalize)
            kms_o, ln_o = eng.kernel_time()
            dog.cancel()
            trainer, elapsed, kernel_ms, launches = over, el_o, kms_o, ln_o
    else:
        for _ in range(args.warmup):
            step_plain()
        fence()
        eng.kernel_time(reset=True)
        elapsed = timed(step_plain, args.steps, 0, None)
        kernel_ms, launches = eng.kernel_time()

    if world > 1 and not args.no_extra_legs and trainer is not None:
        head = (elapsed, trainer, kernel_ms, launches)
        dog = guard(float(os.environ.get("HEAT_BENCH_WATCHDOG_S", "0")) or 240.0,
                    lambda: line(head[0], args.steps, head[1], head[2], head[3], note="an extra leg did not finish: cut short"))
        # (a) the literal reading of configs[3]: one all-reduce every 8192 interactions per GPU
        lit = ItemSync(eng, item_w, world, sync_interactions=8192, mode="sum", overlap=not args.no_overlap,
                       defer_final=not args.no_overlap, pipelined=False)
        k = max(1, min(3, args.steps))
        el = timed(lit.train_one_epoch, k, 1, lit.finalize)
        extra["item_sync_every_8192"] = {"value": T * k / el, "unit": "samples/s", "ms_per_step": el / k * 1e3, "steps": k,
                                         "item_sync": lit.describe(), "exchanges_per_epoch_per_gpu": -(-(T // world) // 8192)}
        # (a') the headline schedule in its other forms, each over <= 10 steps: the direct exchange (slices scattered to their
        # owners, summed there, gathered back) instead of RCCL's all-reduce; the pipelined form (one pass on the training
        # stream, the rest on an exchange stream); one exchange per epoch where the headline runs one every two.  Which one the
        # xGMI links and this node prefer is measured here, not guessed.
        alt_name = "all_reduce" if args.collective == "direct" else "direct"
        legs = [("item_sync_" + alt_name, dict(collective=alt_name)), ("item_sync_pipelined", dict(pipelined=True))]
        if epe != 1:
            legs.append(("item_sync_every_epoch", dict(epochs_per_exchange=1)))
        for leg_name, over_kw in legs:
            try:
                alt = ItemSync(eng, item_w, world, **sync_kwargs(not args.no_overlap, **over_kw))
                k = max(2, min(10, args.steps))
                el = timed(alt.train_one_epoch, k, 2, alt.finalize)
                extra[leg_name] = {"value": T * k / el, "unit": "samples/s", "ms_per_step": el / k * 1e3, "steps": k,
                                   "item_sync": alt.describe()}
            except Exception as err:      # an extra leg must not cost the headline line
                extra[leg_name] = {"error": repr(err)}
        # (b) weak scaling: every rank its own AmazonBooks-shaped graph (different users, same item space)
        g2 = synthetic.make_graph(U, I, T, seed=2022 + 1000 * (rank + 1), with_test=False)
        eng2, item2 = build(g2.clicks, U, uw_h, (rank + 1) * T, None)
        tr2 = ItemSync(eng2, item2, world, **sync_kwargs(not args.no_overlap, epochs_per_exchange=1, force_collective=False))
        k = max(1, min(5, args.steps))
        el = timed(tr2.train_one_epoch, k, 1, tr2.finalize)
        extra["weak_scaling"] = {"value": world * T * k / el, "unit": "samples/s", "ms_per_step": el / k * 1e3, "steps": k,
                                 "workload": f"one {args.shape}-shaped graph PER GPU ({world} x {T} interactions per step)",
                                 "item_sync": tr2.describe()}
        eng2.close()
        dog.cancel()

    if rank == 0:
        out = line(elapsed, args.steps, trainer, kernel_ms, launches)
        out.update(extra)
        if world == 1 and not args.no_extra_legs:
            k = max(2, min(10, args.steps))
            with stdout_to_stderr():
                out["value_host_mode"] = host_mode_leg(graph, d, N, k, pinned=True)
                out["value_host_mode_pageable"] = host_mode_leg(graph, d, N, k, pinned=False)
            out["host_mode_note"] = ("same epoch through cf_c on host numpy buffers, weights written back after every epoch "
                                     "(PCIe-inclusive; never `value`): page-locked arrays as heat_amd.cf.frontend allocates "
                                     "them / plain pageable numpy arrays")
            out["tile_sampler"] = tile_sampler_leg(torch.from_numpy(graph.clicks.view(np.int64)).to(dev), T, uw_h, iw_h, dev, stream,
                                                   U, I, d, N, k)
            with stdout_to_stderr():
                out["accl"] = accl_leg(graph, d, N, max(2, k // 2))
            out["roofline_hbm_resident"] = hbm_resident_leg(dev, stream, 4_000_000)
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
