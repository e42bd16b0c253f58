"""Development aid (GPU box): epoch time of ONE user shard of an N-GPU job (bench.py's strong-scaling leg sees exactly
this per rank) for kernel variants / stream counts — what bounds strong scaling before the collective does.
--sync: the same shard as a device-mode engine with the item-table exchange of heat_amd.cf.distributed.ItemSync on a ONE-rank
RCCL group (the collective itself moves nothing, the element-wise passes and the stream choreography are the real ones):
wall-clock epoch time without exchange, with the blocking exchange, overlapped on the training stream, and pipelined."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic
from heat_amd.cf.distributed import shard_bounds_balanced, shard_clicks

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="amazonbooks")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--streams", type=str, default="0")
ap.add_argument("--epochs", type=int, default=10)
ap.add_argument("--sync", action="store_true")
ap.add_argument("--windows", type=str, default="2")
args = ap.parse_args()
g, d, N = synthetic.make_named(args.shape, with_test=False)
uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d)
bounds = [shard_bounds_balanced(g.train_indptr, args.world, r) for r in range(args.world)]
sizes = [int(g.train_indptr[b] - g.train_indptr[a]) for a, b in bounds]
slow = int(np.argmax(sizes))
shard, lo, hi = shard_clicks(g.clicks, g.num_users, args.world, slow, bounds=bounds[slow])
print(f"slowest shard {slow} of {args.world} (ranges cut at equal interaction counts): users={hi - lo} interactions={shard.shape[0]} "
      f"variant={os.environ.get('HEAT_CF_VARIANT', 'auto')}", flush=True)
if not args.sync:
    for streams in [int(x) for x in args.streams.split(",")]:
        u, i = uw[lo:hi].copy(), iw.copy()
        eng = abi.Engine(shard, u, i, num_negs=N, num_streams=streams, flags=abi.FLAG_LAZY_SYNC)
        eng.train_one_epoch()
        eng.kernel_time(reset=True)
        for _ in range(args.epochs):
            eng.train_one_epoch()
        ms, n = eng.kernel_time()
        per = ms / n
        print(f"  {eng.kernel_name}: {per:.3f} ms/epoch-shard -> {shard.shape[0] / per / 1e3:.1f} M samples/s per GPU; "
              f"x{args.world} = {args.world * shard.shape[0] / per / 1e3:.0f} M/s if the exchange is hidden", flush=True)
        eng.close()
    sys.exit(0)

import torch
import torch.distributed as dist
from heat_amd.cf.distributed import ItemSync

dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
T = shard.shape[0]
for streams in [int(x) for x in args.streams.split(",")]:
    for windows in [int(x) for x in args.windows.split(",")]:
        for mode in ("none", "blocking", "overlap", "pipelined"):
            t_clicks = torch.from_numpy(shard.view(np.int64)).to(dev)
            t_uw, t_iw = torch.from_numpy(np.ascontiguousarray(uw[lo:hi])).to(dev), torch.from_numpy(iw).to(dev)
            eng = abi.Engine.from_device(t_clicks.data_ptr(), T, t_uw.data_ptr(), t_iw.data_ptr(), num_users=hi - lo, num_items=g.num_items,
                                         emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=2022, num_streams=streams,
                                         keep=(t_clicks, t_uw, t_iw))
            tr = None
            if mode != "none":
                tr = ItemSync(eng, t_iw, 1, windows_per_epoch=windows, mode="sum", force_collective=True, overlap=mode != "blocking",
                              defer_final=mode != "blocking", pipelined=mode == "pipelined")

            def step():
                if tr is None:
                    eng.begin_epoch()
                    eng.train_range(0, T, want_loss=False)
                    eng.end_epoch()
                else:
                    tr.train_one_epoch()

            for _ in range(3):
                step()
            if tr:
                tr.finalize()
            torch.cuda.synchronize()
            eng.kernel_time(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.epochs):
                step()
            if tr:
                tr.finalize()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.epochs * 1e3
            kms, kn = eng.kernel_time()
            print(f"  {eng.kernel_name} windows/epoch={windows if tr else 1} exchange={mode:9s}: {wall:.3f} ms per epoch wall "
                  f"(training kernels {kms / args.epochs:.3f} ms) -> x{args.world}: {args.world * T / wall / 1e3:.0f} M samples/s", flush=True)
            eng.close()
dist.destroy_process_group()
