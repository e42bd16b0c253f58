"""Worker for tests/test_distributed_cpu.py: world_size-2 gloo run of heat_amd.cf.distributed.ItemSync with the CPU
oracle standing in for the GPU engine (test infrastructure: the product path never uses the oracle)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heat_amd.cf.distributed import ShardedTrainer  # noqa: E402
from oracle import cf_oracle as orc  # noqa: E402


class OracleWindows:
    """begin_epoch / train_range / end_epoch on top of the oracle, with caller-fed negatives."""

    def __init__(self, clicks, uw, iw, negs, num_negs, lr, **agg):
        self.e = orc.Engine(clicks, uw, iw, num_negs=num_negs, l_r=lr, clip_val=1.0, **agg)
        self.negs = negs
        self.data_rows = clicks.shape[0]

    def begin_epoch(self):
        self.e.lr_step()

    def train_range(self, lo, hi, want_loss=False):
        self.e.train_range(lo, hi, self.negs[lo:hi])

    def end_epoch(self):
        self.e.zero_grad()
        self.e.epoch = self.e.epoch + 1


def main():
    out_dir, mode, window = sys.argv[1], sys.argv[2], int(sys.argv[3])
    overlap = len(sys.argv) > 4 and sys.argv[4] in ("overlap", "defer")
    defer = len(sys.argv) > 4 and sys.argv[4] == "defer"      # bench.py's steady state: the closing exchange of an epoch overlaps too
    collective = sys.argv[5] if len(sys.argv) > 5 else "all_reduce"
    every = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # epochs between exchanges (bench.py at 8 GPUs: 2)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    data = np.load(os.path.join(out_dir, "problem.npz"))
    clicks, num_users = data["clicks"], int(data["num_users"])
    all_negs, N, lr = data["negs"], int(data["num_negs"]), float(data["lr"])

    def oracle_factory(shard, t_user, t_item, base, his=None, masks=None, w0=None):
        # the tensors share memory with the numpy views the oracle borrows and trains in place
        agg = {} if w0 is None else dict(his=his, masks=masks, w0=w0.numpy(), use_aggregator=True)
        return OracleWindows(shard, t_user.numpy(), t_item.numpy(), all_negs[base:base + shard.shape[0]], N, lr, **agg)

    agg = dict(his=data["his"], masks=data["masks"], w0=data["w0"]) if "w0" in data.files else {}
    tr = ShardedTrainer(clicks, data["uw"], data["iw"], num_negs=N, sync_interactions=window, mode=mode,
                        engine_factory=oracle_factory, overlap=overlap, defer_final=defer, collective=collective, epochs_per_exchange=every, **agg)
    for _ in range(int(data["epochs"])):
        tr.train_one_epoch()
    tr.sync.finalize()
    uw, iw = tr.weights()
    full_u = tr.gather_user_weights()
    extra = {} if tr.aggregator_weights() is None else {"w0": tr.aggregator_weights()}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), uw=uw, iw=iw, lo=tr.lo, hi=tr.hi, rows=tr.shard.shape[0], full_u=full_u,
             collective=np.array(tr.sync.describe()["collective"]), exchanges=tr.sync.exchanges, **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
