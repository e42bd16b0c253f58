"""Worker for tests/test_gpu_parity.py::test_two_ranks_on_one_gpu_equal_single_process_training: two ranks (gloo, both
on cuda:0) run heat_amd.cf.distributed.ShardedTrainer with REAL HIP engines (serial walk, caller-fed negatives) on a
problem whose two user shards touch disjoint item rows, so the `sum` sync must reproduce single-process training."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heat_amd import abi  # noqa: E402
from heat_amd.cf.distributed import ShardedTrainer  # noqa: E402


def main():
    out_dir, window = sys.argv[1], int(sys.argv[2])
    abi.load()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    data = np.load(os.path.join(out_dir, "problem.npz"))
    agg = dict(his=data["his"], masks=data["masks"], w0=data["w0"]) if "w0" in data.files else {}
    side = torch.cuda.Stream(device=torch.device("cuda", 0))
    with torch.cuda.stream(side):
        tr = ShardedTrainer(data["clicks"], data["uw"], data["iw"], num_negs=int(data["num_negs"]), sync_interactions=window,
                            mode="sum", negatives=data["negs"], flags=abi.FLAG_SERIAL, l_r=float(data["lr"]), clip_val=1.0,
                            **agg)
        losses = [tr.train_one_epoch(want_loss=True) for _ in range(int(data["epochs"]))]
        uw, iw = tr.weights()
        full_u = tr.gather_user_weights()
        extra = {} if tr.aggregator_weights() is None else {"w0": tr.aggregator_weights()}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), uw=uw, iw=iw, lo=tr.lo, hi=tr.hi, full_u=full_u, losses=np.array(losses),
             **extra)
    dist.barrier()
    dist.destroy_process_group()
    print("SHARD_OK", flush=True)


if __name__ == "__main__":
    main()
