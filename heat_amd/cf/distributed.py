"""Multi-GPU data parallelism for the CCL hot path: user rows (and their interactions) are sharded across ranks,
the item table is replicated and periodically synchronised with ONE collective over the whole table.

Replaces the fork's MPI scaffold (train/engine.cpp:262-286 per-row MPI_Bcast, :366-375 per-row MPI_Allreduce of item
weights then /world_size; cf/main.py:47-70 user-range sharding).  One process per GPU, torch.distributed: backend
"nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the CPU tests.

Synchronisation rule.  Between two syncs every rank trains a window of its own shard on its own replica.  At the sync
    W_item <- W_ref + sum_r (W_item,r - W_ref)            (mode "sum": every rank's updates are applied, the cross-GPU
                                                            analogue of the in-GPU scatter-add; default)
    W_item <- mean_r W_item,r                              (mode "mean": the fork's intent, engine.cpp:366-375)
where W_ref is the table right after the previous sync.  The persistent gradient rows G stay local (the reference
never communicates them).  Window length: `sync_interactions` per rank; by default streams x refresh_interval
(refresh_interval is a per-worker step count in the reference — negative_samplers/random_tile_negative_sampler.cpp:33 —
and a worker here is one wave-stream), capped at one epoch.
"""
import numpy as np


def shard_bounds(num_users, world_size, rank):
    """Contiguous user range of `rank` (cf/main.py:51-57: k = n // P, r = n % P, start = i*k + min(i, r),
    end = start + k + (i < r)); rank 0 uses the same formula (the fork drops user k there when r > 0, main.py:64)."""
    k, r = divmod(num_users, world_size)
    start = rank * k + min(rank, r)
    return start, start + k + (1 if rank < r else 0)


def shard_clicks(clicks, num_users, world_size, rank):
    """Interactions of the rank's users with user ids re-based to the shard (cf/datasets.py:120-137)."""
    lo, hi = shard_bounds(num_users, world_size, rank)
    u = clicks[:, 0]
    a, b = np.searchsorted(u, lo, side="left"), np.searchsorted(u, hi, side="left")
    out = clicks[a:b].copy()
    out[:, 0] -= np.uint64(lo)
    return np.ascontiguousarray(out), lo, hi


class ItemSync:
    """Drives `engine` (heat_amd.abi.Engine in device mode, or any object with begin_epoch / train_range / end_epoch /
    data_rows) through epochs cut into windows, all-reducing `item_w` (a torch tensor aliasing the engine's table)."""

    def __init__(self, engine, item_w, world_size, refresh_interval=8192, sync_interactions=0, mode="sum", streams=0,
                 force_collective=False):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.item_w = item_w
        self.world = world_size
        self.mode = mode
        n = engine.data_rows
        if sync_interactions <= 0:
            streams = streams or getattr(engine, "num_streams", 0) or 3022
            sync_interactions = streams * refresh_interval
        self.window = max(1, int(sync_interactions))   # clamped to the LARGEST shard inside train_one_epoch (same on every rank)
        self.force = bool(force_collective)     # run the collective path even with one rank (tests)
        self.ref = item_w.clone() if (mode == "sum" and (world_size > 1 or self.force)) else None
        self._n_max = None
        if mode not in ("sum", "mean"):
            raise ValueError("mode must be 'sum' or 'mean'")

    def describe(self):
        return {"collective": "all_reduce(item table)", "mode": self.mode,
                "window_interactions_per_gpu": min(self.window, self.engine.data_rows)}

    def sync(self):
        if self.world == 1 and not self.force:
            return
        if self.mode == "mean":
            self.dist.all_reduce(self.item_w, op=self.dist.ReduceOp.SUM)
            self.item_w.div_(self.world)
        else:
            # delta since the last sync, summed over ranks, applied to the common reference
            self.item_w.sub_(self.ref)
            self.dist.all_reduce(self.item_w, op=self.dist.ReduceOp.SUM)
            self.item_w.add_(self.ref)
            self.ref.copy_(self.item_w)

    def train_one_epoch(self):
        e = self.engine
        n = e.data_rows
        # every rank must run the same number of collectives: windows are counted on the largest shard
        if self._n_max is None:            # once: the shard sizes do not change between epochs
            import torch
            if self.world > 1:
                t = torch.tensor([n], dtype=torch.int64, device=self.item_w.device)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                self._n_max = int(t.item())
            else:
                self._n_max = n
        n_max = self._n_max
        window = min(self.window, max(1, n_max))
        n_windows = max(1, -(-n_max // window))
        e.begin_epoch()
        for w in range(n_windows):
            lo = min(n, w * window)
            hi = min(n, (w + 1) * window)
            if hi > lo:
                e.train_range(lo, hi, want_loss=False)
            self.sync()
        e.end_epoch()
