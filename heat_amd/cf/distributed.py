"""Multi-GPU data parallelism for the CCL hot path: user rows (and their interactions) are sharded across ranks,
the item table is replicated and periodically synchronised with ONE collective over the whole table.

Replaces the fork's MPI scaffold (train/engine.cpp:262-286 per-row MPI_Bcast, :366-375 per-row MPI_Allreduce of item
weights then /world_size; cf/main.py:47-70 user-range sharding).  One process per GPU, torch.distributed: backend
"nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the CPU tests.

Synchronisation rule.  Between two syncs every rank trains a window of its own shard on its own replica.  At the sync
    W_item <- W_ref + sum_r (W_item,r - W_ref)            (mode "sum": every rank's updates are applied, the cross-GPU
                                                            analogue of the in-GPU scatter-add; default)
    W_item <- mean_r W_item,r                              (mode "mean": the fork's intent, engine.cpp:366-375)
where W_ref is the table right after the previous sync.  The exchange is delta -> all-reduce -> apply (two fused HIP
passes of the C ABI around ONE collective); with overlap=True the all-reduce of a window runs while the next window
trains and the other ranks' deltas are added one window late (class ItemSync).  The persistent gradient rows G stay local (the reference
never communicates them).  With behaviour aggregation on, the d x d aggregator matrix W0 is replicated and averaged
over ranks at the same points (engine.cpp:355-359: MPI_Allreduce SUM, then / world_size).  Window length: `sync_interactions` per rank; by default streams x refresh_interval
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


def shard_bounds_balanced(user_indptr, world_size, rank):
    """Contiguous user range of `rank` with ~equal INTERACTION counts instead of equal user counts (ranges by
    user count are not ranges by work; a deliberate deviation from cf/main.py:51-57).  `user_indptr` is the CSR row
    pointer of the interaction list ([num_users+1])."""
    indptr = np.asarray(user_indptr, dtype=np.int64)
    n_users, total = indptr.size - 1, int(indptr[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        u = int(np.searchsorted(indptr, target, side="left"))
        cuts.append(min(max(u, cuts[-1]), n_users))
    cuts.append(n_users)
    return cuts[rank], cuts[rank + 1]


def shard_clicks(clicks, num_users, world_size, rank, bounds=None):
    """Interactions of the rank's users with user ids re-based to the shard (cf/datasets.py:120-137)."""
    lo, hi = shard_bounds(num_users, world_size, rank) if bounds is None else bounds
    u = clicks[:, 0]
    if u.size > 1 and np.any(u[1:] < u[:-1]):
        raise ValueError("the interaction list must be grouped by ascending user id (LightGCN order) to be sharded by user range")
    a, b = np.searchsorted(u, lo, side="left"), np.searchsorted(u, hi, side="left")
    out = clicks[a:b].copy()
    out[:, 0] -= np.uint64(lo)
    return np.ascontiguousarray(out), lo, hi


class ItemSync:
    """Drives `engine` (heat_amd.abi.Engine in device mode, or any object with begin_epoch / train_range / end_epoch /
    data_rows) through epochs cut into windows and exchanges the replicated item table `item_w` (a torch tensor aliasing
    the engine's table) between the ranks.

    One exchange = delta (`sum = mine = W - ref`) -> all-reduce(`sum`) -> apply.  With an abi.Engine the two element-wise
    passes are the fused HIP kernels behind heat_cf_sync_delta / heat_cf_sync_apply; with any other engine (the CPU tests
    plug the oracle in) they are the same arithmetic in torch.

    overlap=False: every exchange is completed before the next window starts (`W = ref = ref + scale * sum`, bit-identical
    replicas after every window).
    overlap=True : the all-reduce of window k runs while window k+1 trains (asynchronous collective; torch.distributed
    orders it after the delta kernel and makes the training stream wait for it only at the next exchange); the other
    ranks' deltas then arrive one window late (`W += scale * sum - mine`).  The LAST exchange of an epoch is completed
    before the epoch ends, so the replicas are bit-identical at every epoch boundary — unless defer_final=True, which
    lets it overlap the first window of the next epoch (steady-state throughput runs; call finalize() before reading
    the table).

    pipelined=True (default; takes effect with overlap=True, an abi.Engine and device tensors): only ONE element-wise pass
    stays on the training stream at a window boundary (`W += x; snap = W`, heat_cf_sync_apply_snap); the delta, the
    collective and the preparation of the next `x` (heat_cf_sync_delta_from / heat_cf_sync_finish) run on an exchange
    stream while the next window trains.  Same algebra, bit-identical tables; 94 MB instead of 234 MB of row traffic on the
    training stream per exchange at AmazonBooks shape.

    collective="all_reduce": one RCCL all-reduce of the delta table (rings / trees over the xGMI links, RCCL's choice).
    collective="direct"    : the exchange SURVEY section 5 calls the direct 7-peer one, built from RCCL point-to-point
    collectives: the delta table is cut into world_size slices, slice j of every rank goes straight to rank j
    (all_to_all: every one of the 7 links of a GPU carries 1/8 of the table once), rank j adds the world_size
    contributions, and the summed slices are gathered back (all_gather: 1/8 of the table per link again) — 2 x 7/8 of the
    table over each GPU's links in two steps, against the 2 x 7/8 a ring moves in 14.  Same arithmetic (a sum over
    ranks in rank order), same apply pass; which one is faster is for the 8-GPU run to say (bench.py times both)."""

    def __init__(self, engine, item_w, world_size, refresh_interval=8192, sync_interactions=0, mode="sum", streams=0,
                 force_collective=False, mean_tensors=(), negatives=None, overlap=False, defer_final=False, dist=None,
                 windows_per_epoch=0, collective="all_reduce", pipelined=True, epochs_per_exchange=1):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.item_w = item_w
        self.world = world_size
        self.mode = mode
        self.mean_tensors = tuple(mean_tensors)   # replicated small state averaged at every sync (aggregator W0)
        self.negatives = negatives                # [data_rows, num_negs] u64 fed instead of the on-GPU sampler (parity tests)
        if sync_interactions <= 0:
            streams = streams or getattr(engine, "num_streams", 0) or 3022
            sync_interactions = streams * refresh_interval
        self.window = max(1, int(sync_interactions))   # clamped to the LARGEST shard inside train_one_epoch (same on every rank)
        self.windows_per_epoch = int(windows_per_epoch)   # > 0: the largest shard is cut into this many equal windows instead
        self.epochs_per_exchange = max(1, int(epochs_per_exchange))   # > 1: whole epochs between exchanges (finalize() closes)
        self._epochs_done = 0
        self.force = bool(force_collective)     # run the collective path even with one rank (tests)
        if mode not in ("sum", "mean"):
            raise ValueError("mode must be 'sum' or 'mean'")
        self.scale = 1.0 if mode == "sum" else 1.0 / world_size
        self.active = world_size > 1 or self.force
        self.overlap = bool(overlap) and self.active
        self.defer_final = bool(defer_final) and self.overlap
        self.native = hasattr(engine, "sync_delta") and getattr(item_w, "is_cuda", False)
        if collective not in ("all_reduce", "direct"):
            raise ValueError("collective must be 'all_reduce' or 'direct'")
        self.collective = collective
        self.ref = item_w.clone() if self.active else None
        self.sum = item_w.clone() if self.active else None
        if self.active and collective == "direct":
            import torch
            n, p = item_w.numel(), max(1, world_size)
            self._chunk = -(-n // p)
            # the delta lives at the head of a buffer padded to world_size equal slices (the padding stays zero)
            self._flat = torch.zeros(p * self._chunk, dtype=item_w.dtype, device=item_w.device)
            self.sum = self._flat[:n].view_as(item_w)
            self._recv = torch.empty_like(self._flat)
            self._part = torch.empty(self._chunk, dtype=item_w.dtype, device=item_w.device)
            self._helper = torch.cuda.Stream(device=item_w.device) if item_w.is_cuda else None
        self.mine = item_w.clone() if self.overlap else None
        self.pipelined = bool(pipelined) and self.overlap and self.native and hasattr(engine, "sync_apply_snap")
        self._x_ready = False                    # pipelined: `mine` holds x = scale * sum - mine of the previous exchange
        # An epoch of a shard of an 8-GPU job is ~1 ms: the ~50 us the host needs to issue a collective (and, pipelined, the
        # two passes around it) would leave the training stream empty.  With a HIP engine on device tensors the host therefore
        # only QUEUES the element-wise pass the training stream needs, records an event, and issues the rest — on the exchange
        # stream, behind that event — after the next window's kernel has been queued (`_issue`, called from train_one_epoch).
        self._deferred = None
        self._defer_ok = self.overlap and self.native and getattr(item_w, "is_cuda", False) and hasattr(engine, "device_view")
        if self._defer_ok:
            import torch
            self._xs = torch.cuda.Stream(device=item_w.device)            # the exchange stream
            h = engine.device_view().stream                               # the stream the engine launches on
            self._ts = torch.cuda.ExternalStream(h, device=item_w.device) if h else torch.cuda.default_stream(item_w.device)
            self._ev_snap, self._ev_x = torch.cuda.Event(), torch.cuda.Event()
        if self.pipelined:
            self.snap = item_w.clone()
        self.pending = None                      # (work handle) of an all-reduce in flight
        self._n_max = None
        self.exchanges = 0
        self.track_loss = False   # True: train_range synchronises per window and the local loss sum is returned

    def describe(self):
        name = "all_reduce(item table delta)" if self.collective == "all_reduce" else \
            "all_to_all(delta slices) + local sum + all_gather(summed slices)"
        return {"collective": name + (" + all_reduce(W0)" if self.mean_tensors else ""), "mode": self.mode,
                "overlap": self.overlap, "pipelined": self.pipelined, "fused_delta_apply_kernels": bool(self.native),
                "window_interactions_per_gpu": getattr(self, "last_window", min(self.window, self.engine.data_rows)),
                "epochs_per_exchange": self.epochs_per_exchange,
                "exchanges": self.exchanges}

    # ---- the two element-wise passes --------------------------------------------------------------------------------
    def _delta(self, with_mine):
        if self.native:
            self.engine.sync_delta(self.ref.data_ptr(), self.mine.data_ptr() if with_mine else 0, self.sum.data_ptr())
        else:
            self.sum.copy_(self.item_w).sub_(self.ref)
            if with_mine:
                self.mine.copy_(self.sum)

    def _apply(self, with_mine):
        if self.native:
            self.engine.sync_apply(self.ref.data_ptr(), self.sum.data_ptr(), self.mine.data_ptr() if with_mine else 0, self.scale)
        elif with_mine:
            # the expressions of item_apply_kernel, operation by operation (item_sync.hip is built without FMA contraction):
            # s = scale * sum; W = W + (s - mine); ref = ref + s
            self.sum.mul_(self.scale)
            self.item_w.add_(self.sum - self.mine)
            self.ref.add_(self.sum)
        else:
            self.sum.mul_(self.scale)             # item_apply_exact_kernel: W = ref = ref + scale * sum
            self.ref.add_(self.sum)
            self.item_w.copy_(self.ref)

    def _issue(self):
        """Issue what a non-blocking exchange left for later (see __init__)."""
        if self._deferred is not None:
            fn, self._deferred = self._deferred, None
            fn()

    def _complete(self):
        """Wait for the all-reduce in flight (stream-side for RCCL, host-side for gloo) and apply it."""
        self._issue()
        if self._x_ready:                       # pipelined: the exchange stream has left x in `mine` and moved `ref`
            self._ts.wait_event(self._ev_x)
            self.engine.sync_apply_snap(self.mine.data_ptr(), self.snap.data_ptr())
            self._x_ready = False
        if self.pending is None:
            return
        work, with_mine = self.pending
        if work is not None:
            work.wait()
        self._apply(with_mine)
        self.pending = None

    def _post_pipelined(self):
        """One pass on the training stream (W += x of the previous exchange; snap = W), the rest on the exchange stream."""
        import torch
        for t in self.mean_tensors:             # small replicated state (W0): where it always was, on the training stream
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            t.div_(self.world)
        self._issue()
        if self._x_ready:
            self._ts.wait_event(self._ev_x)     # long done: the previous window's collective had a whole window
        self.engine.sync_apply_snap(self.mine.data_ptr() if self._x_ready else 0, self.snap.data_ptr())
        self._ev_snap.record(self._ts)

        def rest():
            with torch.cuda.stream(self._xs):
                self._xs.wait_event(self._ev_snap)
                self.engine.sync_delta_from(self.snap.data_ptr(), self.ref.data_ptr(), self.mine.data_ptr(), self.sum.data_ptr(),
                                            self._xs.cuda_stream)
                work = self._exchange(blocking=False)
                if work is not None:
                    work.wait()                 # orders the exchange stream (RCCL) / blocks the host (gloo) behind the sum
                self.engine.sync_finish(self.ref.data_ptr(), self.sum.data_ptr(), self.mine.data_ptr(), self.scale, self._xs.cuda_stream)
                self._ev_x.record(self._xs)

        self._deferred = rest                   # issued once the next window's kernel is queued
        self._x_ready = True
        self.exchanges += 1

    def _post(self, blocking):
        """Start an exchange of everything this rank changed since the reference."""
        if self.pipelined and not blocking:
            return self._post_pipelined()
        with_mine = not blocking
        fused = False
        self._issue()
        if (with_mine and self.native and self.pending is not None and self.pending[1] and not self._x_ready
                and hasattr(self.engine, "sync_apply_delta")):
            # steady state of the overlapped schedule: the apply of the exchange in flight and the delta of this one in ONE
            # pass over the tables (heat_cf_sync_apply_delta: same expressions, same bits, one launch and 47 MB less)
            work = self.pending[0]
            if work is not None:
                work.wait()
            self.pending = None
            self.engine.sync_apply_delta(self.ref.data_ptr(), self.sum.data_ptr(), self.mine.data_ptr(), self.scale)
            fused = True
        else:
            self._complete()                    # the reference must be current before the next delta
        for t in self.mean_tensors:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            t.div_(self.world)
        if not fused:
            self._delta(with_mine)
        self.exchanges += 1
        if not blocking and self._defer_ok:
            import torch
            self._ev_snap.record(self._ts)       # the delta is queued; the collective is issued later, behind this event

            def rest():
                with torch.cuda.stream(self._xs):
                    self._xs.wait_event(self._ev_snap)
                    self.pending = (self._exchange(False), True)

            self._deferred = rest
            return
        work = self._exchange(blocking)
        self.pending = (work if not blocking else None, with_mine)
        if blocking:
            self._complete()

    def _exchange(self, blocking):
        """Start the sum of `self.sum` over the ranks (in place); returns the handle to wait on (None when done)."""
        dist = self.dist
        if self.collective == "all_reduce":
            return dist.all_reduce(self.sum, op=dist.ReduceOp.SUM, async_op=not blocking)
        import torch
        p = max(1, self.world)
        if self._helper is None:                             # host tensors (gloo tests): plain sequence
            dist.all_to_all_single(self._recv, self._flat)
            torch.sum(self._recv.view(p, self._chunk), dim=0, out=self._part)
            dist.all_gather_into_tensor(self._flat, self._part)
            return None
        # the scatter is ordered after the delta kernel on the current stream; the local sum and the gather run on a
        # helper stream, so that the training stream is not made to wait for either before the next exchange
        first = dist.all_to_all_single(self._recv, self._flat, async_op=True)
        self._helper.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._helper):
            first.wait()
            torch.sum(self._recv.view(p, self._chunk), dim=0, out=self._part)
            work = dist.all_gather_into_tensor(self._flat, self._part, async_op=True)
        if blocking:
            work.wait()
            return None
        return work

    def sync(self, last=True):
        if not self.active:
            return
        self._post(blocking=not self.overlap or (last and not self.defer_final))

    def finalize(self):
        """Complete an exchange left in flight by defer_final (replicas still differ by their last window: follow with a
        blocking exchange to make them identical)."""
        if not self.active:
            return
        self._complete()
        if self.defer_final or self.epochs_per_exchange > 1:
            self._post(blocking=True)

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
        if self.windows_per_epoch > 0:
            window = max(1, -(-n_max // self.windows_per_epoch))
        n_windows = max(1, -(-n_max // window))
        self.last_window = window
        e.begin_epoch()
        loss_sum = 0.0
        for w in range(n_windows):
            lo = min(n, w * window)
            hi = min(n, (w + 1) * window)
            if hi > lo:
                if self.negatives is None:
                    got = e.train_range(lo, hi, want_loss=self.track_loss)
                else:
                    got = e.train_range(lo, hi, self.negatives[lo:hi], want_loss=self.track_loss)
                self._issue()                    # the window's kernel is queued: now the host has time for the collective
                if self.track_loss:
                    loss_sum += got
            if self.epochs_per_exchange > 1 and (w != n_windows - 1 or (self._epochs_done + 1) % self.epochs_per_exchange != 0):
                continue                         # no exchange at this boundary (the same decision on every rank)
            self.sync(last=(w == n_windows - 1))
        self._epochs_done += 1
        e.end_epoch()
        return loss_sum if self.track_loss else None


class ShardedTrainer:
    """One rank of a user-sharded training job (the fork's intent at cf/main.py:47-70 + engine.cpp:366-375, without
    MPI): takes the WHOLE interaction list and the WHOLE initial tables (identical on every rank), keeps only this
    rank's users, builds a device-mode engine on torch tensors and synchronises the replicated item table with ItemSync.

        trainer = ShardedTrainer(clicks, user_w, item_w, num_negs=16, seed=2022)      # after init_process_group
        for epoch in range(E): trainer.train_one_epoch()
        user_w_shard, item_w = trainer.weights()           # numpy; users [lo, hi) of this rank

    Behaviour aggregation (ACCL): pass `his` [num_users, max_his] u64, `masks` [num_users] u64 and `w0` [d, d] f32; the
    history rows follow their users into the shard, W0 is replicated and averaged at every sync.
    `negatives` [len(clicks), num_negs] u64 replaces the on-GPU sampler with caller-fed ids (parity tests).

    `engine_factory(shard_clicks, user_w_shard_tensor, item_w_tensor, sample_index_base)` may replace the HIP engine
    (the CPU tests plug the oracle in that way); with aggregation it is also handed `his=, masks=, w0=` (shard rows,
    shard lengths, the W0 tensor)."""

    def __init__(self, clicks, user_w, item_w, *, num_negs, rank=None, world_size=None, device=None, seed=2022,
                 refresh_interval=8192, sync_interactions=0, mode="sum", engine_factory=None, balance="users",
                 his=None, masks=None, w0=None, negatives=None, overlap=False, defer_final=False, windows_per_epoch=0,
                 collective="all_reduce", epochs_per_exchange=1, **cfg_kwargs):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world_size is None else world_size
        num_users = user_w.shape[0]
        self.num_users_total = num_users
        self.bounds = None
        if balance == "interactions":      # equal work per rank instead of equal user counts
            indptr = np.concatenate([[0], np.cumsum(np.bincount(clicks[:, 0].astype(np.int64), minlength=num_users))])
            self.all_bounds = [shard_bounds_balanced(indptr, self.world, r) for r in range(self.world)]
        elif balance == "users":
            self.all_bounds = [shard_bounds(num_users, self.world, r) for r in range(self.world)]
        else:
            raise ValueError("balance must be 'users' or 'interactions'")
        self.shard, self.lo, self.hi = shard_clicks(clicks, num_users, self.world, self.rank, bounds=self.all_bounds[self.rank])
        base = int(np.searchsorted(clicks[:, 0], self.lo, side="left"))      # global index of the shard's first interaction
        self.aggregate = w0 is not None
        if self.aggregate and (his is None or masks is None):
            raise ValueError("behaviour aggregation needs his, masks and w0")
        self.t_w0 = None
        if self.aggregate:
            his_shard = np.ascontiguousarray(np.asarray(his, dtype=np.uint64)[self.lo:self.hi])
            masks_shard = np.ascontiguousarray(np.asarray(masks, dtype=np.uint64).reshape(-1)[self.lo:self.hi])
        if engine_factory is None:
            from heat_amd import abi
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            self.t_clicks = torch.from_numpy(self.shard.view(np.int64)).to(dev)
            self.t_user = torch.from_numpy(np.ascontiguousarray(user_w[self.lo:self.hi])).to(dev)
            self.t_item = torch.from_numpy(np.ascontiguousarray(item_w)).to(dev)
            stream = torch.cuda.current_stream().cuda_stream
            flags = cfg_kwargs.pop("flags", 0) | (0 if stream else abi.FLAG_NULL_STREAM)
            agg = {}
            keep_agg = ()
            if self.aggregate:
                t_his = torch.from_numpy(his_shard.view(np.int64)).to(dev)
                t_masks = torch.from_numpy(masks_shard.view(np.int64)).to(dev)
                self.t_w0 = torch.from_numpy(np.ascontiguousarray(w0, dtype=np.float32)).to(dev)
                agg = dict(his_ptr=t_his.data_ptr(), max_his=his_shard.shape[1], masks_ptr=t_masks.data_ptr(),
                           w0_ptr=self.t_w0.data_ptr(), use_aggregator=1)
                keep_agg = (t_his, t_masks, self.t_w0)
            self.engine = abi.Engine.from_device(
                self.t_clicks.data_ptr(), self.shard.shape[0], self.t_user.data_ptr(), self.t_item.data_ptr(),
                num_users=self.hi - self.lo, num_items=item_w.shape[0], emb_dim=item_w.shape[1], num_negs=num_negs,
                stream=stream or None, seed=seed, sample_index_base=base, flags=flags, device=dev.index,
                keep=(self.t_clicks, self.t_user, self.t_item) + keep_agg, **agg, **cfg_kwargs)
        else:
            self.t_user = torch.from_numpy(np.ascontiguousarray(user_w[self.lo:self.hi]).copy())
            self.t_item = torch.from_numpy(np.ascontiguousarray(item_w).copy())
            if self.aggregate:
                self.t_w0 = torch.from_numpy(np.ascontiguousarray(w0, dtype=np.float32).copy())
                self.engine = engine_factory(self.shard, self.t_user, self.t_item, base, his=his_shard, masks=masks_shard,
                                             w0=self.t_w0)
            else:
                self.engine = engine_factory(self.shard, self.t_user, self.t_item, base)
        self.sync = ItemSync(self.engine, self.t_item, self.world, refresh_interval=refresh_interval,
                             sync_interactions=sync_interactions, mode=mode,
                             mean_tensors=(self.t_w0,) if self.aggregate else (), overlap=overlap, defer_final=defer_final,
                             windows_per_epoch=windows_per_epoch,
                             negatives=None if negatives is None else
                             np.ascontiguousarray(negatives[base:base + self.shard.shape[0]], dtype=np.uint64), collective=collective,
                             epochs_per_exchange=epochs_per_exchange)

    def train_one_epoch(self, want_loss=False):
        """One epoch on this rank's shard.  want_loss=True returns the GLOBAL mean loss (loss sums and interaction counts
        summed over ranks, train/engine.cpp:380-385)."""
        self.sync.track_loss = bool(want_loss)
        local = self.sync.train_one_epoch()
        if not want_loss:
            return None
        import torch.distributed as dist
        t = self.torch.tensor([local, float(self.shard.shape[0])], dtype=self.torch.float64, device=self.t_item.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0].item() / max(t[1].item(), 1.0))

    def weights(self):
        """(this rank's user rows [lo, hi), the replicated item table) as numpy arrays."""
        if self.t_item.is_cuda:
            self.torch.cuda.current_stream().synchronize()
        return self.t_user.cpu().numpy(), self.t_item.cpu().numpy()

    def aggregator_weights(self):
        """The replicated W0 [d, d] (None without behaviour aggregation)."""
        if self.t_w0 is None:
            return None
        if self.t_w0.is_cuda:
            self.torch.cuda.current_stream().synchronize()
        return self.t_w0.cpu().numpy()

    def gather_user_weights(self):
        """The full user table on every rank (all_gather of the shards; shards differ by at most one row)."""
        import torch.distributed as dist
        if self.world == 1:
            return self.t_user.cpu().numpy()
        torch = self.torch
        d = self.t_user.shape[1]
        rows = [b - a for a, b in self.all_bounds]
        pad = max(rows)
        mine = torch.zeros((pad, d), dtype=self.t_user.dtype, device=self.t_user.device)
        mine[:self.t_user.shape[0]] = self.t_user
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        return torch.cat([p[:n] for p, n in zip(parts, rows)]).cpu().numpy()
