"""Test infrastructure: P user shards of ONE graph trained by P real HIP engines that take turns on one GPU, with the
item-table exchange of heat_amd.cf.distributed.ItemSync (fused delta / apply kernels of the C ABI) and the all-reduce
replaced by a sum over the P delta buffers on the device.  This is BASELINE.json configs[3] (AmazonBooks user-sharded
across 8 GPUs) minus the wires: same shards, same per-replica stream count and update policy, same exchange rule
(blocking, or `overlap`: the other ranks' deltas arrive one window late), so Recall/NDCG against single-engine training
can be checked on a one-GPU box."""
import numpy as np

from heat_amd import abi
from heat_amd.cf.distributed import shard_bounds, shard_clicks


def train_sharded(graph, uw0, iw0, *, num_negs, world, epochs, windows_per_epoch=1, overlap=False, seed=2022, scale=1.0,
                  exchange_every=1, defer_final=False, **cfg):
    """Returns (user table [num_users, d], item table [num_items, d], mean loss per epoch) after `epochs` epochs."""
    import torch
    dev = torch.device("cuda", 0)
    abi.load()
    side = torch.cuda.Stream(device=dev)
    U, d = uw0.shape
    with torch.cuda.stream(side):
        ranks = []
        for r in range(world):
            shard, lo, hi = shard_clicks(graph.clicks, U, world, r, bounds=shard_bounds(U, world, r))
            base = int(np.searchsorted(graph.clicks[:, 0], lo, side="left"))
            t_clicks = torch.from_numpy(shard.view(np.int64)).to(dev)
            t_user = torch.from_numpy(np.ascontiguousarray(uw0[lo:hi])).to(dev)
            t_item = torch.from_numpy(iw0).to(dev)
            eng = abi.Engine.from_device(t_clicks.data_ptr(), shard.shape[0], t_user.data_ptr(), t_item.data_ptr(),
                                         num_users=hi - lo, num_items=iw0.shape[0], emb_dim=d, num_negs=num_negs,
                                         stream=side.cuda_stream, seed=seed, sample_index_base=base,
                                         keep=(t_clicks, t_user, t_item), **cfg)
            ranks.append(dict(eng=eng, user=t_user, item=t_item, n=shard.shape[0], ref=t_item.clone(), sum=t_item.clone(),
                              mine=t_item.clone(), pending=False))
        n_max = max(r["n"] for r in ranks)
        window = -(-n_max // windows_per_epoch)
        total = torch.zeros_like(ranks[0]["item"])

        def complete():                              # what ItemSync._complete does on every rank
            if not ranks[0]["pending"]:
                return
            for r in ranks:
                r["eng"].sync_apply(r["ref"].data_ptr(), total.data_ptr(), r["mine"].data_ptr(), scale)
                r["pending"] = False

        def post(blocking):                          # ItemSync._post with the all-reduce done by hand
            complete()
            for r in ranks:
                r["eng"].sync_delta(r["ref"].data_ptr(), 0 if blocking else r["mine"].data_ptr(), r["sum"].data_ptr())
            total.zero_()
            for r in ranks:
                total.add_(r["sum"])
            if blocking:
                for r in ranks:
                    r["eng"].sync_apply(r["ref"].data_ptr(), total.data_ptr(), 0, scale)
            else:
                for r in ranks:
                    r["pending"] = True

        losses = []
        for epoch in range(epochs):
            tot = 0.0
            for r in ranks:
                r["eng"].begin_epoch()
            for w in range(windows_per_epoch):
                for r in ranks:
                    lo, hi = min(r["n"], w * window), min(r["n"], (w + 1) * window)
                    if hi > lo:
                        tot += r["eng"].train_range(lo, hi, want_loss=True)
                # exchange_every > 1: the exchange only every so many epochs (ItemSync(epochs_per_exchange=...)), and a closing one
                final = epoch == epochs - 1 and w == windows_per_epoch - 1
                if exchange_every > 1 and not final and ((epoch + 1) % exchange_every != 0 or w != windows_per_epoch - 1):
                    continue
                # defer_final (what bench.py runs in steady state): the closing exchange of an epoch is overlapped too, the other
                # ranks' deltas of an epoch's last window arrive during the next epoch; the very last one is completed
                post(blocking=(not overlap) or (w == windows_per_epoch - 1 and (final or not defer_final)))
            for r in ranks:
                r["eng"].end_epoch()
            losses.append(tot / graph.clicks.shape[0])
        if defer_final and overlap and ranks[0]["pending"]:
            post(blocking=True)                     # ItemSync.finalize()
        side.synchronize()
        name = ranks[0]["eng"].kernel_name
        for a, b in zip(ranks[:-1], ranks[1:]):
            assert torch.equal(a["item"], b["item"]), "replicas must be bit-identical after the closing exchange of an epoch"
        uw = np.concatenate([r["user"].cpu().numpy() for r in ranks])
        iw = ranks[0]["item"].cpu().numpy()
        for r in ranks:
            r["eng"].close()
    return uw, iw, losses, name
