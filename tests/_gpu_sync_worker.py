"""Worker for tests/test_gpu_parity.py::test_device_mode_engine_with_item_sync_on_a_side_stream (GPU, 1-rank RCCL group)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heat_amd import abi  # noqa: E402
from heat_amd.cf import synthetic  # noqa: E402
from heat_amd.cf.distributed import ItemSync  # noqa: E402


def main():
    d, N, U, I, T = 64, 16, 50, 800, 1500
    rng = np.random.default_rng(5)
    g = synthetic.make_graph(U, I, T, seed=5, with_test=False)
    clicks = g.clicks
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    results = []
    for use_sync in (False, True):
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            t_clicks = torch.from_numpy(clicks.view(np.int64)).to(dev)
            t_uw, t_iw = torch.from_numpy(uw).to(dev), torch.from_numpy(iw).to(dev)
            eng = abi.Engine.from_device(t_clicks.data_ptr(), T, t_uw.data_ptr(), t_iw.data_ptr(), num_users=U,
                                         num_items=I, emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=3,
                                         flags=abi.FLAG_SERIAL | abi.FLAG_SAMPLING_CALL,   # window-independent negatives
                                         keep=(t_clicks, t_uw, t_iw))
            if use_sync:
                tr = ItemSync(eng, t_iw, 1, sync_interactions=400, mode="sum", force_collective=True)
                for _ in range(2):
                    tr.train_one_epoch()
            else:
                for _ in range(2):
                    eng.begin_epoch()
                    eng.train_range(0, T, want_loss=False)
                    eng.end_epoch()
            side.synchronize()
            results.append((t_uw.cpu().numpy(), t_iw.cpu().numpy()))
            eng.close()
    (u0, i0), (u1, i1) = results
    assert not np.array_equal(i0, iw)
    np.testing.assert_allclose(i1, i0, rtol=0, atol=2e-5)     # (W - ref) + ref costs 1 ulp per sync
    np.testing.assert_allclose(u1, u0, rtol=0, atol=2e-5)
    # the overlapped exchange in its two forms — delta + apply on the training stream, or only `W += x; snap = W` there and
    # the rest on an exchange stream (pipelined) — with the all-reduce, the direct exchange and a drained pipeline at the
    # end: the same algebra, so bit-identical tables
    outs = {}
    for pipelined in (False, True):
        for collective in ("all_reduce", "direct"):
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                t_clicks = torch.from_numpy(clicks.view(np.int64)).to(dev)
                t_uw, t_iw = torch.from_numpy(uw).to(dev), torch.from_numpy(iw).to(dev)
                eng = abi.Engine.from_device(t_clicks.data_ptr(), T, t_uw.data_ptr(), t_iw.data_ptr(), num_users=U,
                                             num_items=I, emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=3,
                                             flags=abi.FLAG_SERIAL | abi.FLAG_SAMPLING_CALL, keep=(t_clicks, t_uw, t_iw))
                tr = ItemSync(eng, t_iw, 1, sync_interactions=400, mode="sum", force_collective=True, overlap=True,
                              defer_final=True, pipelined=pipelined, collective=collective)
                assert tr.pipelined == pipelined and tr.describe()["pipelined"] == pipelined
                for _ in range(3):
                    tr.train_one_epoch()
                tr.finalize()
                side.synchronize()
                torch.cuda.synchronize()
                outs[(pipelined, collective)] = (t_uw.cpu().numpy(), t_iw.cpu().numpy(), tr.exchanges)
                eng.close()
    base = outs[(False, "all_reduce")]
    assert not np.array_equal(base[1], iw)
    for key, (u, i, n_ex) in outs.items():
        assert np.array_equal(u, base[0]) and np.array_equal(i, base[1]), key
        assert n_ex == base[2]
    print("SYNC_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
