"""The N>1 path on CPU: world_size-2 `gloo` runs of the item-table synchronisation (heat_amd/cf/distributed.py) with
the oracle as the per-rank compute.  Checks user-range sharding (cf/main.py:51-57), that replicas agree after a sync,
the `mean` rule (train/engine.cpp:366-375 intent) and the `sum` rule (every rank's delta applied)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from heat_amd.cf.distributed import shard_bounds, shard_bounds_balanced, shard_clicks
from oracle import cf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_all_users():
    for n, p in [(52643, 8), (10, 3), (7, 8), (100, 1), (29858, 4)]:
        spans = [shard_bounds(n, p, r) for r in range(p)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(p - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)   # main.py:51-57: first r ranks get +1


def test_balanced_shards_equalise_interactions():
    rng = np.random.default_rng(0)
    deg = (rng.pareto(1.5, size=5000) * 20 + 1).astype(np.int64)          # heavy-tailed user degrees
    indptr = np.concatenate([[0], np.cumsum(deg)])
    for p in (2, 4, 8):
        spans = [shard_bounds_balanced(indptr, p, r) for r in range(p)]
        assert spans[0][0] == 0 and spans[-1][1] == 5000 and all(spans[i][1] == spans[i + 1][0] for i in range(p - 1))
        work = np.array([indptr[b] - indptr[a] for a, b in spans], dtype=np.float64)
        by_users = np.array([indptr[b] - indptr[a] for a, b in (shard_bounds(5000, p, r) for r in range(p))], dtype=np.float64)
        assert work.max() / work.mean() <= by_users.max() / by_users.mean() + 1e-9
        assert work.max() / work.mean() < 1.0 + deg.max() * p / indptr[-1] + 1e-9   # within one user's degree of perfect


def test_shard_clicks_rebases_users():
    clicks = np.array([[0, 5], [0, 6], [1, 7], [3, 8], [4, 9], [4, 1]], dtype=np.uint64)
    s0, lo0, hi0 = shard_clicks(clicks, 5, 2, 0)
    s1, lo1, hi1 = shard_clicks(clicks, 5, 2, 1)
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert s0.tolist() == [[0, 5], [0, 6], [1, 7]] and s1.tolist() == [[0, 8], [1, 9], [1, 1]]


def make_problem(tmp_path, disjoint_items, epochs=2):
    rng = np.random.default_rng(0)
    U, I, d, N, T = 40, 200, 16, 4, 600
    users = np.sort(rng.integers(0, U, T))
    half = U // 2
    if disjoint_items:
        # users of rank 0 (ids < 20) only touch items < 100, rank 1 only items >= 100 -> the two replicas never
        # write the same row and the `sum` rule must reproduce single-process training exactly
        pos = np.where(users < half, rng.integers(0, I // 2, T), rng.integers(I // 2, I, T))
        negs = np.where((users < half)[:, None], rng.integers(0, I // 2, (T, N)), rng.integers(I // 2, I, (T, N)))
    else:
        pos = rng.integers(0, I, T)
        negs = rng.integers(0, I, (T, N))
    clicks = np.stack([users, pos], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    np.savez(tmp_path / "problem.npz", clicks=clicks, negs=negs.astype(np.uint64), uw=uw, iw=iw, num_users=U,
             num_negs=N, lr=0.01, epochs=epochs)
    return clicks, negs.astype(np.uint64), uw, iw, U, N


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, mode, window, world=2, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"),
           str(tmp_path), mode, str(window)] + list(extra)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.timeout(400)
def test_sum_sync_equals_single_process_when_replicas_touch_disjoint_rows(tmp_path):
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=True)
    ranks = run_world(tmp_path, "sum", window=64)
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])                 # replicas agree after the last sync
    # single process, whole list, same negatives
    u1, i1 = uw.copy(), iw.copy()
    ref = orc.Engine(clicks, u1, i1, num_negs=N, l_r=0.01, clip_val=1.0)
    for _ in range(2):
        ref.lr_step()
        ref.train_range(0, clicks.shape[0], negs)
        ref.zero_grad()
        ref.epoch = ref.epoch + 1
    np.testing.assert_allclose(ranks[0]["iw"], i1, rtol=0, atol=1e-5)      # (W - ref) + ref costs 1 ulp per sync, carried through training
    got_u = np.concatenate([ranks[0]["uw"], ranks[1]["uw"]])
    assert np.array_equal(ranks[0]["full_u"], got_u) and np.array_equal(ranks[1]["full_u"], got_u)   # all_gather of the shards
    np.testing.assert_allclose(got_u, u1, rtol=0, atol=1e-5)               # user rows are private (never communicated)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("schedule", ["blocking", "overlap"])
def test_direct_exchange_equals_all_reduce(tmp_path, schedule):
    """collective="direct" (delta slices scattered with all_to_all, summed by their owner, gathered back) leaves the same
    replicas as the all-reduce, bit for bit at two ranks (a + b in either order), blocking and one-window-late; the item
    table's size is not a multiple of the rank count times anything (zero-padded last slice)."""
    make_problem(tmp_path, disjoint_items=False, epochs=2)
    extra = ["-"] if schedule == "blocking" else ["overlap"]
    a = run_world(tmp_path, "sum", window=48, extra=extra + ["all_reduce"])
    assert "all_reduce" in str(a[0]["collective"]) and int(a[0]["exchanges"]) > 2
    a = [{k: r[k].copy() for k in ("iw", "uw")} for r in a]
    b = run_world(tmp_path, "sum", window=48, extra=extra + ["direct"])
    assert "all_to_all" in str(b[0]["collective"]) and int(b[0]["exchanges"]) > 2
    assert np.array_equal(b[0]["iw"], b[1]["iw"])
    for r in range(2):
        assert np.array_equal(a[r]["iw"], b[r]["iw"]) and np.array_equal(a[r]["uw"], b[r]["uw"])


@pytest.mark.timeout(400)
def test_direct_exchange_three_ranks_padded_slices(tmp_path):
    """Three ranks: the 200 x 16 table does not divide into three slices (the last one is zero-padded), the owner sums in
    rank order where gloo's ring sums in ring order — replicas identical within a run, equal to the all-reduce run up to
    the last bit of a three-term fp32 sum."""
    make_problem(tmp_path, disjoint_items=False, epochs=2)
    a = run_world(tmp_path, "sum", window=48, world=3, extra=["overlap", "all_reduce"])
    a = [{k: r[k].copy() for k in ("iw", "uw")} for r in a]
    b = run_world(tmp_path, "sum", window=48, world=3, extra=["overlap", "direct"])
    assert "all_to_all" in str(b[0]["collective"])
    assert np.array_equal(b[0]["iw"], b[1]["iw"]) and np.array_equal(b[0]["iw"], b[2]["iw"])
    for r in range(3):
        np.testing.assert_allclose(b[r]["iw"], a[r]["iw"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(b[r]["uw"], a[r]["uw"], rtol=0, atol=2e-6)


@pytest.mark.timeout(400)
def test_mean_sync_averages_replicas(tmp_path):
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=False, epochs=1)
    T = clicks.shape[0]
    ranks = run_world(tmp_path, "mean", window=10 ** 9)                    # one sync, at the end of the epoch
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])
    # host-side restatement: train each shard alone on its own replica, then average (engine.cpp:366-375 intent)
    reps = []
    for r in range(2):
        shard, lo, hi = shard_clicks(clicks, U, 2, r)
        a = int(np.searchsorted(clicks[:, 0], lo))
        u, i = uw[lo:hi].copy(), iw.copy()
        e = orc.Engine(shard, u, i, num_negs=N, l_r=0.01, clip_val=1.0)
        e.lr_step()
        e.train_range(0, shard.shape[0], negs[a:a + shard.shape[0]])
        reps.append(i)
    np.testing.assert_allclose(ranks[0]["iw"], (reps[0] + reps[1]) / 2, rtol=0, atol=1e-7)


@pytest.mark.timeout(400)
def test_sum_sync_overlapping_rows_applies_both_deltas(tmp_path):
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=False, epochs=1)
    ranks = run_world(tmp_path, "sum", window=10 ** 9)
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])
    reps = []
    for r in range(2):
        shard, lo, hi = shard_clicks(clicks, U, 2, r)
        a = int(np.searchsorted(clicks[:, 0], lo))
        u, i = uw[lo:hi].copy(), iw.copy()
        e = orc.Engine(shard, u, i, num_negs=N, l_r=0.01, clip_val=1.0)
        e.lr_step()
        e.train_range(0, shard.shape[0], negs[a:a + shard.shape[0]])
        reps.append(i)
    want = iw + (reps[0] - iw) + (reps[1] - iw)
    np.testing.assert_allclose(ranks[0]["iw"], want, rtol=0, atol=3e-7)


@pytest.mark.timeout(400)
def test_aggregator_weights_are_averaged_with_the_item_table(tmp_path):
    """with behaviour aggregation on, W0 is all-reduce-averaged at the same points as the item table
    (train/engine.cpp:355-359, 366-375)."""
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=False, epochs=1)
    rng = np.random.default_rng(5)
    d, I, max_his = uw.shape[1], iw.shape[0], 6
    masks = rng.integers(1, max_his + 1, U).astype(np.uint64)
    his = rng.integers(0, I, (U, max_his)).astype(np.uint64)
    w0 = (rng.standard_normal((d, d)) * 0.1).astype(np.float32)
    prob = dict(np.load(tmp_path / "problem.npz"))
    np.savez(tmp_path / "problem.npz", his=his, masks=masks, w0=w0, **prob)
    ranks = run_world(tmp_path, "mean", window=10 ** 9)                    # one sync, at the end of the epoch
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"]) and np.array_equal(ranks[0]["w0"], ranks[1]["w0"])
    reps_i, reps_w = [], []
    for r in range(2):
        shard, lo, hi = shard_clicks(clicks, U, 2, r)
        a = int(np.searchsorted(clicks[:, 0], lo))
        u, i, w = uw[lo:hi].copy(), iw.copy(), w0.copy()
        e = orc.Engine(shard, u, i, num_negs=N, l_r=0.01, clip_val=1.0, his=np.ascontiguousarray(his[lo:hi]),
                       masks=np.ascontiguousarray(masks[lo:hi]), w0=w, use_aggregator=True)
        e.lr_step()
        e.train_range(0, shard.shape[0], negs[a:a + shard.shape[0]])
        reps_i.append(i)
        reps_w.append(w)
    assert not np.array_equal(reps_w[0], w0)                               # W0 did move
    np.testing.assert_allclose(ranks[0]["iw"], (reps_i[0] + reps_i[1]) / 2, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ranks[0]["w0"], (reps_w[0] + reps_w[1]) / 2, rtol=0, atol=1e-7)


@pytest.mark.timeout(400)
def test_overlapped_exchange_applies_the_other_ranks_deltas_one_window_late(tmp_path):
    """overlap=True: the all-reduce of window k is in flight while window k+1 trains; the other rank's delta of window k
    is added before window k+2 (`W += sum - mine`), the last exchange of the epoch is completed before the epoch ends
    (bit-identical replicas).  Host-side restatement with three windows per epoch."""
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=False, epochs=1)
    shards = [shard_clicks(clicks, U, 2, r) for r in range(2)]
    n_max = max(sh.shape[0] for sh, _, _ in shards)
    window = -(-n_max // 3)
    ranks = run_world(tmp_path, "sum", window=window, extra=["overlap"])
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])
    reps, engs, bases = [], [], []
    for r, (shard, lo, hi) in enumerate(shards):
        a = int(np.searchsorted(clicks[:, 0], lo))
        u, i = uw[lo:hi].copy(), iw.copy()
        e = orc.Engine(shard, u, i, num_negs=N, l_r=0.01, clip_val=1.0)
        e.lr_step()
        reps.append(i); engs.append(e); bases.append(a)

    def train(r, w):
        n = shards[r][0].shape[0]
        lo, hi = min(n, w * window), min(n, (w + 1) * window)
        if hi > lo:
            engs[r].train_range(lo, hi, negs[bases[r] + lo:bases[r] + hi])

    for r in range(2):
        train(r, 0)
    first = [reps[r] - iw for r in range(2)]                 # window-0 deltas, exchanged while window 1 trains
    for r in range(2):
        train(r, 1)
    for r in range(2):
        reps[r] += first[1 - r]                              # ... and applied before window 2
    for r in range(2):
        train(r, 2)
    ref = iw + first[0] + first[1]
    want = ref + (reps[0] - ref) + (reps[1] - ref)           # the closing (blocking) exchange
    np.testing.assert_allclose(ranks[0]["iw"], want, rtol=0, atol=1e-6)


@pytest.mark.timeout(400)
def test_deferred_closing_exchange_drains_to_identical_replicas(tmp_path):
    """bench.py's steady state (overlap + defer_final): the closing exchange of an epoch is left in flight across the epoch
    boundary; finalize() completes it and makes the replicas bit-identical.  Nothing is lost on the way: with shards that
    touch disjoint item rows the result is single-process training of the whole list."""
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=True, epochs=3)
    ranks = run_world(tmp_path, "sum", window=97, extra=["defer"])
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])
    u1, i1 = uw.copy(), iw.copy()
    ref = orc.Engine(clicks, u1, i1, num_negs=N, l_r=0.01, clip_val=1.0)
    for _ in range(3):
        ref.lr_step()
        ref.train_range(0, clicks.shape[0], negs)
        ref.zero_grad()
        ref.epoch = ref.epoch + 1
    np.testing.assert_allclose(ranks[0]["iw"], i1, rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([ranks[0]["uw"], ranks[1]["uw"]]), u1, rtol=0, atol=2e-5)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("schedule", ["blocking", "defer"])
def test_exchange_every_second_epoch_loses_nothing(tmp_path, schedule):
    """bench.py's schedule at 8 GPUs (ItemSync(epochs_per_exchange=2)): whole epochs between exchanges, finalize() closes.
    Every rank takes the same decision at every boundary (two exchanges for four epochs), the replicas end bit-identical,
    and with shards that touch disjoint item rows the result is single-process training of the whole list."""
    clicks, negs, uw, iw, U, N = make_problem(tmp_path, disjoint_items=True, epochs=4)
    ranks = run_world(tmp_path, "sum", window=0, extra=[schedule, "all_reduce", "2"])
    assert np.array_equal(ranks[0]["iw"], ranks[1]["iw"])
    assert int(ranks[0]["exchanges"]) == int(ranks[1]["exchanges"]) <= 3          # epochs 2 and 4 (+ the closing one of finalize)
    u1, i1 = uw.copy(), iw.copy()
    ref = orc.Engine(clicks, u1, i1, num_negs=N, l_r=0.01, clip_val=1.0)
    for _ in range(4):
        ref.lr_step()
        ref.train_range(0, clicks.shape[0], negs)
        ref.zero_grad()
        ref.epoch = ref.epoch + 1
    np.testing.assert_allclose(ranks[0]["iw"], i1, rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([ranks[0]["uw"], ranks[1]["uw"]]), u1, rtol=0, atol=2e-5)
