"""Oracle forward_backward vs the float64 model and vs finite differences ("parity unpinned" by reference
fixtures for this part: the reference has no tests and its Eigen dependency is absent — see DESIGN.md)."""
import numpy as np
import pytest

from oracle import cf_oracle as orc
from tests.f64_model import loss_only_f64, step_f64


def make_tables(rng, U, I, d, scale=0.01):
    return (rng.standard_normal((U, d)) * scale).astype(np.float32), (rng.standard_normal((I, d)) * scale).astype(np.float32)


@pytest.mark.parametrize("d,N,clip", [(64, 16, 1.0), (128, 64, 0.1), (32, 4, 1.0), (256, 100, 1.0), (20, 3, 0.05)])
def test_single_step_matches_f64(d, N, clip):
    rng = np.random.default_rng(d * 1000 + N)
    U, I = 5, 300
    uw, iw = make_tables(rng, U, I, d)
    clicks = np.array([[2, 7]], dtype=np.uint64)
    eng = orc.Engine(clicks, uw, iw, num_negs=N, clip_val=clip, l_r=0.01)
    # warm persistent gradient rows so the "G <- clip(G + g)" path is exercised
    eng.user_grads()[:] = (rng.standard_normal((U, d)) * 0.02).astype(np.float32)
    eng.item_grads()[:] = (rng.standard_normal((I, d)) * 0.02).astype(np.float32)
    uw64, iw64 = uw.astype(np.float64), iw.astype(np.float64)
    ug64, ig64 = eng.user_grads().astype(np.float64), eng.item_grads().astype(np.float64)
    negs = rng.choice(np.setdiff1d(np.arange(I), [7]), size=N, replace=False).astype(np.uint64)
    loss = eng.forward_backward(2, 7, negs)
    loss64 = step_f64(uw64, iw64, ug64, ig64, 2, 7, [int(x) for x in negs], clip=clip, lr=float(np.float32(0.01)))
    assert abs(loss - loss64) <= 2e-6 * max(1.0, abs(loss64))
    for a, b in [(uw, uw64), (iw, iw64), (eng.user_grads(), ug64), (eng.item_grads(), ig64)]:
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=5e-6)


def test_gradient_is_the_derivative_of_the_loss():
    """With zero persistent grads and a huge clip, G rows after one step hold dLoss/drow: compare to central
    finite differences of the float64 loss."""
    rng = np.random.default_rng(7)
    d, N, U, I = 16, 5, 3, 40
    uw, iw = make_tables(rng, U, I, d, scale=0.1)
    uw0, iw0 = uw.astype(np.float64), iw.astype(np.float64)
    negs = np.array([3, 9, 11, 20, 33], dtype=np.uint64)
    eng = orc.Engine(np.array([[1, 5]], dtype=np.uint64), uw, iw, num_negs=N, clip_val=1e9, l_r=0.0)
    eng.forward_backward(1, 5, negs)
    h = 1e-6

    def fd(get_args):
        g = np.zeros(d)
        for j in range(d):
            lp = loss_only_f64(*get_args(+h, j))
            lm = loss_only_f64(*get_args(-h, j))
            g[j] = (lp - lm) / (2 * h)
        return g

    def pert_u(dh, j):
        u = uw0[1].copy(); u[j] += dh
        return u, iw0[5], iw0[negs.astype(int)]

    def pert_p(dh, j):
        p = iw0[5].copy(); p[j] += dh
        return uw0[1], p, iw0[negs.astype(int)]

    def pert_n2(dh, j):
        nr = iw0[negs.astype(int)].copy(); nr[2, j] += dh
        return uw0[1], iw0[5], nr

    np.testing.assert_allclose(eng.user_grads()[1], fd(pert_u), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(eng.item_grads()[5], fd(pert_p), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(eng.item_grads()[11], fd(pert_n2), rtol=2e-4, atol=1e-5)


def test_duplicate_negatives_last_writer_wins():
    """SURVEY appendix 4 / matrix_factorization.cpp:72-73,147-149: duplicate ids each write their own stale copy
    of W (last writer wins) while G accumulates all contributions."""
    rng = np.random.default_rng(3)
    d, N = 8, 4
    uw, iw = make_tables(rng, 2, 10, d, scale=0.1)
    iw_before = iw.copy()
    eng = orc.Engine(np.array([[0, 1]], dtype=np.uint64), uw, iw, num_negs=N, clip_val=1e9, l_r=0.5)
    eng.forward_backward(0, 1, np.array([4, 4, 6, 4], dtype=np.uint64))
    g4 = eng.item_grads()[4].copy()
    # single-occurrence gradient for comparison
    uw2, iw2 = make_tables(np.random.default_rng(3), 2, 10, d, scale=0.1)
    eng2 = orc.Engine(np.array([[0, 1]], dtype=np.uint64), uw2, iw2, num_negs=N, clip_val=1e9, l_r=0.5)
    eng2.forward_backward(0, 1, np.array([4, 5, 6, 7], dtype=np.uint64))
    # (softmax weights differ between the two neg sets, so compare structure, not values)
    assert np.allclose(iw[4], iw_before[4] - np.float32(0.5) * g4, rtol=1e-6)  # W = stale copy - lr * accumulated G


def test_aggregator_step_matches_f64():
    rng = np.random.default_rng(11)
    d, N, U, I, H = 16, 4, 3, 50, 6
    uw, iw = make_tables(rng, U, I, d, scale=0.1)
    w0 = (rng.standard_normal((d, d)) * 0.1).astype(np.float32)
    his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
    masks = np.array([[H], [3], [1]], dtype=np.uint64)
    clicks = np.array([[1, 9]], dtype=np.uint64)
    eng = orc.Engine(clicks, uw, iw, num_negs=N, his=his, masks=masks, w0=w0, use_aggregator=True, clip_val=1.0,
                     l_r=0.01)
    uw64, iw64, w064 = uw.astype(np.float64), iw.astype(np.float64), w0.astype(np.float64)
    ug64, ig64 = np.zeros_like(uw64), np.zeros_like(iw64)
    agg = dict(w0=w064, gamma=float(np.float32(0.4)), agg_lr=float(np.float32(0.01)),
               state=dict(iteration=0, accu=np.zeros((d, d))))
    negs_all = rng.integers(10, I, size=(40, N)).astype(np.uint64)
    for t in range(40):  # crosses the 32-call W0 update (behavior_aggregators.cpp:141-146)
        user = t % U
        agg["his"] = [int(x) for x in his[user, :int(masks[user, 0])]]
        l32 = eng.forward_backward(user, 9, negs_all[t])
        l64 = step_f64(uw64, iw64, ug64, ig64, user, 9, [int(x) for x in negs_all[t]], clip=1.0,
                       lr=float(np.float32(0.01)), agg=agg)
        assert abs(l32 - l64) < 1e-4
    assert np.abs(w0 - w064).max() < 1e-6 and np.abs(w0.astype(np.float64) - w064).max() > 0  # W0 did move
    np.testing.assert_allclose(uw, uw64, rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(iw, iw64, rtol=1e-3, atol=2e-6)


def test_epoch_loop_single_thread_is_deterministic_and_matches_manual_walk():
    """train/engine.cpp:294-342 at 1 thread == sampler(seed=(epoch+1)*0) + serial forward_backward + zero_grad."""
    rng = np.random.default_rng(5)
    d, N, U, I, T = 16, 4, 20, 60, 300
    users = np.sort(rng.integers(0, U, size=T)).astype(np.uint64)
    clicks = np.stack([users, rng.integers(0, I, size=T).astype(np.uint64)], axis=1).copy()
    uw, iw = make_tables(rng, U, I, d, scale=0.1)
    uw2, iw2 = uw.copy(), iw.copy()
    eng = orc.Engine(clicks, uw, iw, num_negs=N, milestones=(1,), l_r=0.05)
    loss0, negs0 = eng.train_one_epoch(num_threads=1, record_negs=True)
    loss1 = eng.train_one_epoch(num_threads=1)
    assert eng.epoch == 2 and abs(eng.l_r - 0.005) < 1e-9  # StepLR fired at epoch 1 (optimizer.cpp:24-30)
    assert not eng.user_grads().any() and not eng.item_grads().any()  # zero_grad at epoch end
    # manual restatement using the pinned sampler + train_range
    eng2 = orc.Engine(clicks, uw2, iw2, num_negs=N, milestones=(1,), l_r=0.05)
    s = orc.Sampler(I, N, 0)
    negs = np.stack([s.ignore_pos_sampling(int(u), int(p)) for u, p in clicks])
    assert np.array_equal(negs, negs0)
    lsum = eng2.train_range(0, T, negs)
    assert abs(lsum / T - loss0) < 1e-6
    assert loss1 < loss0
