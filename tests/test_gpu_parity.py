"""-m gpu parity tests: the HIP path (through the C ABI, heat_amd.abi) against the CPU oracle on identical
seeded inputs.  fp32 tolerances are stated per test; integer work (negative ids, top-k ids) is bit-exact."""
import numpy as np
import pytest

from heat_amd import abi
from heat_amd.cf import synthetic
from oracle import cf_oracle as orc
from tests import philox_ref

pytestmark = pytest.mark.gpu


# Conditioning note (DESIGN.md "Parity"): at the reference's own hyper-parameters (N(0,0.01^2) init, lr 0.01, clip 1.0)
# one update moves an element by up to lr*clip = 0.01 = the embedding scale and the un-clipped coordinates have
# d(grad)/dW ~ lg/|u|^2 ~ 2000, i.e. lr * d(grad)/dW >> 1: two fp32 implementations that differ in the last bit
# (Eigen vs this oracle vs the GPU) separate exponentially and are O(1) apart after a few hundred dependent steps.
# Trajectory-level comparisons therefore run in a well-conditioned regime (scale 0.1: lr * d(grad)/dW ~ 0.2), where
# thousands of dependent steps agree to ~1e-5; the reference regime is covered over short horizons and statistically.
def small_problem(U, I, T, d, seed, scale=0.1):
    rng = np.random.default_rng(seed)
    g = synthetic.make_graph(U, I, T, seed=seed, with_test=False)
    uw = (rng.standard_normal((U, d)) * scale).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * scale).astype(np.float32)
    return g.clicks, uw, iw


def run_pair(clicks, uw, iw, N, negs, *, clip=1.0, lr=0.01, coherence=abi.COHERENCE_DEFAULT, epochs_of_ranges=1):
    """Serial GPU walk vs oracle walk with identical caller-fed negatives; returns both states."""
    T = clicks.shape[0]
    uw_g, iw_g = uw.copy(), iw.copy()
    uw_o, iw_o = uw.copy(), iw.copy()
    eng = abi.Engine(clicks, uw_g, iw_g, num_negs=N, clip_val=clip, l_r=lr, flags=abi.FLAG_SERIAL, coherence=coherence)
    ora = orc.Engine(clicks, uw_o, iw_o, num_negs=N, clip_val=clip, l_r=lr)
    lg = lo = 0.0
    for _ in range(epochs_of_ranges):
        lg += eng.train_range(0, T, negs)
        lo += ora.train_range(0, T, negs)
    eng.sync_to_host()
    eng.close()
    return (uw_g, iw_g, lg), (uw_o, iw_o, lo)


# fp32 tolerance: both sides are fp32 with different (valid) summation orders and FMA contraction on the GPU;
# after T sequential dependent steps the tables agree to ~1e-5 relative to the embedding scale.
def assert_tables_close(a, b, scale, rtol=2e-4):
    err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    assert err <= rtol * scale, f"max abs err {err:.3e} > {rtol * scale:.3e}"


@pytest.mark.parametrize("d,N,U,I,T", [
    (64, 16, 60, 400, 3000),     # AmazonBooks/Gowalla kernel variant <16,4>
    (128, 64, 40, 600, 800),     # Yelp18 config: <32,4> x 8 waves per workgroup
    (32, 4, 30, 200, 1500),      # <8,1>
    (256, 16, 20, 300, 600),     # <64,16>
    (64, 5, 30, 200, 1000),      # masked negative slots (N not a multiple of rows-per-instruction)
    (20, 3, 25, 300, 800),       # masked columns (emb_dim/4 = 5 of 8 lanes)
    (64, 100, 20, 2000, 300),    # <16,16> x 2 waves per workgroup
    (256, 100, 12, 3000, 150),   # synthetic-HBM config: <64,13> x 8 waves per workgroup
    (128, 100, 12, 3000, 150),   # <32,16> x 4 waves (two id registers per lane)
])
def test_serial_walk_matches_oracle(d, N, U, I, T):
    clicks, uw, iw = small_problem(U, I, T, d, seed=d * 7 + N)
    rng = np.random.default_rng(1)
    negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
    (ug, ig, lg), (uo, io, lo) = run_pair(clicks, uw, iw, N, negs)
    assert abs(lg - lo) <= 1e-5 * abs(lo)
    assert_tables_close(ug, uo, scale=np.abs(uo).max())
    assert_tables_close(ig, io, scale=np.abs(io).max())
    assert not np.array_equal(uo, uw)  # something was trained


@pytest.mark.parametrize("coherence", [abi.COHERENCE_PLAIN, abi.COHERENCE_DEVICE])
def test_collision_heavy_walk_matches_oracle(coherence):
    """17 items only: duplicate negatives, negative == positive and overlaps between consecutive interactions occur
    in almost every step (matrix_factorization.cpp:72-73,147-149,171-174 ordering quirks)."""
    d, N, U, I, T = 64, 16, 10, 17, 1500
    rng = np.random.default_rng(2)
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)  # positives allowed among negatives
    (ug, ig, lg), (uo, io, lo) = run_pair(clicks, uw, iw, N, negs, clip=0.05, lr=0.01, coherence=coherence)
    assert abs(lg - lo) <= 2e-4 * abs(lo)
    assert_tables_close(ug, uo, scale=np.abs(uo).max(), rtol=5e-4)
    assert_tables_close(ig, io, scale=np.abs(io).max(), rtol=5e-4)


def test_reference_hyperparameters_short_horizon():
    """N(0,0.01^2) init, lr 0.01, clip 1.0 (AmazonBooks config0.yaml): 12 dependent steps agree to 1e-4 relative."""
    d, N, U, I, T = 64, 16, 4, 400, 12
    clicks, uw, iw = small_problem(U, I, T, d, seed=77, scale=0.01)
    rng = np.random.default_rng(3)
    negs = np.stack([rng.choice(I, N, replace=False) for _ in range(T)]).astype(np.uint64)
    (ug, ig, lg), (uo, io, lo) = run_pair(clicks, uw, iw, N, negs)
    assert abs(lg - lo) <= 1e-5 * abs(lo)
    assert_tables_close(ug, uo, scale=np.abs(uo).max(), rtol=5e-4)
    assert_tables_close(ig, io, scale=np.abs(io).max(), rtol=5e-4)


@pytest.mark.parametrize("d,N,H", [(64, 16, 100),     # AmazonBooks config: <16,4> x 1 spread over 4 waves (<16,1> x 4)
                                   (128, 64, 50),     # Yelp18 yaml with aggregation: <32,32> x 1 spread to <32,8> x 4
                                   (64, 4, 7), (128, 16, 40), (32, 8, 100),
                                   (256, 100, 100),   # synthetic 10 M x 1 M config: <64,13> x 8 waves, W0 (256 KB) read from L2
                                   (128, 100, 60),    # <32,16> x 4 waves, W0 copy in LDS
                                   (64, 100, 200),    # <16,16> x 2 waves, history longer than 128
                                   (256, 16, 130)])   # single wave at d = 256: W0 from L2, three id registers
def test_aggregator_serial_walk_matches_oracle(d, N, H):
    """ACCL mode (behavior_aggregators.cpp:51-153): history mean -> W0 -> in-place blend of the user row, W0 gradient
    accumulated per worker and applied every 32 calls.  Serial GPU walk vs the oracle, crossing several W0 updates;
    multi-wave workgroups split the history gather and the d x d product over their waves."""
    U, I, T = 12, 400, 200
    rng = np.random.default_rng(d + N + H)
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    w0 = (rng.standard_normal((d, d)) * 0.1).astype(np.float32)
    his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
    masks = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
    masks[0, 0] = H                                             # one user with a full history
    negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
    ug, ig, wg = uw.copy(), iw.copy(), w0.copy()
    uo, io, wo = uw.copy(), iw.copy(), w0.copy()
    eng = abi.Engine(clicks, ug, ig, num_negs=N, his=his, masks=masks, w0=wg, use_aggregator=True, flags=abi.FLAG_SERIAL,
                     l_r=0.01, clip_val=1.0)
    ora = orc.Engine(clicks, uo, io, num_negs=N, his=his, masks=masks, w0=wo, use_aggregator=True, l_r=0.01, clip_val=1.0)
    lg = eng.train_range(0, T, negs)
    lo = ora.train_range(0, T, negs)
    eng.sync_to_host()
    eng.close()
    assert abs(lg - lo) <= 1e-5 * abs(lo)
    assert not np.array_equal(wo, w0)                           # W0 moved (T/32 updates)
    assert_tables_close(wg, wo, scale=np.abs(wo).max(), rtol=1e-4)
    assert_tables_close(ug, uo, scale=np.abs(uo).max(), rtol=5e-4)
    assert_tables_close(ig, io, scale=np.abs(io).max(), rtol=5e-4)


@pytest.mark.parametrize("d,N", [(64, 16), (128, 100)])
def test_aggregator_windows_compose_to_the_whole_range(d, N):
    """The reference's per-worker aggregator lives for a whole epoch (behavior_aggregators.cpp:31,139-146: call counter and
    the W0 gradient accumulated since the last multiple of 32).  An epoch cut into windows (multi-GPU exchanges) therefore
    carries that state from launch to launch: train_range(0,a) + train_range(a,n) must equal train_range(0,n), W0 included,
    for cuts that are not multiples of 32."""
    U, I, T, H = 10, 300, 150, 30
    rng = np.random.default_rng(d + N)
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    w0 = (rng.standard_normal((d, d)) * 0.1).astype(np.float32)
    his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
    masks = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
    negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
    outs = []
    for cuts in ([0, T], [0, 7, 50, 83, T]):
        a, b, w = uw.copy(), iw.copy(), w0.copy()
        eng = abi.Engine(clicks, a, b, num_negs=N, his=his, masks=masks, w0=w, use_aggregator=True, flags=abi.FLAG_SERIAL)
        eng.begin_epoch()
        loss = sum(eng.train_range(lo, hi, negs[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:]))
        eng.sync_to_host()
        eng.close()
        outs.append((a, b, w, loss))
    assert not np.array_equal(outs[0][2], w0)                       # W0 was updated (T // 32 = 4 times)
    for k in range(3):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert abs(outs[0][3] - outs[1][3]) < 1e-9 * abs(outs[0][3])


def test_aggregator_multiwave_hogwild_trains():
    """The multi-wave aggregator outside the serial mode (several streams, atomics on the shared W0): the synthetic
    10 M x 1 M config's kernel variant (d = 256, 100 negatives, history 100) on a small graph — finite, and learning."""
    g = synthetic.make_graph(300, 4000, 12000, seed=3, with_test=False)
    d, N = 256, 100
    uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=3, std=0.05)
    his, masks = synthetic.make_history(g, 100, seed=3)
    w0 = (np.random.default_rng(3).standard_normal((d, d)) * 0.05).astype(np.float32)
    w0_before = w0.copy()
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, his=his, masks=masks, w0=w0, use_aggregator=True, seed=3, num_streams=16)
    assert "<64,13,16,8>" in eng.kernel_name, eng.kernel_name
    losses = [eng.train_one_epoch() for _ in range(3)]
    eng.close()
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert np.isfinite(uw).all() and np.isfinite(iw).all() and np.isfinite(w0).all() and not np.array_equal(w0, w0_before)


def test_aggregator_device_mode_equals_host_mode():
    """Device-resident ACCL (heat_cf_engine_create_device with history, lengths and W0 as device pointers, the form the
    sharded trainer uses) runs the same kernels as the host-mode engine: identical tables and W0 after two epochs."""
    import torch
    d, N, H, U, I, T = 64, 8, 20, 30, 500, 900
    rng = np.random.default_rng(11)
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    w0 = (rng.standard_normal((d, d)) * 0.1).astype(np.float32)
    his = rng.integers(0, I, size=(U, H)).astype(np.uint64)
    masks = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
    uh, ih, wh = uw.copy(), iw.copy(), w0.copy()
    host = abi.Engine(clicks, uh, ih, num_negs=N, his=his, masks=masks, w0=wh, use_aggregator=True, seed=4,
                      flags=abi.FLAG_SERIAL)
    losses_h = [host.train_one_epoch() for _ in range(2)]
    host.close()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        t = {k: torch.from_numpy(v).to(dev) for k, v in dict(c=clicks.view(np.int64), u=uw, i=iw, w=w0, h=his.view(np.int64),
                                                             m=masks.view(np.int64)).items()}
        common = dict(num_users=U, num_items=I, emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=4,
                      flags=abi.FLAG_SERIAL, keep=tuple(t.values()), max_his=H, use_aggregator=1)
        eng = abi.Engine.from_device(t["c"].data_ptr(), T, t["u"].data_ptr(), t["i"].data_ptr(), his_ptr=t["h"].data_ptr(),
                                     masks_ptr=t["m"].data_ptr(), w0_ptr=t["w"].data_ptr(), **common)
        losses_d = [eng.train_one_epoch() for _ in range(2)]
        side.synchronize()
        eng.close()
        assert losses_d == losses_h
        assert np.array_equal(t["u"].cpu().numpy(), uh) and np.array_equal(t["i"].cpu().numpy(), ih)
        assert np.array_equal(t["w"].cpu().numpy(), wh) and not np.array_equal(wh, w0)
        # the device-side checks mirror the host ones
        bad_m = torch.full_like(t["m"], H + 1)
        with pytest.raises(ValueError):
            abi.Engine.from_device(t["c"].data_ptr(), T, t["u"].data_ptr(), t["i"].data_ptr(), his_ptr=t["h"].data_ptr(),
                                   masks_ptr=bad_m.data_ptr(), w0_ptr=t["w"].data_ptr(), **common)
        zero_m = torch.zeros_like(t["m"])
        with pytest.raises(ValueError):
            abi.Engine.from_device(t["c"].data_ptr(), T, t["u"].data_ptr(), t["i"].data_ptr(), his_ptr=t["h"].data_ptr(),
                                   masks_ptr=zero_m.data_ptr(), w0_ptr=t["w"].data_ptr(), **common)
        bad_h = torch.full_like(t["h"], I)
        with pytest.raises(ValueError):
            abi.Engine.from_device(t["c"].data_ptr(), T, t["u"].data_ptr(), t["i"].data_ptr(), his_ptr=bad_h.data_ptr(),
                                   masks_ptr=t["m"].data_ptr(), w0_ptr=t["w"].data_ptr(), **common)
        side.synchronize()


def test_sampler_is_philox_bit_exact():
    d, N, U, I, T = 64, 16, 50, 1000, 2000
    clicks, uw, iw = small_problem(U, I, T, d, seed=9)
    eng = abi.Engine(clicks, uw, iw, num_negs=N, seed=2022, flags=abi.FLAG_SERIAL)
    got = eng.sample_negatives(0, T)
    key = philox_ref.epoch_key(2022, 0)
    want = philox_ref.negatives(clicks, 0, T, N, I, key, per_block=((T + 63) // 64) * 64)
    assert np.array_equal(got, want)
    assert got.max() < I
    # a different epoch draws a different stream
    eng.epoch = 3
    got3 = eng.sample_negatives(0, 64)
    want3 = philox_ref.negatives(clicks, 0, 64, N, I, philox_ref.epoch_key(2022, 3), per_block=64)
    assert np.array_equal(got3, want3) and not np.array_equal(got3, got[:64])
    eng.close()


def test_philox_equals_hiprand_device_api():
    """north_star names hiprand for the on-GPU sampler.  The kernel hand-rolls Philox4x32-10 so that the generator state
    stays in registers; oracle/hiprand_kat.hip runs hipRAND's own device API (hiprand_kernel.h: hiprand_init(key, idx,
    4*slot) + hiprand4) next to the kernel's philox_draw64 on the GPU: the 64-bit draws must be bit-identical, and the ids
    the product engine reports must be mulhi64(hipRAND draw, num_items)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libhiprand_kat.so")
    assert os.path.exists(path), "oracle/libhiprand_kat.so missing: run __graft_entry__.build() (make -C oracle hiprand)"
    abi.load()                                                   # one HIP runtime per process (torch's is loaded first)
    kat = C.CDLL(path)
    kat.hiprand_kat_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    d, N, U, I, T = 64, 100, 20, 90001, 300
    clicks, uw, iw = small_problem(U, I, T, d, seed=5)
    for seed, epoch, base in [(2022, 0, 0), (7, 3, 1 << 33)]:
        key = philox_ref.epoch_key(seed, epoch)
        a = np.empty((T, N), dtype=np.uint64)
        b = np.empty((T, N), dtype=np.uint64)
        assert kat.hiprand_kat_draws(key, base, T, N, a.ctypes.data, b.ctypes.data) == 0
        assert np.array_equal(a, b)                              # hipRAND device API == the kernel's generator, bit for bit
        idx = base + np.arange(T, dtype=np.uint64)
        assert a.ravel().tolist() == philox_ref._draw64(np.tile(np.arange(N, dtype=np.uint32), T), np.repeat(idx, N), key)
        eng = abi.Engine(clicks, uw, iw, num_negs=N, seed=seed, sample_index_base=base,
                         flags=abi.FLAG_SERIAL | abi.FLAG_SAMPLING_CALL)   # sampling(): raw draws, no keep-slot rule
        eng.epoch = epoch
        got = eng.sample_negatives(0, T)
        eng.close()
        want = ((a.astype(object) * I) >> 64).astype(np.uint64)  # mulhi64(draw, num_items) in exact integer arithmetic
        assert np.array_equal(got, want)


def test_sampler_ignore_pos_keeps_previous_slot():
    """3 items: every draw hits the positive with probability 1/3, so the 'slot left unchanged' rule fires often."""
    d, N, T = 64, 16, 640
    rng = np.random.default_rng(4)
    clicks = np.stack([np.sort(rng.integers(0, 5, T)), rng.integers(0, 3, T)], axis=1).astype(np.uint64)
    uw, iw = synthetic.init_embeddings(5, 3, d)
    eng = abi.Engine(clicks, uw, iw, num_negs=N, seed=7, flags=abi.FLAG_SERIAL)
    got = eng.sample_negatives(0, T)
    want = philox_ref.negatives(clicks, 0, T, N, 3, philox_ref.epoch_key(7, 0), per_block=640)
    assert np.array_equal(got, want)
    raw = philox_ref.raw_negatives(np.arange(T, dtype=np.uint64), N, 3, philox_ref.epoch_key(7, 0))
    assert (raw == clicks[:, 1:2]).mean() > 0.2  # the rule was exercised
    eng.close()


def test_random_tile_sampler_bit_exact_and_tile_bounded():
    """neg_sampler=1 with the sampling() call (engine.cpp:333 variant): ids equal the numpy restatement, and inside one
    refresh window a stream draws from at most tile_size distinct items (random_tile_negative_sampler.cpp:23-45)."""
    d, N, U, I, T = 64, 16, 50, 5000, 700
    clicks, uw, iw = small_problem(U, I, T, d, seed=9)
    eng = abi.Engine(clicks, uw, iw, num_negs=N, seed=11, neg_sampler=1, tile_size=32, refresh_interval=100,
                     flags=abi.FLAG_SERIAL | abi.FLAG_SAMPLING_CALL)
    got = eng.sample_negatives(0, T)
    want = philox_ref.tile_negatives(0, T, N, I, philox_ref.epoch_key(11, 0), 32, 100, per_block=((T + 63) // 64) * 64)
    assert np.array_equal(got, want)
    for w in range(0, T, 100):
        assert np.unique(got[w:w + 100]).size <= 32
    assert np.unique(got).size > 32 * 3          # the tile was refreshed
    # ignore_pos_sampling of the tile sampler does not use the tile (random_tile_negative_sampler.cpp:47-57)
    eng2 = abi.Engine(clicks, uw, iw, num_negs=N, seed=11, neg_sampler=1, tile_size=32, refresh_interval=100,
                      flags=abi.FLAG_SERIAL)
    uni = abi.Engine(clicks, uw, iw, num_negs=N, seed=11, flags=abi.FLAG_SERIAL)
    assert np.array_equal(eng2.sample_negatives(0, T), uni.sample_negatives(0, T))
    loss = eng.train_range(0, T)                  # and it trains
    assert np.isfinite(loss)
    for e in (eng, eng2, uni):
        e.close()


def test_tile_in_lds_single_stream_equals_tile_in_global_memory():
    """with the random-tile sampler the tile's accumulated weight deltas live in LDS (12 streams per
    workgroup share them; forward = W_global + delta, backward adds the step to the delta, the deltas are flushed to the
    table by float atomics at the end of the launch).  One stream, positives chosen outside its tile: the resident kernel
    must leave the same tables as the kernel that writes every step straight to the table (same tile, same ids), up to
    the rounding of (W + delta) vs an in-place W."""
    d, N, U, I, T, tile = 64, 16, 12, 5000, 400, 64
    seed = 11
    key = philox_ref.epoch_key(seed, 0)
    in_tile = set(philox_ref.tile_entries(tile, I, key, owner=0).tolist())
    rng = np.random.default_rng(3)
    outside = np.array([i for i in range(I) if i not in in_tile])
    clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.choice(outside, T)], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    out = {}
    for name, extra in (("lds", abi.FLAG_TILE_LDS), ("global", 0)):
        a, b = uw.copy(), iw.copy()
        eng = abi.Engine(clicks, a, b, num_negs=N, seed=seed, neg_sampler=1, tile_size=tile, refresh_interval=100000, num_streams=1,
                         update_mode=abi.UPDATE_ATOMIC_POS, flags=abi.FLAG_SAMPLING_CALL | extra)
        assert ("tile-in-lds" in eng.kernel_name) == (name == "lds"), eng.kernel_name
        loss = eng.train_one_epoch()
        eng.close()
        out[name] = (a, b, loss)
    touched = np.abs(out["global"][1] - iw).max(axis=1) > 0
    assert touched.sum() <= tile + T and touched[list(in_tile)].sum() > tile // 2     # the negatives came from the tile
    assert abs(out["lds"][2] - out["global"][2]) <= 1e-5 * abs(out["global"][2])
    assert_tables_close(out["lds"][0], out["global"][0], scale=np.abs(uw).max(), rtol=1e-4)
    assert_tables_close(out["lds"][1], out["global"][1], scale=np.abs(iw).max(), rtol=1e-4)


def test_tile_in_lds_recall_ndcg_amazonbooks_shape():
    """The resident tile changes who sees a negative update when (the other workgroups: one launch late) and how many
    streams share a tile (12 instead of 1); paper Table 6 reports a Recall drop of at most 1e-3 for random tiling.  At the
    AmazonBooks config (tile 512, refresh 8192, 5 epochs; means over three seeds): Recall@20 / NDCG@20 with the tile in LDS
    within 1e-3 of the same sampler writing straight to the table, and within 2e-3 of the uniform sampler."""
    import types
    from heat_amd.cf import metrics
    g, d, N = synthetic.make_named("amazonbooks")
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]
    runs = {}
    for seed in (2022, 7, 99):     # means over three seeds: one configuration run twice differs by up to 8e-4 in Recall@20
        uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        for name, kw in (("uniform", dict()),
                         ("tile-global", dict(neg_sampler=1, tile_size=512, refresh_interval=8192, flags=abi.FLAG_SAMPLING_CALL | abi.FLAG_LAZY_SYNC)),
                         ("tile-lds", dict(neg_sampler=1, tile_size=512, refresh_interval=8192, flags=abi.FLAG_SAMPLING_CALL | abi.FLAG_TILE_LDS | abi.FLAG_LAZY_SYNC))):
            kw.setdefault("flags", abi.FLAG_LAZY_SYNC)
            uw, iw = uw0.copy(), iw0.copy()
            eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, **kw)
            losses = [eng.train_one_epoch() for _ in range(5)]
            ms_epoch, n = eng.kernel_time()
            eng.sync_to_host()
            kname = eng.kernel_name
            top = eng.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
            eng.close()
            r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
            runs.setdefault(name, []).append((r[ms[0]], r[ms[1]]))
            print(f"seed {seed} {name:12s} {kname}: {ms_epoch / n:.3f} ms/epoch, losses {[round(x, 4) for x in losses]}, Recall@20 {r[ms[0]]:.5f} NDCG@20 {r[ms[1]]:.5f}")
    res = {name: tuple(np.mean(v, axis=0)) for name, v in runs.items()}
    print("means", res)
    assert abs(res["tile-lds"][0] - res["tile-global"][0]) <= 1e-3 and abs(res["tile-lds"][1] - res["tile-global"][1]) <= 1e-3, res
    assert abs(res["tile-lds"][0] - res["uniform"][0]) <= 2e-3 and abs(res["tile-lds"][1] - res["uniform"][1]) <= 2e-3, res


def test_tile_sampler_matches_the_reference_algorithm_at_matched_workers():
    """SURVEY 8 a4 / f2 against an oracle.  The random-tile sampler keeps ONE tile per worker for
    refresh_interval of that worker's calls (random_tile_negative_sampler.cpp:22-45), so what it does to training depends on
    the worker count; a "worker" of the reference is one interaction stream here (the table-writing kernel: one wave, one tile).
    Matched workers: 64 GPU streams against the reference's algorithm with 64 workers — the oracle's forward_backward and the
    oracle's own tile sampler (both pinned by the reference KATs), driven as 64 lockstep workers (tests/tools/stream_sim.c;
    committed fixture tests/golden/tile_stream_model.json), AmazonBooks shape, tile 512 / refresh 8192, 5 epochs, 3 seeds.
    One deliberate difference is part of the fixture: every (epoch, worker) gets its own sampler seed.  The reference seeds
    worker t with (epoch + 1) * t (train/engine.cpp:302): worker 0 draws the same tiles in every epoch and seeds repeat
    across (epoch, worker) pairs, and its tile and index generators share one seed.  The on-GPU sampler the north star asks
    for has no such repeats (a fresh tile per (epoch, stream, tile epoch)), and with that one change the reference's
    algorithm reproduces the GPU's loss curve epoch by epoch (0.96 1.17 1.08 1.00 0.94); with the literal seed rule the
    OpenMP oracle at 64 threads ends at Recall@20 -3e-4 / NDCG@20 +2.7e-3 from the GPU and a different curve
    (0.97 1.06 1.01 0.93 0.93) — profiles/r03_tile_matched_workers.txt."""
    import json
    import os
    import types
    from heat_amd.cf import metrics
    seeds = (2022, 7, 99)
    g, d, N = synthetic.make_named("amazonbooks")
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]
    gpu = []
    for seed in seeds:
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, neg_sampler=1, tile_size=512, refresh_interval=8192, num_streams=64,
                         flags=abi.FLAG_SAMPLING_CALL | abi.FLAG_LAZY_SYNC)
        losses = [eng.train_one_epoch() for _ in range(5)]
        eng.sync_to_host()
        name = eng.kernel_name
        top = eng.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        eng.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        gpu.append([r[ms[0]], r[ms[1]]] + losses)
    gpu = np.array(gpu)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tile_stream_model.json")) as f:
        fixture = [r for r in json.load(f) if r["workers"] == 64 and r["tile_sampler"] and r["seed"] in seeds]
    model = np.array([[r["recall20"], r["ndcg20"]] + r["losses"] for r in fixture])
    assert model.shape == gpu.shape == (3, 7) and "streams=64" in name and "tile-in-lds" not in name, name
    print(f"{name}\n gpu (Recall, NDCG, 5 epoch losses) per seed:\n{gpu}\n reference algorithm, 64 lockstep workers:\n{model}")
    g_, m_ = gpu.mean(axis=0), model.mean(axis=0)
    assert abs(g_[0] - m_[0]) <= 1e-3 and abs(g_[1] - m_[1]) <= 1e-3, (g_, m_)
    assert np.all(np.abs(g_[2:] - m_[2:]) <= 0.03 * m_[2:]), (g_, m_)              # the whole loss curve, epoch by epoch


def test_gpu_sampler_drives_training_like_fed_negatives():
    """train_range with the on-GPU sampler == train_range fed with the ids sample_negatives reports."""
    d, N, U, I, T = 64, 16, 40, 500, 1000
    clicks, uw, iw = small_problem(U, I, T, d, seed=21)
    a_u, a_i, b_u, b_i = uw.copy(), iw.copy(), uw.copy(), iw.copy()
    ea = abi.Engine(clicks, a_u, a_i, num_negs=N, seed=5, flags=abi.FLAG_SERIAL)
    negs = ea.sample_negatives(0, T)
    la = ea.train_range(0, T)
    ea.sync_to_host()
    eb = abi.Engine(clicks, b_u, b_i, num_negs=N, seed=5, flags=abi.FLAG_SERIAL)
    lb = eb.train_range(0, T, negs)
    eb.sync_to_host()
    assert la == lb and np.array_equal(a_u, b_u) and np.array_equal(a_i, b_i)
    ea.close(); eb.close()


def test_epoch_protocol_lr_schedule_zero_grad_and_inplace_weights():
    d, N, U, I, T = 64, 16, 80, 600, 4000
    clicks, uw, iw = small_problem(U, I, T, d, seed=33)
    uw0 = uw.copy()
    eng = abi.Engine(clicks, uw, iw, num_negs=N, milestones=(1,), l_r=0.05, seed=2022)
    l0 = eng.train_one_epoch()
    assert eng.epoch == 1 and not np.array_equal(uw, uw0)        # trained in place, written back
    l1 = eng.train_one_epoch()
    assert abs(eng.l_r - 0.005) < 1e-9                            # StepLR(milestones[0]=1, 0.1) fired at epoch 1
    assert l1 < l0 and np.isfinite(l0) and np.isfinite(l1)
    eng.close()


def test_recall_ndcg_parity_amazonbooks_shape():
    """The north-star parity criterion: Recall@20 / NDCG@20 of the Hogwild GPU engine (thousands of concurrent
    streams, on-GPU Philox negatives) vs the CPU oracle (8 OpenMP threads, mt19937_64 negatives) within +-1e-3 after the
    yaml's 5 epochs, same synthetic AmazonBooks-shaped graph, same N(0,0.01^2) tables, seed 2022.  The runs draw different
    negatives and interleave differently, and the 8-thread oracle itself is not reproducible (dynamic scheduling moves its
    NDCG@20 by ~1e-3 between runs), so the MEAN of three GPU runs is compared with the MEAN of three oracle runs."""
    import types
    from heat_amd.cf import metrics
    g, d, N = synthetic.make_named("amazonbooks")
    uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=2022)
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]

    def rank_and_score(uw, iw):
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)   # both sides are ranked by the same top-k kernel
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        return metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)

    rg_runs, lg_runs = [], []
    for _ in range(3):              # the Hogwild GPU run is not reproducible either (Recall@20 moves by up to 8e-4): mean of three
        uw, iw = uw0.copy(), iw0.copy()
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=2022, flags=abi.FLAG_LAZY_SYNC)
        lg_runs.append([eng.train_one_epoch() for _ in range(5)])
        eng.sync_to_host()
        eng.close()
        rg_runs.append(rank_and_score(uw, iw))
    lg = np.mean(lg_runs, axis=0)
    rg = {m: float(np.mean([r[m] for r in rg_runs])) for m in ms}
    ro_runs, lo_runs = [], []
    for _ in range(3):
        uo, io = uw0.copy(), iw0.copy()
        ora = orc.Engine(g.clicks, uo, io, num_negs=N)
        lo_runs.append([ora.train_one_epoch(num_threads=8) for _ in range(5)])
        ro_runs.append(rank_and_score(uo, io))
    lo = np.mean(lo_runs, axis=0)
    ro = {m: float(np.mean([r[m] for r in ro_runs])) for m in ms}
    print("gpu runs", lg_runs, rg_runs, "oracle runs", lo_runs, ro_runs)
    # epoch losses: the first epoch (tiny N(0,0.01^2) rows, every update computed from slightly stale rows) is the most
    # asynchrony-sensitive one: 2.028 ... 2.034 over five runs vs the oracle's 1.950 ... 1.953 (+4.0 ... +4.3 %); the second
    # +3.1 %, later ones within 2 %.  The bands below are those measurements plus 0.7 %, not a target: the north star fixes
    # Recall / NDCG, asserted next.
    for e_, (a, b) in enumerate(zip(lg, lo)):
        assert abs(a - b) <= (0.05 if e_ == 0 else 0.035) * b, (lg, lo)
    assert ro[ms[0]] > 0.05                                       # the model learned something
    assert abs(rg[ms[0]] - ro[ms[0]]) <= 1e-3, (rg_runs, ro_runs)
    assert abs(rg[ms[1]] - ro[ms[1]]) <= 1e-3, (rg_runs, ro_runs)


def _statistical_parity(shape, *, n_clusters, epochs, clip, seeds, lr=0.01):
    """Hogwild GPU engine (default launch plan, on-GPU Philox negatives) vs the 8-thread oracle (mt19937_64 negatives) on the
    same synthetic graph and the same N(0, 0.01^2) tables, once per seed.  The two sides draw different negatives and
    interleave differently, so the comparison is statistical: Recall@20 / NDCG@20 after `epochs` epochs, both sides ranked
    by the same fused top-k kernel.  Oracle runs for different seeds go through two host threads at a time (the C call
    releases the GIL) to keep the test short."""
    import types
    from concurrent.futures import ThreadPoolExecutor
    from heat_amd.cf import metrics
    g, d, N = synthetic.make_named(shape, n_clusters=n_clusters)
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]

    def rank_and_score(uw, iw):
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        return r[ms[0]], r[ms[1]]

    def oracle_run(seed):
        uo, io = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        ora = orc.Engine(g.clicks, uo, io, num_negs=N, clip_val=clip, l_r=lr)
        losses = [ora.train_one_epoch(num_threads=8) for _ in range(epochs)]
        return uo, io, losses

    with ThreadPoolExecutor(max_workers=2) as pool:
        futures = [pool.submit(oracle_run, s) for s in seeds]
        gpu, name = [], None
        for seed in seeds:
            uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
            eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, clip_val=clip, l_r=lr, flags=abi.FLAG_LAZY_SYNC)
            losses = [eng.train_one_epoch() for _ in range(epochs)]
            eng.sync_to_host()
            name = eng.kernel_name
            eng.close()
            gpu.append(rank_and_score(uw, iw) + (losses[-1],))
        ora = []
        for f in futures:
            uo, io, losses = f.result()
            ora.append(rank_and_score(uo, io) + (losses[-1],))
    gpu, ora = np.array(gpu), np.array(ora)
    print(f"{shape} clusters={n_clusters} {name}\n gpu    (Recall, NDCG, final loss) per seed:\n{gpu}\n oracle:\n{ora}")
    return gpu, ora, name


def test_recall_ndcg_parity_yelp18_config():
    """BASELINE.json configs[2] at the Yelp18 yaml's hyper-parameters (d=128, 64 negatives, clip_val 0.1, lr 0.01, 8 epochs;
    Yelp18/MF_CCL/configs/config0.yaml:8-28) on the Yelp18-shaped graph with latent structure (64 user/item clusters: the
    ranking then depends on the learned geometry, and the 8-thread oracle agrees with itself to ~5e-4 between seeds, which
    the popularity-only graph does not: its Recall moves by +-6e-3 between oracle runs at this config).  The default
    launch plan (220 eight-wave workgroups = 220 interactions in flight, positives by float atomics, negatives by the
    late re-read write-back) must hold the north-star tolerance: mean Recall@20 / NDCG@20 over the seeds
    within +-1e-3 of the oracle's, and the final-epoch training loss within 3 %."""
    gpu, ora, name = _statistical_parity("yelp18", n_clusters=64, epochs=8, clip=0.1, seeds=(1, 2, 3, 4, 5, 6))
    assert "<32,4,16,8>" in name and "upd=0x1c" in name and "streams=220" in name, name
    # single runs of either side scatter by ~+-6e-4 around their mean (profiles/r02_yelp18_policy_sweep.txt): six seeds per
    # side resolve the 1e-3 tolerance on the means (measured differences of the means over four suite runs: -7e-4 ... +3e-4)
    assert np.ptp(ora[:, 0]) < 3e-3 and np.ptp(ora[:, 1]) < 3e-3
    assert ora[:, 0].mean() > 0.02                                            # the model learned something
    assert abs(gpu[:, 0].mean() - ora[:, 0].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 1].mean() - ora[:, 1].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 2].mean() - ora[:, 2].mean()) <= 0.03 * ora[:, 2].mean(), (gpu, ora)


def test_recall_ndcg_parity_gowalla_config():
    """Full-size Gowalla shape at this fork's Gowalla yaml (d=128, 64 negatives, clip_val 0.1, 8 epochs;
    Gowalla/MF_CCL/configs/config0.yaml:8-28), clustered graph, same criterion as the Yelp18 config.  Single runs scatter
    more here than at Yelp18 shape (oracle Recall@20 0.0311 ... 0.0329 over four seeds): six seeds per side."""
    gpu, ora, name = _statistical_parity("gowalla", n_clusters=64, epochs=8, clip=0.1, seeds=(1, 2, 3, 4, 5, 6))
    assert "upd=0x1c" in name and "streams=144" in name, name
    assert abs(gpu[:, 0].mean() - ora[:, 0].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 1].mean() - ora[:, 1].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 2].mean() - ora[:, 2].mean()) <= 0.03 * ora[:, 2].mean(), (gpu, ora)


def test_accl_hogwild_parity_amazonbooks_shape():
    """Behaviour aggregation (ACCL) in Hogwild mode at AmazonBooks shape, yaml hyper-parameters, 5 epochs, the
    engine's default plan (512 four-wave streams: two workgroups per compute unit), three seeds per side.

    Two references, because ACCL's loss curve depends on the NUMBER OF WORKERS even without any asynchrony
    (profiles/r03_accl_worker_count.txt): the 8-thread OpenMP oracle, and the oracle's own forward_backward driven as S
    lockstep workers, sequentially consistent (tests/tools/stream_sim.c; committed fixture tests/golden/accl_stream_model.json,
    S = 8, 438 and 512).  The chain that is asserted:
      * model(8 workers)   vs the 8-thread oracle : Recall@20 +-1e-3, final loss within 5 % — the lockstep model IS the oracle;
      * GPU (512 streams)  vs model(512 workers)  : Recall@20 +-1e-3, final loss within 3.5 % — at a matched worker count the
        Hogwild GPU run adds nothing to what the reference's algorithm does with that many workers;
      * GPU vs the 8-thread oracle: Recall@20 within -1e-3 ... +2e-3 (the model's own 8 -> 512 shift is +1.1e-3: more workers
        rank slightly BETTER), final loss +10 ... +20 % (the model's shift: +13 %; one shared W0 pushed by S users at once).
    NDCG@20 of this mode scatters by +-4e-3 between runs of either side on the popularity-only graph, with occasional
    runs 1e-2 low (the top ranks are the hottest items, whose rows move until the last step), so it is held to 4e-3 on the
    medians over the seeds."""
    import json
    import os
    import types
    from concurrent.futures import ThreadPoolExecutor
    from heat_amd.cf import metrics
    seeds = (2022, 7, 99)
    g, d, N = synthetic.make_named("amazonbooks")
    his, masks = synthetic.make_history(g, 100, seed=2022)
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]

    def rank_and_score(uw, iw):
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        return r[ms[0]], r[ms[1]]

    def tables(seed):
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        w0 = (np.random.default_rng(seed).standard_normal((d, d)) * 0.01).astype(np.float32)
        return uw, iw, w0

    def oracle_run(seed):
        uo, io, w0 = tables(seed)
        ora = orc.Engine(g.clicks, uo, io, num_negs=N, his=his, masks=masks, w0=w0, use_aggregator=True)
        return uo, io, [ora.train_one_epoch(num_threads=8) for _ in range(5)]

    with ThreadPoolExecutor(max_workers=2) as pool:
        futures = [pool.submit(oracle_run, s) for s in seeds]
        gpu, name = [], None
        for seed in seeds:
            uw, iw, w0 = tables(seed)
            eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, his=his, masks=masks, w0=w0, use_aggregator=True,
                             flags=abi.FLAG_LAZY_SYNC)
            losses = [eng.train_one_epoch() for _ in range(5)]
            eng.sync_to_host()
            name = eng.kernel_name
            eng.close()
            gpu.append(rank_and_score(uw, iw) + tuple(losses))
        ora = []
        for f in futures:
            uo, io, losses = f.result()
            ora.append(rank_and_score(uo, io) + tuple(losses))
    gpu, ora = np.array(gpu), np.array(ora)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "accl_stream_model.json")) as f:
        fixture = json.load(f)
    model = {w: np.array([[r["recall20"], r["ndcg20"]] + r["losses"] for r in fixture if r["workers"] == w and r["seed"] in seeds])
             for w in (8, 512)}
    assert model[8].shape == model[512].shape == gpu.shape == ora.shape == (3, 7)
    print(f"{name}\n gpu (Recall, NDCG, 5 epoch losses) per seed:\n{gpu}\n 8-thread oracle:\n{ora}\n model 8 workers:\n{model[8]}\n model 512:\n{model[512]}")
    assert "streams=512" in name and "<16,1,16,4>" in name, name
    g_, o_, m8, m512 = gpu.mean(axis=0), ora.mean(axis=0), model[8].mean(axis=0), model[512].mean(axis=0)
    # the lockstep model at the oracle's worker count is the oracle
    assert abs(m8[0] - o_[0]) <= 1e-3 and abs(m8[6] - o_[6]) <= 0.05 * o_[6], (m8, o_)   # the oracle's final loss moves 1.084 ... 1.095 between boxes
    # the GPU at its worker count is the model at that worker count
    assert abs(g_[0] - m512[0]) <= 1e-3, (g_, m512)
    assert abs(g_[6] - m512[6]) <= 0.035 * m512[6], (g_, m512)
    assert np.all(np.abs(g_[3:] - m512[3:]) <= 0.06 * m512[3:]), (g_, m512)          # every later epoch of the curve, loosely
    # NDCG@20: single runs of EITHER side occasionally land 1e-2 low (one of the few hottest items ends the last epoch
    # displaced: GPU seed 2022 0.2151 in one suite run, the model 0.2178 at 80 workers) — medians over the seeds, 4e-3
    ndcg = {name_: float(np.median(a[:, 1])) for name_, a in (("gpu", gpu), ("oracle", ora), ("m512", model[512]))}
    assert abs(ndcg["gpu"] - ndcg["m512"]) <= 4e-3 and abs(ndcg["gpu"] - ndcg["oracle"]) <= 4e-3, ndcg
    # against the 8-thread oracle: the worker-count shift, no more
    assert -1e-3 <= g_[0] - o_[0] <= 2e-3, (g_, o_)
    assert 1.08 * o_[6] <= g_[6] <= 1.22 * o_[6], (g_, o_)


def test_recall_ndcg_parity_config_s_regime():
    """BASELINE.json configs[4] (10 M x 1 M, d=256, 100 negatives) in Hogwild mode at a size the oracle can run (VERDICT r02
    item 7): 75 000 users x 200 000 items, 1.5 M interactions, clustered, 3 epochs — large enough that the engine's default plan
    is the one of the full shape: `<64,13,16,8>` (eight waves per stream), 256 streams, positives by float atomics
    (0.13 in-flight touches per item row, full shape: 0.03).  Two seeds per side.
      * GPU (256 streams) vs the reference's algorithm with 256 lockstep workers (tests/tools/stream_sim.c, fixture
        tests/golden/config_s_stream_model.json): Recall@20 / NDCG@20 +-1e-3, every epoch loss within 3 %;
      * that model with 8 workers vs the 8-thread oracle run here: +-1e-3 (the model IS the oracle);
      * GPU vs the 8-thread oracle: -1e-3 ... +2e-3 — measured +1.0e-3 ... +1.4e-3 Recall@20 (profiles/r03_config_s_regime_parity.txt),
        of which +4.6e-4 is the worker count (model: 8 -> 256 workers) and +3.6e-4 the sampler (counter-based vs the
        oracle's mt19937_64 with the reference's repeating seeds); three epochs in, Recall still climbs by ~2e-2 per epoch."""
    import json
    import os
    import types
    from concurrent.futures import ThreadPoolExecutor
    from heat_amd.cf import metrics
    seeds = (1, 2)
    _, _, _, d, N = synthetic.SHAPES["synthetic_hbm"]
    g = synthetic.make_graph(75000, 200000, 1500000, seed=2022, n_clusters=64)
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]

    def rank_and_score(uw, iw):
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        return r[ms[0]], r[ms[1]]

    def oracle_run(seed):
        uo, io = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        ora = orc.Engine(g.clicks, uo, io, num_negs=N)
        return uo, io, [ora.train_one_epoch(num_threads=8) for _ in range(3)]

    with ThreadPoolExecutor(max_workers=2) as pool:
        futures = [pool.submit(oracle_run, s) for s in seeds]
        gpu, name = [], None
        for seed in seeds:
            uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
            eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, flags=abi.FLAG_LAZY_SYNC)
            losses = [eng.train_one_epoch() for _ in range(3)]
            eng.sync_to_host()
            name = eng.kernel_name
            eng.close()
            gpu.append(rank_and_score(uw, iw) + tuple(losses))
        ora = []
        for f in futures:
            uo, io, losses = f.result()
            ora.append(rank_and_score(uo, io) + tuple(losses))
    gpu, ora = np.array(gpu), np.array(ora)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_s_stream_model.json")) as f:
        fixture = json.load(f)
    model = {w: np.array([[r["recall20"], r["ndcg20"]] + r["losses"] for r in fixture if r["workers"] == w and r["seed"] in seeds])
             for w in (8, 256)}
    print(f"{name}\n gpu (Recall, NDCG, 3 epoch losses) per seed:\n{gpu}\n 8-thread oracle:\n{ora}\n model 8 workers:\n{model[8]}\n model 256:\n{model[256]}")
    assert "<64,13,16,8>" in name and "upd=0xc" in name and "streams=256" in name, name
    assert model[8].shape == model[256].shape == gpu.shape == ora.shape == (2, 5)
    g_, o_, m8, m256 = gpu.mean(axis=0), ora.mean(axis=0), model[8].mean(axis=0), model[256].mean(axis=0)
    assert o_[0] > 0.03                                                                      # the model learned something
    assert abs(m8[0] - o_[0]) <= 1e-3 and abs(m8[1] - o_[1]) <= 1e-3, (m8, o_)
    assert abs(g_[0] - m256[0]) <= 1e-3 and abs(g_[1] - m256[1]) <= 1e-3, (g_, m256)
    assert np.all(np.abs(g_[2:] - m256[2:]) <= 0.03 * m256[2:]), (g_, m256)
    assert -1e-3 <= g_[0] - o_[0] <= 2e-3 and -1e-3 <= g_[1] - o_[1] <= 2e-3, (g_, o_)
    assert np.all(np.abs(g_[2:] - o_[2:]) <= 0.035 * o_[2:]), (g_, o_)


def test_recall_ndcg_parity_gowalla_pr1_config():
    """BASELINE.json configs[0] ("PR1": Gowalla shape, d=64, 16 negatives, tile 512 in the yaml — which the live loop's
    ignore_pos_sampling never uses, random_tile_negative_sampler.cpp:47-57) at FULL size, config_pr1.yaml's hyper-parameters
    (clip_val 1.0, 5 epochs), six seeds per side, the engine's default plan against the 8-thread oracle: means within +-1e-3,
    final loss within 3.5 %.  (Round 1 measured +5.3e-4 / +5.2e-4 over six seeds, profiles/r01_recall_parity_gowalla_6seeds.txt.)"""
    gpu, ora, name = _statistical_parity("gowalla_pr1", n_clusters=0, epochs=5, clip=1.0, seeds=(1, 2, 3, 4, 5, 6))
    assert "<16,4,16,1>" in name and "upd=0xc" in name, name
    assert ora[:, 0].mean() > 0.1                                             # the model learned something
    assert abs(gpu[:, 0].mean() - ora[:, 0].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 1].mean() - ora[:, 1].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 2].mean() - ora[:, 2].mean()) <= 0.035 * ora[:, 2].mean(), (gpu, ora)


def test_recall_ndcg_parity_amazonbooks_clustered():
    """The headline config (AmazonBooks yaml: d=64, 16 negatives, clip_val 1.0, 5 epochs) on the clustered variant of the
    AmazonBooks-shaped graph — the more discriminating twin of test_recall_ndcg_parity_amazonbooks_shape."""
    gpu, ora, name = _statistical_parity("amazonbooks", n_clusters=64, epochs=5, clip=1.0, seeds=(2022, 7))
    assert "<16,4,16,1>" in name and "upd=0xc" in name, name
    assert abs(gpu[:, 0].mean() - ora[:, 0].mean()) <= 1e-3, (gpu, ora)
    assert abs(gpu[:, 1].mean() - ora[:, 1].mean()) <= 1e-3, (gpu, ora)


def test_eight_user_shards_match_single_engine_recall_ndcg():
    """BASELINE.json configs[3] without the wires: the AmazonBooks-shaped graph cut into 8 user shards (cf/main.py:51-57),
    one real HIP engine per shard taking turns on this GPU, the item-table deltas exchanged by the product's own delta /
    apply kernels with the all-reduce replaced by a device-side sum over the 8 delta buffers (tests/shard_sim.py) — exactly
    what `bench.py --gpus 8` runs per rank, minus the wires.  Three exchange schedules, each with every shard engine on its
    own default launch plan (a stream walks >= 256 interactions: about 1170 streams per shard):
      * two exchanges per epoch, the other shards' deltas arriving one window late, the closing one completed (round 2);
      * ONE exchange per epoch, overlapped with the next epoch (the other shards' deltas arrive one EPOCH late);
      * one exchange every TWO epochs (what `bench.py --gpus 8` runs: the exchange window held at >= ~500 k interactions per
        GPU, i.e. one epoch at 4 GPUs, two at 8), with the closing exchange of the job completed.
    Recall@20 / NDCG@20 after the yaml's 5 epochs (means over three seeds) must stay within +-1e-3 of single-engine training
    on the whole graph for each of them."""
    import types
    from heat_amd.cf import metrics
    from tests.shard_sim import train_sharded
    g, d, N = synthetic.make_named("amazonbooks")
    ep = g.test_indptr.astype(np.int64)
    test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist()
                                                 for u in range(g.num_users) if ep[u + 1] > ep[u]})
    ms = ["Recall(k=20)", "NDCG(k=20)"]

    def rank_and_score(uw, iw):
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        return r[ms[0]], r[ms[1]]

    schedules = {"2 per epoch": dict(windows_per_epoch=2, overlap=True),
                 "1 per epoch, one epoch late": dict(windows_per_epoch=1, overlap=True, defer_final=True),
                 "every 2 epochs": dict(windows_per_epoch=1, overlap=True, defer_final=True, exchange_every=2)}
    # means over three seeds on both sides: two runs of ONE configuration differ by up to 8e-4 in Recall@20 (Hogwild
    # interleaving), so a single pair cannot resolve the 1e-3 bar
    singles, shardeds = [], {name: [] for name in schedules}
    for seed in (2022, 7, 99):
        uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
        uw, iw = uw0.copy(), iw0.copy()
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, flags=abi.FLAG_LAZY_SYNC)
        for _ in range(5):
            eng.train_one_epoch()
        eng.sync_to_host()
        eng.close()
        singles.append(rank_and_score(uw, iw))
        for sname, kw in schedules.items():
            su, si, losses, name = train_sharded(g, uw0, iw0, num_negs=N, world=8, epochs=5, seed=seed, **kw)
            shardeds[sname].append(rank_and_score(su, si))
            print("seed", seed, "single", singles[-1], sname, shardeds[sname][-1], name, [round(x, 4) for x in losses])
            assert "<16,4,16,1>/upd=0xc" in name and 1024 <= int(name.split("streams=")[1]) <= 1200, name   # shard 0: 300 770 // 256
    single = np.mean(singles, axis=0)
    for sname, runs in shardeds.items():
        sharded = np.mean(runs, axis=0)
        print("means: single", single, sname, sharded)
        assert abs(sharded[0] - single[0]) <= 1e-3 and abs(sharded[1] - single[1]) <= 1e-3, (sname, singles, runs)


def test_overwrite_mode_loses_updates_at_gpu_concurrency():
    """Documents WHY the default is the atomic write-back: the reference's literal overwrite, run with more than a thousand
    concurrent streams (forced: on its own the overwrite policy stays at 64), drops a large share of the updates of
    popular rows and the epoch loss stays visibly higher."""
    g, d, N = synthetic.make_named("gowalla_pr1", scale=0.5)
    losses = {}
    for mode in (abi.UPDATE_OVERWRITE, abi.UPDATE_ATOMIC_WG):
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=1)
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=1, update_mode=mode, num_streams=1300, flags=abi.FLAG_LAZY_SYNC)
        losses[mode] = [eng.train_one_epoch() for _ in range(3)]
        eng.close()
    assert losses[abi.UPDATE_ATOMIC_WG][-1] < losses[abi.UPDATE_OVERWRITE][-1]


def test_evaluate0_and_topk():
    d, U, I = 64, 70, 333
    uw, iw = synthetic.init_embeddings(U, I, d, seed=5)
    clicks = np.array([[0, 1]], dtype=np.uint64)
    eng = abi.Engine(clicks, uw, iw, num_negs=4)
    sim = eng.evaluate0()
    ref = orc.Engine(clicks, uw.copy(), iw.copy(), num_negs=4).evaluate0()
    assert np.array_equal(sim, ref)  # unfused fp32 multiply-add, k left to right: bit-identical to the oracle
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 12, size=U)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    items = np.concatenate([rng.choice(I, size=n, replace=False) for n in lens]).astype(np.uint32)
    top = eng.topk(20, mask_indptr=indptr, mask_items=items)
    masked = ref.copy()
    for u in range(U):
        masked[u, items[indptr[u]:indptr[u + 1]]] = -np.inf
    want = np.argsort(-masked, axis=1, kind="stable")[:, :20]
    assert np.array_equal(top, want.astype(np.uint32))
    eng.close()


def _topk_want(sim, indptr, items, k):
    masked = sim.copy()
    if indptr is not None:
        for u in range(sim.shape[0]):
            masked[u, items[int(indptr[u]):int(indptr[u + 1])]] = -np.inf
    return np.argsort(-masked, axis=1, kind="stable")[:, :k].astype(np.uint32)


@pytest.mark.parametrize("U,I,d,k,case", [
    (70, 333, 64, 20, "random"),        # one item split, ragged last tile, unsorted mask rows
    (100, 5000, 64, 64, "random"),      # many item splits, k at the list capacity
    (100, 5000, 128, 1, "random"),
    (65, 700, 20, 7, "random"),         # emb_dim not a multiple of the 16-wide k slab
    (3, 30, 4, 20, "mostly_masked"),    # fewer than k unmasked items: -inf entries fill the list in id order
    (130, 3000, 64, 20, "ascending"),   # every tile beats the thresholds: every score of every tile is a candidate
    (64, 900, 64, 20, "ties"),          # repeated item rows: equal scores rank by item id
    (1, 129, 8, 5, "random"),
    (300, 2000, 256, 33, "random"),     # 128-user workgroups: 4 item slabs per tile, 64-slot lists, single-buffered slab
    (257, 1500, 96, 32, "random"),      # a slab that ends inside emb_dim, k at the 32-slot list capacity, one user in the last block
    (129, 4000, 64, 20, "ascending_users"),  # scores ascending in item id: every score is inserted, in both halves of a wave
    (200, 2500, 64, 32, "random"),      # 128-user kernel at its list capacity (all 32 lanes of a half hold a slot)
    (90, 1000, 32, 1, "random"),        # ... and at k = 1 (slot 0 is the threshold)
])
def test_topk_fused_matches_numpy_and_panel_path(U, I, d, k, case, monkeypatch):
    """the fused U*V^T + mask + top-k (no score matrix) returns exactly the ids numpy's stable
    argsort picks from the oracle's dense scores (cf/metrics.py:21-29 semantics), and exactly what the materialised
    panel path returns."""
    rng = np.random.default_rng(U * 1000 + I)
    uw, iw = synthetic.init_embeddings(U, I, d, seed=7)
    if case in ("ascending", "ascending_users"):
        base = np.abs(uw[0]).astype(np.float32) + 0.01
        uw[:] = base * rng.uniform(0.5, 2.0, size=(U, 1)).astype(np.float32)
        iw[:] = base * (1.0 + np.arange(I, dtype=np.float32)[:, None] / I)
    if case == "ties":
        iw[1::2] = iw[0::2][: len(iw[1::2])]
    clicks = np.array([[0, 1]], dtype=np.uint64)
    lens = rng.integers(0, 12, size=U) if case != "mostly_masked" else np.full(U, I - 5)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    items = np.concatenate([rng.choice(I, size=n, replace=False) for n in lens] + [np.zeros(0, np.int64)]).astype(np.uint32)
    ref = orc.Engine(clicks, uw.copy(), iw.copy(), num_negs=1).evaluate0()
    eng = abi.Engine(clicks, uw, iw, num_negs=1)
    got = eng.topk(k, mask_indptr=indptr, mask_items=items)
    assert np.array_equal(got, _topk_want(ref, indptr, items, k))
    nomask = eng.topk(k)
    assert np.array_equal(nomask, _topk_want(ref, None, None, k))
    sub = eng.topk(k, mask_indptr=indptr, mask_items=items, u_begin=U // 2, u_end=U)
    assert np.array_equal(sub, got[U // 2:])
    monkeypatch.setenv("HEAT_CF_TOPK_PANEL", "37")           # users walked in panels of 37: same ids
    assert np.array_equal(eng.topk(k, mask_indptr=indptr, mask_items=items), got)
    monkeypatch.delenv("HEAT_CF_TOPK_PANEL")
    monkeypatch.setenv("HEAT_CF_TOPK_KERNEL", "v1")          # the 64 x 128 kernel of round 2 (kept for A/B runs): same ids
    assert np.array_equal(eng.topk(k, mask_indptr=indptr, mask_items=items), got)
    monkeypatch.delenv("HEAT_CF_TOPK_KERNEL")
    monkeypatch.setenv("HEAT_CF_TOPK_PATH", "panel")
    assert np.array_equal(eng.topk(k, mask_indptr=indptr, mask_items=items), got)
    eng.close()


def test_topk_fused_large_equals_panel_path(monkeypatch):
    """At a size the dense numpy check is too slow for, the fused path must agree id for id with the panel path."""
    g, d, N = synthetic.make_named("gowalla_pr1", scale=0.25)
    uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=3)
    eng = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
    fused = eng.topk(50, mask_indptr=g.train_indptr, mask_items=g.train_items)
    monkeypatch.setenv("HEAT_CF_TOPK_KERNEL", "v1")
    assert np.array_equal(eng.topk(50, mask_indptr=g.train_indptr, mask_items=g.train_items), fused)
    monkeypatch.delenv("HEAT_CF_TOPK_KERNEL")
    monkeypatch.setenv("HEAT_CF_TOPK_PATH", "panel")
    panel = eng.topk(50, mask_indptr=g.train_indptr, mask_items=g.train_items)
    assert np.array_equal(fused, panel)
    eng.close()


def test_bad_arguments_raise():
    uw, iw = synthetic.init_embeddings(4, 8, 64)
    with pytest.raises(ValueError):
        abi.Engine(np.array([[0, 8]], dtype=np.uint64), uw, iw, num_negs=4)       # item id out of range
    with pytest.raises(ValueError):
        abi.Engine(np.array([[0, 1]], dtype=np.int64), uw, iw, num_negs=4)        # wrong dtype
    with pytest.raises(abi.HeatError):
        abi.Engine(np.array([[0, 1]], dtype=np.uint64), np.zeros((4, 6), np.float32), np.zeros((8, 6), np.float32),
                   num_negs=4)                                                      # emb_dim % 4 != 0
    eng = abi.Engine(np.array([[0, 1]], dtype=np.uint64), uw, iw, num_negs=4)
    with pytest.raises(ValueError):
        eng.train_range(0, 2)
    with pytest.raises(ValueError):
        eng.train_range(0, 1, np.full((1, 4), 8, dtype=np.uint64))
    eng.close()


def test_frontend_main_runs_through_cf_c(tmp_path, capsys):
    """The reference-shaped driver (heat_amd/cf/main.py = cf/main.py:19-124 without mpi4py) on a synthetic graph through
    the pybind11 cf_c module: yaml -> CFConfig -> ClickDataset -> MatrixFactorization -> Engine -> epochs -> evaluate0 ->
    Recall, both evaluation paths (dense evaluate0 + metrics, fused GPU top-k)."""
    import yaml
    from heat_amd.cf import main as cf_main
    cfg = yaml.safe_load(open("heat_amd/cf/benchmarks/Gowalla/MF_CCL/configs/config_pr1.yaml"))
    cfg["model_config"]["epochs"] = 3
    cfg["model_config"]["eval_interval"] = 2
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    r_dense = cf_main.main(["--config", str(path), "--synthetic", "gowalla", "--scale", "0.1", "--dense-eval"])
    r_topk = cf_main.main(["--config", str(path), "--synthetic", "gowalla", "--scale", "0.1"])
    out = capsys.readouterr().out
    assert "epoch: 2; loss:" in out and "[Metrics] Recall(k=20)" in out
    assert 0.02 < r_dense["Recall(k=20)"] < 1.0
    # same seed, same data; the two runs differ only by Hogwild interleaving and by how ties are ranked
    assert abs(r_dense["Recall(k=20)"] - r_topk["Recall(k=20)"]) < 0.02


def test_full_size_properties_amazonbooks_shape():
    """Size-independent properties at BASELINE.json's full AmazonBooks size (2 380 730 interactions, Hogwild):
    (a) lr = 0 leaves both weight tables bit-identical and the mean loss equals the oracle's forward-only loss within 1 %;
    (b) after a real epoch every element of a user row moved by at most lr*clip*degree(user) (clip bounds each step);
    (c) both gradient tables are zero after the epoch (engine.cpp:345-347);  (d) no NaN/Inf anywhere."""
    g, d, N = synthetic.make_named("amazonbooks", with_test=False)
    uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=2022)
    # (a)
    uw, iw = uw0.copy(), iw0.copy()
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=2022, l_r=0.0)
    l_gpu = eng.train_one_epoch()
    assert np.array_equal(uw, uw0) and np.array_equal(iw, iw0)
    eng.close()
    uo, io = uw0.copy(), iw0.copy()
    ora = orc.Engine(g.clicks, uo, io, num_negs=N, l_r=0.0)
    l_cpu = ora.train_one_epoch(num_threads=8)
    assert abs(l_gpu - l_cpu) <= 0.01 * l_cpu, (l_gpu, l_cpu)
    # (b), (c), (d)
    lr, clip = 0.01, 1.0
    uw, iw = uw0.copy(), iw0.copy()
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=2022, l_r=lr, clip_val=clip)
    loss = eng.train_one_epoch()
    assert np.isfinite(loss) and np.isfinite(uw).all() and np.isfinite(iw).all()
    deg = np.diff(g.train_indptr.astype(np.int64)).astype(np.float64)
    moved = np.abs(uw.astype(np.float64) - uw0.astype(np.float64)).max(axis=1)
    assert (moved <= lr * clip * deg * (1 + 1e-5) + 1e-7).all()
    assert moved.max() > 0.05                                   # and they did move
    view = eng.device_view()
    assert not eng.read_device(view.user_g, (g.num_users, d)).any()
    assert not eng.read_device(view.item_g, (g.num_items, d)).any()
    eng.close()


def test_full_size_properties_synthetic_hbm_shape():
    """BASELINE.json configs[4] at its full TABLE size (10 M users x 1 M items, d = 256, 100 negatives: 22.5 GB of W and G,
    far beyond the Infinity Cache; device-resident engine on torch tensors) over a 1 M-interaction sample of its 200 M list
    (50 000 users spread over the whole table x 20 interactions).  Size-independent properties: (a) lr = 0 leaves both
    tables bit-identical and the mean loss is that of near-orthogonal rows, log(1 + 100) up to the init's scatter;
    (b) a real epoch changes exactly the user rows that have interactions, keeps everything finite, and leaves both
    gradient tables zero (engine.cpp:345-347); (c) the kernel is the 8-wave variant the launch plan picks for this shape."""
    import torch
    U, I, _, d, N = synthetic.SHAPES["synthetic_hbm"]
    T = 1_000_000
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 40 * (1 << 30):
        pytest.skip("needs 40 GB of free device memory")
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        clicks = synthetic.make_clicks_torch(U, I, T, dev, seed=5, per_user=20)
        gen = torch.Generator(device=dev)
        gen.manual_seed(5)
        user_w = torch.empty((U, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=gen)
        item_w = torch.empty((I, d), device=dev, dtype=torch.float32).normal_(0.0, 0.01, generator=gen)
        u0 = user_w.clone()
        i0 = item_w.clone()
        common = dict(num_users=U, num_items=I, emb_dim=d, num_negs=N, stream=side.cuda_stream, seed=5,
                      keep=(clicks, user_w, item_w))
        eng = abi.Engine.from_device(clicks.data_ptr(), T, user_w.data_ptr(), item_w.data_ptr(), l_r=0.0, **common)
        assert "<64,13,16,8>" in eng.kernel_name, eng.kernel_name                                      # (c)
        loss0 = eng.train_one_epoch()
        side.synchronize()
        assert torch.equal(user_w, u0) and torch.equal(item_w, i0)                                     # (a)
        assert abs(loss0 - np.log(1.0 + N)) < 0.5, loss0
        eng.close()
        eng = abi.Engine.from_device(clicks.data_ptr(), T, user_w.data_ptr(), item_w.data_ptr(), l_r=0.01, clip_val=1.0, **common)
        loss1 = eng.train_one_epoch()
        side.synchronize()
        assert np.isfinite(loss1) and bool(torch.isfinite(user_w).all()) and bool(torch.isfinite(item_w).all())   # (b)
        touched = torch.zeros(U, dtype=torch.bool, device=dev)
        touched[clicks[:, 0]] = True
        moved = (user_w != u0).any(dim=1)
        assert bool((moved == touched).all()) and int(touched.sum()) > 40_000
        assert bool((item_w != i0).any())
        view = eng.device_view()
        # the engine's G tables are its own allocations: read through the C ABI, the user side in five 256 MB slabs
        slab = 250_000
        for lo in range(0, U, slab * 8):
            assert not eng.read_device(view.user_g + lo * d * 4, (min(slab, U - lo), d)).any()
        assert not eng.read_device(view.item_g, (I, d)).any()
        eng.close()
    del user_w, item_w, u0, i0, clicks
    torch.cuda.empty_cache()


def test_topk_full_size_properties_amazonbooks_shape():
    """Size-independent properties of the fused top-k at the full AmazonBooks size (52 643 users x 91 599 items, the matrix
    the reference's evaluate0 materialises in 19.3 GB): (a) no train item is ever returned; (b) the ids of a row are
    distinct and in range; (c) for sampled users the returned ids are exactly the oracle's top-20 of the masked dense
    row, in the same order; (d) user sub-ranges and a second call return the same ids (no state between calls)."""
    g, d, N = synthetic.make_named("amazonbooks", with_test=False)
    uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=11, std=0.1)
    eng = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
    top = eng.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
    assert top.shape == (g.num_users, 20) and top.max() < g.num_items
    srt = np.sort(top, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                                         # (b)
    indptr, items = g.train_indptr.astype(np.int64), g.train_items
    seen = np.zeros(g.num_items, dtype=bool)
    rng = np.random.default_rng(0)
    sample = np.concatenate([[0, g.num_users - 1], rng.integers(0, g.num_users, 2000)])
    for u in sample:                                                                 # (a) on 2002 users
        seen[:] = False
        seen[items[indptr[u]:indptr[u + 1]]] = True
        assert not seen[top[u]].any()
    one = orc.Engine(g.clicks[:1].copy(), uw[:1].copy(), iw.copy(), num_negs=N)
    for u in sample[:40]:                                                            # (c)
        one.user_w[0] = uw[u]
        row = one.evaluate0()[0]
        row[items[indptr[u]:indptr[u + 1]]] = -np.inf
        assert np.array_equal(top[u], np.argsort(-row, kind="stable")[:20].astype(np.uint32)), u
    lo, hi = g.num_users // 3, g.num_users // 3 + 1000
    assert np.array_equal(eng.topk(20, u_begin=lo, u_end=hi, mask_indptr=g.train_indptr, mask_items=g.train_items), top[lo:hi])
    assert np.array_equal(eng.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items), top)    # (d)
    eng.close()


def test_device_mode_engine_with_item_sync_on_a_side_stream():
    """Device mode (torch tensors handed over as raw pointers, engine launching on a torch side stream) + the item-table
    sync path of heat_amd.cf.distributed with a 1-rank RCCL group: `W <- ref + allreduce(W - ref)` must be the identity,
    and the torch element-wise ops / the collective must be ordered after the training kernel on that stream.  Runs in a
    child process (tests/_gpu_sync_worker.py) so that RCCL / process-group teardown cannot take the test runner with it."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_sync_worker.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0 and "SYNC_OK" in res.stdout, res.stdout[-3000:]


def test_item_sync_kernels_equal_the_torch_arithmetic_bit_for_bit():
    """heat_cf_sync_delta / heat_cf_sync_apply (item_sync.hip) against the torch arithmetic ItemSync falls back to without a
    HIP engine (heat_amd/cf/distributed.py: _delta, _apply) on the same tensors: bit for bit, with `mine` NULL (blocking
    exchange: W = ref = ref + scale * sum) and non-NULL (overlapped exchange: W += scale * sum - mine), scale 1 and 1/3."""
    import torch
    from heat_amd.cf.distributed import ItemSync
    dev = torch.device("cuda", 0)
    d, U, I, T = 64, 6, 1237, 64
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    clicks = torch.zeros((T, 2), dtype=torch.int64, device=dev)
    uw = torch.zeros((U, d), device=dev)

    class Plain:                  # an engine without the fused passes: ItemSync then runs the torch expressions
        data_rows = T

    for scale_mode, world in (("sum", 2), ("mean", 3)):
        for with_mine in (False, True):
            w0 = torch.randn((I, d), device=dev, generator=g) * 0.1
            ref0 = w0 + torch.randn((I, d), device=dev, generator=g) * 1e-3
            others = torch.randn((I, d), device=dev, generator=g) * 1e-3          # what the other ranks would add to the sum
            trained = torch.randn((I, d), device=dev, generator=g) * 1e-3         # progress made while the exchange was in flight
            w_n, w_t = w0.clone(), w0.clone()
            eng = abi.Engine.from_device(clicks.data_ptr(), T, uw.data_ptr(), w_n.data_ptr(), num_users=U, num_items=I, emb_dim=d,
                                         num_negs=4, keep=(clicks, uw, w_n))
            nat = ItemSync(eng, w_n, world, mode=scale_mode, overlap=with_mine, dist=object())
            ref = ItemSync(Plain(), w_t, world, mode=scale_mode, overlap=with_mine, dist=object())
            assert nat.native and not ref.native
            for s_ in (nat, ref):
                s_.ref.copy_(ref0)
            torch.cuda.synchronize()
            nat._delta(with_mine)
            ref._delta(with_mine)
            eng.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(nat.sum, ref.sum) and (not with_mine or torch.equal(nat.mine, ref.mine))
            for s_, w_ in ((nat, w_n), (ref, w_t)):
                s_.sum.add_(others)                                                # "all-reduce"
                if with_mine:
                    w_.add_(trained)
            torch.cuda.synchronize()
            nat._apply(with_mine)
            ref._apply(with_mine)
            eng.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(w_n, w_t) and torch.equal(nat.ref, ref.ref), (scale_mode, with_mine)
            assert not torch.equal(w_n, w0)
            if not with_mine:
                assert torch.equal(w_n, nat.ref)                                   # blocking form: replicas identical
            else:
                # the fused pass of the overlapped schedule (apply of one exchange + delta of the next, heat_cf_sync_apply_delta)
                # against apply followed by delta in torch
                nat._delta(True)
                ref._delta(True)
                eng.synchronize()
                torch.cuda.synchronize()
                for s_, w_ in ((nat, w_n), (ref, w_t)):
                    s_.sum.add_(others)
                    w_.add_(trained)
                torch.cuda.synchronize()
                eng.sync_apply_delta(nat.ref.data_ptr(), nat.sum.data_ptr(), nat.mine.data_ptr(), nat.scale)
                ref._apply(True)
                ref._delta(True)
                eng.synchronize()
                torch.cuda.synchronize()
                assert torch.equal(w_n, w_t) and torch.equal(nat.ref, ref.ref) and torch.equal(nat.sum, ref.sum) and torch.equal(nat.mine, ref.mine)
            eng.close()


def test_edge_shapes_empty_single_tiny():
    """Edge cases the domain has: an empty interaction list, a single interaction, one negative, emb_dim = 4, more
    negatives than items (duplicates guaranteed), a ragged tail (n % 64 != 0), users without interactions."""
    # empty
    uw, iw = synthetic.init_embeddings(3, 5, 64)
    u0 = uw.copy()
    eng = abi.Engine(np.zeros((0, 2), dtype=np.uint64), uw, iw, num_negs=4)
    assert eng.train_one_epoch() == 0.0 and eng.epoch == 1 and np.array_equal(uw, u0)
    eng.close()
    # single interaction, one negative, emb_dim 4, vs the oracle
    for d, N, I in [(4, 1, 6), (64, 16, 3), (8, 5, 2)]:
        rng = np.random.default_rng(d)
        uw = (rng.standard_normal((2, d)) * 0.1).astype(np.float32)
        iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        uo, io = uw.copy(), iw.copy()
        clicks = np.array([[1, 0]], dtype=np.uint64)
        negs = rng.integers(0, I, size=(1, N)).astype(np.uint64)
        eng = abi.Engine(clicks, uw, iw, num_negs=N, flags=abi.FLAG_SERIAL)
        lg = eng.train_range(0, 1, negs)
        eng.sync_to_host(); eng.close()
        lo = orc.Engine(clicks, uo, io, num_negs=N).train_range(0, 1, negs)
        assert abs(lg - lo) <= 1e-5 * max(1.0, abs(lo))
        np.testing.assert_allclose(uw, uo, rtol=0, atol=2e-7)
        np.testing.assert_allclose(iw, io, rtol=0, atol=2e-7)
        assert np.array_equal(uw[0], uo[0])                      # user 0 has no interactions: untouched


def test_windows_compose_to_the_whole_range():
    """train_range(0,a) + train_range(a,n) == train_range(0,n) in serial mode (multi-GPU windows cut epochs this way);
    sampling() call so that the negatives do not depend on where a window starts."""
    d, N, U, I, T = 64, 16, 30, 500, 1000
    clicks, uw, iw = small_problem(U, I, T, d, seed=12)
    outs = []
    for cuts in ([0, T], [0, 1, 65, 300, 777, T]):
        a, b = uw.copy(), iw.copy()
        eng = abi.Engine(clicks, a, b, num_negs=N, seed=9, flags=abi.FLAG_SERIAL | abi.FLAG_SAMPLING_CALL)
        loss = sum(eng.train_range(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]))
        eng.sync_to_host(); eng.close()
        outs.append((a, b, loss))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert abs(outs[0][2] - outs[1][2]) < 1e-9 * abs(outs[0][2])


def test_randomized_serial_parity_sweep():
    """40 random configurations (emb_dim multiple of 4 up to 256, 1..100 negatives, every update policy, both samplers'
    id ranges, ragged lengths; every third one with behaviour aggregation: random histories of 1..119 items, W0 updates
    every 32 calls): the serial GPU walk vs the oracle on caller-fed negatives, duplicates and negative == positive
    collisions included.  Catches variant-specific indexing errors (masked lanes / slots, multi-wave workgroups, atomics vs
    stores, the history gather and the d x d product).
    Aggregation cases carry a second arbiter: the compounding user blend drives the user rows to small norms, where two
    fp32 implementations separate quickly (tools/serial_sweep.py 400 11: 3 of 133 such cases end 1.3x - 9x beyond the table
    tolerance; profiles/r03_serial_arbiter.txt).  Such a case passes when the GPU is at least as close to the float64
    trajectory (tests/f64_model.py) as the fp32 oracle is — the oracle's own distance from it is then what the two differ by."""
    from tests import serial_cases
    rng = np.random.default_rng(2024)
    modes = [abi.UPDATE_OVERWRITE, abi.UPDATE_ATOMIC_W, abi.UPDATE_ATOMIC_WG, abi.UPDATE_ATOMIC_POS, abi.UPDATE_AUTO,
             16 + 0x1C, 16 + 0x10]   # raw bits: late re-read of negative rows (positives atomic / nothing atomic)
    seen_agg = 0
    for case in range(40):
        d = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 96, 128, 160, 256]))
        N = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 50, 64, 100]))
        U = int(rng.integers(2, 12))
        I = int(rng.integers(max(3, N // 4), 400))
        T = int(rng.integers(1, 150))
        agg = case % 3 == 2
        mode = modes[case % len(modes)] if not agg else abi.UPDATE_AUTO
        clicks = np.stack([np.sort(rng.integers(0, U, T)), rng.integers(0, I, T)], axis=1).astype(np.uint64)
        uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        negs = rng.integers(0, I, size=(T, N)).astype(np.uint64)
        c = dict(d=d, N=N, U=U, I=I, T=T, agg=agg, mode=mode, clicks=clicks, uw=uw, iw=iw, negs=negs, his=None, masks=None, w0=None)
        if agg:
            H = int(rng.integers(1, 120))
            c["masks"] = rng.integers(1, H + 1, size=(U, 1)).astype(np.uint64)
            c["his"] = rng.integers(0, I, size=(U, H)).astype(np.uint64)
            c["w0"] = (rng.standard_normal((d, d)) * 0.05).astype(np.float32)
            seen_agg += 1
        ug, ig, w0g, name = serial_cases.run_gpu(c, want_loss=True)
        lg = serial_cases.run_gpu.last_loss
        uo, io, w0o = serial_cases.run_oracle(c, want_loss=True)
        lo = serial_cases.run_oracle.last_loss
        ctx = f"case {case}: d={d} N={N} U={U} I={I} T={T} mode={mode} agg={agg} {name}"
        e_go = serial_cases.dist((ug, ig, w0g), (uo, io, w0o))
        if agg and e_go > 1.0:
            f = serial_cases.run_f64(c)
            e_gf, e_of = serial_cases.dist((ug, ig, w0g), f), serial_cases.dist((uo, io, w0o), f)
            print(ctx, f"gpu-oracle {e_go:.3g} gpu-f64 {e_gf:.3g} oracle-f64 {e_of:.3g} (tolerance units)")
            assert e_gf <= max(1.0, e_of), ctx
            continue
        assert abs(lg - lo) <= 2e-5 * max(1.0, abs(lo)), ctx
        assert e_go <= 1.0, ctx       # tables (and W0) within 3e-4 of their largest entry
    assert seen_agg == 13


def test_serial_aggregation_outliers_are_conditioning():
    """The three aggregation cases of `tools/serial_sweep.py 400 11` that end beyond the suite's table tolerance (cases 68, 344,
    362: emb_dim 8 / 20 / 20, 5-lane or 2-lane rows with masked columns, 89-92 dependent steps on 3-6 users), regenerated
    from the sweep's own generator and arbitrated: the serial GPU walk, the fp32 oracle and the float64 model on the same
    inputs.  Measured (profiles/r03_serial_arbiter.txt, tolerance units): GPU-float64 3.8 / 0.15 / 1.8 against
    oracle-float64 9.2 / 1.16 / 7.3 — in every case it is the fp32 ORACLE that drifts from the exact trajectory, the GPU
    (fused multiply-adds) stays closer to it; after the first 8 steps all three agree to 1e-3 of the tolerance or better,
    with no jump at the 32-call W0 update.  So: conditioning of the compounding blend, not the <8,*> aggregation path."""
    from tests import serial_cases
    want = {68: "<8,4,16,1>", 344: "<8,8,16,1>", 362: "<8,4,16,1>"}
    for case, c in serial_cases.sweep_cases(363, 11):
        if case not in want:
            continue
        g = serial_cases.run_gpu(c)
        assert want[case] in g[3] and c["agg"], g[3]
        o, f = serial_cases.run_oracle(c), serial_cases.run_f64(c)
        e_go, e_gf, e_of = serial_cases.dist(g[:3], o), serial_cases.dist(g[:3], f), serial_cases.dist(o, f)
        print(f"case {case} {g[3]}: gpu-oracle {e_go:.3g} gpu-f64 {e_gf:.3g} oracle-f64 {e_of:.3g} (units of the 3e-4 table tolerance)")
        assert e_go > 1.0                                   # the case is an outlier of the sweep ...
        assert e_gf <= e_of                                 # ... in which the GPU is the one closer to the exact trajectory
        # short horizon: no indexing error shows in the first steps, including the first W0 update at call 32
        for steps in (1, 8, 33):
            gs, os_, fs = serial_cases.run_gpu(c, steps), serial_cases.run_oracle(c, steps), serial_cases.run_f64(c, steps)
            assert serial_cases.dist(gs[:3], fs) <= max(0.25, 1.5 * serial_cases.dist(os_, fs)), (case, steps)


@pytest.mark.parametrize("accl,nproc", [(False, 1), (True, 1), (False, 2), (True, 2)])
def test_distributed_main_under_torchrun(tmp_path, accl, nproc):
    """`torchrun -m heat_amd.cf.main --distributed`: the user-sharded trainer (device-mode engine on torch tensors, item
    sync, loss and Recall reduced over ranks) end to end with one rank on this GPU (child process); with `accl` the
    behaviour aggregator runs device-resident and its W0 joins the synchronised state.  nproc=2 puts two ranks on this one
    GPU with the gloo backend (RCCL refuses two ranks per device): real HIP engines on real shards, windows agreed
    across ranks, item table (and W0) all-reduced, loss and Recall reduced over ranks."""
    import os
    import subprocess
    import sys
    import yaml
    cfg = yaml.safe_load(open("heat_amd/cf/benchmarks/Gowalla/MF_CCL/configs/config_pr1.yaml"))
    cfg["model_config"]["epochs"] = 3
    cfg["model_config"]["eval_interval"] = 2
    cfg["model_config"]["use_aggregator"] = bool(accl)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(29581 + int(accl) + 2 * (nproc - 1)), "-m", "heat_amd.cf.main", "--config", str(path), "--synthetic", "gowalla", "--scale", "0.1",
           "--distributed"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1", **({"HEAT_CF_DIST_BACKEND": "gloo"} if nproc > 1 else {})))
    out = res.stdout
    assert res.returncode == 0, out[-3000:]
    losses = [float(l.split("loss:")[1].split(";")[0]) for l in out.splitlines() if l.startswith("epoch:")]
    assert len(losses) == 3 and losses[2] < losses[0]
    rec = [float(l.split("Recall(k=20):")[1]) for l in out.splitlines() if l.startswith("[Metrics]")]
    assert rec and (0.005 if accl else 0.02) < rec[-1] < 1.0


def test_two_ranks_on_one_gpu_equal_single_process_training(tmp_path):
    """The multi-GPU path with real HIP engines: two ranks (gloo; both on this GPU, RCCL refuses two ranks per device)
    train their user shards in windows of 64 interactions and all-reduce the item table after every window.  The shards
    touch disjoint item rows, so the `sum` rule must reproduce single-process training of the whole list (oracle), the
    replicas must agree bit for bit, and the all-gathered user table must be the concatenation of the shards."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(0)
    U, I, d, N, T, epochs = 40, 200, 64, 4, 600, 2
    users = np.sort(rng.integers(0, U, T))
    half = U // 2
    pos = np.where(users < half, rng.integers(0, I // 2, T), rng.integers(I // 2, I, T))
    negs = np.where((users < half)[:, None], rng.integers(0, I // 2, (T, N)), rng.integers(I // 2, I, (T, N))).astype(np.uint64)
    clicks = np.stack([users, pos], axis=1).astype(np.uint64)
    uw = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    iw = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    np.savez(tmp_path / "problem.npz", clicks=clicks, negs=negs, uw=uw, iw=iw, num_negs=N, lr=0.01, epochs=epochs)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29591", os.path.join(os.path.dirname(__file__), "_gpu_shard_worker.py"), str(tmp_path), "64"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert res.returncode == 0 and res.stdout.count("SHARD_OK") == 2, res.stdout[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["iw"], r1["iw"])
    u1, i1 = uw.copy(), iw.copy()
    ref = orc.Engine(clicks, u1, i1, num_negs=N, l_r=0.01, clip_val=1.0)
    want_losses = []
    for _ in range(epochs):
        ref.lr_step()
        want_losses.append(ref.train_range(0, T, negs) / T)
        ref.zero_grad()
        ref.epoch = ref.epoch + 1
    got_u = np.concatenate([r0["uw"], r1["uw"]])
    assert np.array_equal(r0["full_u"], got_u) and np.array_equal(r1["full_u"], got_u)
    assert_tables_close(r0["iw"], i1, scale=np.abs(i1).max(), rtol=2e-4)
    assert_tables_close(got_u, u1, scale=np.abs(u1).max(), rtol=2e-4)
    np.testing.assert_allclose(r0["losses"], want_losses, rtol=1e-4)      # global mean loss, reduced over the two ranks
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
