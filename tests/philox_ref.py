"""numpy restatement of Philox4x32-10 (Salmon et al., SC'11 — the generator hipRAND's device API calls
hiprandStatePhilox4_32_10_t) and of the engine's id mapping; test infrastructure for the on-GPU sampler."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32).copy() for x in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32).copy()
    k1 = np.asarray(k1, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + W0
            k1 = k1 + W1
    return c0, c1, c2, c3


def epoch_key(seed, epoch):
    return (int(seed) + 0x9E3779B97F4A7C15 * (int(epoch) + 1)) & 0xFFFFFFFFFFFFFFFF


def raw_negatives(idx, num_negs, num_items, key):
    """ids[len(idx), num_negs] before the ignore-positive rule: slot k of interaction i is
    mulhi64(philox(counter={k,0,i_lo,i_hi}, key).xy, num_items)."""
    idx = np.asarray(idx, dtype=np.uint64)
    n = idx.size
    slot = np.tile(np.arange(num_negs, dtype=np.uint32), n)
    ii = np.repeat(idx, num_negs)
    c2 = (ii & MASK).astype(np.uint32)
    c3 = (ii >> np.uint64(32)).astype(np.uint32)
    k0 = np.full(slot.shape, key & 0xFFFFFFFF, dtype=np.uint32)
    k1 = np.full(slot.shape, key >> 32, dtype=np.uint32)
    x, y, _, _ = philox4x32_10(slot, np.zeros_like(slot), c2, c3, k0, k1)
    draw = [(int(a) | (int(b) << 32)) for a, b in zip(x.tolist(), y.tolist())]
    ids = np.array([(d * num_items) >> 64 for d in draw], dtype=np.uint64)
    return ids.reshape(n, num_negs)


def negatives(clicks, begin, end, num_negs, num_items, key, per_block, sample_base=0, sampling_call=False):
    """What the training kernel draws for interactions [begin,end): raw ids with the reference's
    ignore_pos_sampling rule (uniform_random_negative_sampler.cpp:26-36) applied per sequential stream of
    `per_block` interactions (slot state starts at 0 at the head of every stream)."""
    idx = np.arange(begin, end, dtype=np.uint64)
    raw = raw_negatives(idx + np.uint64(sample_base), num_negs, num_items, key)
    if sampling_call:
        return raw
    out = raw.copy()
    prev = np.zeros(num_negs, dtype=np.uint64)
    for r in range(end - begin):
        if r % per_block == 0:
            prev[:] = 0
        pos = clicks[begin + r, 1]
        hit = raw[r] == pos
        out[r] = np.where(hit, prev, raw[r])
        prev = out[r]
    return out


def _mulhi64(draws, n):
    return np.array([(int(d) * int(n)) >> 64 for d in draws], dtype=np.uint64)


def _draw64(slot, idx, key):
    slot = np.asarray(slot, dtype=np.uint32)
    idx = np.asarray(idx, dtype=np.uint64)
    k0 = np.full(slot.shape, key & 0xFFFFFFFF, dtype=np.uint32)
    k1 = np.full(slot.shape, key >> 32, dtype=np.uint32)
    x, y, _, _ = philox4x32_10(slot, np.zeros_like(slot), (idx & MASK).astype(np.uint32),
                               (idx >> np.uint64(32)).astype(np.uint32), k0, k1)
    return [(int(a) | (int(b) << 32)) for a, b in zip(x.tolist(), y.tolist())]


def tile_negatives(begin, end, num_negs, num_items, key, tile_size, refresh_interval, per_block, sample_base=0):
    """sampling() of the random-tile sampler as the kernel computes it (ccl_device.hpp: tile_item), one stream per
    `per_block` interactions: tile index j from the per-interaction draw, tile entry from the (stream, tile_epoch) domain."""
    n = end - begin
    idx = np.repeat(np.arange(begin, end, dtype=np.uint64) + np.uint64(sample_base), num_negs)
    slot = np.tile(np.arange(num_negs, dtype=np.uint32), n)
    j = _mulhi64(_draw64(slot, idx, key), tile_size)
    rel = np.repeat(np.arange(n, dtype=np.uint64), num_negs)
    stream = rel // np.uint64(per_block)
    call = rel % np.uint64(per_block)
    tile_epoch = call // np.uint64(refresh_interval)
    tidx = (np.uint64(1) << np.uint64(63)) | (stream << np.uint64(32)) | tile_epoch
    ids = _mulhi64(_draw64(j.astype(np.uint32), tidx, key), num_items)
    return ids.reshape(n, num_negs)


def tile_entries(tile_size, num_items, key, owner=0, tile_epoch=0):
    """item ids of the tile of (owner, tile_epoch): entry j = mulhi64(philox(j, 2^63 | owner << 32 | tile_epoch), num_items)
    (ccl_device.hpp: tile_entry)."""
    j = np.arange(tile_size, dtype=np.uint32)
    tidx = np.full(tile_size, (1 << 63) | (int(owner) << 32) | int(tile_epoch), dtype=np.uint64)
    return _mulhi64(_draw64(j, tidx, key), num_items)
