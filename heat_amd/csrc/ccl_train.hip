// ccl_train.hip — the fused per-interaction SimpleX/CCL step as hand-written gfx950 HIP.
//
// What it replaces (paths relative to /root/reference/cf_cpu/src):
//   models/matrix_factorization.cpp:15-181   forward_backward (gather, 3+2N dots, cosine, softmax-CE, grads)
//   optimizers/sgd.cpp:14-26, optimizer.cpp:17-22   clip + SGD on persistent gradient rows
//   negative_samplers/uniform_random_negative_sampler.cpp:17-36   uniform negatives (here: Philox on the GPU)
//   train/engine.cpp:327-340   the per-thread sequential walk over a chunk of interactions
//   memory/array.hpp:46-55     row gather / scatter (here: 16-byte-per-lane buffer loads/stores)
//
// Mapping to CDNA4
//   * one workgroup = one sequential interaction stream (the analogue of one OpenMP thread walking a
//     `schedule(dynamic,512)` chunk): all interactions of a stream are processed in stored order, so a user's
//     run is updated sequentially and the user row lives in registers while the user does not change; a workgroup is
//     ONE 64-lane wavefront up to 17 rows per interaction and NW = 2..16 wavefronts for wider interactions;
//   * an embedding row is read by LPR = emb_dim/4 (rounded up to 8/16/32/64) consecutive lanes, 16 B per lane
//     (one `buffer_load_dwordx4` fetches R = 64/LPR whole rows); row-wise dot products are 4 DPP adds inside
//     the 16-lane DPP row (+ ds_bpermute steps for 128/256-wide rows);
//   * the N negatives sit in NGW x NW = ceil(N/R) register groups; per-negative scalars (norms, cosines, softmax
//     weights) are computed once per row group, the softmax runs across groups and rows with two cross-row
//     shuffles;
//   * write-back is Hogwild: per kind of row either the reference's literal overwrite or float atomic adds
//     (W += -lr*G, G += G_new - G_read), shaped as 256 contiguous bytes per wave instruction through a 1 KiB LDS
//     transpose tile; default policy: positive row atomic, negative rows overwritten (DESIGN.md "Hogwild at GPU
//     concurrency").  With AUX = sc1 all row traffic is device-coherent across the 8 XCD L2s;
//   * large num_negs: NW waves per workgroup share one stream (negative slots split across waves, every wave draws all
//     ids itself, softmax statistics and user gradient exchanged through LDS: two barriers per interaction); RR = late
//     re-read of the negative rows (short read-modify-write window); behaviour aggregation (AGG) keeps W0 (while it
//     fits) and the last 32 gradient pairs in LDS and splits its history gather and d x d product over the waves;
//     TS = 12 single-wave streams per workgroup sharing the random-tile sampler's tile, whose weight deltas live in LDS;
//   * negatives come from Philox4x32-10 keyed by (seed, epoch) with the interaction index as counter: no host
//     round trip, no sampler state in memory, schedule-independent.
#include "ccl_device.hpp"
#include "ccl_train.hpp"

#ifndef HEATCF_PART
#define HEATCF_PART 0
#endif

#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace heatcf
{

// Stream boundaries follow user runs: a stream owns every run that STARTS inside its nominal slice, so one user's
// interactions are never split between two concurrently running waves (each would keep the user row in registers and
// overwrite the other's updates).  x is moved forward to the next run start, looking at most `cap` entries ahead; a
// run longer than that is split at x.  Both neighbours evaluate the same function, so slices stay disjoint.
__device__ __forceinline__ uint64_t align_to_user_run(const uint2* clicks, uint64_t x, uint64_t begin, uint64_t end,
                                                      uint32_t cap, int lane)
{
    if (x >= end) return end;
    if (x <= begin) return begin;
    const uint32_t prev_user = clicks[x - 1].x;
    for (uint32_t off = 0; off < cap; off += 64)
    {
        const uint64_t i = x + off + (uint64_t)lane;
        const bool differs = (i >= end) || (clicks[i].x != prev_user);
        const uint64_t m = __builtin_amdgcn_ballot_w64(differs);
        if (m != 0ull) return x + off + (uint64_t)__builtin_ctzll(m);
    }
    return x;
}

// Float atomic adds run at full rate only when one wave instruction covers 256 contiguous bytes (one dword per lane);
// the row layout of this kernel is 16 B per lane, so a component-wise atomic would touch 4 dwords of every 64-B
// segment (4x the memory-side atomic requests per byte: measured 0.31 TB/s vs ~1.3 TB/s).  The deltas of one register
// group (64 lanes x 16 B = 1 KiB = R rows) are therefore transposed through a wave-private 1 KiB LDS tile: lane l of
// chunk q adds dword l of bytes [256q, 256q+256) of the tile, whose owner in the row layout is lane 16q + l/4.
struct AtomicOffsets
{
    uint32_t o[4];
};

__device__ __forceinline__ AtomicOffsets atomic_offsets(uint32_t off, int lane)
{
    AtomicOffsets a;
#pragma unroll
    for (int q = 0; q < 4; ++q) a.o[q] = lane_get(off, 16 * q + (lane >> 2)) + 4u * (uint32_t)(lane & 3); // OOB stays OOB
    return a;
}

template <int NCHUNK>
__device__ __forceinline__ void atomic_add_tile(__amdgpu_buffer_rsrc_t rsrc, const AtomicOffsets& ao, f32x4 delta,
                                                float* tile, int lane)
{
    *reinterpret_cast<f32x4*>(tile + lane * 4) = delta;
#pragma unroll
    for (int q = 0; q < NCHUNK; ++q)
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tile[q * 64 + lane], rsrc, (int)ao.o[q], 0, 0);
}

// does any other lane of my 16-lane DPP row hold the same id?  (row_ror:1..15)
template <int ROT>
__device__ __forceinline__ void dup_rot_step(uint32_t id, bool& hit)
{
    const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id, 0x120 + ROT, 0xF, 0xF, true);
    hit = hit || (other == id);
}
__device__ __forceinline__ void dup_rotations(uint32_t id, bool& hit)
{
    dup_rot_step<1>(id, hit);  dup_rot_step<2>(id, hit);  dup_rot_step<3>(id, hit);  dup_rot_step<4>(id, hit);
    dup_rot_step<5>(id, hit);  dup_rot_step<6>(id, hit);  dup_rot_step<7>(id, hit);  dup_rot_step<8>(id, hit);
    dup_rot_step<9>(id, hit);  dup_rot_step<10>(id, hit); dup_rot_step<11>(id, hit); dup_rot_step<12>(id, hit);
    dup_rot_step<13>(id, hit); dup_rot_step<14>(id, hit); dup_rot_step<15>(id, hit);
}

// ---- multi-wave workgroups (NW > 1): every wave HOLDS all negative slots of the interaction (who evaluates the generator: below) --
// Lane k, register v holds slot v*64 + k.  The ~100-instruction generator is cheaper than sharing ids through LDS: no
// barrier, and every wave can count the multiplicity of ITS slots against all ids with one compare + ballot per slot
// instead of a num_negs-iteration scan.  The draws do not depend on table data, so the kernel evaluates them for the
// NEXT interaction in the shadow of the current gather.
// EXT = false compiles the caller-fed path (a global load) away: the draw that runs in the shadow of a gather must not
// hand the compiler a value that MAY come from memory, or it waits for every outstanding load (vmcnt counts in order)
// before the multiplicity count and the shadow is gone.
// With <= 16 negatives (one id register, uniform sampler) ONE generator evaluation serves four consecutive interactions,
// as in the single-wave variants: lane l computes slot (l & 15) of interaction idx + (l >> 4) when jj (the interaction's
// place in its batch of 64) is a multiple of four, `raw4` keeps the result and the other three interactions only fetch
// their 16 lanes from it.  Same (slot, interaction index) counters, hence the same ids.
// With more than 16 negatives the generator call is shared ACROSS THE WAVES instead: wave w evaluates it for interaction
// first + g NW + w of the stream's g-th group of NW interactions and leaves the raw ids in an LDS ring (two groups deep,
// group g + 1 is drawn while group g runs; the workgroup barriers of the interactions in between order writes and reads);
// every wave then fetches the ids of the interaction at hand from the ring — one evaluation per wave and NW interactions
// instead of one per interaction.  In the Yelp18 yaml's kernel the generator + multiplicity count ran longer than the gather
// whose shadow they were meant to fill (profiles/r03_train_timeline.txt): 26.9 -> 26.1 ms per epoch, Gowalla yaml 25.3 -> 24.1.
// Same (slot, interaction index) counters, hence the same ids.
template <int NIDA, int NW>
__device__ __forceinline__ void draw_group(const TrainArgs& a, uint64_t first, uint32_t g, int wave, int lane, uint32_t* ring)
{
    const uint64_t idx = first + (uint64_t)g * NW + (uint64_t)wave;
    uint32_t* dst = ring + ((g & 1u) * NW + (uint32_t)wave) * (uint32_t)(NIDA * 64);
#pragma unroll
    for (int v = 0; v < NIDA; ++v)
        dst[v * 64 + lane] = uniform_item(philox_draw64((uint32_t)(v * 64 + lane), a.sample_base + idx, a.key), a.num_items);
}

template <int NIDA, int NW, bool EXT = true>
__device__ __forceinline__ void draw_all_ids(const TrainArgs& a, uint64_t idx, uint32_t pos, uint64_t first, int lane,
                                             uint32_t (&nid)[NIDA], uint32_t& raw4, int jj, const uint32_t* ring, bool use_ring)
{
    if (use_ring)
    {
        const uint32_t k = (uint32_t)(idx - first);
        const uint32_t* src = ring + (((k / NW) & 1u) * NW + k % NW) * (uint32_t)(NIDA * 64);
#pragma unroll
        for (int v = 0; v < NIDA; ++v)
        {
            uint32_t id = src[v * 64 + lane];
            if (!a.sampling_call && id == pos) id = nid[v];      // ignore_pos_sampling keeps the previous id
            nid[v] = id;
        }
        return;
    }
    if constexpr (NIDA == 1)
    {
        if (!(EXT && a.ext_negs != nullptr) && !(a.tile_size != 0u && a.sampling_call) && a.num_negs <= 16u)
        {
            if ((jj & 3) == 0)
                raw4 = uniform_item(philox_draw64((uint32_t)(lane & 15), a.sample_base + idx + (uint64_t)(lane >> 4), a.key), a.num_items);
            uint32_t id = lane_get(raw4, ((jj & 3) << 4) | (lane & 15));
            if (!a.sampling_call && id == pos) id = nid[0];      // ignore_pos_sampling keeps the previous id
            nid[0] = id;
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < NIDA; ++v)
    {
        const uint32_t slot = (uint32_t)(v * 64 + lane);
        uint32_t id;
        if (EXT && a.ext_negs != nullptr)
        {
            id = slot < a.num_negs ? a.ext_negs[(idx - a.ext_base) * a.num_negs + slot] : 0u;
        }
        else
        {
            if (a.tile_size != 0u && a.sampling_call)
                id = tile_item(slot, a.sample_base + idx, a.key, blockIdx.x, idx - first, a.tile_size, a.refresh_interval,
                               a.num_items);
            else
                id = uniform_item(philox_draw64(slot, a.sample_base + idx, a.key), a.num_items);
            if (!a.sampling_call && id == pos) id = nid[v];      // ignore_pos_sampling keeps the previous id
        }
        nid[v] = id;
    }
}

// mult[g] = (copies of this lane's row id among all slots) | (copies in EARLIER slots) << 16 for the slot (g, rr) this lane
// serves; returns the largest multiplicity among the wave's slots (wave-uniform).
template <int NIDA, int NGW, int R>
__device__ __forceinline__ uint32_t slot_multiplicity(const uint32_t (&nid)[NIDA], uint32_t N, uint32_t wave_base, int lane,
                                                      int rr, uint32_t (&mult)[NGW])
{
    uint32_t cmax = 1u;
    // pass 1, branch-free scalar code: the largest multiplicity among the wave's slots (duplicates are rare: N^2/2I of the
    // interactions); pass 2 below, the per-slot bookkeeping, only runs when there is one
    uint32_t most = 0u;
#pragma unroll
    for (int g = 0; g < NGW; ++g)
    {
        mult[g] = 1u;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const uint32_t gslot = wave_base + (uint32_t)(g * R + r);          // wave-uniform
            uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)nid[0], (int)(gslot & 63u));
#pragma unroll
            for (int v = 1; v < NIDA; ++v)
            {
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)nid[v], (int)(gslot & 63u));
                sid = (int)(gslot >> 6) == v ? t : sid;
            }
            uint32_t eq = 0u;
#pragma unroll
            for (int v = 0; v < NIDA; ++v)
                eq += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nid[v] == sid && (uint32_t)(v * 64 + lane) < N));
            eq = gslot < N ? eq : 0u;
            most = eq > most ? eq : most;
        }
    }
    if (most <= 1u) return cmax;
#pragma unroll
    for (int g = 0; g < NGW; ++g)
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const uint32_t gslot = wave_base + (uint32_t)(g * R + r);          // wave-uniform
            if (gslot < N)
            {
                uint32_t sid = 0u;
#pragma unroll
                for (int v = 0; v < NIDA; ++v)
                    if ((int)(gslot >> 6) == v) sid = (uint32_t)__builtin_amdgcn_readlane((int)nid[v], (int)(gslot & 63u));
                uint64_t same[NIDA];
                uint32_t eq = 0u;
#pragma unroll
                for (int v = 0; v < NIDA; ++v)
                {
                    same[v] = __builtin_amdgcn_ballot_w64(nid[v] == sid && (uint32_t)(v * 64 + lane) < N);
                    eq += (uint32_t)__builtin_popcountll(same[v]);
                }
                if (eq > 1u)                                                     // rare, wave-uniform
                {
                    uint32_t earlier = 0u;
#pragma unroll
                    for (int v = 0; v < NIDA; ++v)
                    {
                        const uint32_t lo = (uint32_t)(v * 64);
                        const uint64_t below = gslot >= lo + 64u ? ~0ull : (gslot <= lo ? 0ull : ((1ull << (gslot - lo)) - 1ull));
                        earlier += (uint32_t)__builtin_popcountll(same[v] & below);
                    }
                    if (rr == r) mult[g] = eq | (earlier << 16);
                    cmax = eq > cmax ? eq : cmax;
                }
            }
        }
    }
    return cmax;
}

template <int LPR, int AUX>
__device__ __forceinline__ void flush_user_row(const TrainArgs& a, uint32_t user, bool cut, f32x4 u4, f32x4 gu4, f32x4 u4_in,
                                               f32x4 gu4_in, float* tile, int lane, int rr, bool col_ok, uint32_t col_off)
{
    const size_t o = (size_t)user * a.row_bytes;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char*)a.user_w + o, a.row_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc((const char*)a.user_g + o, a.row_bytes);
    const uint32_t off = (rr == 0 && col_ok) ? col_off : OOB_OFF;
    if (cut)
    {
        constexpr int NCH = LPR >= 16 ? LPR / 16 : 1;
        const AtomicOffsets ao = atomic_offsets(off, lane);
        atomic_add_tile<NCH>(rw, ao, u4 - u4_in, tile, lane);
        atomic_add_tile<NCH>(rg, ao, gu4 - gu4_in, tile, lane);
    }
    else
    {
        buf_store<AUX>(rw, off, u4);
        buf_store<AUX>(rg, off, gu4);
    }
}

// a.upd_bits selects, per kind of row, how the update is written back (wave-uniform branches):
//   bit 0: negative W by atomic add   bit 1: negative G by atomic add
//   bit 2: positive W by atomic add   bit 3: positive G by atomic add      (0 = the reference's literal overwrite)
//
// NW waves form one workgroup and share ONE interaction stream: wave w owns the negative slots
// [w*NGW*R, (w+1)*NGW*R); user and positive rows are replicated in every wave's registers (every wave computes the
// identical update of them, wave 0 writes them back).  Softmax statistics and the user-gradient partial sums cross
// waves through LDS (two workgroup barriers per interaction: softmax statistics, user-gradient partials); every wave
// holds all negative ids of the interaction (no exchange of the ids the dots need; since round 3 the GENERATOR CALL is shared:
// one evaluation per four interactions with <= 16 negatives, one per wave and NW interactions through an LDS ring otherwise,
// draw_group).  NW = 1 compiles all of that away.
//
// AGG = true adds the reference's behaviour aggregation (behavior_aggregators/behavior_aggregators.cpp:51-153, called
// unconditionally at matrix_factorization.cpp:38,152): u <- 0.4 u + 0.6 (mean of the user's history item rows) W0, in
// place, before the dots; after the negative sweep the W0 gradient means (x) (0.6 g_u) is accumulated per stream and
// applied every 32 calls (W0 -= lr * acc/32, here by float atomic adds on the shared W0), and g_u *= 0.4.
// The stream keeps a private LDS copy of W0 (refreshed after each of its own W0 updates) and the last <= 32
// (means, 0.6 g_u) pairs; the accumulation order of the reference (call by call) is preserved when the pairs are summed.
//
// RR = true ("late re-read", a.upd_bits bit 4): a negative row's W is read a second time together with its G row, a few
// groups ahead of its update in the backward sweep, and the update is applied to THAT value (W_fresh - lr*G_new).  The
// gradient itself is still the one of the forward pass.  This shrinks the read-modify-write window of a negative row
// from the whole interaction (gather -> softmax -> sweep) to one memory round trip, i.e. the share of negative updates
// lost to a concurrent writer by the same factor, without the memory-side atomic rate.
//
// TS > 1 ("tile-resident", SURVEY 8f row 2: the random-tile sampler with its tile held in LDS): TS single-wave streams
// form one workgroup and share ONE tile of `tile_size` item rows for the whole launch
// (negative_samplers/random_tile_negative_sampler.cpp:22-45 keeps a tile per worker for refresh_interval calls).  What
// lives in LDS is the tile's accumulated weight DELTA, D[tile_size][emb_dim] fp32 (512 x 64 x 4 B = 128 KB of the 160 KB):
//   forward  : a negative row is W_global[id] + D[j]   (j = its tile index);
//   backward : the clipped-SGD step of a negative row is added to D[j] (a 16-byte LDS read-modify-write per lane) instead
//              of being stored to W; G is read-modify-written in global memory as always;
//   flush    : at the end of the launch (= tile refresh; the engine only selects this kernel when a stream makes fewer
//              than refresh_interval calls per launch) every D row is added to its W row by 256-byte float atomics.
// Nothing is lost and no base copy is needed (there is no room for one); other workgroups see this tile's negative updates
// one launch late.  A negative row then costs 3 row transfers instead of 4 (the W store stays on chip).
template <int LPR, int NGW, int AUX, int NW, bool AGG, bool RR = false, int TS = 1>
__global__ __launch_bounds__(64 * NW * TS) void ccl_train_kernel(TrainArgs a)
{
    static_assert(TS == 1 || (NW == 1 && !AGG && !RR), "the tile-resident mode is built for plain single-wave variants");
    const bool neg_w_atomic = (a.upd_bits & 1u) != 0u, neg_g_atomic = (a.upd_bits & 2u) != 0u;
    const bool pos_w_atomic = (a.upd_bits & 4u) != 0u, pos_g_atomic = (a.upd_bits & 8u) != 0u;
    const bool any_atomic = (a.upd_bits & 0xFu) != 0u;
    constexpr int R = 64 / LPR;            // rows fetched by one wave instruction
    constexpr int WCAP = NGW * R;          // negative slots held by one wave
    constexpr int NIDV = (WCAP + 63) / 64; // id registers per lane (lane k, register v: slot wave_base + v*64 + k)
    constexpr int NIDA = NW > 1 ? (NW * WCAP + 63) / 64 : 1; // NW > 1: every wave holds ALL slots (lane k, register v: slot v*64 + k)
    // how many groups ahead the G rows are fetched in the backward sweep: a short look-ahead keeps the read-modify-write
    // window of a negative's G row small (fewer Hogwild collisions) where rows are small and hot; the multi-wave
    // variants (large rows, large tables) fetch deep to cover HBM latency
    // RR (late re-read): the point is the SHORT window, so W and G of a group are requested only two groups ahead of its
    // update and not before the softmax statistics are known
    constexpr int GPF = RR ? (NGW < 2 ? NGW : 2) : (NW > 1 ? (NGW < 16 ? NGW : 16) : (NGW < 4 ? NGW : 4));
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int swave = TS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // stream inside the workgroup
    const uint32_t stream_id = blockIdx.x * (uint32_t)TS + (uint32_t)swave;
    const uint32_t wave_base = (uint32_t)(wave * WCAP);
    const int sub = lane & (LPR - 1);      // 16-byte column of the row
    const int rr = lane / LPR;             // which of the R rows of a group this lane serves
    const uint32_t col_off = (uint32_t)sub * 16u;
    const bool col_ok = col_off < a.row_bytes;
    const uint32_t N = a.num_negs;
    const float lr = a.lr, clip = a.clip;
    const float eps = 1e-8f;               // matrix_factorization.cpp:53
    const float score_mul = (float)(1.0 / 0.07); // :101-103 (double converted to the array's scalar type)

    const __amdgpu_buffer_rsrc_t item_w = make_rsrc(a.item_w, a.item_bytes);
    const __amdgpu_buffer_rsrc_t item_g = make_rsrc(a.item_g, a.item_bytes);

    uint64_t first = a.begin + (uint64_t)stream_id * a.per_block;
    if (first > a.end) first = a.end;
    uint64_t last = first + a.per_block;
    if (last > a.end) last = a.end;
    if (a.align_cap != 0u)
    {
        first = align_to_user_run(a.clicks, first, a.begin, a.end, a.align_cap, lane);
        last = align_to_user_run(a.clicks, last, a.begin, a.end, a.align_cap, lane);
    }
    __shared__ __attribute__((aligned(16))) float tile_all[NW * TS * 256]; // per-wave transpose tile for the atomics
    __shared__ float sh_stat[NW > 1 ? NW * 2 : 1];                    // per-wave (max, sum of exp)
    __shared__ __attribute__((aligned(16))) float sh_gu[NW > 1 ? NW * 64 * 4 : 4]; // per-wave user-gradient partials
    __shared__ float sh_slg[NW > 1 ? NW : 1];
    float* tile = tile_all + (wave + swave) * 256;
    // aggregation state (dynamic LDS: W0 copy [D*D] (when it fits, a.agg_w0_lds) | pair ring [32][2][DP] | means [DP] |
    // two cross-wave partial buffers [NW][DP] each (NW > 1)); D = emb_dim, DP = 4*LPR
    extern __shared__ __attribute__((aligned(16))) float agg_lds[];
    constexpr int DP = 4 * LPR;
    const int D = (int)a.emb_dim;
    const bool w0_in_lds = AGG && a.agg_w0_lds != 0u;
    float* agg_w0 = agg_lds;
    float* agg_pairs = agg_lds + (w0_in_lds ? D * D : 0);
    float* agg_means = agg_pairs + (AGG ? 32 * 2 * DP : 0);
    float* agg_part_h = agg_means + (AGG ? DP : 0);              // per-wave partial history sums
    float* agg_part_f = agg_part_h + (AGG && NW > 1 ? NW * DP : 0); // per-wave partial products
    uint32_t agg_iter = 0u;                 // behavior_aggregators.cpp:31 (per worker, per epoch)
    uint32_t agg_H = 0u;
    uint32_t hid[4] = {0u, 0u, 0u, 0u};     // history ids of the current user: lane l holds his[l], his[64 + l], ...
    f32x4 means4 = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t w0_rsrc = make_rsrc(AGG ? a.w0 : a.item_w, AGG ? (uint32_t)(D * D * 4) : 16u);
    if (w0_in_lds)
    {
        for (int t = (int)threadIdx.x; t < D * D / 4; t += 64 * NW)   // W0 is [D,D] row-major, D % 4 == 0
            reinterpret_cast<f32x4*>(agg_w0)[t] = buf_load<AUX>(w0_rsrc, (uint32_t)t * 16u);
        if (NW > 1) __syncthreads();
    }
    float* agg_saved = (AGG && a.agg_state) ? a.agg_state + (size_t)stream_id * agg_state_floats(LPR) : nullptr;
    if (AGG && agg_saved)
    {
        // the stream's aggregator as the previous launch of this epoch left it: call counter and the pairs not yet applied
        agg_iter = reinterpret_cast<const uint32_t*>(agg_saved)[0];
        const int pending = (int)(agg_iter & 31u) * 2 * DP;
        for (int t = (int)threadIdx.x; t < pending; t += 64 * NW) agg_pairs[t] = agg_saved[4 + t];
        if (NW > 1) __syncthreads();
    }
    float* tile_delta = agg_lds;            // TS > 1: D[tile_size][emb_dim]
    uint32_t nidj[NIDV];                    // TS > 1: tile index of each negative slot
    uint32_t raw_batch_j = 0u;
#pragma unroll
    for (int v = 0; v < NIDV; ++v) nidj[v] = 0u;
    if constexpr (TS > 1)
    {
        for (uint32_t t = threadIdx.x; t < a.tile_size * (uint32_t)D / 4u; t += 64u * TS)
            reinterpret_cast<f32x4*>(tile_delta)[t] = f32x4{0, 0, 0, 0};
        __syncthreads();
    }
    // A user run cut by a stream boundary is also being updated by the neighbouring stream: its row is then written
    // back as an atomic delta (nothing lost); a run owned entirely by this stream is written back with plain stores.
    uint32_t cut_head_user = 0xFFFFFFFFu, cut_tail_user = 0xFFFFFFFFu;
    if (any_atomic && first < last)
    {
        if (first > a.begin && a.clicks[first - 1].x == a.clicks[first].x) cut_head_user = a.clicks[first].x;
        if (last < a.end && a.clicks[last].x == a.clicks[last - 1].x) cut_tail_user = a.clicks[last - 1].x;
    }

    uint32_t cur_user = 0xFFFFFFFFu;
    f32x4 u4 = {0, 0, 0, 0}, gu4 = {0, 0, 0, 0};
    f32x4 u4_in = {0, 0, 0, 0}, gu4_in = {0, 0, 0, 0};
    uint32_t nid[NIDV];
#pragma unroll
    for (int v = 0; v < NIDV; ++v) nid[v] = 0u; // engine.cpp:298: neg_ids zero-initialised per worker
    // NW > 1: all slots in every wave, multiplicities of the wave's own slots, and the same for the NEXT interaction
    uint32_t nid_all[NIDA], nxt_nid[NIDA], mult[NW > 1 ? NGW : 1], nxt_mult[NW > 1 ? NGW : 1];
    uint32_t cmax_w = 1u, nxt_cmax = 1u;
    bool have_next = false;
#pragma unroll
    for (int v = 0; v < NIDA; ++v) { nid_all[v] = 0u; nxt_nid[v] = 0u; }
    double loss_acc = 0.0;
    uint32_t raw_batch = 0u;               // raw draws of up to four interactions (BATCH4)
    // multi-wave variants with more than 16 negatives share the generator through an LDS ring (draw_group)
    __shared__ uint32_t id_ring[NW > 1 ? 2 * NW * NIDA * 64 : 1];
    const bool use_ring = NW > 1 && a.ext_negs == nullptr && !(a.tile_size != 0u && a.sampling_call) && a.num_negs > 16u;
    if (NW > 1 && use_ring)
    {
        draw_group<NIDA, NW>(a, first, 0u, wave, lane, id_ring);
        __syncthreads();
    }


    for (uint64_t base = first; base < last; base += 64)
    {
        // one coalesced load brings the (user,item) pairs of the next 64 interactions (datasets/click_dataset.cpp:17-22)
        uint2 pair = make_uint2(0u, 0u);
        if (base + (uint64_t)lane < last) pair = a.clicks[base + (uint64_t)lane];
        const int cnt = (last - base) < 64 ? (int)(last - base) : 64;

        for (int j = 0; j < cnt; ++j)
        {
            const uint32_t user = (uint32_t)__builtin_amdgcn_readlane((int)pair.x, j);
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)pair.y, j);
            const uint64_t idx = base + (uint64_t)j;
            if (NW > 1 && a.exact_order)
            {
                // B0 (serial / parity mode only): the previous interaction's row writes of EVERY wave are performed before
                // any wave gathers again (within one wave program order already guarantees this; across waves nothing
                // does).  Hogwild runs skip it: the next gather overlaps the draining stores.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }

            // ---- negatives: lane k, register v owns slot wave_base + v*64 + k --------------------------------
            // Single-wave variants with <= 16 slots draw FOUR interactions per Philox evaluation: lane l computes slot
            // (l & 15) of interaction idx + (l >> 4); the same (slot, interaction index) counter as always, so the ids
            // are unchanged — 3 of 4 interactions skip the ~95-instruction generator.
            constexpr bool BATCH4 = (NW == 1 && WCAP <= 16 && NIDV == 1);
            if (BATCH4 && a.ext_negs == nullptr)
            {
                if ((j & 3) == 0)
                {
                    const uint64_t bidx = a.sample_base + idx + (uint64_t)(lane >> 4);
                    if (a.tile_size != 0u && a.sampling_call)
                        raw_batch = tile_item((uint32_t)(lane & 15), bidx, a.key, blockIdx.x, idx + (uint64_t)(lane >> 4) - first,
                                              a.tile_size, a.refresh_interval, a.num_items, &raw_batch_j);
                    else
                        raw_batch = uniform_item(philox_draw64((uint32_t)(lane & 15), bidx, a.key), a.num_items);
                }
                uint32_t id = lane_get(raw_batch, ((j & 3) << 4) | (lane & 15));
                if (TS > 1) nidj[0] = lane_get(raw_batch_j, ((j & 3) << 4) | (lane & 15));
                if (!a.sampling_call && id == pos) id = nid[0];      // ignore_pos_sampling: keep the previous id
                nid[0] = id;
                if (a.neg_out != nullptr && (uint32_t)lane < N) a.neg_out[(idx - a.neg_out_base) * N + (uint32_t)lane] = id;
            }
            else if constexpr (NW > 1)
            {
                if (have_next)                       // drawn in the shadow of the previous interaction's gather
                {
#pragma unroll
                    for (int v = 0; v < NIDA; ++v) nid_all[v] = nxt_nid[v];
#pragma unroll
                    for (int g = 0; g < NGW; ++g) mult[g] = nxt_mult[g];
                    cmax_w = nxt_cmax;
                }
                else
                {
                    draw_all_ids<NIDA, NW>(a, idx, pos, first, lane, nid_all, raw_batch, j, id_ring, use_ring);
                    cmax_w = slot_multiplicity<NIDA, NGW, R>(nid_all, N, wave_base, lane, rr, mult);
                }
                if (a.neg_out != nullptr && wave == 0)
                {
#pragma unroll
                    for (int v = 0; v < NIDA; ++v)
                        if ((uint32_t)(v * 64 + lane) < N) a.neg_out[(idx - a.neg_out_base) * N + (uint32_t)(v * 64 + lane)] = nid_all[v];
                }
            }
            else
#pragma unroll
            for (int v = 0; v < NIDV; ++v)
            {
                const uint32_t wslot = (uint32_t)(v * 64 + lane);
                const uint32_t slot = wave_base + wslot;
                const bool mine = wslot < (uint32_t)WCAP && slot < N;
                uint32_t id;
                if (a.ext_negs != nullptr)
                {
                    id = mine ? a.ext_negs[(idx - a.ext_base) * N + slot] : 0u;
                }
                else
                {
                    // the tile sampler's ignore_pos_sampling() does not use the tile
                    // (random_tile_negative_sampler.cpp:47-57), only its sampling() does (:23-45)
                    if (a.tile_size != 0u && a.sampling_call)
                        id = tile_item(slot, a.sample_base + idx, a.key, blockIdx.x, idx - first, a.tile_size,
                                       a.refresh_interval, a.num_items, &nidj[v]);
                    else
                        id = uniform_item(philox_draw64(slot, a.sample_base + idx, a.key), a.num_items);
                    // ignore_pos_sampling (uniform_random_negative_sampler.cpp:26-36): a draw equal to the
                    // positive leaves the slot unchanged (previous interaction's id, initially 0)
                    if (!a.sampling_call && id == pos) id = nid[v];
                }
                nid[v] = id;
                if (a.neg_out != nullptr && mine) a.neg_out[(idx - a.neg_out_base) * N + slot] = id;
            }

            // ---- user row: registers while the user does not change (write back on change) -----------------
            if (user != cur_user)
            {
                if (cur_user != 0xFFFFFFFFu && wave == 0)
                    flush_user_row<LPR, AUX>(a, cur_user, cur_user == cut_head_user || cur_user == cut_tail_user, u4, gu4,
                                             u4_in, gu4_in, tile, lane, rr, col_ok, col_off);
                cur_user = user;
                const size_t o = (size_t)user * a.row_bytes;
                const uint32_t uo = col_ok ? col_off : OOB_OFF;
                u4 = buf_load<AUX>(make_rsrc((const char*)a.user_w + o, a.row_bytes), uo);
                gu4 = buf_load<AUX>(make_rsrc((const char*)a.user_g + o, a.row_bytes), uo);
                u4_in = u4;
                gu4_in = gu4;
                if (AGG)
                {
                    agg_H = a.masks[user];                                        // behavior_aggregators.cpp:61-62
                    const uint32_t* hrow = a.his + (size_t)user * a.max_his;       // :60
#pragma unroll
                    for (int q = 0; q < 4; ++q) hid[q] = (uint32_t)(64 * q + lane) < agg_H ? hrow[64 * q + lane] : 0u;
                }
            }
            // ---- behaviour aggregation: the first pass of the history gather is requested before anything else, the
            //      positive / negative rows right behind it, and the next interaction's ids are drawn in their shadow —
            //      the aggregator's two barriers then stand behind ONE memory round trip instead of following a second
            // history slot hh = hh0 + rr with hh0 a multiple of R, the same for the whole wave: the id register (hh / 64) is
            // wave-uniform, so ONE cross-lane fetch serves (not one per id register and a select)
            auto his_id = [&](uint32_t hh0) {
                const uint32_t reg = hh0 >> 6;
                const uint32_t src = reg == 0u ? hid[0] : reg == 1u ? hid[1] : reg == 2u ? hid[2] : hid[3];
                return lane_get(src, (int)((hh0 & 63u) + (uint32_t)rr));
            };
            f32x4 his_part[AGG ? 8 : 1];
            if (AGG)
            {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const uint32_t hh0 = (uint32_t)(wave + q * NW) * (uint32_t)R, hh = hh0 + (uint32_t)rr;
                    const uint32_t id = his_id(hh0);
                    his_part[q] = buf_load<AUX>(item_w, (hh < agg_H && col_ok) ? id * a.row_bytes + col_off : OOB_OFF);
                }
            }

            // ---- gather: positive row (replicated in every row group) + this wave's negative rows (W only; the
            //      G rows are fetched a few groups ahead of their use in the backward sweep) ----------------------
            const uint32_t poff = col_ok ? pos * a.row_bytes + col_off : OOB_OFF;
            f32x4 p4 = buf_load<AUX>(item_w, poff);
            f32x4 gp4 = buf_load<AUX>(item_g, poff);

            f32x4 n4[NGW];
            uint32_t noff[NGW];
            uint32_t doff[TS > 1 ? NGW : 1];                 // TS > 1: float offset of the row's delta in LDS (or ~0)
#pragma unroll
            for (int g = 0; g < NGW; ++g)
            {
                const int wk = g * R + rr;                   // slot of this lane's row inside the wave
                uint32_t id;
                if constexpr (NW > 1)
                {
                    const uint32_t gslot = wave_base + (uint32_t)wk;
                    id = lane_get(nid_all[0], (int)(gslot & 63u));
#pragma unroll
                    for (int v = 1; v < NIDA; ++v)
                    {
                        const uint32_t t = lane_get(nid_all[v], (int)(gslot & 63u));
                        id = (int)(gslot >> 6) == v ? t : id;
                    }
                }
                else
                {
                    id = lane_get(nid[(g * R) / 64], wk & 63);
                }
                const bool valid = wave_base + (uint32_t)wk < N;
                noff[g] = (valid && col_ok) ? id * a.row_bytes + col_off : OOB_OFF;
                n4[g] = buf_load<AUX>(item_w, noff[g]);
                if constexpr (TS > 1)
                {
                    const uint32_t jrow = lane_get(nidj[(g * R) / 64], wk & 63);
                    doff[g] = (valid && col_ok) ? jrow * a.emb_dim + (uint32_t)sub * 4u : 0xFFFFFFFFu;
                }
            }
            if constexpr (TS > 1)
            {
                // the row as this workgroup sees it: global weights + the tile's accumulated delta
#pragma unroll
                for (int g = 0; g < NGW; ++g)
                    if (doff[g] != 0xFFFFFFFFu) n4[g] += *reinterpret_cast<const f32x4*>(tile_delta + doff[g]);
            }

            if constexpr (NW > 1)
            {
                // while the gather is in flight, draw the NEXT interaction
                have_next = (j + 1 < cnt) && a.ext_negs == nullptr;      // caller-fed ids (tests) are fetched at the top instead
                if (have_next)
                {
                    const uint32_t pos_n = (uint32_t)__builtin_amdgcn_readlane((int)pair.y, j + 1);
#pragma unroll
                    for (int v = 0; v < NIDA; ++v) nxt_nid[v] = nid_all[v];
                    draw_all_ids<NIDA, NW, false>(a, idx + 1, pos_n, first, lane, nxt_nid, raw_batch, j + 1, id_ring, use_ring);
                    nxt_cmax = slot_multiplicity<NIDA, NGW, R>(nxt_nid, N, wave_base, lane, rr, nxt_mult);
                }
                // the first interaction of a group of NW draws the next group (this wave: its w-th interaction) into the other
                // half of the ring: last read two interactions ago, first read NW - 1 interactions from now
                if (use_ring && (uint32_t)(idx - first) % (uint32_t)NW == 0u)
                    draw_group<NIDA, NW>(a, first, (uint32_t)(idx - first) / (uint32_t)NW + 1u, wave, lane, id_ring);
            }

            if (AGG)
            {
                // ---- aggregator forward (behavior_aggregators.cpp:96-122) -----------------------------------------
                // The H history rows are fetched R per wave instruction, instruction q by wave q % NW (first pass: above); the
                // d x d product likewise by rows of W0.  With NW > 1 the per-wave partial sums cross through LDS (two barriers)
                // and every wave adds them in the same order, so the replicated user row stays identical across waves.
                f32x4 hs = {0, 0, 0, 0};
                const uint32_t n_inst = (agg_H + (uint32_t)R - 1u) / (uint32_t)R;
#pragma unroll
                for (int q = 0; q < 8; ++q) hs += his_part[q];
                for (uint32_t q0 = (uint32_t)(wave + NW * 8); q0 < n_inst; q0 += (uint32_t)(NW * 8))
                {
                    f32x4 part[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                    {
                        const uint32_t hh0 = (q0 + (uint32_t)(q * NW)) * (uint32_t)R, hh = hh0 + (uint32_t)rr;
                        const uint32_t id = his_id(hh0);
                        part[q] = buf_load<AUX>(item_w, (hh < agg_H && col_ok) ? id * a.row_bytes + col_off : OOB_OFF);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) hs += part[q];
                }
                hs.x = cross_sum<LPR>(hs.x);
                hs.y = cross_sum<LPR>(hs.y);
                hs.z = cross_sum<LPR>(hs.z);
                hs.w = cross_sum<LPR>(hs.w);
                if (NW > 1)
                {
                    if (rr == 0) *reinterpret_cast<f32x4*>(agg_part_h + wave * DP + sub * 4) = hs;
                    __syncthreads();                                                  // BA1
                    hs = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int w = 0; w < NW; ++w) hs += *reinterpret_cast<const f32x4*>(agg_part_h + w * DP + sub * 4);
                }
                means4 = hs * (1.0f / (float)agg_H);                                  // :63,105
                if (rr == 0) *reinterpret_cast<f32x4*>(agg_means + sub * 4) = means4;   // every wave writes the same values
                f32x4 f4 = {0, 0, 0, 0};
                for (int i = wave * R + rr; i < D; i += NW * R)                       // :118 f = means (1xD) * W0 (DxD)
                {
                    const float m = agg_means[i];
                    f32x4 wrow = {0, 0, 0, 0};
                    if (col_ok)
                        wrow = w0_in_lds ? *reinterpret_cast<const f32x4*>(agg_w0 + i * D + sub * 4)
                                         : buf_load<AUX>(w0_rsrc, (uint32_t)(i * D * 4) + col_off);
                    f4 += m * wrow;
                }
                f4.x = cross_sum<LPR>(f4.x);
                f4.y = cross_sum<LPR>(f4.y);
                f4.z = cross_sum<LPR>(f4.z);
                f4.w = cross_sum<LPR>(f4.w);
                if (NW > 1)
                {
                    if (rr == 0) *reinterpret_cast<f32x4*>(agg_part_f + wave * DP + sub * 4) = f4;
                    __syncthreads();                                                  // BA2
                    f4 = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int w = 0; w < NW; ++w) f4 += *reinterpret_cast<const f32x4*>(agg_part_f + w * DP + sub * 4);
                }
                const float gamma = 0.4f, omg = 1.0f - gamma;                          // :37, :122
                u4 = gamma * u4 + omg * f4;
                agg_iter += 1u;                                                       // :124
            }

            // ---- duplicate negatives inside one interaction (rare): multiplicity per slot ------------------
            // Reference semantics (matrix_factorization.cpp:127-150): slot k re-reads G fresh, so c copies of one
            // row apply G <- clip(G + g) c times; every copy writes W_stale - lr*G, last writer wins.  All copies
            // compute the identical c-fold result here, so whichever store lands last is the reference's value.
            uint32_t eq[NIDV], earlier[NIDV];
            uint32_t cmax = 1u;
            if constexpr (NW > 1)
            {
                cmax = cmax_w;                                           // multiplicities were counted with the draw
#pragma unroll
                for (int v = 0; v < NIDV; ++v) { eq[v] = 0u; earlier[v] = 0u; }
            }
            else
            {
#pragma unroll
                for (int v = 0; v < NIDV; ++v) { eq[v] = 0u; earlier[v] = 0u; }
                // Fast rejection when the interaction's N <= 16 slots sit in one 16-lane DPP row (lanes 0..15 of a
                // single-wave variant): 15 row rotations compare every pair; the counting scan below only runs when a
                // duplicate exists (0.14 % of the interactions at AmazonBooks shape).
                bool need_scan = true;
                if (NW == 1 && WCAP <= 16)
                {
                    const uint32_t mine = (uint32_t)lane < N ? nid[0] : 0xFFFFFF00u + (uint32_t)lane; // unique sentinels
                    bool hit = false;
                    dup_rotations(mine, hit);
                    need_scan = __builtin_amdgcn_ballot_w64(hit && lane < 16) != 0ull;
                }
                for (uint32_t s = 0; need_scan && s < N; ++s)
                {
                    uint32_t sid = 0u;
#pragma unroll
                    for (int v = 0; v < NIDV; ++v)
                        if ((int)(s >> 6) == v) sid = (uint32_t)__builtin_amdgcn_readlane((int)nid[v], (int)(s & 63u));
#pragma unroll
                    for (int v = 0; v < NIDV; ++v)
                    {
                        const uint32_t wslot = (uint32_t)(v * 64 + lane);
                        const uint32_t slot = wave_base + wslot;
                        const bool same = nid[v] == sid && wslot < (uint32_t)WCAP && slot < N;
                        eq[v] += same ? 1u : 0u;
                        earlier[v] += (same && s < slot) ? 1u : 0u;
                    }
                }
                bool any_dup = false;
#pragma unroll
                for (int v = 0; v < NIDV; ++v) any_dup = any_dup || (eq[v] > 1u);
                if (__builtin_amdgcn_ballot_w64(any_dup) != 0ull)
                {
#pragma unroll
                    for (int v = 0; v < NIDV; ++v) cmax = eq[v] > cmax ? eq[v] : cmax;
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1)
                    {
                        const uint32_t o = lane_get(cmax, lane ^ m);
                        cmax = o > cmax ? o : cmax;
                    }
                    cmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmax); // provably uniform loop bound
                }
            }

            // ---- forward: dots, cosines, softmax cross-entropy (matrix_factorization.cpp:43-109) -----------
            const float uu = row_sum<LPR>(dot4(u4, u4));
            const float pp = row_sum<LPR>(dot4(p4, p4));
            const float up = row_sum<LPR>(dot4(u4, p4));
            const float unorm = fast_sqrt(fmaxf(uu, eps));
            const float pnorm = fast_sqrt(fmaxf(pp, eps));
            const float unorm3 = unorm * unorm * unorm;
            const float pnorm3 = pnorm * pnorm * pnorm;
            const float r_u3_p = fast_rcp(unorm3 * pnorm);
            const float r_u_p3 = fast_rcp(unorm * pnorm3);
            const float upcos = up * fast_rcp(unorm * pnorm);

            float un[NGW], nn[NGW], es[NGW];
            float mx = -INFINITY;
#pragma unroll
            for (int g = 0; g < NGW; ++g)
            {
                un[g] = row_sum<LPR>(dot4(u4, n4[g]));
                nn[g] = row_sum<LPR>(dot4(n4[g], n4[g]));
                const float nnorm = fast_sqrt(nn[g] < eps ? eps : nn[g]);
                const float c = un[g] * fast_rcp(unorm * nnorm);
                const bool valid = wave_base + (uint32_t)(g * R + rr) < N;
                es[g] = valid ? (c - upcos) * score_mul : -INFINITY;    // the score; becomes exp(score - max) below
                mx = fmaxf(mx, es[g]);
            }
            // the first G rows are requested here, so that their latency overlaps the softmax arithmetic; the rest follow
            // GPF groups ahead of their use in the backward sweep
            f32x4 gpf[GPF];                                             // G rows in flight
            f32x4 wpf[RR ? GPF : 1];                                    // late re-read of the W rows (RR)
            if (!RR)
            {
#pragma unroll
                for (int g = 0; g < GPF && g < NGW; ++g) gpf[g] = buf_load<AUX>(item_g, noff[g]);
            }
            mx = cross_max<LPR>(mx);                                    // maximum over this wave's slots
            float ssum = 0.0f;
#pragma unroll
            for (int g = 0; g < NGW; ++g)
            {
                es[g] = (es[g] == -INFINITY) ? 0.0f : expf(es[g] - mx);
                ssum += es[g];
            }
            ssum = cross_sum<LPR>(ssum);
            if (NW > 1)
            {
                // ONE exchange: every wave publishes (its maximum, its sum of exp(score - its maximum)); the workgroup's
                // maximum M and sum follow by rescaling with exp(m_w - M) (the streaming-softmax identity)
                if (lane == 0)
                {
                    sh_stat[wave * 2] = mx;
                    sh_stat[wave * 2 + 1] = ssum;
                }
                __syncthreads();                                        // B2
                // lane l takes the pair of wave l % NW: one expf per lane and a log2(NW)-step DPP tree instead of NW expf's
                // per wave (every wave runs the same tree on the same pairs, so the replicated results stay identical)
                const float mw = sh_stat[(lane & (NW - 1)) * 2], sw = sh_stat[(lane & (NW - 1)) * 2 + 1];
                float big = mw;
                big = fmaxf(big, dpp_mov<0xB1>(big));                               // xor 1
                if (NW >= 4) big = fmaxf(big, dpp_mov<0x4E>(big));                  // xor 2
                if (NW >= 8) big = fmaxf(big, dpp_mov<0x141>(big));                 // other quad of the 8 lanes
                if (NW >= 16) big = fmaxf(big, dpp_mov<0x140>(big));                // other half of the 16 lanes
                float tot = (mw == -INFINITY) ? 0.0f : sw * expf(mw - big);
                tot += dpp_mov<0xB1>(tot);
                if (NW >= 4) tot += dpp_mov<0x4E>(tot);
                if (NW >= 8) tot += dpp_mov<0x141>(tot);
                if (NW >= 16) tot += dpp_mov<0x140>(tot);
                const float scale = (mx == -INFINITY) ? 0.0f : expf(mx - big);
#pragma unroll
                for (int g = 0; g < NGW; ++g) es[g] *= scale;
                mx = big;
                ssum = tot;
            }
            // :106 adds exp(-max) computed in double; fp32 expf differs by <= 1 ulp of the sum
            const float Z = ssum + expf(-mx);
            const float rcp_z = fast_rcp(Z);
            const float loss = mx + logf(Z);
            loss_acc += (double)loss;

            // ---- backward + clipped SGD + scatter (matrix_factorization.cpp:118-174, sgd.cpp:14-26) ----------
            // u_p_cos_u_grad, u_p_cos_p_grad (:62-63)
            const f32x4 upu = (uu * p4 - up * u4) * r_u3_p;
            const f32x4 upp = -(pp * u4 - up * p4) * r_u_p3;
            f32x4 gu_acc = {0, 0, 0, 0};
            float slg = 0.0f;
            if (RR)
            {
                __builtin_amdgcn_sched_barrier(0);                      // not hoisted above the softmax exchange
#pragma unroll
                for (int g = 0; g < GPF && g < NGW; ++g)
                {
                    gpf[g] = buf_load<AUX>(item_g, noff[g]);
                    wpf[RR ? g : 0] = buf_load<AUX>(item_w, noff[g]);
                }
            }
            // The sweep exists in three forms (MODE): 0 = plain stores, 1 = negative-row atomics, 2 = the policy bits tested
            // per row group.  Where the G (and W) rows are fetched a few groups ahead of their use, the (wave-uniform) policy
            // picks form 0 or 1 OUTSIDE the loop: with the test inside, the compiler's vmcnt bookkeeping merged the atomic and
            // the plain branch after every row group and made each group wait for the stores of the one before (Yelp18 shape:
            // 34.2 -> 29.6 ms per epoch).  The wide multi-wave kernels without the late re-read fetch all their G rows before
            // the first store, lose nothing to that merge, and keep form 2 (the straight-line forms cost them registers).
            auto sweep = [&](auto mode_tag) __attribute__((always_inline))
            {
            constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
            for (int g = 0; g < NGW; ++g)
            {
                const f32x4 g_read = gpf[g % GPF];
                const f32x4 w_base = RR ? wpf[RR ? g % GPF : 0] : n4[g];       // the W value the update is applied to
                if (g + GPF < NGW)
                {
                    gpf[g % GPF] = buf_load<AUX>(item_g, noff[g + GPF]);
                    if (RR) wpf[RR ? g % GPF : 0] = buf_load<AUX>(item_w, noff[g + GPF]);
                }
                const float lg = (es[g] * rcp_z) * score_mul;                   // :109
                const float nnorm = fast_sqrt(nn[g] < eps ? eps : nn[g]);
                const float nnorm3 = nnorm * nnorm * nnorm;
                const float r_u3_n = fast_rcp(unorm3 * nnorm);                  // :136
                const float r_u_n3 = fast_rcp(unorm * nnorm3);                  // :137
                const f32x4 unu = (uu * n4[g] - un[g] * u4) * r_u3_n;           // :138
                const f32x4 unn = (nn[g] * u4 - un[g] * n4[g]) * r_u_n3;        // :139 (raw nn, not eps-clamped)
                gu_acc += lg * (unu - upu);                                     // :141
                slg += lg;                                                      // :142 (pos grad += lg * upp)
                const f32x4 t = lg * unn;
                f32x4 gn = clip4(g_read + t, clip);                             // :143,147 + sgd.cpp:22
                uint32_t woff = noff[g];
                if (cmax > 1u)                                                  // duplicates present (wave-uniform)
                {
                    const int src = (g * R + rr) & 63;
                    const uint32_t cm = NW > 1 ? (mult[NW > 1 ? g : 0] & 0xFFFFu) : lane_get(eq[(g * R) / 64], src);
                    for (uint32_t c = 1; c < cmax; ++c)
                    {
                        const f32x4 g2 = clip4(gn + t, clip);
                        if (c < cm) gn = g2;
                    }
                    // only the first copy of a duplicated row writes its (c-fold) result: the reference's copies all
                    // write the same W_stale - lr*G_c / G_c (last writer wins = one effective update), and a later
                    // copy's streamed G fetch may already see the first copy's write
                    const uint32_t before = NW > 1 ? (mult[NW > 1 ? g : 0] >> 16) : lane_get(earlier[(g * R) / 64], src);
                    if (before != 0u) woff = OOB_OFF;
                }
                // a negative that equals the positive is not written: the positive's write-back comes last in the
                // reference (matrix_factorization.cpp:171-174) and overwrites it
                if (woff != OOB_OFF && woff - col_off == pos * a.row_bytes) woff = OOB_OFF;
                if constexpr (TS > 1)
                {
                    // sgd.cpp:23 as a delta: the step goes into the tile's LDS accumulator, re-read just before (one 16-byte
                    // read-modify-write per lane; four scalar LDS float atomics per lane ran 2x slower for the whole epoch,
                    // and two of the TS streams meeting in one row inside these few cycles is rarer than the collisions the
                    // plain stores of the table path accept)
                    if (woff != OOB_OFF)
                    {
                        f32x4* dst = reinterpret_cast<f32x4*>(tile_delta + doff[g]);
                        *dst = *dst - lr * gn;
                    }
                    buf_store<AUX>(item_g, woff, gn);                           // :149
                }
                else if (MODE == 0 || (MODE == 2 && !neg_w_atomic && !neg_g_atomic))
                {
                    buf_store<AUX>(item_w, woff, w_base - lr * gn);             // sgd.cpp:23, :148
                    buf_store<AUX>(item_g, woff, gn);                           // :149
                }
                else
                {
                    const AtomicOffsets ao = atomic_offsets(woff, lane);
                    if (neg_w_atomic) atomic_add_tile<4>(item_w, ao, -(lr * gn), tile, lane); // W += -(lr*G)
                    else buf_store<AUX>(item_w, woff, w_base - lr * gn);
                    if (neg_g_atomic) atomic_add_tile<4>(item_g, ao, gn - g_read, tile, lane);
                    else buf_store<AUX>(item_g, woff, gn);
                }
                if ((g % GPF) == GPF - 1) __builtin_amdgcn_sched_barrier(0);    // bound how far G fetches are hoisted
            }
            };
            if constexpr (!RR && NW == 1 && GPF >= NGW)
            {
                // Every G row of the interaction is already in flight (GPF == NGW) and the rows arrive together.  Waiting
                // for all of them HERE, where only loads are outstanding, is free — and it spares the sweep its in-loop
                // waits: the compiler cannot count stores behind the wave-uniform duplicate branch, so it guarded the last
                // group's G row with `s_waitcnt vmcnt(0)` AFTER the stores of the groups before it, i.e. one full
                // write-through store drain per interaction (round 3, read off the ISA of <16,4,16,1>).
                // simm16: vmcnt = 0 (bits 3:0 and 15:14), expcnt = 7 and lgkmcnt = 15 (no wait)
                __builtin_amdgcn_sched_barrier(0);   // the softmax arithmetic above stays above: it is the shadow the G rows arrive in
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (RR || TS > 1) sweep(std::integral_constant<int, 0>{});    // bits 0-1 are rejected with bit 4 (engine.cpp)
            else if constexpr (NW > 1 || NGW > 16) sweep(std::integral_constant<int, 2>{});
            else if (!neg_w_atomic && !neg_g_atomic) sweep(std::integral_constant<int, 0>{});
            else sweep(std::integral_constant<int, 1>{});
            gu_acc.x = cross_sum<LPR>(gu_acc.x);
            gu_acc.y = cross_sum<LPR>(gu_acc.y);
            gu_acc.z = cross_sum<LPR>(gu_acc.z);
            gu_acc.w = cross_sum<LPR>(gu_acc.w);
            slg = cross_sum<LPR>(slg);
            if (NW > 1)
            {
                *reinterpret_cast<f32x4*>(sh_gu + (wave * 64 + lane) * 4) = gu_acc;
                if (lane == 0) sh_slg[wave] = slg;
                __syncthreads();                                                // B3
                gu_acc = f32x4{0, 0, 0, 0};
                slg = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w)
                {
                    gu_acc += *reinterpret_cast<const f32x4*>(sh_gu + (w * 64 + lane) * 4);
                    slg += sh_slg[w];
                }
            }

            if (AGG)
            {
                // ---- aggregator backward (behavior_aggregators.cpp:129-153) ----------------------------------------
                const float gamma = 0.4f, omg = 1.0f - gamma;
                const f32x4 gfull = gu4 + gu_acc;                                    // outs_grad = persistent G + this step
                const f32x4 fgrad = gfull * omg;                                     // :132
                const uint32_t slot32 = (agg_iter - 1u) & 31u;
                if (rr == 0 && wave == 0)
                {
                    *reinterpret_cast<f32x4*>(agg_pairs + (slot32 * 2 + 0) * DP + sub * 4) = means4;
                    *reinterpret_cast<f32x4*>(agg_pairs + (slot32 * 2 + 1) * DP + sub * 4) = fgrad;
                }
                if ((agg_iter & 31u) == 0u)                                          // :141 (iteration > 0 holds here)
                {
                    if (NW > 1) __syncthreads();                                     // the ring is complete and visible
                    for (int i0 = wave * R; i0 < D; i0 += NW * R)                    // R rows of W0 per tile, rows split over waves
                    {
                        const int i = i0 + rr;
                        f32x4 acc = {0, 0, 0, 0};
                        for (int c = 0; c < 32; ++c)                                 // :134-139 in call order
                        {
                            const float m = agg_pairs[(c * 2 + 0) * DP + (i < D ? i : 0)];
                            const f32x4 fg = *reinterpret_cast<const f32x4*>(agg_pairs + (c * 2 + 1) * DP + sub * 4);
                            acc += m * fg;
                        }
                        const f32x4 delta = -(a.agg_lr * (acc * 0.03125f));          // :143-144 (/32 is exact as *2^-5)
                        const uint32_t off = (i < D && col_ok) ? (uint32_t)(i * D * 4) + col_off : OOB_OFF;
                        const AtomicOffsets ao = atomic_offsets(off, lane);
                        atomic_add_tile<4>(w0_rsrc, ao, delta, tile, lane);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // own W0 updates performed ...
                    if (NW > 1) __syncthreads();                                     // ... by every wave, ring free again
                    if (w0_in_lds)
                    {
                        for (int t = (int)threadIdx.x; t < D * D / 4; t += 64 * NW)  // ... then refresh the private copy
                            reinterpret_cast<f32x4*>(agg_w0)[t] = buf_load<AUX>(w0_rsrc, (uint32_t)t * 16u);
                        if (NW > 1) __syncthreads();
                    }
                }
                gu4 = clip4(gfull * gamma, clip);                                    // :148-152 then sgd.cpp:22
            }
            else
            {
                gu4 = clip4(gu4 + gu_acc, clip);                                     // :166
            }
            u4 = u4 - lr * gu4;
            const f32x4 gp_old = gp4;
            gp4 = clip4(gp4 + slg * upp, clip);                                 // :169
            const uint32_t pst = (wave == 0 && rr == 0 && col_ok) ? pos * a.row_bytes + col_off : OOB_OFF;
            if (!pos_w_atomic && !pos_g_atomic)
            {
                buf_store<AUX>(item_w, pst, p4 - lr * gp4);                     // :173
                buf_store<AUX>(item_g, pst, gp4);                               // :174
            }
            else if (wave == 0)
            {
                constexpr int NCH = LPR >= 16 ? LPR / 16 : 1;                   // row 0 of the tile only
                const AtomicOffsets ao = atomic_offsets(pst, lane);
                if (pos_w_atomic) atomic_add_tile<NCH>(item_w, ao, -(lr * gp4), tile, lane);
                else buf_store<AUX>(item_w, pst, p4 - lr * gp4);
                if (pos_g_atomic) atomic_add_tile<NCH>(item_g, ao, gp4 - gp_old, tile, lane);
                else buf_store<AUX>(item_g, pst, gp4);
            }
        }
    }

    if (cur_user != 0xFFFFFFFFu && wave == 0)                                              // :171-172
        flush_user_row<LPR, AUX>(a, cur_user, cur_user == cut_head_user || cur_user == cut_tail_user, u4, gu4, u4_in,
                                 gu4_in, tile, lane, rr, col_ok, col_off);
    if (AGG && agg_saved)
    {
        if (NW > 1) __syncthreads();
        const int pending = (int)(agg_iter & 31u) * 2 * DP;
        for (int t = (int)threadIdx.x; t < pending; t += 64 * NW) agg_saved[4 + t] = agg_pairs[t];
        if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(agg_saved)[0] = agg_iter;
    }
    if (wave == 0 && lane == 0) a.loss_part[stream_id] = loss_acc;
    if constexpr (TS > 1)
    {
        // tile refresh = end of the launch: every accumulated delta row goes to its W row, 256 contiguous bytes per atomic
        __syncthreads();
        for (uint32_t jrow = (uint32_t)swave; jrow < a.tile_size; jrow += (uint32_t)TS)
        {
            const uint32_t id = tile_entry(jrow, a.key, blockIdx.x, 0ull, a.num_items);
            for (uint32_t c = (uint32_t)lane; c < a.emb_dim; c += 64u)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tile_delta[jrow * a.emb_dim + c], item_w, (int)(id * a.row_bytes + c * 4u), 0, 0);
        }
    }
}

#if HEATCF_PART == 0
// Deterministic fixed-order reduction of the per-stream loss partials (fp64), accumulated into *out.
__global__ __launch_bounds__(256) void loss_reduce_kernel(const double* part, uint32_t n, double* out)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1)
    {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += sh[0];
}

// u64 (user,item) pairs -> packed u32 pairs + range check (max ids written with atomicMax)
__global__ void pack_clicks_kernel(const uint64_t* in, uint2* out, uint64_t n, uint32_t* max_user, uint32_t* max_item,
                                   uint32_t* overflow)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t u = in[2 * i], it = in[2 * i + 1];
    if ((u >> 32) || (it >> 32)) atomicOr(overflow, 1u);
    out[i] = make_uint2((uint32_t)u, (uint32_t)it);
    atomicMax(max_user, (uint32_t)u);
    atomicMax(max_item, (uint32_t)it);
}

// Device-mode twin of the host loop in heat_cf_engine_create: history ids and lengths to u32, slots past the length
// zeroed, bad[0] |= 1 for a length above max_his, |= 2 for an id outside the item table.
__global__ void pack_history_kernel(const uint64_t* his, const uint64_t* masks, uint32_t* his32, uint32_t* masks32,
                                    uint64_t num_users, uint32_t max_his, uint64_t num_items, uint32_t* bad)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= num_users * max_his) return;
    const uint64_t u = t / max_his, k = t % max_his;
    const uint64_t h = masks[u];
    if (k == 0)
    {
        masks32[u] = (uint32_t)(h > max_his ? max_his : h);
        if (h > max_his) atomicOr(bad, 1u);
    }
    const uint64_t it = his[t];
    if (k < h && it >= num_items) atomicOr(bad, 2u);
    his32[t] = (k < h && it < num_items) ? (uint32_t)it : 0u;
}

// behavior_aggregators.cpp:63 divides by masks[u]: bad[0] |= 4 when a user with interactions has an empty history
__global__ void check_history_kernel(const uint2* clicks, uint64_t n, const uint32_t* masks32, uint64_t num_users, uint32_t* bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = clicks[i].x;
    if (u < num_users && masks32[u] == 0) atomicOr(bad, 4u);
}

// Sampler-only kernel: the ids the training kernel draws for interactions [begin,end) (tests, oracle feeding).
// One wave walks the range sequentially so that the ignore_pos "slot keeps previous id" state is reproduced for the
// stream layout given by per_block.
__global__ __launch_bounds__(64) void sample_negs_kernel(TrainArgs a, uint64_t out_base, uint64_t* out)
{
    const int lane = (int)threadIdx.x;
    const uint32_t N = a.num_negs;
    uint64_t first = a.begin + (uint64_t)blockIdx.x * a.per_block;
    uint64_t last = first + a.per_block;
    if (last > a.end) last = a.end;
    first = align_to_user_run(a.clicks, first, a.begin, a.end, a.align_cap, lane);
    last = align_to_user_run(a.clicks, last, a.begin, a.end, a.align_cap, lane);
    const int nidv = (int)((N + 63u) / 64u);
    uint32_t prev[4] = {0u, 0u, 0u, 0u};
    for (uint64_t idx = first; idx < last; ++idx)
    {
        const uint32_t pos = a.clicks[idx].y;
        for (int v = 0; v < nidv && v < 4; ++v)
        {
            const uint32_t slot = (uint32_t)(v * 64 + lane);
            uint32_t id;
            if (a.tile_size != 0u && a.sampling_call)
                id = tile_item(slot, a.sample_base + idx, a.key, blockIdx.x, idx - first, a.tile_size, a.refresh_interval,
                               a.num_items);
            else
                id = uniform_item(philox_draw64(slot, a.sample_base + idx, a.key), a.num_items);
            if (!a.sampling_call && id == pos) id = prev[v];
            prev[v] = id;
            if (slot < N) out[(idx - out_base) * N + slot] = (uint64_t)id;
        }
    }
}

#endif // HEATCF_PART == 0

// ---- dispatch -------------------------------------------------------------------------------------------------
template <int LPR, int NGW, int NW>
static hipError_t launch_variant(const TrainArgs& a, uint32_t grid, int aux, hipStream_t s)
{
    if (a.agg)
    {
        if (aux != AUX_SC1 || (a.upd_bits & 0x10u)) return hipErrorInvalidValue;
        const size_t lds = agg_lds_bytes(a.emb_dim, LPR, NW, a.agg_w0_lds != 0u);
        auto kern = ccl_train_kernel<LPR, NGW, AUX_SC1, NW, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a);
        return hipGetLastError();
    }
    if (a.tile_streams > 1u)
    {
        if constexpr (NW == 1 && LPR <= 16 && NGW <= 4)
        {
            if (aux != AUX_SC1 || a.tile_streams != (uint32_t)TILE_STREAMS || a.tile_size == 0u || !a.sampling_call || a.exact_order ||
                (a.upd_bits & 0x3Fu) != 0xCu)
                return hipErrorInvalidValue;
            const size_t lds = (size_t)a.tile_size * a.emb_dim * sizeof(float);
            auto kern = ccl_train_kernel<LPR, NGW, AUX_SC1, 1, false, false, TILE_STREAMS>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * TILE_STREAMS), lds, s, a);
            return hipGetLastError();
        }
        else
        {
            return hipErrorInvalidValue;
        }
    }
    if (a.upd_bits & 16u)
    {
        if (aux != AUX_SC1) return hipErrorInvalidValue;            // a fresh value needs device-coherent loads
        hipLaunchKernelGGL((ccl_train_kernel<LPR, NGW, AUX_SC1, NW, false, true>), dim3(grid), dim3(64 * NW), 0, s, a);
    }
    else if (aux == AUX_PLAIN)
        hipLaunchKernelGGL((ccl_train_kernel<LPR, NGW, AUX_PLAIN, NW, false>), dim3(grid), dim3(64 * NW), 0, s, a);
    else
        hipLaunchKernelGGL((ccl_train_kernel<LPR, NGW, AUX_SC1, NW, false>), dim3(grid), dim3(64 * NW), 0, s, a);
    return hipGetLastError();
}

// (lanes per row, register groups per wave, waves per workgroup): capacity = NGW * (64/LPR) * NW negatives.
// Per-wave register budget ~ 8 VGPRs per group + ~60; more waves per workgroup instead of more groups per wave once a
// wave would need > ~20 groups.
// The kernel family is compiled as three translation units of this one source (-DHEATCF_PART=0|1|2: rows of <= 16, 32, 64
// lanes), so that `make -j` builds them side by side; part 0 also carries everything that is not a template.
#define HEATCF_VARIANTS_P0(X) \
    X(8, 1, 1) X(8, 2, 1) X(8, 4, 1) X(8, 8, 1) X(8, 16, 1) \
    X(16, 2, 1) X(16, 4, 1) X(16, 8, 1) X(16, 16, 1) X(16, 16, 2) X(16, 2, 2) X(16, 1, 4)
#define HEATCF_VARIANTS_P1(X) \
    X(32, 4, 1) X(32, 8, 1) X(32, 16, 1) X(32, 32, 1) X(32, 16, 2) X(32, 8, 4) X(32, 16, 4) X(32, 4, 8) X(32, 2, 16)
#define HEATCF_VARIANTS_P2(X) \
    X(64, 8, 1) X(64, 16, 1) X(64, 16, 2) X(64, 16, 4) X(64, 25, 4) X(64, 13, 8) X(64, 16, 8)
#define HEATCF_VARIANTS(X) HEATCF_VARIANTS_P0(X) HEATCF_VARIANTS_P1(X) HEATCF_VARIANTS_P2(X)
#if HEATCF_PART == 0
#define HEATCF_MY_VARIANTS(X) HEATCF_VARIANTS_P0(X)
#define HEATCF_PART_FN(name) name##_part0
#elif HEATCF_PART == 1
#define HEATCF_MY_VARIANTS(X) HEATCF_VARIANTS_P1(X)
#define HEATCF_PART_FN(name) name##_part1
#else
#define HEATCF_MY_VARIANTS(X) HEATCF_VARIANTS_P2(X)
#define HEATCF_PART_FN(name) name##_part2
#endif
hipError_t launch_train_part0(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s);
hipError_t launch_train_part1(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s);
hipError_t launch_train_part2(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s);
int occupancy_part0(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim);
int occupancy_part1(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim);
int occupancy_part2(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim);

#if HEATCF_PART == 0
bool pick_variant(uint32_t emb_dim, uint32_t num_negs, bool single_wave, int* lpr_out, int* ng_out, int* nw_out)
{
    if (emb_dim == 0 || emb_dim % 4 != 0 || emb_dim > 256 || num_negs == 0) return false;
    const uint32_t need = emb_dim / 4;
    int lpr = 8;
    while ((uint32_t)lpr < need) lpr <<= 1;
    const int R = 64 / lpr;
    // sizing sweeps (BASELINE.json configs[2]): HEAT_CF_VARIANT="groups,waves" forces a compiled variant with enough capacity
    if (const char* ov = std::getenv("HEAT_CF_VARIANT"))
    {
        int g = 0, w = 0;
        if (std::sscanf(ov, "%d,%d", &g, &w) == 2)
        {
            bool ok = false;
#define X(L, G, W) if (L == lpr && G == g && W == w && (uint32_t)(G * R * W) >= num_negs && (!single_wave || W == 1)) ok = true;
            HEATCF_VARIANTS(X)
#undef X
            if (ok)
            {
                *lpr_out = lpr;
                *ng_out = g;
                *nw_out = w;
                return true;
            }
            return false;
        }
    }
    int best_cap = 0, best_g = 0, best_w = 0;
    // The tightest capacity that fits decides the candidates (up to 10 % slack); among them the fewest register groups per
    // wave, down to 4 (= most waves per workgroup).  Measured (profiles/r02_variant_sweep.txt): Yelp18 shape with 256
    // streams <32,4,8> 31.1 ms > <32,2,16> 37.2 = <32,8,4> 37.4 ms per epoch; synthetic 10 M x 1 M shape (d 256, 100
    // negatives, HBM-resident) <64,13,8> 12.8 M > <64,25,4> 9.3 M = <64,16,8> 9.1 M samples/s; splitting a small
    // interaction (<= 4 groups) over several waves is slower (AmazonBooks shard, 1162 streams: <16,4,1> 1.08 ms >
    // <16,2,2> 1.20 > <16,1,4> 1.37 ms).
    int tight = 0;
#define X(L, G, W) \
    if (L == lpr && (uint32_t)(G * R * W) >= num_negs && (!single_wave || W == 1) && (tight == 0 || G * R * W < tight)) tight = G * R * W;
    HEATCF_VARIANTS(X)
#undef X
    const int slack = tight + tight / 10;
#define X(L, G, W)                                                                                                       \
    if (L == lpr && (uint32_t)(G * R * W) >= num_negs && G * R * W <= slack && (!single_wave || W == 1) &&                 \
        (best_cap == 0 || (G >= 4 && (G < best_g || best_g < 4)) || (G == best_g && G * R * W < best_cap) ||               \
         (best_g < 4 && G < 4 && G * R * W < best_cap)))                                                                  \
    {                                                                                                                    \
        best_cap = G * R * W;                                                                                            \
        best_g = G;                                                                                                      \
        best_w = W;                                                                                                      \
    }
    HEATCF_VARIANTS(X)
#undef X
    if (best_cap == 0) return false;
    *lpr_out = lpr;
    *ng_out = best_g;
    *nw_out = best_w;
    return true;
}

// Behaviour aggregation: the same capacity spread over up to 4 waves (<L, G, 1> -> <L, G/W, W>), where such a variant is
// compiled.  The aggregator's history gather and d x d product split over the waves, and its streams are latency-bound at
// the count the Recall bound allows: AmazonBooks shape, 438 streams, <16,4,1> 35.1 ms per epoch, <16,2,2> 34.2, <16,1,4> 25.7.
void widen_for_aggregator(int lpr, int* ng, int* nw)
{
    if (std::getenv("HEAT_CF_VARIANT") || *nw != 1) return;
    int best_w = 1, best_g = *ng;
#define X(L, G, W) if (L == lpr && W > best_w && W <= 4 && G * W == *ng) { best_w = W; best_g = G; }
    HEATCF_VARIANTS(X)
#undef X
    *ng = best_g;
    *nw = best_w;
}

hipError_t launch_train(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s)
{
    if (lpr <= 16) return launch_train_part0(a, lpr, ng, nw, grid, aux, s);
    if (lpr == 32) return launch_train_part1(a, lpr, ng, nw, grid, aux, s);
    return launch_train_part2(a, lpr, ng, nw, grid, aux, s);
}
#endif // HEATCF_PART == 0

hipError_t HEATCF_PART_FN(launch_train)(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s)
{
#define X(L, G, W) if (lpr == L && ng == G && nw == W) return launch_variant<L, G, W>(a, grid, aux, s);
    HEATCF_MY_VARIANTS(X)
#undef X
    return hipErrorInvalidValue;
}

// resident workgroups per CU of the variant the engine will launch (register / LDS limited), from the runtime
template <int LPR, int NGW, int NW>
static int occupancy_variant(int aux, bool agg, uint32_t emb_dim)
{
    int blocks = 0;
    hipError_t e = hipErrorInvalidValue;
    if (agg)
    {
        const size_t lds = agg_lds_bytes(emb_dim, LPR, NW, agg_w0_fits_lds(emb_dim, LPR, NW));
        auto kern = ccl_train_kernel<LPR, NGW, AUX_SC1, NW, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 64 * NW, lds);
    }
    else if (aux == AUX_PLAIN)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, ccl_train_kernel<LPR, NGW, AUX_PLAIN, NW, false>, 64 * NW, 0);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, ccl_train_kernel<LPR, NGW, AUX_SC1, NW, false>, 64 * NW, 0);
    return e == hipSuccess ? blocks : 0;
}

int HEATCF_PART_FN(occupancy)(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim)
{
#define X(L, G, W) if (lpr == L && ng == G && nw == W) return occupancy_variant<L, G, W>(aux, agg, emb_dim);
    HEATCF_MY_VARIANTS(X)
#undef X
    return 0;
}

#if HEATCF_PART == 0
int query_blocks_per_cu(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim)
{
    if (lpr <= 16) return occupancy_part0(lpr, ng, nw, aux, agg, emb_dim);
    if (lpr == 32) return occupancy_part1(lpr, ng, nw, aux, agg, emb_dim);
    return occupancy_part2(lpr, ng, nw, aux, agg, emb_dim);
}

hipError_t launch_loss_reduce(const double* part, uint32_t n, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, part, n, out);
    return hipGetLastError();
}

hipError_t launch_pack_clicks(const uint64_t* in, uint2* out, uint64_t n, uint32_t* stats, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(pack_clicks_kernel, dim3(blocks), dim3(256), 0, s, in, out, n, stats, stats + 1, stats + 2);
    return hipGetLastError();
}

hipError_t launch_pack_history(const uint64_t* his, const uint64_t* masks, uint32_t* his32, uint32_t* masks32,
                               uint64_t num_users, uint32_t max_his, uint64_t num_items, const uint2* clicks,
                               uint64_t data_rows, uint32_t* bad, hipStream_t s)
{
    const uint64_t n = num_users * max_his;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_history_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, his, masks, his32, masks32,
                       num_users, max_his, num_items, bad);
    if (data_rows)
        hipLaunchKernelGGL(check_history_kernel, dim3((uint32_t)((data_rows + 255) / 256)), dim3(256), 0, s, clicks, data_rows,
                           masks32, num_users, bad);
    return hipGetLastError();
}

hipError_t launch_sample_negs(const TrainArgs& a, uint32_t grid, uint64_t out_base, uint64_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(sample_negs_kernel, dim3(grid), dim3(64), 0, s, a, out_base, out);
    return hipGetLastError();
}
#endif // HEATCF_PART == 0

} // namespace heatcf
