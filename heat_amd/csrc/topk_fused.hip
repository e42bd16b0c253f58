// topk_fused.hip — evaluation without the [num_users, num_items] matrix (gfx950).
//
// Replaces the chain  Engine::evaluate0 (train/engine.cpp:388-400)  ->  PyMatrix copy (pybind/init_modules.cpp:122-129)
// -> sim[train] = -inf, argpartition, argsort (cf/metrics.py:21-29)  by one pass that keeps, per user, the running
// k best (score, item) pairs in LDS while 64 x 128 score tiles are produced in registers and never stored.
//
// Arithmetic: every score is the fp32 dot the oracle's evaluate0 defines (oracle/cf_oracle.c dot_fma): one fused
// multiply-add per k, k ascending, starting from 0 — the reference's own order is unspecified (Eigen GEMM,
// train/engine.cpp:394-398).  That chain is exactly what the fp32 matrix core computes: v_mfma_f32_32x32x2_f32
// accumulates its two k steps as two fmaf's, so the 64 x 128 score tiles come from MFMA (157 TFLOP/s peak, the same as
// the packed-FMA VALU peak, but it leaves the VALU free for staging and selection) and still rank bit-identically to the
// dense path (sim_panel_kernel uses the same fmaf chain).
//
// Order: pairs are ranked by (score descending, item id ascending); NaN scores are never selected; masked items score
// -inf and so can still fill the list when fewer than k unmasked items exist — exactly what topk_rows_kernel yields on
// the materialised panel (eval_kernels.hip), which stays as the k > 64 path and as the cross-check in tests.
#include "eval_kernels.hpp"

#include <math.h>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace heatcf
{
namespace
{
constexpr int TU = 64;     // users per workgroup
constexpr int TI = 128;    // items per tile
constexpr int KS = 32;     // k-slab staged through LDS
constexpr int LDA = TU + 4;
constexpr int LDB = TI + 4;
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr uint32_t NONE = 0xFFFFFFFFu;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int QW = 128;    // candidate queue entries per wave and tile

// CAP = list slots per user (32 or 64): the short form leaves room for a fourth workgroup per CU
template <int CAP> struct __attribute__((aligned(16))) SharedT
{
    float    a[KS / 2][LDA][2];    // [k pair][user][k parity]: the two lane halves of an MFMA read one 64-float span
    float    b[KS / 2][LDB][2];
    float    topv[TU][CAP]; // per user: the k best so far, best first
    uint32_t topi[TU][CAP];
    float    thr_v[TU];                  // = entry k-1, the one a candidate has to beat
    float    thr_sh[TU];                 // the best k-th score ANY item split of these users has reached (global exchange)
    uint32_t thr_i[TU];
    uint32_t mbits[TU][4];
    float    qv[4][QW];
    uint32_t qi[4][QW];
    uint8_t  qu[4][QW];
    uint32_t qn[4];
};

// true when (v, i) ranks ahead of (w, j)
__device__ __forceinline__ bool ahead(float v, uint32_t i, float w, uint32_t j) { return v > w || (v == w && i < j); }

// local user u belongs to wave owner(u): candidates of one user are only ever handled by one wave
__device__ __forceinline__ int owner(uint32_t u) { return (int)((u >> 2) & 3u); }

// One candidate for local user u, handled by a whole wave: lane l < k owns list slot l.  The entries ahead of the
// candidate are a prefix of the sorted list, so its position is a ballot + popcount and the tail moves down one slot.
template <class Shared> __device__ __forceinline__ void insert(Shared& s, uint32_t u, uint32_t item, float v, uint32_t k, int lane)
{
    // threshold and list slots are fetched together: one LDS round trip per candidate
    const bool have = (uint32_t)lane < k;
    const float    tv = s.thr_v[u];
    const uint32_t ti = s.thr_i[u];
    const float    ev = have ? s.topv[u][lane] : 0.0f;
    const uint32_t ei = have ? s.topi[u][lane] : 0u;
    const float    pv = (have && lane > 0) ? s.topv[u][lane - 1] : 0.0f;
    const uint32_t pi = (have && lane > 0) ? s.topi[u][lane - 1] : 0u;
    if (v != v || !ahead(v, item, tv, ti)) return;
    const uint32_t pos = (uint32_t)__popcll(__ballot(have && ahead(ev, ei, v, item)));
    if (have && (uint32_t)lane >= pos)
    {
        const float    nv = (uint32_t)lane == pos ? v : pv;
        const uint32_t ni = (uint32_t)lane == pos ? item : pi;
        s.topv[u][lane] = nv;
        s.topi[u][lane] = ni;
        if ((uint32_t)lane == k - 1)
        {
            s.thr_v[u] = nv;
            s.thr_i[u] = ni;
        }
    }
    // LDS operations of one wave complete in issue order: the next candidate's reads see these writes without a
    // wait; the compiler only must not move them
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

// every wave empties its own queue: 64 entries are fetched at once, then handed out lane by lane
template <class Shared> __device__ __forceinline__ void drain(Shared& s, uint32_t k, int wave, int lane)
{
    const uint32_t n = min(s.qn[wave], (uint32_t)QW);
    for (uint32_t base = 0; base < n; base += 64)
    {
        const uint32_t e = base + (uint32_t)lane;
        const uint32_t cu = e < n ? (uint32_t)s.qu[wave][e] : 0u;
        const uint32_t ci = e < n ? s.qi[wave][e] : 0u;
        const float    cv = e < n ? s.qv[wave][e] : 0.0f;
        const int cnt = (int)min(64u, n - base);
        for (int j = 0; j < cnt; ++j)
            insert(s, (uint32_t)__builtin_amdgcn_readlane((int)cu, j), (uint32_t)__builtin_amdgcn_readlane((int)ci, j),
                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), j)), k, lane);
    }
}

template <class Shared> __device__ __forceinline__ void push(Shared& s, uint32_t u, uint32_t item, float v)
{
    const int w = owner(u);
    const uint32_t slot = atomicAdd(&s.qn[w], 1u);
    if (slot < (uint32_t)QW)
    {
        s.qv[w][slot] = v;
        s.qi[w][slot] = item;
        s.qu[w][slot] = (uint8_t)u;
    }
}

// One k-slab on the matrix core.  Wave (wu, wi) owns users [32 wu, 32 wu + 32) x items [64 wi, 64 wi + 64) of the tile: two
// 32 x 32 accumulators.  v_mfma_f32_32x32x2_f32 takes, per lane l, A[l % 32][l / 32] and B[l / 32][l % 32]: lane half h
// supplies k = 2 kp + h, so k runs in ascending order through the chain of MFMAs.
template <class Shared>
__device__ __forceinline__ void mfma_step(const Shared& s, int kp, int wu, int wi, int n, int h, f16v& acc0, f16v& acc1)
{
    const float a = s.a[kp][wu * 32 + n][h];
    const float b0 = s.b[kp][wi * 64 + n][h];
    const float b1 = s.b[kp][wi * 64 + 32 + n][h];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
}

template <bool FULL, class Shared>
__device__ __forceinline__ void tile_slab_mfma(const Shared& s, int kpairs, int wu, int wi, int n, int h, f16v& acc0, f16v& acc1)
{
    if constexpr (FULL)
    {
#pragma unroll
        for (int kp = 0; kp < KS / 2; ++kp) mfma_step(s, kp, wu, wi, n, h, acc0, acc1);
    }
    else
    {
        for (int kp = 0; kp < kpairs; ++kp) mfma_step(s, kp, wu, wi, n, h, acc0, acc1);
    }
}

struct FusedArgs
{
    const float*    U;        // first user row of the range
    const float*    V;
    uint32_t        rows, num_items, d, k;
    uint32_t        tiles_per_split;
    const uint64_t* indptr;   // [rows + 1], relative to `items`; NULL = no masking; rows sorted ascending
    const uint32_t* items;
    float*          thr_shared; // [rows]: max over the item splits of a user's k-th best score so far (-inf at launch)
    float*          part_v;   // [splits, rows, k]
    uint32_t*       part_i;
};

// Registers of one slab in flight: two float4 of user row u0+sr and of item rows i0+sr, i0+64+sr at columns
// [k0+sc, k0+sc+4) and [k0+16+sc, k0+16+sc+4).  No branches: rows past the end of a table are clamped to its last row and
// columns past emb_dim to its last float4 — such scores are computed but never ranked (the filter checks user < rows,
// item < num_items; the k loop stops at emb_dim).  emb_dim % 4 == 0 and 16-byte aligned rows are checked on the host.
struct SlabRegs { f4 a[2], b0[2], b1[2]; };

__device__ __forceinline__ void load_slab(const FusedArgs& p, uint32_t u0, uint32_t i0, uint32_t k0, int sr, int sc, SlabRegs& g)
{
    const uint32_t d = p.d, last = p.num_items - 1u;
    const float* ua = p.U + (size_t)min(u0 + (uint32_t)sr, p.rows - 1u) * d;
    const float* v0 = p.V + (size_t)min(i0 + (uint32_t)sr, last) * d;
    const float* v1 = p.V + (size_t)min(i0 + 64u + (uint32_t)sr, last) * d;
#pragma unroll
    for (int j = 0; j < 2; ++j)
    {
        const uint32_t kk = min(k0 + (uint32_t)(sc + 16 * j), d - 4u);
        g.a[j] = *(const f4*)(ua + kk);
        g.b0[j] = *(const f4*)(v0 + kk);
        g.b1[j] = *(const f4*)(v1 + kk);
    }
}

template <class Shared> __device__ __forceinline__ void store_slab(Shared& s, int sr, int sc, const SlabRegs& g)
{
#pragma unroll
    for (int j = 0; j < 2; ++j)
    {
        const int kp = sc / 2 + 8 * j;
        *(f2*)&s.a[kp][sr][0] = f2{g.a[j][0], g.a[j][1]};
        *(f2*)&s.a[kp + 1][sr][0] = f2{g.a[j][2], g.a[j][3]};
        *(f2*)&s.b[kp][sr][0] = f2{g.b0[j][0], g.b0[j][1]};
        *(f2*)&s.b[kp + 1][sr][0] = f2{g.b0[j][2], g.b0[j][3]};
        *(f2*)&s.b[kp][64 + sr][0] = f2{g.b1[j][0], g.b1[j][1]};
        *(f2*)&s.b[kp + 1][64 + sr][0] = f2{g.b1[j][2], g.b1[j][3]};
    }
}

template <int CAP> __global__ __launch_bounds__(256, 2) void topk_fused_kernel(FusedArgs p)
{
    typedef SharedT<CAP> Shared;
    __shared__ Shared s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wu = wave & 1, wi = wave >> 1;   // the wave's 32-user x 64-item corner of the tile
    const int n = lane & 31, h = lane >> 5;    // MFMA lane coordinates: column / k parity (operands), column / row half (results)
    const uint32_t u0 = blockIdx.x * (uint32_t)TU;
    const uint32_t ntiles = (p.num_items + TI - 1) / TI;
    const uint32_t t_begin = blockIdx.y * p.tiles_per_split;
    const uint32_t t_end = min(ntiles, t_begin + p.tiles_per_split);
    const uint32_t d = p.d, k = p.k;
    uint32_t ulive = 0u;       // bits 2r, 2r + 1: the user of result register r exists
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (u0 + (uint32_t)(wu * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) < p.rows) ulive |= 3u << (2 * r);

    for (int t = tid; t < TU * CAP; t += 256)
    {
        (&s.topv[0][0])[t] = -INFINITY;
        (&s.topi[0][0])[t] = NONE;
    }
    if (tid < TU)
    {
        s.thr_v[tid] = -INFINITY;
        // a split that starts late inherits what the other splits of its users have already found
        s.thr_sh[tid] = (p.thr_shared && u0 + tid < p.rows) ? __builtin_nontemporal_load(p.thr_shared + u0 + tid) : -INFINITY;
        s.thr_i[tid] = NONE;
    }
    if (tid < 4) s.qn[tid] = 0;

    // wave 0: lane u walks user u's sorted train items; cur/nxt/nx2 = cursor, its item, the one after (prefetched)
    uint64_t cur = 0, hi = 0;
    uint32_t nxt = NONE, nx2 = NONE;
    if (wave == 0 && p.indptr && u0 + lane < p.rows)
    {
        uint64_t lo = p.indptr[u0 + lane];
        hi = p.indptr[u0 + lane + 1];
        const uint32_t first = t_begin * (uint32_t)TI;
        uint64_t a = lo, b = hi; // first entry >= first
        while (a < b)
        {
            const uint64_t m = (a + b) >> 1;
            if (p.items[m] < first) a = m + 1;
            else b = m;
        }
        cur = a;
        nxt = cur < hi ? p.items[cur] : NONE;
        nx2 = cur + 1 < hi ? p.items[cur + 1] : NONE;
    }
    __syncthreads();

    // staging map: thread (sr, sc) carries columns sc..sc+3 and 16+sc..16+sc+3 of user row sr and item rows sr, 64+sr.
    // Two slabs are always in flight (ring[0] = even slabs, ring[1] = odd slabs of a tile): a load is issued two slabs —
    // at emb_dim 64 one whole tile — before it is consumed, so the L2 / Infinity-Cache latency hides behind the MFMAs.
    const int sr = tid >> 2, sc = (tid & 3) * 4;
    SlabRegs ring[2];
    ring[0].a[0] = ring[0].a[1] = ring[0].b0[0] = ring[0].b0[1] = ring[0].b1[0] = ring[0].b1[1] = f4{0, 0, 0, 0};
    ring[1] = ring[0];
    if (t_begin < t_end)
    {
        load_slab(p, u0, t_begin * (uint32_t)TI, 0, sr, sc, ring[0]);
        if ((uint32_t)KS < d) load_slab(p, u0, t_begin * (uint32_t)TI, KS, sr, sc, ring[1]);
    }

    for (uint32_t tile = t_begin; tile < t_end; ++tile)
    {
        const uint32_t i0 = tile * (uint32_t)TI;
        if (wave == 0)
        {
            s.mbits[lane][0] = 0; s.mbits[lane][1] = 0; s.mbits[lane][2] = 0; s.mbits[lane][3] = 0;
            const uint32_t tile_end = i0 + TI; // ids < 2^32 - TI by the host check
            while (nxt < tile_end)
            {
                const uint32_t bit = nxt - i0;
                s.mbits[lane][bit >> 5] |= 1u << (bit & 31);
                ++cur;
                nxt = nx2;
                nx2 = cur + 1 < hi ? p.items[cur + 1] : NONE;
            }
        }

        f16v acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }

        for (uint32_t k0 = 0; k0 < d; k0 += 2 * KS)
        {
#pragma unroll
            for (int slot = 0; slot < 2; ++slot)
            {
                const uint32_t ks = k0 + (uint32_t)(slot * KS);
                if (ks >= d) break;
                __syncthreads(); // previous slab fully consumed
                store_slab(s, sr, sc, ring[slot]);
                __syncthreads();
                // this slot's next slab: two slabs further in this tile, else the same slot of the next tile
                const uint32_t nk = ks + 2u * (uint32_t)KS;
#ifndef TOPK_ABLATE_LOADS
                if (nk < d) load_slab(p, u0, i0, nk, sr, sc, ring[slot]);
                else if (tile + 1 < t_end && (uint32_t)(slot * KS) < d) load_slab(p, u0, i0 + TI, slot * KS, sr, sc, ring[slot]);
#endif
                const int kmax = (d - ks) < (uint32_t)KS ? (int)(d - ks) : KS;   // emb_dim % 4 == 0: always even
#ifndef TOPK_ABLATE_MFMA
                if (kmax == KS) tile_slab_mfma<true>(s, KS / 2, wu, wi, n, h, acc0, acc1);
                else tile_slab_mfma<false>(s, kmax / 2, wu, wi, n, h, acc0, acc1);
#else
                acc0[0] += s.a[0][wu * 32 + n][h] * (float)kmax; acc1[0] += s.b[0][wi * 64 + n][h];
#endif
            }
        }
        __syncthreads(); // mbits of this tile visible; LDS slabs free

        // lane's results: acc_c[r] = user 32 wu + 8 (r / 4) + 4 h + (r % 4), item i0 + 64 wi + 32 c + n.
        // (1) branch-free filter of the 32 raw scores against their users' thresholds: bit 2r + c of `cand`.  After the
        //     first tiles a candidate is rare, so this — 4 LDS reads, 32 compares — is all the selection costs per tile.
        const uint32_t it0 = i0 + (uint32_t)(wi * 64 + n), it1 = it0 + 32u;
        uint32_t cand = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const f4 tl = *(const f4*)&s.thr_v[wu * 32 + 8 * q + 4 * h];
            const f4 ts = *(const f4*)&s.thr_sh[wu * 32 + 8 * q + 4 * h];
            const f4 t4 = f4{fmaxf(tl[0], ts[0]), fmaxf(tl[1], ts[1]), fmaxf(tl[2], ts[2]), fmaxf(tl[3], ts[3])};
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int r = 4 * q + j;
                cand |= (!(acc0[r] < t4[j]) ? 1u : 0u) << (2 * r);
                cand |= (!(acc1[r] < t4[j]) ? 1u : 0u) << (2 * r + 1);
            }
        }
        cand &= ulive & ((it0 < p.num_items ? 0x55555555u : 0u) | (it1 < p.num_items ? 0xAAAAAAAAu : 0u));
#ifdef TOPK_ABLATE_SELECT
        if (acc0[0] != 12345.678f) cand = 0u;     // timing experiment: scores are produced, nothing is ranked
#endif
        // (2) the rare part: a train item scores -inf (metrics.py:24) and is queued only while its user's list is not full
        if (__ballot(cand != 0u) != 0ull)
        {
            uint32_t todo = cand;
            while (todo != 0u)
            {
                const int b = __builtin_ctz(todo);
                todo &= todo - 1u;
                const int r = b >> 1, c = b & 1;
                const uint32_t u = (uint32_t)(wu * 32 + 8 * (r >> 2) + 4 * h + (r & 3));
                float v = 0.0f;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2)
                    if (r2 == r) v = c ? acc1[r2] : acc0[r2];
                if ((s.mbits[u][wi * 2 + c] >> n) & 1u) v = -INFINITY;
                if (!(v < s.thr_v[u]) && !(v < s.thr_sh[u])) push(s, u, c ? it1 : it0, v);
            }
        }
        __syncthreads();
        const uint32_t nq = max(max(s.qn[0], s.qn[1]), max(s.qn[2], s.qn[3]));
        if (nq <= (uint32_t)QW)
        {
            if (nq) drain(s, k, wave, lane);
        }
        else
        {
            // more candidates than a queue holds (first tiles, or scores arriving in ascending order): one output
            // per thread and round, i.e. at most 64 per wave, against the thresholds the earlier rounds raised
            for (int rnd = 0; rnd < 32; ++rnd)
            {
                __syncthreads();
                if (tid < 4) s.qn[tid] = 0;
                __syncthreads();
                const int r = rnd >> 1, c = rnd & 1;
                const uint32_t u = (uint32_t)(wu * 32 + 8 * (r >> 2) + 4 * h + (r & 3));
                const uint32_t item = i0 + (uint32_t)(wi * 64 + 32 * c + n);
                float v = 0.0f;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2)
                    if (r2 == r) v = c ? acc1[r2] : acc0[r2];
                if ((s.mbits[u][wi * 2 + c] >> n) & 1u) v = -INFINITY;
                if (u0 + u < p.rows && item < p.num_items && !(v < s.thr_v[u]) && !(v < s.thr_sh[u])) push(s, u, item, v);
                __syncthreads();
                drain(s, k, wave, lane);
            }
        }
        __syncthreads();
        if (tid < 4) s.qn[tid] = 0;
        // Threshold exchange between the item splits of these users.  The k-th best score of ONE split is a valid bound for
        // all of them (that split alone already holds k better items), so the splits prune like a single pass: ~k ln(n/k)
        // candidates per user in total instead of per split, and a split that starts late skips its warm-up flood.
        if (p.thr_shared && tid < TU && u0 + tid < p.rows)
        {
            const float mine = s.thr_v[tid];
            float* g = p.thr_shared + u0 + tid;
            if (mine > s.thr_sh[tid])
            {
                if (mine >= 0.0f) atomicMax(reinterpret_cast<int*>(g), __float_as_int(mine));
                else atomicMin(reinterpret_cast<unsigned int*>(g), __float_as_uint(mine));
            }
            s.thr_sh[tid] = fmaxf(mine, __builtin_nontemporal_load(g));
        }
    }
    __syncthreads();
    for (int t = tid; t < TU * (int)k; t += 256)
    {
        const uint32_t u = (uint32_t)t / k, j = (uint32_t)t % k;
        if (u0 + u >= p.rows) continue;
        const size_t o = ((size_t)blockIdx.y * p.rows + (u0 + u)) * k + j;
        p.part_v[o] = s.topv[u][j];
        p.part_i[o] = s.topi[u][j];
    }
}

// ---- round 3: 128 users x 128 items per workgroup, every user owned by ONE wave (k <= 32, emb_dim <= 64) ----------------
// What the 64 x 128 kernel above spends beyond its arithmetic (profiles/r02_topk_sq_counters.txt: matrix pipe busy 3.9 of
// 11.9 ms at AmazonBooks shape) is per-tile fixed cost: seven workgroup barriers, both operands restaged through LDS, and a
// candidate path in which four waves push into each other's queues, meet at a barrier and insert one candidate per LDS
// round trip.  This form removes the sharing instead of tuning it:
//   * wave w owns users [32 w, 32 w + 32) of the tile and ALL 128 items: four 32 x 32 accumulators;
//   * the A operand (the wave's 32 user rows) never changes over the item loop: it lives in REGISTERS for the whole kernel
//     (lane (n, h): U[32 w + n][2 kp + h], kp = 0 .. d/2: 32 registers at emb_dim 64) — no staging, no LDS reads, no barrier;
//   * only the item tile goes through LDS (128 items x 64 k-values = 33.8 KB): two barriers per tile instead of seven, the
//     next tile's global loads in flight under the MFMAs;
//   * selection is wave-local and has no queue: the scores that pass their user's threshold are one bit each in four words
//     per lane (a compare and an add-with-carry per score); per step the first lane of EACH HALF of the wave that holds a
//     candidate hands it over (the two halves never hold the same user) and both are inserted at once into their users'
//     sorted lists — lane n < k owns slot n, position = popcount of its half of one ballot, one LDS round trip for both.
//     Leader election, the register index of the score (s_set_gpr_idx), user and column of both candidates are computed
//     on the SCALAR unit: measured per phase with s_memtime (profiles/r03_topk_timeline.txt), a step cost ~2000 cycles when
//     its ~60 vector instructions (a 16-way select among them) waited behind the co-resident workgroup's 64-cycle MFMAs;
//   * two workgroups per compute unit (69 KB of LDS, 256 registers each): one's selection runs under the other's MFMAs.
// AmazonBooks shape top-20: 9.0-9.2 ms against 11.9 in round 2 (profiles/r03_topk_kernel_stats.csv; 10.1 with the per-wave
// candidate queue this selection replaced).  Per wave and tile (25 k cycles): MFMA phase 9.4 k, insert steps 7.2 x 1.25 k,
// waiting at the tile barrier for the slowest wave's steps 2.5 k, filter 1.2 k, tile store / mask bits / loads 1.3 k.
// Measured on the way and removed: one workgroup per CU with the tile double-buffered, 13.7 ms;
// per-user append buffers merged by rank counting, 14.5; threshold exchange consumed a tile late, 15.6; filter + queueing as
// one static sweep over the 64 (column, row) pairs, 25; one step loop over all four column blocks (dynamic block index: the
// accumulators went to scratch), 84; s_setprio around the selection, no change.  At k > 32 or emb_dim > 64 the 64-user kernel
// stays faster and keeps those shapes.
// Scores, order and ids are those of the kernel above (same fmaf chain per score, same (score desc, id asc) comparator).
constexpr int TU2 = 128;          // users per workgroup
constexpr int KP2 = 32;           // k pairs the kernel holds: emb_dim <= 64
constexpr int CAP2 = 32;          // list slots per user: k <= 32
constexpr int LDB2 = TI + 2;      // 130: the store's four k-pair groups of a half-wave land 8 banks apart (2 * 2 * 130 mod 32)

struct TopEntry { float v; uint32_t i; };   // one list slot: read and written as 8 bytes

// Small arrays first: everything the candidate path touches sits below 64 KB, where the LDS instructions' own 16-bit offset
// field reaches it from one per-user base address.
struct __attribute__((aligned(16))) Shared2
{
    float    thr_v[TU2];                  // = entry k - 1, the one a candidate has to beat
    float    thr_sh[TU2];                 // best k-th score any item split of these users has published
    uint32_t thr_i[TU2];
    uint32_t mbits[TU2][4];               // train items of the current tile
    TopEntry top[TU2][CAP2];              // per user: the k best so far, best first
    float    b[KP2][2][LDB2];             // the item tile: [k pair][k parity][item] — the 32 lanes of an MFMA half read 32
                                          // consecutive floats (32 LDS banks: [item][parity] made every read a 2-way conflict,
                                          // 39 % of the LDS cycles by SQ_LDS_BANK_CONFLICT)
};

// Two candidates per step, one in each half of the wave.  Result register r of lane (n, h) belongs to user
// 8 (r / 4) + 4 h + (r % 4): the two halves NEVER hold the same user, so half 0 can insert its candidate into its user's
// sorted list (lane n < k owns slot n) while half 1 does the same for its own — one LDS round trip for both, each half's
// position from its 32 bits of one ballot.
__device__ __forceinline__ void insert_half(Shared2& s, uint32_t u, int c, int ns, uint32_t item, float v, uint32_t k, uint64_t kmask, int lane)
{
    // branch-free on purpose: every lane reads valid slots (CAP2 = 32 of them per user), the conditions are combined as lane
    // masks on the scalar unit; a half without a candidate is handed v = NaN, which fails every comparison below
    const int l = lane & 31, lm = l > 0 ? l - 1 : 0;
    const uint32_t mb = s.mbits[u][c];
    const float    tv = s.thr_v[u];
    const uint32_t ti = s.thr_i[u];
    const float    ts = s.thr_sh[u];
    TopEntry e = s.top[u][l], pe = s.top[u][lm];
    __asm__ volatile("" : "+v"(pe.v), "+v"(pe.i));                                  // fetched with the rest, not inside the store path
    const bool masked = ((mb >> ns) & 1u) != 0u;                                   // a train item scores -inf (metrics.py:24)
    v = masked ? (v == v ? -INFINITY : v) : v;
    // one compare per ballot, combined as lane masks on the scalar unit
    const uint64_t ahead_thr = __builtin_amdgcn_ballot_w64(v > tv) | (__builtin_amdgcn_ballot_w64(v == tv) & __builtin_amdgcn_ballot_w64(item < ti));
    const uint64_t go = kmask & ahead_thr & __builtin_amdgcn_ballot_w64(!(v < ts)) & __builtin_amdgcn_ballot_w64(v == v);
    const uint64_t before = go & (__builtin_amdgcn_ballot_w64(e.v > v) | (__builtin_amdgcn_ballot_w64(e.v == v) & __builtin_amdgcn_ballot_w64(e.i < item)));
    uint32_t p0 = (uint32_t)__builtin_popcount((uint32_t)before), p1 = (uint32_t)__builtin_popcount((uint32_t)(before >> 32));
    __asm__("" : "+s"(p0), "+s"(p1));                                              // two scalar popcounts, not a vector shift
    const uint32_t pos = p0 ^ ((p0 ^ p1) & (lane < 32 ? 0u : 0xFFFFFFFFu));
    if (__builtin_amdgcn_inverse_ballot_w64(go & __builtin_amdgcn_ballot_w64((uint32_t)l >= pos)))
    {
        const TopEntry ne = (uint32_t)l == pos ? TopEntry{v, item} : pe;
        s.top[u][l] = ne;
        if ((uint32_t)l == k - 1)
        {
            s.thr_v[u] = ne.v;
            s.thr_i[u] = ne.i;
        }
    }
    // LDS operations of one wave complete in issue order: the next step's reads see these writes
    __builtin_amdgcn_wave_barrier();
    __asm__ volatile("" ::: "memory");
}

struct Tile2Regs { f4 v[8]; };   // thread (sr, sc): item rows sr and 64 + sr, columns sc*4 + 16 j (j < 4)

__device__ __forceinline__ void load_tile2(const FusedArgs& p, uint32_t i0, int sr, int sc, Tile2Regs& g)
{
    const uint32_t d = p.d, last = p.num_items - 1u;
    const float* v0 = p.V + (size_t)min(i0 + (uint32_t)sr, last) * d;
    const float* v1 = p.V + (size_t)min(i0 + 64u + (uint32_t)sr, last) * d;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        const uint32_t kk = min((uint32_t)(sc * 4 + 16 * j), d - 4u);   // columns past emb_dim: clamped, never multiplied
        g.v[j] = *(const f4*)(v0 + kk);
        g.v[4 + j] = *(const f4*)(v1 + kk);
    }
}

__device__ __forceinline__ void store_tile2(Shared2& s, int sr, int sc, const Tile2Regs& g)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        const int kp = (sc * 4 + 16 * j) / 2;
        s.b[kp][0][sr] = g.v[j][0];          s.b[kp][1][sr] = g.v[j][1];
        s.b[kp + 1][0][sr] = g.v[j][2];      s.b[kp + 1][1][sr] = g.v[j][3];
        s.b[kp][0][64 + sr] = g.v[4 + j][0]; s.b[kp][1][64 + sr] = g.v[4 + j][1];
        s.b[kp + 1][0][64 + sr] = g.v[4 + j][2]; s.b[kp + 1][1][64 + sr] = g.v[4 + j][3];
    }
}

__global__ __launch_bounds__(256, 2) void topk_fused2_kernel(FusedArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    Shared2& s = *reinterpret_cast<Shared2*>(smem2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const uint32_t hshift = h ? 16u : 0u, hmask = h ? 0xFFFFFFFFu : 0u;
    const uint32_t u0 = blockIdx.x * (uint32_t)TU2;
    const uint32_t ub = u0 + (uint32_t)(wave * 32);            // first user of this wave
    const uint32_t ntiles = (p.num_items + TI - 1) / TI;
    const uint32_t t_begin = blockIdx.y * p.tiles_per_split;
    const uint32_t t_end = min(ntiles, t_begin + p.tiles_per_split);
    const uint32_t d = p.d, k = p.k;
    const uint32_t kpairs = d / 2;                              // emb_dim % 4 == 0

    // A operand: lane (n, h) keeps U[ub + n][2 kp + h] for every k pair
    float a[KP2];
    {
        const float* ur = p.U + (size_t)min(ub + (uint32_t)n, p.rows - 1u) * d;
#pragma unroll
        for (int kp = 0; kp < KP2; ++kp) a[kp] = (uint32_t)kp < kpairs ? ur[2 * kp + h] : 0.0f;
    }
    uint32_t ulive = 0u;       // bit r: the user of result register r exists
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (ub + (uint32_t)(8 * (r >> 2) + 4 * h + (r & 3)) < p.rows) ulive |= 1u << r;

    for (int t = tid; t < TU2 * CAP2; t += 256) (&s.top[0][0])[t] = TopEntry{-INFINITY, NONE};
    const uint64_t kslots = (1ull << k) - 1ull, kmask = kslots | (kslots << 32);   // lanes whose slot (lane % 32) is below k
    if (tid < TU2)
    {
        s.thr_v[tid] = -INFINITY;
        s.thr_sh[tid] = (p.thr_shared && u0 + tid < p.rows) ? __builtin_nontemporal_load(p.thr_shared + u0 + tid) : -INFINITY;
        s.thr_i[tid] = NONE;
    }

    // lane ul < 32 of every wave walks the sorted train items of its user: cur / nxt / nx2 = cursor, its item, the one after
    uint64_t cur = 0, hi = 0;
    uint32_t nxt = NONE, nx2 = NONE;
    if (lane < 32 && p.indptr && ub + lane < p.rows)
    {
        uint64_t lo = p.indptr[ub + lane];
        hi = p.indptr[ub + lane + 1];
        const uint32_t first = t_begin * (uint32_t)TI;
        uint64_t x = lo, y = hi;
        while (x < y)
        {
            const uint64_t m = (x + y) >> 1;
            if (p.items[m] < first) x = m + 1;
            else y = m;
        }
        cur = x;
        nxt = cur < hi ? p.items[cur] : NONE;
        nx2 = cur + 1 < hi ? p.items[cur + 1] : NONE;
    }

    const int sr = tid >> 2, sc = tid & 3;
    Tile2Regs regs;
#pragma unroll
    for (int j = 0; j < 8; ++j) regs.v[j] = f4{0, 0, 0, 0};
    if (t_begin < t_end) load_tile2(p, t_begin * (uint32_t)TI, sr, sc, regs);
    __syncthreads();

    for (uint32_t tile = t_begin; tile < t_end; ++tile)
    {
        const uint32_t i0 = tile * (uint32_t)TI;
        if (lane < 32)
        {
            const uint32_t u = (uint32_t)(wave * 32 + lane);
            *reinterpret_cast<u4*>(&s.mbits[u][0]) = u4{0u, 0u, 0u, 0u};
            const uint32_t tile_end = i0 + TI;
            while (nxt < tile_end)
            {
                const uint32_t bit = nxt - i0;
                s.mbits[u][bit >> 5] |= 1u << (bit & 31);
                ++cur;
                nxt = nx2;
                nx2 = cur + 1 < hi ? p.items[cur + 1] : NONE;
            }
        }

        store_tile2(s, sr, sc, regs);                    // the tile loaded one iteration ago
        __syncthreads();
        if (tile + 1 < t_end) load_tile2(p, i0 + TI, sr, sc, regs);

        f16v acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
        auto tile_mfma = [&](auto full_tag) __attribute__((always_inline))
        {
#pragma unroll
            for (int kp = 0; kp < KP2; ++kp)
            {
                if (decltype(full_tag)::value || (uint32_t)kp < kpairs)
                {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], s.b[kp][h][32 * c + n], acc[c], 0, 0, 0);
                }
            }
        };
        if (kpairs == (uint32_t)KP2) tile_mfma(std::true_type{});
        else tile_mfma(std::false_type{});
        __syncthreads();                                 // every wave is done with the tile in LDS

        // ---- selection, wave-local: lane's results acc[c][r] = user ub + 8 (r / 4) + 4 h + (r % 4), item i0 + 32 c + n
        // bit r of cand[c] = !(acc[c][r] < threshold of r's user), built from r = 15 down as cand = 2 cand + carry: one compare
        // and one add-with-carry per score (vector instructions are what this phase pays for, see the insert loop below)
        uint32_t cand[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 3; q >= 0; --q)
        {
            const f4 tl = *(const f4*)&s.thr_v[wave * 32 + 8 * q + 4 * h];
            const f4 ts = *(const f4*)&s.thr_sh[wave * 32 + 8 * q + 4 * h];
            const f4 t4 = f4{fmaxf(tl[0], ts[0]), fmaxf(tl[1], ts[1]), fmaxf(tl[2], ts[2]), fmaxf(tl[3], ts[3])};
#pragma unroll
            for (int j = 3; j >= 0; --j)
            {
                const int r = 4 * q + j;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    __asm__("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(cand[c]) : "v"(acc[c][r]), "v"(t4[j]) : "vcc");
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            uint32_t todo = i0 + (uint32_t)(32 * c + n) < p.num_items ? (cand[c] & ulive) : 0u;
            uint64_t bal = __builtin_amdgcn_ballot_w64(todo != 0u);
            while (bal != 0ull)
            {
                // Per step one candidate from each half of the wave is inserted (insert_half).  The first lane of a half that
                // has a candidate is its leader; everything that can be is computed on the scalar unit from the leaders' masks.
                uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
                __asm__("" : "+s"(lo), "+s"(hi));                                  // two 32-bit scalars (not one 64-bit compare)
                const int l0 = lo != 0u ? __builtin_ctz(lo) : 0, l1 = hi != 0u ? 32 + __builtin_ctz(hi) : 32;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)todo, l0);
                const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)todo, l1);
                const int r0 = t0 != 0u ? __builtin_ctz(t0) : 0, r1 = t1 != 0u ? __builtin_ctz(t1) : 0;
                uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(acc[c][r0]), l0);   // uniform register index
                uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(acc[c][r1]), l1);
                b0 = lo != 0u ? b0 : 0x7FC00000u;                                  // no candidate in this half: NaN
                b1 = hi != 0u ? b1 : 0x7FC00000u;
                // (user, item column) of both leaders packed into one scalar, each half shifts its own 16 bits out
                const uint32_t u0 = (uint32_t)(wave * 32 + 8 * (r0 >> 2) + (r0 & 3));
                const uint32_t u1 = (uint32_t)(wave * 32 + 8 * (r1 >> 2) + 4 + (r1 & 3));
                const uint32_t pack = u0 | ((uint32_t)l0 << 8) | (u1 << 16) | ((uint32_t)(l1 - 32) << 24);
                const uint32_t mine = pack >> hshift;
                const uint32_t u = mine & 0xFFu;
                const int ns = (int)((mine >> 8) & 31u);
                const float v = __uint_as_float(b0 ^ ((b0 ^ b1) & hmask));
                insert_half(s, u, c, ns, i0 + (uint32_t)(32 * c) + (uint32_t)ns, v, k, kmask, lane);
                if ((lane == l0 && lo != 0u) || (lane == l1 && hi != 0u)) todo &= todo - 1u;
                bal = __builtin_amdgcn_ballot_w64(todo != 0u);
            }
        }

        // threshold exchange between the item splits of these users (see the 64 x 128 kernel)
        if (p.thr_shared && lane < 32 && ub + lane < p.rows)
        {
            const uint32_t u = (uint32_t)(wave * 32 + lane);
            const float mine = s.thr_v[u];
            float* gthr = p.thr_shared + ub + lane;
            if (mine > s.thr_sh[u])
            {
                if (mine >= 0.0f) atomicMax(reinterpret_cast<int*>(gthr), __float_as_int(mine));
                else atomicMin(reinterpret_cast<unsigned int*>(gthr), __float_as_uint(mine));
            }
            s.thr_sh[u] = fmaxf(mine, __builtin_nontemporal_load(gthr));
        }
        __builtin_amdgcn_wave_barrier();
        __asm__ volatile("" ::: "memory");
    }
    __syncthreads();
    for (int t = tid; t < TU2 * (int)k; t += 256)
    {
        const uint32_t u = (uint32_t)t / k, j = (uint32_t)t % k;
        if (u0 + u >= p.rows) continue;
        const size_t o = ((size_t)blockIdx.y * p.rows + (u0 + u)) * k + j;
        p.part_v[o] = s.top[u][j].v;
        p.part_i[o] = s.top[u][j].i;
    }
}

// One wave per user: rank the splits*k partial entries, emit the first k ids.
__global__ __launch_bounds__(64) void topk_merge_kernel(const float* part_v, const uint32_t* part_i, uint32_t rows,
                                                        uint32_t k, uint32_t splits, uint32_t* topk)
{
    __shared__ float    cv[TOPK_FUSED_MAX_SPLITS * TOPK_FUSED_MAX_K];
    __shared__ uint32_t ci[TOPK_FUSED_MAX_SPLITS * TOPK_FUSED_MAX_K];
    const uint32_t u = blockIdx.x;
    if (u >= rows) return;
    const uint32_t n = splits * k;
    for (uint32_t c = threadIdx.x; c < n; c += 64)
    {
        const uint32_t z = c / k, j = c % k;
        const size_t o = ((size_t)z * rows + u) * k + j;
        cv[c] = part_v[o];
        ci[c] = part_i[o];
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < n; c += 64)
    {
        const float v = cv[c];
        const uint32_t i = ci[c];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < n; ++o)
        {
            const float w = cv[o];
            const uint32_t j = ci[o];
            rank += (ahead(w, j, v, i) || (w == v && j == i && o < c)) ? 1u : 0u;
        }
        if (rank < k) topk[(size_t)u * k + rank] = i;
    }
}
} // namespace

// which kernel runs: the 128 x 128 one (round 3) unless HEAT_CF_TOPK_KERNEL=v1 asks for the 64 x 128 one (A/B runs)
// Which kernel runs.  The 128-user kernel (two workgroups per compute unit, item slab single-buffered, 32-slot lists, the
// whole user panel of a wave in 32 registers) is compiled for k <= 32 and emb_dim <= 64, where it measured faster
// (AmazonBooks shape top-20: 10.4 ms against 11.9); everything else runs the 64-user kernel, which stays faster there
// (its one- and two-workgroup 128-user twins were measured at emb_dim 128 / 256 and k = 50 and lost by 4-25 %).
// HEAT_CF_TOPK_KERNEL=v1 forces the 64-user kernel (A/B runs).
static bool use_v2(uint32_t emb_dim, uint32_t k)
{
    const char* e = std::getenv("HEAT_CF_TOPK_KERNEL");
    if (e && std::strcmp(e, "v1") == 0) return false;
    return k <= 32u && emb_dim <= 64u;
}

// Item-range splits per user block for a chip with `cus` compute units.  Workgroups are dispatched as slots free up, so a
// launch costs about ceil(workgroups / slots) rounds of one split's work (a fractional last round is a tail in which most
// of the chip idles), and one split's work is its tiles PLUS its warm-up: until a split's lists are full every score is a
// candidate, and of the first 128 items about k + k ln(128 / k) per user are inserted — measured, about 40 tiles' worth of
// time per split at k = 20.  So fewer, longer splits are preferred unless they leave slots empty: AmazonBooks shape on the
// 128-user kernel (412 user blocks, 512 slots) runs ONE split per block, the 64-user kernel (823 blocks) three.
uint32_t topk_fused_splits(uint32_t rows, uint32_t num_items, uint32_t cus, uint32_t emb_dim, uint32_t k)
{
    const uint32_t tu = use_v2(emb_dim, k) ? (uint32_t)TU2 : (uint32_t)TU, slots = 2u * cus;   // both kernels: two workgroups per CU
    const uint32_t nblocks = (rows + tu - 1) / tu, ntiles = (num_items + TI - 1) / TI;
    if (nblocks == 0 || ntiles == 0 || slots == 0) return 1;
    uint32_t best = 1;
    double best_cost = 0.0;
    for (uint32_t z = 1; z <= (uint32_t)TOPK_FUSED_MAX_SPLITS && z <= ntiles; ++z)
    {
        const uint32_t per = (ntiles + z - 1) / z, ze = (ntiles + per - 1) / per;     // no empty split
        if (ze != z) continue;
        const uint64_t wgs = (uint64_t)nblocks * ze, rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * ((double)per + 40.0) * (1.0 + 0.002 * (double)ze);
        if (best_cost == 0.0 || cost < best_cost) { best_cost = cost; best = ze; }
    }
    return best;
}

static hipError_t launch_v2(const FusedArgs& p, uint32_t splits, hipStream_t s)
{
    const size_t lds = sizeof(Shared2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_fused2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(topk_fused2_kernel, dim3((p.rows + TU2 - 1) / TU2, splits), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_topk_fused(const float* user_rows, const float* item_w, uint32_t rows, uint32_t num_items,
                             uint32_t emb_dim, uint32_t k, const uint64_t* indptr, const uint32_t* items, uint32_t splits,
                             float* part_v, uint32_t* part_i, uint32_t* topk, float* thr_shared, hipStream_t s)
{
    if (rows == 0 || num_items == 0) return hipSuccess;
    if (k == 0 || k > (uint32_t)TOPK_FUSED_MAX_K || splits == 0 || splits > (uint32_t)TOPK_FUSED_MAX_SPLITS || (emb_dim & 3u))
        return hipErrorInvalidValue;
    const uint32_t ntiles = (num_items + TI - 1) / TI;
    FusedArgs p;
    p.U = user_rows; p.V = item_w; p.rows = rows; p.num_items = num_items; p.d = emb_dim; p.k = k;
    p.tiles_per_split = (ntiles + splits - 1) / splits;
    if ((uint64_t)p.tiles_per_split * (splits - 1) >= ntiles && splits > 1) return hipErrorInvalidValue; // empty split
    p.indptr = indptr; p.items = items; p.part_v = part_v; p.part_i = part_i;
    p.thr_shared = splits > 1 ? thr_shared : nullptr;
    if (p.thr_shared)
    {
        hipError_t e0 = hipMemsetD32Async((hipDeviceptr_t)thr_shared, (int)0xFF800000u, rows, s);   // -inf
        if (e0 != hipSuccess) return e0;
    }
    hipError_t err;
    if (use_v2(emb_dim, k))
    {
        err = launch_v2(p, splits, s);
    }
    else
    {
        if (k <= 32) hipLaunchKernelGGL(topk_fused_kernel<32>, dim3((rows + TU - 1) / TU, splits), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(topk_fused_kernel<64>, dim3((rows + TU - 1) / TU, splits), dim3(256), 0, s, p);
        err = hipGetLastError();
    }
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(rows), dim3(64), 0, s, part_v, part_i, rows, k, splits, topk);
    return hipGetLastError();
}

} // namespace heatcf
