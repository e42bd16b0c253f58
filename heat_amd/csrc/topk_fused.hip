// topk_fused.hip — evaluation without the [num_users, num_items] matrix (gfx950).
//
// Replaces the chain  Engine::evaluate0 (train/engine.cpp:388-400)  ->  PyMatrix copy (pybind/init_modules.cpp:122-129)
// -> sim[train] = -inf, argpartition, argsort (cf/metrics.py:21-29)  by one pass that keeps, per user, the running
// k best (score, item) pairs in LDS while 64 x 128 score tiles are produced in registers and never stored.
//
// Arithmetic: every score is the fp32 dot of the oracle (oracle/cf_oracle.c dotf): multiply and add UNFUSED, k left to
// right.  The tile loop therefore uses packed fp32 multiply + packed fp32 add (v_pk_mul_f32 / v_pk_add_f32, exact per
// element) and not MFMA: on gfx950 the fp32 MFMA peak equals the packed-FMA VALU peak (157 TFLOP/s), so exact unfused
// arithmetic costs a factor 2 against that bound and keeps the ranking bit-identical to the dense path.
//
// Order: pairs are ranked by (score descending, item id ascending); NaN scores are never selected; masked items score
// -inf and so can still fill the list when fewer than k unmasked items exist — exactly what topk_rows_kernel yields on
// the materialised panel (eval_kernels.hip), which stays as the k > 64 path and as the cross-check in tests.
#include "eval_kernels.hpp"

#include <math.h>

namespace heatcf
{
namespace
{
constexpr int TU = 64;     // users per workgroup
constexpr int TI = 128;    // items per tile
constexpr int KS = 16;     // k-slab staged through LDS
constexpr int LDA = TU + 4;
constexpr int LDB = TI + 4;
constexpr uint32_t NONE = 0xFFFFFFFFu;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int QW = 128;    // candidate queue entries per wave and tile

// CAP = list slots per user (32 or 64): the short form leaves room for a fourth workgroup per CU
template <int CAP> struct __attribute__((aligned(16))) SharedT
{
    float    a[KS][LDA];
    float    b[KS][LDB];
    float    topv[TU][CAP]; // per user: the k best so far, best first
    uint32_t topi[TU][CAP];
    float    thr_v[TU];                  // = entry k-1, the one a candidate has to beat
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

// operands of one k step: 4 user values (broadcast over the 16 lanes of a row) and 8 item values
struct StepOps { f4 av, b0, b1; };

template <class Shared> __device__ __forceinline__ StepOps tile_fetch(const Shared& s, int kq, int tx, int ty)
{
    StepOps o;
    o.av = *(const f4*)&s.a[kq][ty * 4];
    o.b0 = *(const f4*)&s.b[kq][tx * 4];
    o.b1 = *(const f4*)&s.b[kq][64 + tx * 4];
    return o;
}

// acc[a][c] += a-row value * b-row pair, product and sum rounded separately
__device__ __forceinline__ void tile_fma(const StepOps& o, f2 (&acc)[4][4])
{
#pragma unroll
    for (int a = 0; a < 4; ++a)
    {
        const f2 aa = f2{o.av[a], o.av[a]};
        acc[a][0] = acc[a][0] + aa * f2{o.b0[0], o.b0[1]};
        acc[a][1] = acc[a][1] + aa * f2{o.b0[2], o.b0[3]};
        acc[a][2] = acc[a][2] + aa * f2{o.b1[0], o.b1[1]};
        acc[a][3] = acc[a][3] + aa * f2{o.b1[2], o.b1[3]};
    }
}

template <class Shared> __device__ __forceinline__ void tile_step(const Shared& s, int kq, int tx, int ty, f2 (&acc)[4][4])
{
    tile_fma(tile_fetch(s, kq, tx, ty), acc);
}

// a full slab with the LDS reads of step k+1 in flight while step k multiplies
template <class Shared> __device__ __forceinline__ void tile_slab(const Shared& s, int tx, int ty, f2 (&acc)[4][4])
{
    StepOps cur = tile_fetch(s, 0, tx, ty);
#pragma unroll
    for (int kq = 0; kq < KS; ++kq)
    {
        StepOps nxt = cur;
        if (kq + 1 < KS) nxt = tile_fetch(s, kq + 1, tx, ty);
        __builtin_amdgcn_sched_barrier(0); // keep the fetch ahead of the arithmetic it overlaps with
        tile_fma(cur, acc);
        cur = nxt;
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
    float*          part_v;   // [splits, rows, k]
    uint32_t*       part_i;
};

// One float4 of user row u0+sr and of item rows i0+sr, i0+64+sr at columns [k0+sc, k0+sc+4).  No branches: rows past
// the end of a table are clamped to its last row and columns past emb_dim to its last float4 — such scores are
// computed but never ranked (the filter checks user < rows, item < num_items; the k loop stops at emb_dim).
// emb_dim % 4 == 0 and 16-byte aligned rows are checked on the host.
__device__ __forceinline__ void load_slab(const FusedArgs& p, uint32_t u0, uint32_t i0, uint32_t k0, int sr, int sc,
                                          f4& ga, f4& gb0, f4& gb1)
{
    const uint32_t d = p.d, kk = min(k0 + (uint32_t)sc, d - 4u);
    const uint32_t last = p.num_items - 1u;
    ga  = *(const f4*)(p.U + ((size_t)min(u0 + (uint32_t)sr, p.rows - 1u) * d + kk));
    gb0 = *(const f4*)(p.V + ((size_t)min(i0 + (uint32_t)sr, last) * d + kk));
    gb1 = *(const f4*)(p.V + ((size_t)min(i0 + 64u + (uint32_t)sr, last) * d + kk));
}

template <int CAP> __global__ __launch_bounds__(256, CAP == 32 ? 4 : 3) void topk_fused_kernel(FusedArgs p)
{
    typedef SharedT<CAP> Shared;
    __shared__ Shared s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tid & 15, ty = tid >> 4;
    const uint32_t u0 = blockIdx.x * (uint32_t)TU;
    const uint32_t ntiles = (p.num_items + TI - 1) / TI;
    const uint32_t t_begin = blockIdx.y * p.tiles_per_split;
    const uint32_t t_end = min(ntiles, t_begin + p.tiles_per_split);
    const uint32_t d = p.d, k = p.k;

    for (int t = tid; t < TU * CAP; t += 256)
    {
        (&s.topv[0][0])[t] = -INFINITY;
        (&s.topi[0][0])[t] = NONE;
    }
    if (tid < TU)
    {
        s.thr_v[tid] = -INFINITY;
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

    // staging map: one float4 of a user row, two of item rows, per thread and slab
    const int sr = tid >> 2, sc = (tid & 3) * 4;
    f4 ga = f4{0, 0, 0, 0}, gb0 = ga, gb1 = ga;

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

        f2 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f2{0.0f, 0.0f};

        // slab s+1 travels from L2 into registers while slab s is multiplied out of LDS
        if (tile == t_begin) load_slab(p, u0, i0, 0, sr, sc, ga, gb0, gb1);
        for (uint32_t k0 = 0; k0 < d; k0 += KS)
        {
            __syncthreads(); // previous slab fully consumed
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                s.a[sc + j][sr] = ga[j];
                s.b[sc + j][sr] = gb0[j];
                s.b[sc + j][64 + sr] = gb1[j];
            }
            __syncthreads();
            if (k0 + KS < d) load_slab(p, u0, i0, k0 + KS, sr, sc, ga, gb0, gb1);
            else if (tile + 1 < t_end) load_slab(p, u0, i0 + TI, 0, sr, sc, ga, gb0, gb1); // lands during selection
            const int kmax = (d - k0) < (uint32_t)KS ? (int)(d - k0) : KS;
            if (kmax == KS)
            {
                tile_slab(s, tx, ty, acc);
            }
            else
            {
                for (int kq = 0; kq < kmax; ++kq) tile_step(s, kq, tx, ty, acc);
            }
        }
        __syncthreads(); // mbits of this tile visible; LDS slabs free

        // thread's outputs: out[a][c] = user ty*4 + a, item i0 + (c>>2)*64 + tx*4 + (c&3); train items score -inf
        float out[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
        {
            const uint32_t u = (uint32_t)(ty * 4 + a);
#pragma unroll
            for (int c = 0; c < 8; ++c) out[a][c] = acc[a][c >> 1][c & 1];
            const uint32_t m8 = ((s.mbits[u][tx >> 3] >> ((tx & 7) * 4)) & 0xFu) |
                                (((s.mbits[u][2 + (tx >> 3)] >> ((tx & 7) * 4)) & 0xFu) << 4);
            if (m8)
            {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if ((m8 >> c) & 1u) out[a][c] = -INFINITY;
            }
        }
        // a candidate is rare after the first tiles: one max and one compare per user row decide for all 8 outputs
#pragma unroll
        for (int a = 0; a < 4; ++a)
        {
            const uint32_t u = (uint32_t)(ty * 4 + a);
            const float tv = s.thr_v[u];
            const float vmax = fmaxf(fmaxf(fmaxf(out[a][0], out[a][1]), fmaxf(out[a][2], out[a][3])),
                                     fmaxf(fmaxf(out[a][4], out[a][5]), fmaxf(out[a][6], out[a][7])));
#ifdef TOPK_EXPERIMENT_NO_SELECT // timing experiment only: scores are produced, nothing is ranked
            if (vmax == 12345.678f)
#else
            if (!(vmax < tv) && u0 + u < p.rows)
#endif
            {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                {
                    const uint32_t item = i0 + (uint32_t)((c >> 2) * 64 + tx * 4 + (c & 3));
                    if (item < p.num_items && !(out[a][c] < tv)) push(s, u, item, out[a][c]);
                }
            }
        }
        __syncthreads();
        const uint32_t n = max(max(s.qn[0], s.qn[1]), max(s.qn[2], s.qn[3]));
        if (n <= (uint32_t)QW)
        {
            if (n) drain(s, k, wave, lane);
        }
        else
        {
            // more candidates than a queue holds (first tiles, or scores arriving in ascending order): one output
            // per thread and round, i.e. at most 64 per wave, against the thresholds the earlier rounds raised
            for (int r = 0; r < 32; ++r)
            {
                __syncthreads();
                if (tid < 4) s.qn[tid] = 0;
                __syncthreads();
                const int a = r >> 3, c = r & 7;
                const uint32_t u = (uint32_t)(ty * 4 + a);
                const uint32_t item = i0 + (uint32_t)((c >> 2) * 64 + tx * 4 + (c & 3));
                float v = 0.0f;
#pragma unroll
                for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2)
                        if (a2 == a && c2 == c) v = out[a2][c2];
                if (u0 + u < p.rows && item < p.num_items && !(v < s.thr_v[u])) push(s, u, item, v);
                __syncthreads();
                drain(s, k, wave, lane);
            }
        }
        __syncthreads();
        if (tid < 4) s.qn[tid] = 0;
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

uint32_t topk_fused_splits(uint32_t rows, uint32_t num_items, uint32_t slots)
{
    const uint32_t nblocks = (rows + TU - 1) / TU, ntiles = (num_items + TI - 1) / TI;
    if (nblocks == 0 || ntiles == 0) return 1;
    uint32_t z = (3 * slots + nblocks - 1) / nblocks; // at least three rounds of workgroups over the chip
    z = z < 1 ? 1 : z;
    z = z > (uint32_t)TOPK_FUSED_MAX_SPLITS ? (uint32_t)TOPK_FUSED_MAX_SPLITS : z;
    z = z > ntiles ? ntiles : z;
    const uint32_t per = (ntiles + z - 1) / z;
    return (ntiles + per - 1) / per;
}

hipError_t launch_topk_fused(const float* user_rows, const float* item_w, uint32_t rows, uint32_t num_items,
                             uint32_t emb_dim, uint32_t k, const uint64_t* indptr, const uint32_t* items, uint32_t splits,
                             float* part_v, uint32_t* part_i, uint32_t* topk, hipStream_t s)
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
    if (k <= 32) hipLaunchKernelGGL(topk_fused_kernel<32>, dim3((rows + TU - 1) / TU, splits), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(topk_fused_kernel<64>, dim3((rows + TU - 1) / TU, splits), dim3(256), 0, s, p);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(rows), dim3(64), 0, s, part_v, part_i, rows, k, splits, topk);
    return hipGetLastError();
}

} // namespace heatcf
