// eval_kernels.hip — the materialised evaluation path (gfx950): dense U*V^T panels for evaluate0 (the reference surface
// returns the full matrix), train-item masking and k rounds of arg-max for k > 64.  The path the product uses for top-k is
// the fused matrix-core kernel of topk_fused.hip; this one doubles as its cross-check in the tests.
#include "eval_kernels.hpp"

#include <math.h>

namespace heatcf
{

// 64x64 output tile per 256-thread block, 4x4 outputs per thread, K staged through LDS in slabs of 32.
// One fused multiply-add per k, k left to right: bit-identical to the oracle's evaluate0 dot (oracle/cf_oracle.c dot_fma)
// and to the matrix-core chain of the fused top-k (topk_fused.hip); the reference's Eigen GEMM order is unspecified.
__global__ __launch_bounds__(256) void sim_panel_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                        float* __restrict__ S, uint32_t rows, uint32_t num_items,
                                                        uint32_t d)
{
    __shared__ float As[64][33];
    __shared__ float Bs[64][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint32_t u0 = blockIdx.y * 64u, i0 = blockIdx.x * 64u;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    for (uint32_t k0 = 0; k0 < d; k0 += 32)
    {
        for (int t = threadIdx.x; t < 64 * 32; t += 256)
        {
            const int r = t >> 5, c = t & 31;
            const uint32_t k = k0 + (uint32_t)c;
            As[r][c] = (u0 + r < rows && k < d) ? U[(size_t)(u0 + r) * d + k] : 0.0f;
            Bs[r][c] = (i0 + r < num_items && k < d) ? V[(size_t)(i0 + r) * d + k] : 0.0f;
        }
        __syncthreads();
        const int kmax = (d - k0) < 32u ? (int)(d - k0) : 32;
        for (int k = 0; k < kmax; ++k)
        {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[ty * 4 + a][k];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = Bs[tx * 4 + b][k];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __fmaf_rn(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
        {
            const uint32_t u = u0 + ty * 4 + a, i = i0 + tx * 4 + b;
            if (u < rows && i < num_items) S[(size_t)u * num_items + i] = acc[a][b];
        }
}

__global__ void mask_panel_kernel(float* sim, uint32_t rows, uint32_t num_items, const uint64_t* indptr,
                                  const uint32_t* items)
{
    const uint32_t u = blockIdx.x;
    if (u >= rows) return;
    const uint64_t lo = indptr[u], hi = indptr[u + 1];
    for (uint64_t j = lo + threadIdx.x; j < hi; j += blockDim.x) sim[(size_t)u * num_items + items[j]] = -INFINITY;
}

// k rounds of a block-wide arg-max; a taken entry is replaced by NaN and skipped afterwards.
__global__ __launch_bounds__(256) void topk_rows_kernel(float* sim, uint32_t rows, uint32_t num_items, uint32_t k,
                                                        uint32_t* topk)
{
    __shared__ float sv[256];
    __shared__ uint32_t si[256];
    const uint32_t u = blockIdx.x;
    if (u >= rows) return;
    float* row = sim + (size_t)u * num_items;
    for (uint32_t r = 0; r < k; ++r)
    {
        float bv = 0.0f;
        uint32_t bi = 0xFFFFFFFFu;
        for (uint32_t i = threadIdx.x; i < num_items; i += 256)
        {
            const float v = row[i];
            if (v != v) continue; // taken (or NaN score)
            if (bi == 0xFFFFFFFFu || v > bv) { bv = v; bi = i; }
        }
        sv[threadIdx.x] = bv;
        si[threadIdx.x] = bi;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1)
        {
            if ((int)threadIdx.x < w)
            {
                const float ov = sv[threadIdx.x + w];
                const uint32_t oi = si[threadIdx.x + w];
                const float mv = sv[threadIdx.x];
                const uint32_t mi = si[threadIdx.x];
                const bool take = (oi != 0xFFFFFFFFu) && (mi == 0xFFFFFFFFu || ov > mv || (ov == mv && oi < mi));
                if (take) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            const uint32_t w = si[0];
            topk[(size_t)u * k + r] = w;
            if (w != 0xFFFFFFFFu) row[w] = __builtin_nanf("");
        }
        __syncthreads();
    }
}

hipError_t launch_sim_panel(const float* user_rows, const float* item_w, float* sim, uint32_t rows, uint32_t num_items,
                            uint32_t emb_dim, hipStream_t s)
{
    if (rows == 0 || num_items == 0) return hipSuccess;
    dim3 grid((num_items + 63) / 64, (rows + 63) / 64);
    hipLaunchKernelGGL(sim_panel_kernel, grid, dim3(256), 0, s, user_rows, item_w, sim, rows, num_items, emb_dim);
    return hipGetLastError();
}

hipError_t launch_mask_panel(float* sim, uint32_t rows, uint32_t num_items, const uint64_t* indptr, const uint32_t* items,
                             hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(mask_panel_kernel, dim3(rows), dim3(64), 0, s, sim, rows, num_items, indptr, items);
    return hipGetLastError();
}

hipError_t launch_topk_rows(float* sim, uint32_t rows, uint32_t num_items, uint32_t k, uint32_t* topk, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, s, sim, rows, num_items, k, topk);
    return hipGetLastError();
}

} // namespace heatcf
