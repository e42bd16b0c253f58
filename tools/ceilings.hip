// ceilings.hip — what this box's memory system delivers for the access patterns of the training kernel
// ("measure the box's achievable stream-copy GB/s ... and the random-256-B-row gather ceiling").
//   hipcc --offload-arch=gfx950 -O3 tools/ceilings.hip -o tools/ceilings && ./tools/ceilings
// 1. float4 stream copy (read + write), 2 GiB
// 2. random-row gather (read only) and random-row read-modify-write (read W,G / write W,G with sc1 buffer ops, the
//    kernel's policy) of 256-B, 512-B and 1-KiB rows, from a table that fits the 256 MB Infinity Cache (47 MB, the
//    AmazonBooks item W+G footprint) and from one that does not (4 GiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../heat_amd/csrc/ccl_device.hpp"
using namespace heatcf;

__global__ void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// one wave per chain of `rows_per_wave` groups; each group = 64 lanes x 16 B = 1 KiB = 1024/ROWB random rows
template <int ROWB, bool RMW, int AUX>
__global__ __launch_bounds__(64) void row_kernel(float* table, uint32_t num_rows, uint32_t groups_per_wave, uint64_t seed, float* sink)
{
    constexpr int LPR = ROWB / 16;
    const int lane = threadIdx.x;
    const int sub = lane % LPR, rr = lane / LPR;
    auto rs = make_rsrc(table, (uint32_t)((uint64_t)num_rows * ROWB));
    f32x4 acc = {0, 0, 0, 0};
    for (uint32_t g0 = 0; g0 < groups_per_wave; g0 += 8)
    {
        f32x4 v[8];
        uint32_t off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
            const uint64_t idx = ((uint64_t)blockIdx.x * groups_per_wave + g0 + q) * (64 / LPR) + rr;
            const uint32_t row = uniform_item(philox_draw64(0, idx, seed), num_rows);
            off[q] = row * ROWB + sub * 16;
            v[q] = buf_load<AUX>(rs, off[q]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
            if (RMW) buf_store<AUX>(rs, off[q], v[q] * 1.0001f);
            else acc += v[q];
        }
    }
    if (!RMW && acc.x == 123.456f) sink[0] = acc.x;
}

template <int ROWB, bool RMW>
void run_rows(const char* what, size_t table_bytes, int waves, uint32_t groups_per_wave)
{
    float *table, *sink;
    hipMalloc(&table, table_bytes);
    hipMemset(table, 0, table_bytes);
    hipMalloc(&sink, 16);
    const uint32_t rows = (uint32_t)(table_bytes / ROWB);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it)
    {
        hipEventRecord(a);
        hipLaunchKernelGGL((row_kernel<ROWB, RMW, AUX_SC1>), dim3(waves), dim3(64), 0, 0, table, rows, groups_per_wave, 1234 + it, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)waves * groups_per_wave * 1024.0 * (RMW ? 2.0 : 1.0);
    printf("%-58s rows of %4d B, table %7.1f MB: %8.1f GB/s (%s)\n", what, ROWB, table_bytes / 1e6, bytes / ms / 1e6, RMW ? "read+write bytes" : "read bytes");
    hipFree(table); hipFree(sink);
}

int main(int argc, char** argv)
{
    {
        const size_t n = (size_t)1 << 27; // 2 GiB of float4
        f32x4 *in, *out;
        hipMalloc(&in, n * 16); hipMalloc(&out, n * 16);
        hipMemset(in, 1, n * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 3; ++it)
        {
            hipEventRecord(a);
            hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, in, out, n);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-58s %8.1f GB/s (read+write bytes)\n", "float4 stream copy, 2 GiB -> 2 GiB", 2.0 * n * 16 / ms / 1e6);
        hipFree(in); hipFree(out);
    }
    // waves: argv[1], default 256 * 12 = what the round-1 training kernel kept resident at AmazonBooks shape (round 2: 256 * 8)
    const int waves = argc > 1 ? atoi(argv[1]) : 256 * 12;
    const uint32_t gpw = (uint32_t)(2048ull * 256 * 12 / (unsigned)waves);   // same number of rows moved
    printf("waves = %d\n", waves);
    run_rows<256, false>("random-row gather, Infinity-Cache resident", 47u << 20, waves, gpw);
    run_rows<256, true>("random-row read-modify-write, Infinity-Cache resident", 47u << 20, waves, gpw);
    run_rows<256, false>("random-row gather, HBM resident", (size_t)4000 << 20, waves, gpw);
    run_rows<256, true>("random-row read-modify-write, HBM resident", (size_t)4000 << 20, waves, gpw);
    run_rows<512, true>("random-row read-modify-write, HBM resident", (size_t)4000 << 20, waves, gpw);
    run_rows<1024, false>("random-row gather, HBM resident", (size_t)4000 << 20, waves, gpw);
    run_rows<1024, true>("random-row read-modify-write, HBM resident", (size_t)4000 << 20, waves, gpw);
    return 0;
}
