// Development aid: issue rate of scalar vs packed fp32 multiply / add / fma on gfx950 (one wave per SIMD and four).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float x, float y)
{
    f2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{x + i, y + threadIdx.x};
    const f2 m = f2{x, y};
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            if (MODE == 0) { acc[i] = acc[i] + acc[i] * m; }                                  // pk_mul + pk_add
            if (MODE == 1) { acc[i].x = acc[i].x + acc[i].x * m.x; acc[i].y = acc[i].y + acc[i].y * m.y; } // may pack too
            if (MODE == 2) { acc[i] = __builtin_elementwise_fma(acc[i], m, acc[i]); }           // pk_fma
            if (MODE == 3) { acc[i].x = __builtin_fmaf(acc[i].x, m.x, acc[i].x); acc[i].y = __builtin_fmaf(acc[i].y, m.y, acc[i].y); }
        }
    }
    f2 s = f2{0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) s = s + acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <int MODE> void run(const char* name, int blocks, float flop_per_lane_iter)
{
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.9999f);
    hipEventRecord(a);
    rate_kernel<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 256 * iters * flop_per_lane_iter;
    printf("%-28s blocks=%5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int blocks : {256, 1024, 2048})
    {
        run<0>("packed mul + packed add", blocks, 16 * 2 * 2);
        run<1>("scalar mul + scalar add", blocks, 16 * 2 * 2);
        run<2>("packed fma", blocks, 16 * 2 * 2);
        run<3>("scalar fma", blocks, 16 * 2 * 2);
    }
    return 0;
}
