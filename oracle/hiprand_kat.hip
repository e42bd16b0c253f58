// TEST INFRASTRUCTURE — known-answer probe for the on-GPU negative sampler (north_star: "uniform negative sampler
// on-GPU (hiprand)").  The product kernel hand-rolls Philox4x32-10 (heat_amd/csrc/ccl_device.hpp: philox_draw64) and
// documents it as identical to hipRAND's device generator:
//     hiprand_init(key, /*subsequence*/ idx, /*offset*/ 4 * slot, &st);  hiprand4(&st)  ->  (.x | .y << 32)
// This translation unit runs BOTH on the GPU, for the same (key, idx, slot) triples, and hands the two streams to the
// test (tests/test_gpu_parity.py::test_philox_equals_hiprand_device_api), which demands bit equality.  Only tests load it.
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>
#include <stdint.h>

#include "../heat_amd/csrc/ccl_device.hpp"

__global__ void hiprand_kat_kernel(uint64_t key, uint64_t idx_base, uint32_t n_idx, uint32_t n_slots, uint64_t* out_hiprand,
                                   uint64_t* out_heat)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * n_slots) return;
    const uint64_t idx = idx_base + t / n_slots;
    const uint32_t slot = t % n_slots;
    hiprandStatePhilox4_32_10_t st;
    hiprand_init(key, idx, 4ull * slot, &st);
    const uint4 r = hiprand4(&st);
    out_hiprand[t] = (uint64_t)r.x | ((uint64_t)r.y << 32);
    out_heat[t] = heatcf::philox_draw64(slot, idx, key);
}

extern "C" int hiprand_kat_draws(uint64_t key, uint64_t idx_base, uint32_t n_idx, uint32_t n_slots, uint64_t* out_hiprand,
                                 uint64_t* out_heat)
{
    const size_t n = (size_t)n_idx * n_slots;
    if (n == 0) return 0;
    uint64_t *d_a = nullptr, *d_b = nullptr;
    if (hipMalloc(&d_a, n * 8) != hipSuccess) return -1;
    if (hipMalloc(&d_b, n * 8) != hipSuccess) { (void)hipFree(d_a); return -1; }
    hipLaunchKernelGGL(hiprand_kat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, key, idx_base, n_idx, n_slots, d_a, d_b);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_hiprand, d_a, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_heat, d_b, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_a);
    (void)hipFree(d_b);
    return e == hipSuccess ? 0 : -2;
}
