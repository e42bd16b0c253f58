// coherence_probe.hip — does a CU on another XCD see a plain / sc1 store while both kernels are still running?
// Build: hipcc --offload-arch=gfx950 -O2 tools/coherence_probe.hip -o tools/coherence_probe ; run on the GPU box.
// One producer workgroup rewrites a 16-byte word K times; consumer workgroups (spread over all XCDs) poll it with a
// given load policy and report the last value they saw and how many distinct values they observed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../heat_amd/csrc/ccl_device.hpp"
using namespace heatcf;

template <int LD, int ST>
__global__ __launch_bounds__(64) void probe(float* word, uint32_t* out, int K, int M, volatile int* go)
{
    const int lane = threadIdx.x;
    auto rs = make_rsrc(word, 64);
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID bits [3:0]
    if (blockIdx.x == 0)
    {
        for (int i = 1; i <= K; ++i)
        {
            f32x4 v = {(float)i, (float)i, (float)i, (float)i};
            if (lane == 0) buf_store<ST>(rs, 0, v);
            __builtin_amdgcn_s_sleep(100);
            __builtin_amdgcn_s_sleep(100);
        }
        if (lane == 0) { out[0] = K; out[1] = 0; out[2] = xcc & 0xF; }
        return;
    }
    // consumers: warm the line with a plain load first
    f32x4 w = buf_load<0>(rs, lane == 0 ? 0 : OOB_OFF);
    float last = w.x;
    uint32_t changes = 0;
    for (int i = 0; i < M; ++i)
    {
        f32x4 v = buf_load<LD>(rs, lane == 0 ? 0 : OOB_OFF);
        if (v.x != last) { ++changes; last = v.x; }
        __builtin_amdgcn_s_sleep(40);
    }
    if (lane == 0)
    {
        out[blockIdx.x * 4 + 0] = (uint32_t)last;
        out[blockIdx.x * 4 + 1] = changes;
        out[blockIdx.x * 4 + 2] = xcc & 0xF;
    }
}

template <int LD, int ST>
void run(const char* name)
{
    const int blocks = 33, K = 4000, M = 20000;
    float* word; uint32_t* out;
    hipMalloc(&word, 256); hipMemset(word, 0, 256);
    hipMalloc(&out, blocks * 16); hipMemset(out, 0, blocks * 16);
    hipLaunchKernelGGL((probe<LD, ST>), dim3(blocks), dim3(64), 0, 0, word, out, K, M, nullptr);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(blocks * 4);
    hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
    uint32_t min_last = ~0u, max_last = 0, min_ch = ~0u, max_ch = 0; int same_xcd_ch = -1;
    for (int b = 1; b < blocks; ++b)
    {
        min_last = std::min(min_last, h[b * 4]); max_last = std::max(max_last, h[b * 4]);
        min_ch = std::min(min_ch, h[b * 4 + 1]); max_ch = std::max(max_ch, h[b * 4 + 1]);
        if (h[b * 4 + 2] == h[2]) same_xcd_ch = (int)h[b * 4 + 1];
    }
    printf("%-28s producer_xcc=%u consumers: last seen min=%u max=%u (of %d)  distinct values seen min=%u max=%u  same-xcd consumer changes=%d\n",
           name, h[2], min_last, max_last, K, min_ch, max_ch, same_xcd_ch);
    hipFree(word); hipFree(out);
}

int main()
{
    run<0, 0>("plain load / plain store");
    run<16, 0>("sc1 load / plain store");
    run<0, 16>("plain load / sc1 store");
    run<16, 16>("sc1 load / sc1 store");
    run<17, 17>("sc0sc1 load / sc0sc1 store");
    run<1, 1>("sc0 load / sc0 store");
    run<2, 2>("nt load / nt store");
    return 0;
}
