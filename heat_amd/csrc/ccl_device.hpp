// ccl_device.hpp — gfx950 device helpers shared by the training / sampling / evaluation kernels.
// CDNA4 only: 64-lane wavefronts, DPP row operations (a DPP "row" is 16 lanes), buffer instructions
// with gfx940-family cache-policy bits (sc0/sc1/nt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace heatcf
{

typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// gfx940-family buffer cache-policy bits (LLVM AMDGPU CPol): sc0 = 1, nt = 2, sc1 = 16.
constexpr int AUX_PLAIN = 0;
constexpr int AUX_SC1   = 16; // agent ("device") scope: loads bypass the per-CU L1, stores write through the XCD L2
constexpr int AUX_SC01  = 17;

// ---- DPP / cross-lane ------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float lane_xor(float x, int mask)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, __builtin_bit_cast(int, x)));
}

__device__ __forceinline__ uint32_t lane_get(uint32_t x, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)x);
}

// Sum over the LPR consecutive lanes that hold one embedding row; every lane of the group gets the total.
// 16 lanes = one DPP row: 4 full-rate VALU adds, no LDS traffic.
template <int LPR>
__device__ __forceinline__ float row_sum(float x)
{
    x += dpp_mov<0xB1>(x);               // quad_perm [1,0,3,2]  (xor 1)
    x += dpp_mov<0x4E>(x);               // quad_perm [2,3,0,1]  (xor 2)
    if (LPR >= 8) x += dpp_mov<0x141>(x);  // row_half_mirror      (other quad of the 8-lane half)
    if (LPR >= 16) x += dpp_mov<0x140>(x); // row_mirror           (other half of the 16-lane row)
    if (LPR >= 32) x += lane_xor(x, 16);
    if (LPR >= 64) x += lane_xor(x, 32);
    return x;
}

// Reductions across the R = 64/LPR row groups of a wave (lanes l, l^LPR, l^2LPR, ...).
template <int LPR>
__device__ __forceinline__ float cross_sum(float x)
{
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) x += lane_xor(x, m);
    return x;
}
template <int LPR>
__device__ __forceinline__ float cross_max(float x)
{
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) x = fmaxf(x, lane_xor(x, m));
    return x;
}

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__device__ __forceinline__ f32x4 clip4(f32x4 g, float c)
{
    // optimizers/optimizer.cpp:17-22: max(min(g, c), -c) == median(g, -c, c) for c >= 0: one v_med3_f32 per element
    f32x4 r;
    r.x = __builtin_amdgcn_fmed3f(g.x, -c, c);
    r.y = __builtin_amdgcn_fmed3f(g.y, -c, c);
    r.z = __builtin_amdgcn_fmed3f(g.z, -c, c);
    r.w = __builtin_amdgcn_fmed3f(g.w, -c, c);
    return r;
}

// 1/x and sqrt(x) as single hardware instructions (v_rcp_f32 / v_sqrt_f32, 1 ulp) instead of the ~10-instruction
// correctly rounded sequences: the kernel evaluates 19 divisions and 6 square roots per interaction, all on values that
// are clamped away from 0 (eps = 1e-8) and far from the denormal range.  The difference to IEEE division is <= 2 ulp.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// ---- buffer (SRD) access with cache policy ---------------------------------------------------------------
// An out-of-range byte offset makes a raw buffer load return 0 and a store be dropped: masked lanes simply
// use OOB_OFF, no divergent branches around memory instructions.
constexpr uint32_t OOB_OFF = 0xFFFFFFF0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

template <int AUX>
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, AUX);
    return __builtin_bit_cast(f32x4, v);
}

template <int AUX>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)byte_off, 0, AUX);
}

// ---- Philox4x32-10 (Salmon et al., SC'11), identical to hipRAND/rocRAND's device generator -------------------
// philox(counter = {slot, 0, idx_lo, idx_hi}, key = {seed_lo, seed_hi}) equals
//   hiprand_init(seed, /*subsequence*/ idx, /*offset*/ 4 * slot, &st); hiprand4(&st)
// (rocrand_philox4x32_10.h: seed() sets the key, discard_subsequence adds to counter.zw, discard(4*slot) adds
// slot to counter.xy).  Checked on the GPU against hiprand_kernel.h itself: oracle/hiprand_kat.hip +
// tests/test_gpu_parity.py::test_philox_equals_hiprand_device_api.
__device__ __forceinline__ uint64_t philox_draw64(uint32_t slot, uint64_t idx, uint64_t seed)
{
    uint32_t c0 = slot, c1 = 0u, c2 = (uint32_t)idx, c3 = (uint32_t)(idx >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    constexpr int ROUNDS = 10;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
    {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (uint64_t)c0 | ((uint64_t)c1 << 32);
}

// Uniform id in [0, num_items): high 64 bits of draw64 * num_items (Lemire's multiply-shift; the bias is
// < num_items / 2^64, i.e. < 1e-12 for every shape in scope).  Replaces random/uniform.hpp:27-30.
__device__ __forceinline__ uint32_t uniform_item(uint64_t draw, uint32_t num_items)
{
    return (uint32_t)__umul64hi(draw, (uint64_t)num_items);
}

// Random-tile negative sampler (negative_samplers/random_tile_negative_sampler.cpp:23-45), stateless form.
// A worker (= one stream) keeps a tile of `tile_size` uniformly drawn ids and refreshes it every `refresh_interval`
// sampling() calls; a negative is tile[j] with j uniform in [0, tile_size).  Here tile entry j of (stream, tile_epoch)
// is itself a Philox draw from a separate counter domain (bit 63 of the index set), so no tile is ever stored:
//   j  = mulhi64(philox(slot, sample_base + idx), tile_size)
//   id = mulhi64(philox(j, 2^63 | stream << 32 | tile_epoch), num_items)
// id of tile entry j of (tile owner, tile_epoch)
__device__ __forceinline__ uint32_t tile_entry(uint32_t j, uint64_t key, uint32_t owner, uint64_t tile_epoch, uint32_t num_items)
{
    const uint64_t tidx = (1ull << 63) | ((uint64_t)owner << 32) | (tile_epoch & 0xFFFFFFFFull);
    return uniform_item(philox_draw64(j, tidx, key), num_items);
}

__device__ __forceinline__ uint32_t tile_item(uint32_t slot, uint64_t idx, uint64_t key, uint32_t stream, uint64_t call,
                                              uint32_t tile_size, uint32_t refresh_interval, uint32_t num_items,
                                              uint32_t* tile_index = nullptr)
{
    const uint32_t j = (uint32_t)__umul64hi(philox_draw64(slot, idx, key), (uint64_t)tile_size);
    if (tile_index) *tile_index = j;
    return tile_entry(j, key, stream, call / (uint64_t)refresh_interval, num_items);
}

} // namespace heatcf
