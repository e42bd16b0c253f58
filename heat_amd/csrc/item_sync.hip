// item_sync.hip — the two element-wise passes of the multi-GPU item-table exchange, fused.
//
// Replaces the fork's per-row MPI_Allreduce + "/ world_size" loop (train/engine.cpp:366-375) together with ONE collective
// over the whole table that the caller issues (RCCL over xGMI through torch.distributed).  Between two exchanges every
// rank trains its user shard on its own replica of the item table; at an exchange
//     delta :  mine = sum = W - ref                      (what this rank changed since the last common reference)
//     [caller: all-reduce(sum) over the ranks, possibly still in flight while the next window trains]
//     apply :  W += scale * sum - mine ;  ref += scale * sum      (mine == NULL, nothing trained meanwhile: W = ref = ref + scale * sum)
// With scale = 1 every rank's updates are applied (the cross-GPU analogue of the in-GPU scatter-add), with
// scale = 1 / world_size the replicas are averaged (the fork's intent).  Because `apply` adds the OTHER ranks' deltas on
// top of whatever W has become, the all-reduce of window k may overlap the training of window k+1: the next delta,
// W - ref, is then exactly this rank's progress since the snapshot of window k.
// Both passes are pure streaming (16 B per lane, grid-stride), HBM / Infinity-Cache bandwidth bound:
// delta moves 16 B per element (2 reads, 2 writes), apply 24 B (4 reads, 2 writes).
#include "ccl_train.hpp"

namespace heatcf
{

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void item_delta_kernel(const f4* __restrict__ w, const f4* __restrict__ ref,
                                                         f4* __restrict__ mine, f4* __restrict__ sum, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        const f4 d = w[i] - ref[i];
        if (mine) mine[i] = d;                         // NULL in the blocking form (apply with mine == NULL)
        sum[i] = d;
    }
}

__global__ __launch_bounds__(256) void item_apply_kernel(f4* __restrict__ w, f4* __restrict__ ref, const f4* __restrict__ sum,
                                                         const f4* __restrict__ mine, float scale, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        const f4 s = scale * sum[i];
        w[i] = w[i] + (s - mine[i]);
        ref[i] = ref[i] + s;
    }
}

// Blocking form (nothing trained since the delta was taken): W = ref = ref + scale * sum — the same expression on every
// rank, so the replicas are bit-identical afterwards.
__global__ __launch_bounds__(256) void item_apply_exact_kernel(f4* __restrict__ w, f4* __restrict__ ref, const f4* __restrict__ sum,
                                                               float scale, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        const f4 r = ref[i] + scale * sum[i];
        w[i] = r;
        ref[i] = r;
    }
}

// apply of exchange k and delta of exchange k + 1 in ONE pass (the overlapped schedule runs them back to back at every window
// boundary): W += s - mine ; ref += s ; mine = sum = W - ref  — 4 tables read, 4 written (187 MB at AmazonBooks shape)
// instead of 234 MB in two launches; the same expressions in the same order, so the same bits.
__global__ __launch_bounds__(256) void item_apply_delta_kernel(f4* __restrict__ w, f4* __restrict__ ref, f4* __restrict__ sum,
                                                               f4* __restrict__ mine, float scale, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        // (non-temporal accesses for the three exchange buffers were measured: no change — what an exchange costs the next
        // training kernel is not these buffers displacing the tables)
        const f4 s = scale * sum[i];
        const f4 wn = w[i] + (s - mine[i]);
        const f4 rn = ref[i] + s;
        const f4 d = wn - rn;
        w[i] = wn;
        ref[i] = rn;
        mine[i] = d;
        sum[i] = d;
    }
}

// ---- the pipelined form (round 3): only ONE pass stays on the training stream ---------------------------------------------
// The overlapped exchange above still runs delta (94 MB at AmazonBooks shape) and apply (140 MB) on the training stream at
// every window boundary: 0.17-0.19 ms per epoch next to a 0.96 ms shard epoch of an 8-GPU job.  The same algebra cut so that
// the training stream only does
//     apply_snap :  W += x ; snap = W            (x = what the previous exchange brought from the OTHER ranks; 4 row passes)
// and everything else works on `snap` on an exchange stream of the caller, concurrently with the next window:
//     delta_from :  mine = sum = snap - ref
//     [all-reduce(sum)]
//     finish     :  s = scale * sum ; x = s - mine (written over `mine`) ; ref += s
// Element for element these are the expressions of item_delta_kernel / item_apply_kernel (W + (s - mine), ref + s,
// W - ref), so the tables are bit-identical to the overlapped form's.
__global__ __launch_bounds__(256) void item_apply_snap_kernel(f4* __restrict__ w, const f4* __restrict__ x, f4* __restrict__ snap,
                                                              size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        f4 v = w[i];
        if (x)
        {
            v = v + x[i];
            w[i] = v;
        }
        snap[i] = v;
    }
}

__global__ __launch_bounds__(256) void item_finish_kernel(f4* __restrict__ ref, const f4* __restrict__ sum, f4* __restrict__ mine_x,
                                                          float scale, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    {
        const f4 s = scale * sum[i];
        mine_x[i] = s - mine_x[i];
        ref[i] = ref[i] + s;
    }
}

static uint32_t stream_grid(size_t n4)
{
    const size_t want = (n4 + 255) / 256;
    return (uint32_t)(want < 1 ? 1 : (want > 256u * 16u ? 256u * 16u : want));   // <= 16 workgroups per CU
}

hipError_t launch_item_delta(const float* w, const float* ref, float* mine, float* sum, size_t n_floats, hipStream_t s)
{
    const size_t n4 = n_floats / 4;
    if (n4 == 0) return hipSuccess;
    hipLaunchKernelGGL(item_delta_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (const f4*)w, (const f4*)ref, (f4*)mine, (f4*)sum, n4);
    return hipGetLastError();
}

hipError_t launch_item_apply(float* w, float* ref, const float* sum, const float* mine, float scale, size_t n_floats, hipStream_t s)
{
    const size_t n4 = n_floats / 4;
    if (n4 == 0) return hipSuccess;
    if (mine)
        hipLaunchKernelGGL(item_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (f4*)w, (f4*)ref, (const f4*)sum, (const f4*)mine, scale, n4);
    else
        hipLaunchKernelGGL(item_apply_exact_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (f4*)w, (f4*)ref, (const f4*)sum, scale, n4);
    return hipGetLastError();
}

hipError_t launch_item_apply_delta(float* w, float* ref, float* sum, float* mine, float scale, size_t n_floats, hipStream_t s)
{
    const size_t n4 = n_floats / 4;
    if (n4 == 0) return hipSuccess;
    hipLaunchKernelGGL(item_apply_delta_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (f4*)w, (f4*)ref, (f4*)sum, (f4*)mine, scale, n4);
    return hipGetLastError();
}

hipError_t launch_item_apply_snap(float* w, const float* x, float* snap, size_t n_floats, hipStream_t s)
{
    const size_t n4 = n_floats / 4;
    if (n4 == 0) return hipSuccess;
    hipLaunchKernelGGL(item_apply_snap_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (f4*)w, (const f4*)x, (f4*)snap, n4);
    return hipGetLastError();
}

hipError_t launch_item_finish(float* ref, const float* sum, float* mine_x, float scale, size_t n_floats, hipStream_t s)
{
    const size_t n4 = n_floats / 4;
    if (n4 == 0) return hipSuccess;
    hipLaunchKernelGGL(item_finish_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (f4*)ref, (const f4*)sum, (f4*)mine_x, scale, n4);
    return hipGetLastError();
}

} // namespace heatcf
