// ccl_train.hpp — launch interface between the host engine (engine.cpp) and the gfx950 kernels (ccl_train.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace heatcf
{

struct TrainArgs
{
    const uint2* clicks;     // [data_rows] packed {user, item} (u32)
    float*       user_w;     // [num_users, emb_dim]
    float*       user_g;
    float*       item_w;     // [num_items, emb_dim]
    float*       item_g;
    uint64_t     begin, end; // interaction range of this launch
    uint64_t     per_block;  // consecutive interactions walked by one wave (multiple of 64)
    uint32_t     num_items, num_negs, emb_dim, row_bytes;
    uint32_t     item_bytes; // num_items * row_bytes (< 4 GiB: 32-bit buffer offsets)
    uint32_t     sampling_call;
    uint32_t     exact_order; // serial/parity mode: order cross-wave row writes before the next gather
    uint32_t     tile_size;  // 0: uniform sampler; >0: random-tile sampler (used by the sampling() call only)
    uint32_t     tile_streams; // > 1: tile-resident kernel — this many single-wave streams per workgroup share one tile in LDS
    uint32_t     refresh_interval;
    uint32_t     upd_bits;   // bit0 neg W atomic, bit1 neg G atomic, bit2 pos W atomic, bit3 pos G atomic
    uint32_t     align_cap;  // how far a stream boundary may move forward to the next user-run start
    float        lr, clip;
    uint64_t     key;         // Philox key for this epoch
    uint64_t     sample_base; // added to the interaction index in the Philox counter
    const uint32_t* ext_negs; // optional caller-fed negatives [., num_negs], row (idx - ext_base)
    uint64_t     ext_base;
    uint32_t*    neg_out;     // optional record of the negatives used, row (idx - neg_out_base)
    uint64_t     neg_out_base;
    double*      loss_part;   // [grid] per-stream loss sums
    // behaviour aggregation (use_aggregator)
    uint32_t        agg;      // 0 / 1
    uint32_t        max_his;
    const uint32_t* his;      // [num_users, max_his] packed u32 history item ids
    const uint32_t* masks;    // [num_users] history lengths
    float*          w0;       // [emb_dim, emb_dim] shared aggregator weights
    float           agg_lr;   // frozen at the CONFIG learning rate (behavior_aggregators.cpp:38)
    uint32_t        agg_w0_lds; // 1: the workgroup keeps a copy of W0 in LDS; 0: W0 rows are read from L2 (emb_dim 256)
    float*          agg_state;  // [streams][agg_state_floats]: call counter + unflushed (means, gradient) pairs of every stream,
                                // carried from launch to launch within an epoch (behavior_aggregators.cpp:31,139-146: the
                                // reference's per-worker aggregator lives for the whole epoch)
};

// streams per workgroup of the tile-resident kernel (3 waves per SIMD; 12 KB of transpose tiles beside the 128 KB tile)
constexpr int TILE_STREAMS = 12;

// dynamic LDS of the aggregator kernels: [W0 copy d*d] | pair ring 32 x 2 x DP | means DP | 2 x NW x DP partials (NW > 1)
inline size_t agg_lds_bytes(uint32_t emb_dim, int lpr, int nw, bool w0_in_lds)
{
    const size_t dp = 4u * (size_t)lpr;
    return ((w0_in_lds ? (size_t)emb_dim * emb_dim : 0) + 32 * 2 * dp + dp + (nw > 1 ? 2 * (size_t)nw * dp : 0)) * sizeof(float);
}
// floats of one stream's aggregator state: [0] = call counter (as u32), then the ring of 32 x 2 x DP
__host__ __device__ inline size_t agg_state_floats(int lpr) { return 4 + 32 * 2 * 4 * (size_t)lpr; }
// the W0 copy is kept in LDS when everything fits 128 KB (the static LDS of the kernel needs the rest)
inline bool agg_w0_fits_lds(uint32_t emb_dim, int lpr, int nw) { return agg_lds_bytes(emb_dim, lpr, nw, true) <= 128u * 1024u; }

// per-epoch sampler key: (seed, epoch) -> 64-bit Philox key (= the `seed` argument of hiprand_init)
inline uint64_t epoch_key(uint64_t seed, uint64_t epoch) { return seed + 0x9E3779B97F4A7C15ull * (epoch + 1ull); }

bool       pick_variant(uint32_t emb_dim, uint32_t num_negs, bool single_wave, int* lpr, int* ng, int* nw);
void       widen_for_aggregator(int lpr, int* ng, int* nw);
hipError_t launch_train(const TrainArgs& a, int lpr, int ng, int nw, uint32_t grid, int aux, hipStream_t s);
int        query_blocks_per_cu(int lpr, int ng, int nw, int aux, bool agg, uint32_t emb_dim);
hipError_t launch_loss_reduce(const double* part, uint32_t n, double* out, hipStream_t s);
hipError_t launch_pack_clicks(const uint64_t* in, uint2* out, uint64_t n, uint32_t* stats, hipStream_t s);
hipError_t launch_pack_history(const uint64_t* his, const uint64_t* masks, uint32_t* his32, uint32_t* masks32,
                               uint64_t num_users, uint32_t max_his, uint64_t num_items, const uint2* clicks,
                               uint64_t data_rows, uint32_t* bad, hipStream_t s);
// item_sync.hip
hipError_t launch_item_delta(const float* w, const float* ref, float* mine, float* sum, size_t n_floats, hipStream_t s);
hipError_t launch_item_apply(float* w, float* ref, const float* sum, const float* mine, float scale, size_t n_floats, hipStream_t s);
hipError_t launch_item_apply_delta(float* w, float* ref, float* sum, float* mine, float scale, size_t n_floats, hipStream_t s);
hipError_t launch_item_apply_snap(float* w, const float* x, float* snap, size_t n_floats, hipStream_t s);
hipError_t launch_item_finish(float* ref, const float* sum, float* mine_x, float scale, size_t n_floats, hipStream_t s);
hipError_t launch_sample_negs(const TrainArgs& a, uint32_t grid, uint64_t out_base, uint64_t* out, hipStream_t s);

} // namespace heatcf
