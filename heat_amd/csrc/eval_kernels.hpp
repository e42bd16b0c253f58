// eval_kernels.hpp — evaluation path: dense U*V^T panels (Engine::evaluate0, train/engine.cpp:388-400), train-item
// masking and per-user top-k (cf/metrics.py:21-29).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace heatcf
{
// sim[rows, num_items] = U[rows, d] * V[num_items, d]^T (fp32, k summed left to right like the oracle's dot)
hipError_t launch_sim_panel(const float* user_rows, const float* item_w, float* sim, uint32_t rows, uint32_t num_items,
                            uint32_t emb_dim, hipStream_t s);
// sim[u, items[indptr[u] .. indptr[u+1])] = -inf  (metrics.py:24)
hipError_t launch_mask_panel(float* sim, uint32_t rows, uint32_t num_items, const uint64_t* indptr, const uint32_t* items,
                             hipStream_t s);
// topk[u, 0..k) = ids of the k largest entries of row u, descending; ties -> lower id first.  Destroys `sim`.
hipError_t launch_topk_rows(float* sim, uint32_t rows, uint32_t num_items, uint32_t k, uint32_t* topk, hipStream_t s);

// Fused path (topk_fused.hip): no score matrix; k <= TOPK_FUSED_MAX_K, emb_dim % 4 == 0, mask rows sorted ascending.
#define TOPK_FUSED_MAX_K 64
#define TOPK_FUSED_MAX_SPLITS 16
// how many item-range splits to run for `rows` users on a chip with `cus` compute units (depends on the kernel (emb_dim, k) select)
uint32_t topk_fused_splits(uint32_t rows, uint32_t num_items, uint32_t cus, uint32_t emb_dim, uint32_t k);
// part_v / part_i: scratch [splits, rows, k]; thr_shared: scratch [rows] (threshold exchange between item splits);
// topk: DEVICE [rows, k]; indptr (relative, [rows+1]) / items may be NULL
hipError_t launch_topk_fused(const float* user_rows, const float* item_w, uint32_t rows, uint32_t num_items,
                             uint32_t emb_dim, uint32_t k, const uint64_t* indptr, const uint32_t* items, uint32_t splits,
                             float* part_v, uint32_t* part_i, uint32_t* topk, float* thr_shared, hipStream_t s);
// items_out[indptr[r] .. indptr[r+1]) = sorted(items_in[same range]) for every row r (mask_sort.hip).  Call once with
// temp == NULL to learn *temp_bytes, then with the scratch.  id_bits = bits needed for the largest id.
hipError_t sort_mask_rows(const uint32_t* items_in, uint32_t* items_out, uint32_t n_items, uint32_t rows,
                          const uint64_t* indptr, uint32_t id_bits, void* temp, size_t* temp_bytes, hipStream_t s);
} // namespace heatcf
