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
} // namespace heatcf
