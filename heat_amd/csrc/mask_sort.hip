// mask_sort.hip — ascending order inside every row of the train-item CSR the fused top-k walks (topk_fused.hip).
// LightGCN files list a user's items in arbitrary order (cf/datasets.py:31-79 keeps file order), so the rows arrive
// unsorted; rocPRIM's segmented radix sort orders all rows in one launch on the engine's stream.
#include "eval_kernels.hpp"

#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace heatcf
{
hipError_t sort_mask_rows(const uint32_t* items_in, uint32_t* items_out, uint32_t n_items, uint32_t rows,
                          const uint64_t* indptr, uint32_t id_bits, void* temp, size_t* temp_bytes, hipStream_t s)
{
    size_t bytes = temp ? *temp_bytes : 0;
    const hipError_t err = rocprim::segmented_radix_sort_keys(temp, bytes, items_in, items_out, n_items, rows, indptr,
                                                              indptr + 1, 0u, id_bits, s);
    if (!temp) *temp_bytes = bytes;
    return err;
}
} // namespace heatcf
