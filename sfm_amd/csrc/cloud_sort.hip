// cloud_sort.hip -- the one place that uses a ROCm library primitive: rocPRIM's device radix sort
// orders the output cloud by packed voxel key for sfm_fusion_read_points (boundary plumbing after
// the run, not part of the fusion path).  Its own translation unit keeps the template-heavy
// header out of the kernels' compile.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace sfm {

// keys: packed voxel keys (bits 0..62); idx: their positions.  tmp == nullptr: size query.
hipError_t sort_keys_with_index(void *tmp, size_t &tmp_bytes, const unsigned long long *keys_in,
                                unsigned long long *keys_out, const uint32_t *idx_in, uint32_t *idx_out, size_t n,
                                hipStream_t stream)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0u, 63u, stream);
}

}  // namespace sfm
