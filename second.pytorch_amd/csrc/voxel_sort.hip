// Stable device radix sort of (voxel row, point index) pairs for the voxeliser's many-points-per-voxel path (pillars:
// max_points 60, nuscenes/all.pp.largea.config:10).  rocPRIM's device-wide radix sort is the one library primitive of the
// library (a plain key sort, like the plain GEMMs left to hipBLASLt); it lives in its own translation unit so that the header's
// compile time stays out of the other kernels.
#include <cstring>
#include <string.h>
#include "common.hpp"
#include <rocprim/device/device_radix_sort.hpp>

namespace sec {

size_t vox_sort_temp_bytes(int n, int bits) {
    size_t b = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr,
                                                   (int *)nullptr, (unsigned)(n > 0 ? n : 1), 0u, (unsigned)bits, (hipStream_t)0);
    if (e != hipSuccess || b == 0) b = (size_t)(n > 0 ? n : 1) * 16 + (1u << 20);   // no device to ask (build container): an upper bound
    (void)hipGetLastError();
    return align_up(b);
}

int vox_sort_pairs(void *tmp, size_t tmp_bytes, const unsigned *kin, unsigned *kout, const int *vin, int *vout, int n, int bits,
                   hipStream_t st) {
    return hip_ok(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (unsigned)n, 0u, (unsigned)bits, st));
}

}  // namespace sec
