// Sparse -> dense scatters (HBM-bound byte movers).
//   sec_sparse_to_dense : SparseConvTensor.dense() feeding the RPN (second/pytorch/models/middle.py:206-210)
//   sec_pillar_scatter  : PointPillarsScatter.forward (second/pytorch/models/pointpillars.py:444-476),
//                         replacing its per-sample Python loop with one launch.
// The destination is addressed through element strides so the same kernel writes NCDHW-contiguous
// (API parity) or the channels-last [B, C*D, H, W] image the bf16 RPN consumes (no permute copy).
#include "common.hpp"

namespace sec {

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scatter_rows(const T *__restrict__ feat, const int *__restrict__ idx, int n,
                                                        const int *__restrict__ num_dev, int c, T *__restrict__ out,
                                                        long long sb, long long sc, long long sz, long long sy,
                                                        long long sx) {
    if (num_dev) n = *num_dev;
    long long total = (long long)n * c;
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < total; g += (long long)gridDim.x * kBlock) {
        int i = (int)(g / c), ch = (int)(g % c);
        int4 q = *reinterpret_cast<const int4 *>(idx + (size_t)i * 4);
        out[q.x * sb + ch * sc + q.y * sz + q.z * sy + q.w * sx] = feat[g];
    }
}

// adjoint of k_scatter_rows: rows[i, ch] = dense[idx[i]] (the backward of SparseConvTensor.dense())
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_rows(const T *__restrict__ dense, const int *__restrict__ idx, int n,
                                                       int c, T *__restrict__ rows, long long sb, long long sc,
                                                       long long sz, long long sy, long long sx, const int *__restrict__ num_dev) {
    if (num_dev) { const int live = *num_dev; n = live < n ? (live > 0 ? live : 0) : n; }   // static capacity: indices past the live rows are unspecified
    long long total = (long long)n * c;
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < total; g += (long long)gridDim.x * kBlock) {
        int i = (int)(g / c), ch = (int)(g % c);
        int4 q = *reinterpret_cast<const int4 *>(idx + (size_t)i * 4);
        rows[g] = dense[q.x * sb + ch * sc + q.y * sz + q.z * sy + q.w * sx];
    }
}

template <typename T>
static int run_gather(const void *dense, const int *idx, int n, int c, void *rows, long long sb, long long sc,
                      long long sz, long long sy, long long sx, const int *num_dev, hipStream_t st) {
    if (n == 0) return SEC_OK;
    int blocks = div_up((long long)n * c, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_gather_rows<T>, dim3(blocks), dim3(kBlock), 0, st, (const T *)dense, idx, n, c, (T *)rows, sb, sc,
                       sz, sy, sx, num_dev);
    return check_launch();
}

// Zero fill of the dense image as a KERNEL, not hipMemsetAsync: inside a captured hipGraph the runtime's memset node was
// measured (tools/inflight_stress.py, ROCm 7.2) to start writing a small non-zero pattern (bf16 0x0180, 0x01c0, ...) instead of
// zeros after some tens of replays of the same graph -- numerically invisible (4.7e-38) but it defeats the all-zero-tile test of
// the first RPN layer and flips rounding in a tie-dominated top-k once in a few hundred steps.
__global__ __launch_bounds__(kBlock) void k_zero_fill(uint4 *__restrict__ p, long long n16, unsigned char *__restrict__ tail, int ntail) {
    const long long stride = (long long)gridDim.x * kBlock;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

static int zero_fill(void *out, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SEC_OK;
    if (((uintptr_t)out & 15) != 0) return hip_ok(hipMemsetAsync(out, 0, bytes, st));     // never for torch tensors (256-byte aligned)
    const long long n16 = (long long)(bytes / 16);
    const int ntail = (int)(bytes - (size_t)n16 * 16);
    long long blocks = div_up(n16 > 0 ? n16 : 1, (long long)kBlock * 4);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(kBlock), 0, st, (uint4 *)out, n16, (unsigned char *)out + (size_t)n16 * 16, ntail);
    return check_launch();
}

template <typename T>
static int run_scatter(const void *feat, const int *idx, int n, const int *num_dev, int c, void *out, size_t out_elems,
                       long long sb, long long sc, long long sz, long long sy, long long sx, hipStream_t st) {
    int rc;
    if ((rc = zero_fill(out, out_elems * sizeof(T), st))) return rc;
    if (n == 0) return SEC_OK;
    int blocks = div_up((long long)n * c, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_scatter_rows<T>, dim3(blocks), dim3(kBlock), 0, st, (const T *)feat, idx, n, num_dev, c, (T *)out,
                       sb, sc, sz, sy, sx);
    return check_launch();
}

// Site map of a sparse tensor: map[b][z][y][x] = row + 1 (0 = no active site).  What sec_conv2d_nhwc_gather reads instead
// of a dense image.
__global__ __launch_bounds__(kBlock) void k_site_map(const int *__restrict__ idx, int n, const int *__restrict__ num_dev, int batch, int d,
                                                    int h, int w, int *__restrict__ map) {
    if (num_dev) n = *num_dev;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int4 q = *reinterpret_cast<const int4 *>(idx + (size_t)i * 4);
        if ((unsigned)q.x < (unsigned)batch && (unsigned)q.y < (unsigned)d && (unsigned)q.z < (unsigned)h && (unsigned)q.w < (unsigned)w)
            map[(((size_t)q.x * d + q.y) * h + q.z) * w + q.w] = i + 1;
    }
}

}  // namespace sec

using namespace sec;

SEC_API int sec_sparse_site_map(const int *indices, int n, const int *num_dev, int batch, int d, int h, int w, int *site_map,
                                void *stream) {
    if (n < 0 || batch <= 0 || d <= 0 || h <= 0 || w <= 0 || !site_map || (n > 0 && !indices)) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = zero_fill(site_map, (size_t)batch * d * h * w * sizeof(int), st))) return rc;
    if (n == 0) return SEC_OK;
    int blocks = div_up(n, kBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_site_map, dim3(blocks), dim3(kBlock), 0, st, indices, n, num_dev, batch, d, h, w, site_map);
    return check_launch();
}

SEC_API int sec_sparse_to_dense(const void *features, const int *indices, int n, int c, const int *num_dev, void *out,
                                size_t out_elems, int64_t stride_b, int64_t stride_c, int64_t stride_z, int64_t stride_y,
                                int64_t stride_x, int dtype, void *stream) {
    if (n < 0 || c <= 0 || !out || (n > 0 && (!features || !indices))) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32)
        return run_scatter<float>(features, indices, n, num_dev, c, out, out_elems, stride_b, stride_c, stride_z, stride_y, stride_x, st);
    if (dtype == SEC_F16 || dtype == SEC_BF16)  // pure byte movement: both are 16-bit
        return run_scatter<unsigned short>(features, indices, n, num_dev, c, out, out_elems, stride_b, stride_c, stride_z, stride_y, stride_x, st);
    return SEC_E_UNSUPPORTED;
}

SEC_API int sec_dense_to_sparse(const void *dense, const int *indices, int n, int c, const int *num_dev, void *rows, int64_t stride_b,
                                int64_t stride_c, int64_t stride_z, int64_t stride_y, int64_t stride_x, int dtype,
                                void *stream) {
    if (n < 0 || c <= 0 || (n > 0 && (!dense || !indices || !rows))) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32)
        return run_gather<float>(dense, indices, n, c, rows, stride_b, stride_c, stride_z, stride_y, stride_x, num_dev, st);
    if (dtype == SEC_F16 || dtype == SEC_BF16)
        return run_gather<unsigned short>(dense, indices, n, c, rows, stride_b, stride_c, stride_z, stride_y, stride_x, num_dev, st);
    return SEC_E_UNSUPPORTED;
}

SEC_API int sec_pillar_scatter(const void *features, const int *coords, int p, int c, const int *num_dev, void *out,
                               size_t out_elems, int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                               int dtype, void *stream) {
    // coords are (b, z, y, x) with z == 0 for pillars; the reference ignores z (pointpillars.py:462)
    return sec_sparse_to_dense(features, coords, p, c, num_dev, out, out_elems, stride_b, stride_c, 0, stride_y, stride_x,
                               dtype, stream);
}
