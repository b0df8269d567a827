// PointPillars front end + NuScenes block filter on gfx950.
//   sec_pfn_fwd                : PillarFeatureNet.forward with one PFNLayer, eval mode
//                                (second/pytorch/models/pointpillars.py:203-237 + :51-65): ~15 torch kernels on a
//                                [P,60,9] tensor in the reference, ONE launch here.  One wave64 per pillar, lane =
//                                output channel (64 channels == one wave), points broadcast with v_readlane.
//   sec_voxel_block_filter_f32 : height-span filter of points_to_voxel_3d_with_filtering (SURVEY A.2,
//                                enabled by second/configs/nuscenes/all.fhd.config:9-12) + order-preserving
//                                compaction of the voxel arrays (flag + exclusive scan).
#include "common.hpp"

namespace sec {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <typename OT> __device__ __forceinline__ OT cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 cvt_out(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half cvt_out(float v) { return __float2half_rn(v); }

// F = 4 point features (x, y, z, r/dt), IN = 9 decorated features
template <typename OT>
__global__ __launch_bounds__(kBlock) void k_pfn_fwd(const float *__restrict__ voxels, const int *__restrict__ num_points,
                                                   const int *__restrict__ coords, int P, const int *__restrict__ num_dev,
                                                   int T, const float *__restrict__ wt, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, int C, float vx, float vy, float xo,
                                                   float yo, OT *__restrict__ out) {
    if (num_dev) P = *num_dev < P ? *num_dev : P;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= P) return;
    const float4 *pv = reinterpret_cast<const float4 *>(voxels) + (size_t)p * T;
    const int n = num_points[p];
    const int4 co = *reinterpret_cast<const int4 *>(coords + (size_t)p * 4);
    const float cx = __fadd_rn(__fmul_rn((float)co.w, vx), xo), cy = __fadd_rn(__fmul_rn((float)co.z, vy), yo);
    // pillar mean over all T slots (padded slots are zero) / n
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int t0 = 0; t0 < T; t0 += 64) {
        float4 q = (t0 + lane < T) ? pv[t0 + lane] : make_float4(0, 0, 0, 0);
        sx += q.x; sy += q.y; sz += q.z;
    }
    const float inv = (float)n;
    const float mx = wave_sum(sx) / inv, my = wave_sum(sy) / inv, mz = wave_sum(sz) / inv;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < C;
        float w[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) w[j] = live ? wt[(size_t)j * C + c] : 0.f;
        const float sc = live ? scale[c] : 0.f, sh = live ? shift[c] : 0.f;
        float best = (n < T) ? fmaxf(sh, 0.f) : -INFINITY;   // padded slots: relu(bn(linear(0)))
        for (int t = 0; t < n && t < T; ++t) {
            const float4 q = pv[t];                       // wave-uniform address: one broadcast load
            float f[9] = {q.x, q.y, q.z, q.w, q.x - mx, q.y - my, q.z - mz, q.x - cx, q.y - cy};
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) y = fmaf(f[j], w[j], y);
            y = fmaxf(fmaf(y, sc, sh), 0.f);
            best = fmaxf(best, y);
        }
        if (live) out[(size_t)p * C + c] = cvt_out<OT>(best);
    }
}

// ---- block filter ---------------------------------------------------------------------------------
__device__ __forceinline__ int f2ord(float f) {  // monotonic float -> int map for atomicMin/Max
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct BfParams {
    int T, F, batch, bx, by, block_factor, block_size;
    float thr, high;
};

__global__ __launch_bounds__(kBlock) void k_bf_init(int *__restrict__ mins, int *__restrict__ maxs, long long n) {
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    mins[g] = f2ord(99999999.0f);
    maxs[g] = f2ord(-99999999.0f);
}

__global__ __launch_bounds__(kBlock) void k_bf_minmax(const float *__restrict__ voxels, const int *__restrict__ coors,
                                                     const int *__restrict__ num_points, const int *__restrict__ voff,
                                                     BfParams p, int *__restrict__ mins, int *__restrict__ maxs) {
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    int v = (int)(g / p.T), t = (int)(g % p.T);
    if (v >= voff[p.batch] || t >= num_points[v]) return;
    int4 c = *reinterpret_cast<const int4 *>(coors + (size_t)v * 4);
    int cell = (c.x * p.by + c.z / p.block_factor) * p.bx + c.w / p.block_factor;
    float z = voxels[((size_t)v * p.T + t) * p.F + 2];
    atomicMin(&mins[cell], f2ord(z));
    atomicMax(&maxs[cell], f2ord(z));
}

__global__ __launch_bounds__(kBlock) void k_bf_mask(const int *__restrict__ coors, const int *__restrict__ voff, BfParams p,
                                                   const int *__restrict__ mins, const int *__restrict__ maxs, int rows,
                                                   int *__restrict__ keep) {
    int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= rows) return;
    if (v >= voff[p.batch]) { keep[v] = 0; return; }
    int4 c = *reinterpret_cast<const int4 *>(coors + (size_t)v * 4);
    int cy = c.z / p.block_factor, cx = c.w / p.block_factor;
    int y0 = max(cy - p.block_size / 2, 0), y1 = min(cy + p.block_size - p.block_size / 2, p.by);
    int x0 = max(cx - p.block_size / 2, 0), x1 = min(cx + p.block_size - p.block_size / 2, p.bx);
    float hmin = 99999999.0f, hmax = -99999999.0f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            int cell = (c.x * p.by + y) * p.bx + x;
            hmin = fminf(hmin, ord2f(mins[cell]));
            hmax = fmaxf(hmax, ord2f(maxs[cell]));
        }
    float span = __fsub_rn(hmax, hmin);
    keep[v] = (span > p.thr && span < p.high) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_bf_compact(const float *__restrict__ voxels, const int *__restrict__ coors,
                                                      const int *__restrict__ num_points, const int *__restrict__ voff,
                                                      const int *__restrict__ keep, const int *__restrict__ pos,
                                                      const int *__restrict__ total, BfParams p, int rows,
                                                      float *__restrict__ ov, int *__restrict__ oc, int *__restrict__ on,
                                                      int *__restrict__ ooff) {
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    int per = p.T * p.F;
    int v = (int)(g / per), e = (int)(g % per);
    if (g <= p.batch) {  // new per-cloud offsets: kept voxels before the cloud's first voxel
        int o = voff[g];
        ooff[g] = o < rows ? pos[o] : *total;
    }
    if (v >= voff[p.batch] || !keep[v]) return;
    int d = pos[v];
    ov[(size_t)d * per + e] = voxels[(size_t)v * per + e];
    if (e < 4) oc[(size_t)d * 4 + e] = coors[(size_t)v * 4 + e];
    if (e == 0) on[d] = num_points[v];
}

struct BfWorkspace {
    int *mins, *maxs, *keep, *pos, *scan, *total;
    size_t bytes;
};
static BfWorkspace carve_bf(void *ws, size_t cap, int rows, int batch, int bx, int by) {
    BfWorkspace w;
    Arena a(ws, cap);
    size_t cells = (size_t)batch * bx * by;
    w.mins = a.take<int>(cells);
    w.maxs = a.take<int>(cells);
    w.keep = a.take<int>(rows > 0 ? rows : 1);
    w.pos = a.take<int>(rows > 0 ? rows : 1);
    w.scan = a.take<int>(scan_scratch_ints(rows));
    w.total = a.take<int>(1);
    w.bytes = align_up(a.used);
    return w;
}

}  // namespace sec

using namespace sec;

SEC_API int sec_pfn_fwd(const float *voxels, const int *num_points, const int *coords, int num_pillars, const int *num_dev,
                        int max_points, int num_features, const float *weight_t, const float *scale, const float *shift,
                        int channels, float vx, float vy, float x_offset, float y_offset, void *out, int out_dtype,
                        void *stream) {
    if (num_pillars < 0 || max_points <= 0 || channels <= 0 || !weight_t || !scale || !shift || !out) return SEC_E_INVALID;
    if (num_features != 4) return SEC_E_UNSUPPORTED;  // x, y, z + one extra (reflectance / dt): every shipped config
    if (num_pillars == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(div_up(num_pillars, kBlock / 64)), block(kBlock);
#define SEC_PFN(OT) hipLaunchKernelGGL(k_pfn_fwd<OT>, grid, block, 0, st, voxels, num_points, coords, num_pillars, num_dev, \
                                       max_points, weight_t, scale, shift, channels, vx, vy, x_offset, y_offset, (OT *)out)
    if (out_dtype == SEC_F32) SEC_PFN(float);
    else if (out_dtype == SEC_BF16) SEC_PFN(__hip_bfloat16);
    else if (out_dtype == SEC_F16) SEC_PFN(__half);
    else return SEC_E_UNSUPPORTED;
#undef SEC_PFN
    return check_launch();
}

SEC_API size_t sec_block_filter_workspace_bytes(int rows, int batch, int grid_x, int grid_y, int block_factor) {
    if (rows < 0 || batch <= 0 || block_factor <= 0) return 0;
    return carve_bf(nullptr, 0, rows, batch, div_up(grid_x, block_factor), div_up(grid_y, block_factor)).bytes;
}

SEC_API int sec_voxel_block_filter_f32(const float *voxels, const int *coors, const int *num_points,
                                       const int *voxel_offsets, int rows, int batch, int max_points, int num_features,
                                       int grid_x, int grid_y, int block_factor, int block_size, float height_threshold,
                                       float height_high_threshold, float *out_voxels, int *out_coors,
                                       int *out_num_points, int *out_offsets, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (rows < 0 || batch <= 0 || max_points <= 0 || num_features < 3 || block_factor <= 0 || block_size <= 0 ||
        !voxel_offsets || !out_offsets)
        return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    BfParams p{max_points, num_features, batch, div_up(grid_x, block_factor), div_up(grid_y, block_factor), block_factor,
               block_size, height_threshold, height_high_threshold};
    BfWorkspace w = carve_bf(workspace, workspace_bytes, rows, batch, p.bx, p.by);
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    long long cells = (long long)batch * p.bx * p.by;
    hipLaunchKernelGGL(k_bf_init, dim3(div_up(cells, kBlock)), dim3(kBlock), 0, st, w.mins, w.maxs, cells);
    int rc;
    if (rows > 0) {
        hipLaunchKernelGGL(k_bf_minmax, dim3(div_up((long long)rows * max_points, kBlock)), dim3(kBlock), 0, st, voxels,
                           coors, num_points, voxel_offsets, p, w.mins, w.maxs);
        hipLaunchKernelGGL(k_bf_mask, dim3(div_up(rows, kBlock)), dim3(kBlock), 0, st, coors, voxel_offsets, p, w.mins,
                           w.maxs, rows, w.keep);
    }
    if ((rc = exclusive_scan_i32(w.keep, w.pos, rows, w.total, w.scan, st))) return rc;
    long long work = (long long)(rows > 0 ? rows : 1) * max_points * num_features;
    if (work < batch + 1) work = batch + 1;
    hipLaunchKernelGGL(k_bf_compact, dim3(div_up(work, kBlock)), dim3(kBlock), 0, st, voxels, coors, num_points,
                       voxel_offsets, w.keep, w.pos, w.total, p, rows, out_voxels, out_coors, out_num_points, out_offsets);
    return check_launch();
}
