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
// SLOTS: `voxels` is the caller's flat POINT array and `slots` the voxeliser's point lists (slots[p * T + t] = index of pillar p's
// t-th point, vox_slots_of): the pillar's points are fetched through the list instead of from a [P, T, 4] tensor -- the same
// values in the same positions (slots >= n read as zero), so the result is bit-identical, and the voxeliser need not write (nor
// this kernel re-read) that tensor: 98 MB each way at 100 k pillars x 60 slots.
template <typename OT, bool SLOTS = false>
__global__ __launch_bounds__(kBlock) void k_pfn_fwd(const float *__restrict__ voxels, const int *__restrict__ num_points,
                                                   const int *__restrict__ coords, int P, const int *__restrict__ num_dev,
                                                   int T, const float *__restrict__ wt, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, int C, float vx, float vy, float xo,
                                                   float yo, OT *__restrict__ out, const int *__restrict__ slots = nullptr) {
    if (num_dev) P = *num_dev < P ? *num_dev : P;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= P) return;
    const float4 *pv = reinterpret_cast<const float4 *>(voxels) + (SLOTS ? (size_t)0 : (size_t)p * T);
    const int *sl = SLOTS ? slots + (size_t)p * T : nullptr;
    const int n = num_points[p];
    const int4 co = *reinterpret_cast<const int4 *>(coords + (size_t)p * 4);
    const float cx = __fadd_rn(__fmul_rn((float)co.w, vx), xo), cy = __fadd_rn(__fmul_rn((float)co.z, vy), yo);
    // pillar mean over all T slots (padded slots are zero) / n
    float sx = 0.f, sy = 0.f, sz = 0.f;
    float4 mine = make_float4(0, 0, 0, 0);            // T <= 64: lane t keeps point t for the channel loop below
    for (int t0 = 0; t0 < T; t0 += 64) {
        float4 q = make_float4(0, 0, 0, 0);
        if (t0 + lane < T) {
            if (SLOTS) { if (t0 + lane < n) q = pv[sl[t0 + lane]]; }
            else q = pv[t0 + lane];
        }
        if (t0 == 0) mine = q;
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
            // SLOTS, T <= 64 (every shipped config): the point sits in lane t's registers since the mean pass -- four v_readlane
            // instead of a doubly dependent broadcast load (slot index, then point) per point: 105 -> 78 us at 100 k pillars
            float4 q;
            if (SLOTS && T <= 64) {                       // (the tensor form is 4 us FASTER with its plain broadcast loads: 68 vs 73 us)
                q.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), t));
                q.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), t));
                q.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.z), t));
                q.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.w), t));
            } else {
                q = SLOTS ? pv[sl[t]] : pv[t];            // wave-uniform address: one broadcast load
            }
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

// ---- PillarFeatureNet in TRAINING mode: batch statistics, forward and backward -------------------------------------------
// Reference: PFNLayer.forward (pointpillars.py:51-65) under train(): x = Linear(9, C, bias=False)(decorated, masked points);
// BatchNorm1d over ALL P * T rows (the zero rows of the padded slots included), ReLU, max over the T slots; autograd then runs
// ~25 kernels over a [P, T, C] tensor (1.1 GB in fp32 at 75 k pillars x 60 points x 64 channels) forward and again backward.
// Here the [P, T, C] tensor never exists: lane = channel, a wave walks the points of a pillar.
//   k_pfn_stats     per channel: sum x, sum x^2, M[c][j] = sum x_c * dec_j, and S1[j] = sum dec_j (block partials)
//   k_pfn_finalize  mean, 1 / sqrt(var + eps) (biased variance over P * T rows, as BatchNorm does), running statistics
//   k_pfn_fwd<.., TRAIN>  normalise, ReLU, max over the slots, slot of the maximum (-1: a padded slot won)
//   k_pfn_bwd       the gradient of the max reaches ONE row per (pillar, channel): dz = g * [out > 0]; block partials of
//                   dbeta = sum dz, dgamma = sum dz * xhat, A[c][j] = sum dz * dec_j over those rows
//   k_pfn_bwd_finalize  BatchNorm's backward spreads dbeta / dgamma over every row (dx = gamma * invstd * (dz - dbeta / N -
//                   xhat * dgamma / N)); summed against the inputs that is closed form in S1 and M:
//                   dW[c][j] = gamma_c invstd_c (A[c][j] - dbeta_c / N * S1[j] - dgamma_c / N * invstd_c (M[c][j] - mean_c S1[j]))
// No gradient with respect to the points (they are data).
constexpr int kPfnIn = 9;
constexpr int kPfnStatVals = 2 + kPfnIn;          // per channel: sum x, sum x^2, M[c][0..8]
constexpr int kPfnBwdVals = 2 + kPfnIn;           // per channel: dbeta, dgamma, A[c][0..8]
constexpr int kPfnMaxBlocks = 1024;

struct PfnPillar {
    float mx, my, mz, cx, cy;
    int n;
};
// the pillar constants exactly as k_pfn_fwd derives them (mean over all T slots / n, pillar centre from its coordinates)
__device__ __forceinline__ PfnPillar pfn_pillar(const float4 *__restrict__ pv, int T, int n, const int *__restrict__ coords, int p,
                                                float vx, float vy, float xo, float yo, int lane) {
    const int4 co = *reinterpret_cast<const int4 *>(coords + (size_t)p * 4);
    PfnPillar r;
    r.cx = __fadd_rn(__fmul_rn((float)co.w, vx), xo);
    r.cy = __fadd_rn(__fmul_rn((float)co.z, vy), yo);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int t0 = 0; t0 < T; t0 += 64) {
        float4 q = (t0 + lane < T) ? pv[t0 + lane] : make_float4(0, 0, 0, 0);
        sx += q.x; sy += q.y; sz += q.z;
    }
    const float inv = (float)n;
    r.mx = wave_sum(sx) / inv; r.my = wave_sum(sy) / inv; r.mz = wave_sum(sz) / inv;
    r.n = n < T ? n : T;
    return r;
}
__device__ __forceinline__ void pfn_decorate(const float4 q, const PfnPillar &pl, float (&f)[kPfnIn]) {
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    f[4] = q.x - pl.mx; f[5] = q.y - pl.my; f[6] = q.z - pl.mz;
    f[7] = q.x - pl.cx; f[8] = q.y - pl.cy;
}

// partial[block][c][kPfnStatVals], partial_s1[block][kPfnIn]; C <= 64 (one wave = all channels)
__global__ __launch_bounds__(kBlock) void k_pfn_stats(const float *__restrict__ voxels, const int *__restrict__ num_points,
                                                     const int *__restrict__ coords, int P, int T, const float *__restrict__ wt, int C,
                                                     float vx, float vy, float xo, float yo, float *__restrict__ partial,
                                                     float *__restrict__ partial_s1) {
    __shared__ float red[kBlock / 64][64][kPfnStatVals + 1];
    __shared__ float red1[kBlock / 64][kPfnIn];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = lane < C;
    float w[kPfnIn];
#pragma unroll
    for (int j = 0; j < kPfnIn; ++j) w[j] = live ? wt[(size_t)j * C + lane] : 0.f;
    float acc[kPfnStatVals], s1[kPfnIn];
#pragma unroll
    for (int j = 0; j < kPfnStatVals; ++j) acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < kPfnIn; ++j) s1[j] = 0.f;
    for (int p = blockIdx.x * (kBlock / 64) + wv; p < P; p += gridDim.x * (kBlock / 64)) {
        const float4 *pv = reinterpret_cast<const float4 *>(voxels) + (size_t)p * T;
        const PfnPillar pl = pfn_pillar(pv, T, num_points[p], coords, p, vx, vy, xo, yo, lane);
        for (int t = 0; t < pl.n; ++t) {
            float f[kPfnIn];
            pfn_decorate(pv[t], pl, f);
            float x = 0.f;
#pragma unroll
            for (int j = 0; j < kPfnIn; ++j) x = fmaf(f[j], w[j], x);
            acc[0] += x;
            acc[1] = fmaf(x, x, acc[1]);
#pragma unroll
            for (int j = 0; j < kPfnIn; ++j) { acc[2 + j] = fmaf(x, f[j], acc[2 + j]); s1[j] += f[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < kPfnStatVals; ++j) red[wv][lane][j] = acc[j];
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < kPfnIn; ++j) red1[wv][j] = s1[j];
    __syncthreads();
    if (wv == 0) {
        if (live)
#pragma unroll
            for (int j = 0; j < kPfnStatVals; ++j) {
                float v = 0.f;
                for (int k = 0; k < kBlock / 64; ++k) v += red[k][lane][j];
                partial[((size_t)blockIdx.x * C + lane) * kPfnStatVals + j] = v;
            }
        if (lane < kPfnIn) {
            float v = 0.f;
            for (int k = 0; k < kBlock / 64; ++k) v += red1[k][lane];
            partial_s1[(size_t)blockIdx.x * kPfnIn + lane] = v;
        }
    }
}

__device__ __forceinline__ double block_sum_f64(double v, double *sh) {   // 256 threads
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int k = 0; k < kBlock / 64; ++k) r += sh[k];
    __syncthreads();
    return r;
}

// one block per channel (+ one for S1).  stats[c] layout: mean[C], invstd[C], M[C][9], S1[9]
__global__ __launch_bounds__(kBlock) void k_pfn_finalize(const float *__restrict__ partial, const float *__restrict__ partial_s1,
                                                        int blocks, int C, double rows, float eps, float momentum,
                                                        float *__restrict__ running_mean, float *__restrict__ running_var,
                                                        float *__restrict__ stats) {
    __shared__ double sh[kBlock / 64];
    const int c = blockIdx.x;
    if (c == C) {                                     // S1
        for (int j = 0; j < kPfnIn; ++j) {
            double v = 0.0;
            for (int b = threadIdx.x; b < blocks; b += kBlock) v += (double)partial_s1[(size_t)b * kPfnIn + j];
            v = block_sum_f64(v, sh);
            if (threadIdx.x == 0) stats[(size_t)C * (2 + kPfnIn) + j] = (float)v;
        }
        return;
    }
    double tot[kPfnStatVals];
    for (int j = 0; j < kPfnStatVals; ++j) {
        double v = 0.0;
        for (int b = threadIdx.x; b < blocks; b += kBlock) v += (double)partial[((size_t)b * C + c) * kPfnStatVals + j];
        tot[j] = block_sum_f64(v, sh);
    }
    if (threadIdx.x == 0) {
        const double mean = tot[0] / rows;
        double var = tot[1] / rows - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[c] = (float)mean;
        stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        for (int j = 0; j < kPfnIn; ++j) stats[(size_t)2 * C + (size_t)c * kPfnIn + j] = (float)tot[2 + j];
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (rows > 1.0 ? rows / (rows - 1.0) : 1.0));
    }
}

// normalise with the batch statistics, ReLU, max over the slots; argmax[p][c] = slot of the maximum (first one; -1 = a padded slot)
__global__ __launch_bounds__(kBlock) void k_pfn_fwd_train(const float *__restrict__ voxels, const int *__restrict__ num_points,
                                                         const int *__restrict__ coords, int P, int T, const float *__restrict__ wt,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         const float *__restrict__ stats, int C, float vx, float vy, float xo,
                                                         float yo, float *__restrict__ out, signed char *__restrict__ argmax) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= P) return;
    const bool live = lane < C;
    float w[kPfnIn];
#pragma unroll
    for (int j = 0; j < kPfnIn; ++j) w[j] = live ? wt[(size_t)j * C + lane] : 0.f;
    const float mean = live ? stats[lane] : 0.f, invstd = live ? stats[C + lane] : 0.f;
    const float ga = live ? gamma[lane] : 0.f, be = live ? beta[lane] : 0.f;
    const float4 *pv = reinterpret_cast<const float4 *>(voxels) + (size_t)p * T;
    const PfnPillar pl = pfn_pillar(pv, T, num_points[p], coords, p, vx, vy, xo, yo, lane);
    float best = -INFINITY;
    int arg = -1;
    for (int t = 0; t < pl.n; ++t) {
        float f[kPfnIn];
        pfn_decorate(pv[t], pl, f);
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < kPfnIn; ++j) x = fmaf(f[j], w[j], x);
        const float y = fmaxf(fmaf((x - mean) * invstd, ga, be), 0.f);
        if (y > best) { best = y; arg = t; }
    }
    if (pl.n < T) {                                   // padded slots: x = 0
        const float y = fmaxf(fmaf((0.f - mean) * invstd, ga, be), 0.f);
        if (y > best) { best = y; arg = -1; }
    }
    if (live) {
        out[(size_t)p * C + lane] = best;
        argmax[(size_t)p * C + lane] = (signed char)arg;
    }
}

// partial[block][c][kPfnBwdVals] = (dbeta, dgamma, A[c][0..8])
__global__ __launch_bounds__(kBlock) void k_pfn_bwd(const float *__restrict__ voxels, const int *__restrict__ num_points,
                                                   const int *__restrict__ coords, int P, int T, const float *__restrict__ wt,
                                                   const float *__restrict__ stats, int C, float vx, float vy, float xo, float yo,
                                                   const float *__restrict__ grad_out, const float *__restrict__ out,
                                                   const signed char *__restrict__ argmax, float *__restrict__ partial) {
    __shared__ float red[kBlock / 64][64][kPfnBwdVals + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = lane < C;
    float w[kPfnIn];
#pragma unroll
    for (int j = 0; j < kPfnIn; ++j) w[j] = live ? wt[(size_t)j * C + lane] : 0.f;
    const float mean = live ? stats[lane] : 0.f, invstd = live ? stats[C + lane] : 0.f;
    float acc[kPfnBwdVals];
#pragma unroll
    for (int j = 0; j < kPfnBwdVals; ++j) acc[j] = 0.f;
    for (int p = blockIdx.x * (kBlock / 64) + wv; p < P; p += gridDim.x * (kBlock / 64)) {
        const float4 *pv = reinterpret_cast<const float4 *>(voxels) + (size_t)p * T;
        const PfnPillar pl = pfn_pillar(pv, T, num_points[p], coords, p, vx, vy, xo, yo, lane);
        if (!live) continue;
        const float g = grad_out[(size_t)p * C + lane];
        const float dz = out[(size_t)p * C + lane] > 0.f ? g : 0.f;
        const int t = argmax[(size_t)p * C + lane];
        float x = 0.f;
        float f[kPfnIn];
#pragma unroll
        for (int j = 0; j < kPfnIn; ++j) f[j] = 0.f;
        if (t >= 0) {
            pfn_decorate(pv[t], pl, f);
#pragma unroll
            for (int j = 0; j < kPfnIn; ++j) x = fmaf(f[j], w[j], x);
        }
        acc[0] += dz;
        acc[1] = fmaf(dz, (x - mean) * invstd, acc[1]);
#pragma unroll
        for (int j = 0; j < kPfnIn; ++j) acc[2 + j] = fmaf(dz, f[j], acc[2 + j]);
    }
#pragma unroll
    for (int j = 0; j < kPfnBwdVals; ++j) red[wv][lane][j] = acc[j];
    __syncthreads();
    if (wv == 0 && live)
#pragma unroll
        for (int j = 0; j < kPfnBwdVals; ++j) {
            float v = 0.f;
            for (int k = 0; k < kBlock / 64; ++k) v += red[k][lane][j];
            partial[((size_t)blockIdx.x * C + lane) * kPfnBwdVals + j] = v;
        }
}

// one block per channel: dweight_t[j][c], dgamma[c], dbeta[c]
__global__ __launch_bounds__(kBlock) void k_pfn_bwd_finalize(const float *__restrict__ partial, int blocks, int C, double rows,
                                                            const float *__restrict__ gamma, const float *__restrict__ stats,
                                                            float *__restrict__ dweight_t, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta) {
    __shared__ double sh[kBlock / 64];
    const int c = blockIdx.x;
    double tot[kPfnBwdVals];
    for (int j = 0; j < kPfnBwdVals; ++j) {
        double v = 0.0;
        for (int b = threadIdx.x; b < blocks; b += kBlock) v += (double)partial[((size_t)b * C + c) * kPfnBwdVals + j];
        tot[j] = block_sum_f64(v, sh);
    }
    if (threadIdx.x == 0) {
        const double mean = stats[c], invstd = stats[C + c], ga = gamma[c];
        const double db = tot[0], dg = tot[1];
        dbeta[c] = (float)db;
        dgamma[c] = (float)dg;
        for (int j = 0; j < kPfnIn; ++j) {
            const double s1 = stats[(size_t)C * (2 + kPfnIn) + j], m = stats[(size_t)2 * C + (size_t)c * kPfnIn + j];
            dweight_t[(size_t)j * C + c] = (float)(ga * invstd * (tot[2 + j] - db / rows * s1 - dg / rows * invstd * (m - mean * s1)));
        }
    }
}

static int pfn_blocks(int P) {
    int b = div_up(P > 0 ? P : 1, kBlock / 64);
    return b > kPfnMaxBlocks ? kPfnMaxBlocks : b;
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

// The same test with one WAVE per voxel: lane l takes cell l of the (block_size x block_size <= 64-cell) window, the wave reduces
// min / max with shuffles.  The thread-per-voxel form above walks its 64 cells x 2 arrays itself -- 128 loads on one thread's
// dependency chain: 214 us for the 360 k voxels of a nuscenes/all.fhd batch; min and max do not depend on the order, so the result
// is bit-identical.
__global__ __launch_bounds__(kBlock) void k_bf_mask_wave(const int *__restrict__ coors, const int *__restrict__ voff, BfParams p,
                                                        const int *__restrict__ mins, const int *__restrict__ maxs, int rows,
                                                        int *__restrict__ keep) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (v >= rows) return;
    if (v >= voff[p.batch]) { if (lane == 0) keep[v] = 0; return; }
    const int4 c = *reinterpret_cast<const int4 *>(coors + (size_t)v * 4);
    const int cy = c.z / p.block_factor, cx = c.w / p.block_factor;
    const int y0 = max(cy - p.block_size / 2, 0), y1 = min(cy + p.block_size - p.block_size / 2, p.by);
    const int x0 = max(cx - p.block_size / 2, 0), x1 = min(cx + p.block_size - p.block_size / 2, p.bx);
    float hmin = 99999999.0f, hmax = -99999999.0f;
    const int wx = x1 - x0, cells = wx * (y1 - y0);
    for (int e = lane; e < cells; e += 64) {
        const int y = y0 + e / wx, x = x0 + e % wx;
        const int cell = (c.x * p.by + y) * p.bx + x;
        hmin = fminf(hmin, ord2f(mins[cell]));
        hmax = fmaxf(hmax, ord2f(maxs[cell]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        hmin = fminf(hmin, __shfl_xor(hmin, d, 64));
        hmax = fmaxf(hmax, __shfl_xor(hmax, d, 64));
    }
    if (lane == 0) {
        const float span = __fsub_rn(hmax, hmin);
        keep[v] = (span > p.thr && span < p.high) ? 1 : 0;
    }
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

SEC_API int sec_pfn_fwd_slots(const float *points, const void *vox_workspace, size_t vox_workspace_bytes, int vox_num_points,
                              int vox_batch, int vox_max_voxels, int max_points, int num_features, const int *num_points_per_voxel,
                              const int *coords, int num_pillars, const int *num_dev, const float *weight_t, const float *scale,
                              const float *shift, int channels, float vx, float vy, float x_offset, float y_offset, void *out,
                              int out_dtype, void *stream) {
    if (num_pillars < 0 || max_points <= 0 || channels <= 0 || !points || !vox_workspace || !num_points_per_voxel || !coords ||
        !weight_t || !scale || !shift || !out)
        return SEC_E_INVALID;
    if (num_features != 4) return SEC_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(points) & 15) != 0) return SEC_E_UNSUPPORTED;      // float4 gathers
    const int *count, *slots;
    if (!vox_slots_of(vox_workspace, vox_workspace_bytes, vox_num_points, vox_batch, vox_max_voxels, max_points, &count, &slots))
        return SEC_E_WORKSPACE;
    if (num_pillars == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(div_up(num_pillars, kBlock / 64)), block(kBlock);
#define SEC_PFN(OT) hipLaunchKernelGGL((k_pfn_fwd<OT, true>), grid, block, 0, st, points, num_points_per_voxel, coords, num_pillars, num_dev, \
                                       max_points, weight_t, scale, shift, channels, vx, vy, x_offset, y_offset, (OT *)out, slots)
    if (out_dtype == SEC_F32) SEC_PFN(float);
    else if (out_dtype == SEC_BF16) SEC_PFN(__hip_bfloat16);
    else if (out_dtype == SEC_F16) SEC_PFN(__half);
    else return SEC_E_UNSUPPORTED;
#undef SEC_PFN
    return check_launch();
}

SEC_API size_t sec_pfn_train_workspace_bytes(int num_pillars, int channels) {
    if (num_pillars < 0 || channels <= 0 || channels > 64) return 0;
    const size_t b = (size_t)pfn_blocks(num_pillars);
    return align_up(b * channels * kPfnStatVals * sizeof(float)) + align_up(b * kPfnIn * sizeof(float));
}

SEC_API int sec_pfn_train_fwd(const float *voxels, const int *num_points, const int *coords, int num_pillars, int max_points,
                              int num_features, const float *weight_t, const float *gamma, const float *beta, float eps,
                              float momentum, float *running_mean, float *running_var, int channels, float vx, float vy,
                              float x_offset, float y_offset, float *out, signed char *argmax, float *stats, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (num_pillars < 0 || max_points <= 0 || max_points > 127 || channels <= 0 || !weight_t || !gamma || !beta || !out || !argmax || !stats)
        return SEC_E_INVALID;
    if (num_features != 4 || channels > 64) return SEC_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < sec_pfn_train_workspace_bytes(num_pillars, channels)) return SEC_E_WORKSPACE;
    if (num_pillars == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = pfn_blocks(num_pillars);
    float *partial = (float *)workspace;
    float *partial_s1 = (float *)((char *)workspace + align_up((size_t)blocks * channels * kPfnStatVals * sizeof(float)));
    hipLaunchKernelGGL(k_pfn_stats, dim3(blocks), dim3(kBlock), 0, st, voxels, num_points, coords, num_pillars, max_points, weight_t,
                       channels, vx, vy, x_offset, y_offset, partial, partial_s1);
    const double rows = (double)num_pillars * (double)max_points;
    hipLaunchKernelGGL(k_pfn_finalize, dim3(channels + 1), dim3(kBlock), 0, st, partial, partial_s1, blocks, channels, rows, eps,
                       momentum, running_mean, running_var, stats);
    hipLaunchKernelGGL(k_pfn_fwd_train, dim3(div_up(num_pillars, kBlock / 64)), dim3(kBlock), 0, st, voxels, num_points, coords,
                       num_pillars, max_points, weight_t, gamma, beta, stats, channels, vx, vy, x_offset, y_offset, out, argmax);
    return check_launch();
}

SEC_API int sec_pfn_train_bwd(const float *voxels, const int *num_points, const int *coords, int num_pillars, int max_points,
                              int num_features, const float *weight_t, const float *gamma, const float *stats, int channels, float vx,
                              float vy, float x_offset, float y_offset, const float *grad_out, const float *out,
                              const signed char *argmax, float *dweight_t, float *dgamma, float *dbeta, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (num_pillars < 0 || max_points <= 0 || max_points > 127 || channels <= 0 || !weight_t || !gamma || !stats || !grad_out || !out ||
        !argmax || !dweight_t || !dgamma || !dbeta)
        return SEC_E_INVALID;
    if (num_features != 4 || channels > 64) return SEC_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < sec_pfn_train_workspace_bytes(num_pillars, channels)) return SEC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (num_pillars == 0) {
        int rc = fill_words(dweight_t, sizeof(float) * kPfnIn * channels, 0u, st);
        if (!rc) rc = fill_words(dgamma, sizeof(float) * channels, 0u, st);
        if (!rc) rc = fill_words(dbeta, sizeof(float) * channels, 0u, st);
        return rc;
    }
    const int blocks = pfn_blocks(num_pillars);
    float *partial = (float *)workspace;
    hipLaunchKernelGGL(k_pfn_bwd, dim3(blocks), dim3(kBlock), 0, st, voxels, num_points, coords, num_pillars, max_points, weight_t, stats,
                       channels, vx, vy, x_offset, y_offset, grad_out, out, argmax, partial);
    hipLaunchKernelGGL(k_pfn_bwd_finalize, dim3(channels), dim3(kBlock), 0, st, partial, blocks, channels,
                       (double)num_pillars * (double)max_points, gamma, stats, dweight_t, dgamma, dbeta);
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
        // one wave per voxel (wave_form 0: one thread per voxel, the round-2 form)
        static int wave_form = -1;
        if (wave_form < 0) wave_form = 1;
        if (wave_form)
            hipLaunchKernelGGL(k_bf_mask_wave, dim3(div_up(rows, kBlock / 64)), dim3(kBlock), 0, st, coors, voxel_offsets, p, w.mins,
                               w.maxs, rows, w.keep);
        else
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
