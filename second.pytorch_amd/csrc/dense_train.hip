// Training kernels of the dense RPN (second/pytorch/models/rpn.py:468-497 trained by second/pytorch/train.py:316-325), 16-bit
// activations over fp32 master weights (the reference's mixed-precision mode; fp32 training keeps MIOpen, whose Winograd kernel
// out-runs any direct fp32-MFMA form: 64 FLOP/clk/SIMD):
//
//   conv 3x3 / s1 / p1 forward      sec_conv2d_nhwc (k_conv2d_halo_reg), no bias, no ReLU
//   its data gradient               the SAME kernel on dY with the weights flipped and transposed (ops.conv2d_dgrad_weight)
//   its weight gradient             k_conv2d_wgrad3x3 + k_conv2d_wgrad_reduce   (here)
//   BatchNorm2d (batch statistics) + ReLU, forward and backward, channels-last 16-bit   (here)
//
// replacing MIOpen's igemm_wrw / igemm_bwd kernels and ~8 torch kernels per BatchNorm + ReLU pair of the reference path.
#include "common.hpp"
#include <type_traits>

namespace sec {

typedef float tf32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct MfmaT;
template <> struct MfmaT<__hip_bfloat16> {
    static __device__ __forceinline__ tf32x16 run(uint4 a, uint4 b, tf32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tbf16x8, a), __builtin_bit_cast(tbf16x8, b), c, 0, 0, 0);
    }
};
template <> struct MfmaT<__half> {
    static __device__ __forceinline__ tf32x16 run(uint4 a, uint4 b, tf32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tf16x8, a), __builtin_bit_cast(tf16x8, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ float t2f(T v);
template <> __device__ __forceinline__ float t2f(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float t2f(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T f2t(float v);
template <> __device__ __forceinline__ __hip_bfloat16 f2t(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half f2t(float v) { return __float2half_rn(v); }

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / pad 1 convolution, Cin = Cout = 128 (every 3x3 layer of the car.fhd / nuScenes-fhd RPN):
//     dW[tap][ci][co] = sum over pixels p of  X[p + offset(tap)][ci] * dY[p][co]
// a GEMM whose contraction runs over the PIXELS (K = B*H*W = 140 800 at batch 4) and whose result is tiny (9 x 128 x 128), so the
// work is split over (pixel range, tap): blockIdx.y = tap, blockIdx.x = a run of `steps` 64-pixel steps.  Both MFMA operands need 8
// consecutive pixels per lane for one channel -- the transpose of channels-last memory -- so every step stages X^T and dY^T in LDS
// ([channel][pixel], 144-byte pitch: conflict-free ds_read_b128): a thread loads the same 8 channels of 4 consecutive pixels (four
// 16-byte loads), transposes the 4 x 8 block in registers (v_perm_b32) and stores eight 8-byte runs.  Double-buffered: the loads of
// step s + 1 are in flight while the 64 MFMAs of step s run; one barrier per step.  A wave owns a 32-row ci tile x all 128 co
// (4 accumulators).  Every workgroup writes its 128 x 128 fp32 partial to the workspace; k_conv2d_wgrad_reduce sums the partials of a
// tap in a fixed order (deterministic -- no float atomics) into torch's [Cout][Cin][3][3] layout.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv2d_wgrad3x3(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part,
                                                            int B, int H, int W, int steps, long long P, int slices, int ntaps) {
    constexpr int C = 128, LD = 72;                        // LD: 64 pixels + 8 pad (144-byte rows)
    __shared__ __attribute__((aligned(16))) T sX[2][C][LD];
    __shared__ __attribute__((aligned(16))) T sD[2][C][LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // The nine taps of a pixel slice read the same 2 x 650 KB of x and dy: workgroup id = (group of 8 slices) * 72 + tap * 8 + xcd puts
    // them on ONE XCD (workgroups go to XCD id % 8) and next to each other in dispatch order, so eight of the nine reads are hits
    // in that XCD's L2 instead of nine trips to the fabric.
    // (ntaps == 1: a 1x1 convolution -- the centre tap alone.)
    const int per = ntaps * 8;
    const int within = blockIdx.x % per, slice = (blockIdx.x / per) * 8 + within % 8;
    const int tslot = within / 8, tap = ntaps == 1 ? 4 : tslot, ty = tap / 3 - 1, tx = tap % 3 - 1;
    if (slice >= slices) return;
    const long long step0 = (long long)slice * steps;
    const long long total_steps = (P + 63) / 64;
    const int nst = (int)(step0 + steps <= total_steps ? steps : (total_steps > step0 ? total_steps - step0 : 0));
    // this thread stages channels chg*8.. of pixels pg*4..pg*4+3 of a step.  The PIXEL group runs fastest across lanes: the sixteen
    // lanes of a ds_write_b64 group then store 128 contiguous bytes of one channel row (conflict-free); with the channel group fastest
    // (fully coalesced 256-byte global reads) all sixteen hit ONE bank -- rows are 8 x 144 bytes apart -- and the kernel ran at 175 us
    // instead of ~50.  The global reads are 64-byte segments this way (four channel groups per wave and pixel); the other waves read the
    // rest of the same lines.
    const int pg = tid & 15, chg = tid >> 4;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x), *d4 = reinterpret_cast<const uint4 *>(dy);
    const long long HW = (long long)H * W;
    uint4 rx[4], rd[4];
    auto fetch = [&](int s) {
        const long long q0 = (step0 + s) * 64 + pg * 4;
        const int b = (int)(q0 / HW);
        const int rem = (int)(q0 - (long long)b * HW);
        int yy = rem / W, xx = rem - yy * W, bb = b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long q = q0 + i;
            const bool okp = q < P;
            rd[i] = okp ? d4[q * (C / 8) + chg] : make_uint4(0, 0, 0, 0);
            const int sy = yy + ty, sx = xx + tx;
            const bool okx = okp && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            rx[i] = okx ? x4[(((long long)bb * H + sy) * W + sx) * (C / 8) + chg] : make_uint4(0, 0, 0, 0);
            if (++xx == W) { xx = 0; if (++yy == H) { yy = 0; ++bb; } }
        }
    };
    auto put = [&](int buf) {
        // 4 pixels x 8 channels -> 8 channels x 4 pixels: dword j of pixel i holds channels 2j, 2j+1
        const unsigned *px[4] = {reinterpret_cast<const unsigned *>(&rx[0]), reinterpret_cast<const unsigned *>(&rx[1]),
                                 reinterpret_cast<const unsigned *>(&rx[2]), reinterpret_cast<const unsigned *>(&rx[3])};
        const unsigned *pd[4] = {reinterpret_cast<const unsigned *>(&rd[0]), reinterpret_cast<const unsigned *>(&rd[1]),
                                 reinterpret_cast<const unsigned *>(&rd[2]), reinterpret_cast<const unsigned *>(&rd[3])};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // __builtin_amdgcn_perm(hi, lo, sel): bytes 0-3 of lo, 4-7 of hi
            const uint2 xe = make_uint2(__builtin_amdgcn_perm(px[1][j], px[0][j], 0x05040100u), __builtin_amdgcn_perm(px[3][j], px[2][j], 0x05040100u));
            const uint2 xo = make_uint2(__builtin_amdgcn_perm(px[1][j], px[0][j], 0x07060302u), __builtin_amdgcn_perm(px[3][j], px[2][j], 0x07060302u));
            *reinterpret_cast<uint2 *>(&sX[buf][chg * 8 + 2 * j][pg * 4]) = xe;
            *reinterpret_cast<uint2 *>(&sX[buf][chg * 8 + 2 * j + 1][pg * 4]) = xo;
            const uint2 de = make_uint2(__builtin_amdgcn_perm(pd[1][j], pd[0][j], 0x05040100u), __builtin_amdgcn_perm(pd[3][j], pd[2][j], 0x05040100u));
            const uint2 dO = make_uint2(__builtin_amdgcn_perm(pd[1][j], pd[0][j], 0x07060302u), __builtin_amdgcn_perm(pd[3][j], pd[2][j], 0x07060302u));
            *reinterpret_cast<uint2 *>(&sD[buf][chg * 8 + 2 * j][pg * 4]) = de;
            *reinterpret_cast<uint2 *>(&sD[buf][chg * 8 + 2 * j + 1][pg * 4]) = dO;
        }
    };
    tf32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    if (nst > 0) {
        fetch(0);
        put(0);
        __syncthreads();
        for (int s = 0; s < nst; ++s) {
            const int buf = s & 1;
            if (s + 1 < nst) fetch(s + 1);                  // in flight during this step's MFMAs
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4 *>(&sX[buf][wv * 32 + r][ks * 16 + h * 8]);     // A[ci][8 pixels]
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint4 bq = *reinterpret_cast<const uint4 *>(&sD[buf][t * 32 + r][ks * 16 + h * 8]);   // B[8 pixels][co]
                    acc[t] = MfmaT<T>::run(a, bq, acc[t]);
                }
            }
            if (s + 1 < nst) put(buf ^ 1);                  // the other buffer was last read before the previous barrier
            __syncthreads();
        }
    }
    // D layout: column (co) = lane & 31, rows (ci) = (i & 3) + 8 (i >> 2) + 4 h
    float *dst = part + ((size_t)slice * ntaps + tslot) * C * C;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = wv * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, co = t * 32 + r;
            dst[(size_t)ci * C + co] = acc[t][i];
        }
}

// dw[co][ci][tap] (torch's [Cout][Cin][3][3]) = sum_g part[g][tap][ci][co], g ascending (fixed order: run-to-run identical)
__global__ __launch_bounds__(256) void k_conv2d_wgrad_reduce(const float *__restrict__ part, int groups, float *__restrict__ dw, int ntaps) {
    constexpr int C = 128;
    const int e = blockIdx.x * 256 + threadIdx.x;           // e = (tap * C + ci) * C + co
    if (e >= ntaps * C * C) return;
    float s = 0.0f;
    for (int g = 0; g < groups; ++g) s += part[(size_t)g * ntaps * C * C + e];
    const int co = e % C, ci = (e / C) % C, tap = e / (C * C);
    dw[((size_t)co * C + ci) * ntaps + tap] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d with BATCH statistics + ReLU on a channels-last 16-bit activation [P pixels][C] (rpn.py:486-497 in training mode:
// nn.BatchNorm2d(eps 1e-3, momentum 0.01) followed by nn.ReLU).  Forward: partial (sum, sum of squares) per workgroup -> one finalize
// workgroup (fixed order; also updates running_mean / running_var with torch's unbiased-variance convention) -> one normalise + ReLU
// pass.  Backward: with g = dz * [z > 0], xh = (y - mean) * invstd:  dbeta = sum g, dgamma = sum g * xh,
// dy = gamma * invstd * (g - dbeta / P - xh * dgamma / P): one partial-sum pass, one finalize, one apply pass.
// Thread layout: a thread owns 8 channels (one 16-byte chunk) of every (256 / (C/8))-th pixel.
// static-capacity training (DeviceTrainer with a captured step): the activation holds `P` rows of CAPACITY, the first *p_dev are
// live; statistics, normalisation and the backward sums run over the live rows only (NULL: all P rows)
__device__ __forceinline__ long long bn_live_rows(long long P, const int *p_dev) {
    if (!p_dev) return P;
    const long long live = *p_dev;
    return live < P ? (live > 0 ? live : 0) : P;
}
template <typename T, int MODE>     // MODE 0: (sum y, sum y^2);  MODE 1: (sum g, sum g * xh) with mean / invstd / gamma / beta given
__global__ __launch_bounds__(256) void k_bn_partial(const T *__restrict__ y, const T *__restrict__ dz, long long P, int C,
                                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                   float *__restrict__ part, const int *__restrict__ p_dev) {
    __shared__ float red[2][256][8 + 1];
    P = bn_live_rows(P, p_dev);
    const int cg = C / 8, tid = threadIdx.x;
    const int chg = tid % cg, pl = tid / cg, ppi = 256 / cg;      // pixels per iteration of this workgroup
    float a[8], b2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = b2[j] = 0.0f;
    float mu[8], is[8], ga[8], be[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chg * 8 + j;
            mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
        }
    }
    if (pl < ppi) {
        for (long long p = (long long)blockIdx.x * ppi + pl; p < P; p += (long long)gridDim.x * ppi) {
            const uint4 v = reinterpret_cast<const uint4 *>(y)[p * cg + chg];
            const T *e = reinterpret_cast<const T *>(&v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = t2f<T>(e[j]);
                    a[j] += f;
                    b2[j] += f * f;
                }
            } else {
                const uint4 gq = reinterpret_cast<const uint4 *>(dz)[p * cg + chg];
                const T *ge = reinterpret_cast<const T *>(&gq);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (t2f<T>(e[j]) - mu[j]) * is[j];
                    float g = t2f<T>(ge[j]);
                    if (relu && !(xh * ga[j] + be[j] > 0.0f)) g = 0.0f;
                    a[j] += g;
                    b2[j] += g * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][tid][j] = a[j]; red[1][tid][j] = b2[j]; }
    __syncthreads();
    // channel c = chg * 8 + j is held by the threads {pl * cg + chg}: thread c < C sums them in ascending pl (fixed order)
    if (tid < C) {
        const int oc = tid / 8, oj = tid % 8;
        float s0 = 0.0f, s1 = 0.0f;
        for (int q = 0; q < ppi; ++q) { s0 += red[0][q * cg + oc][oj]; s1 += red[1][q * cg + oc][oj]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + tid] = s0;
        part[((size_t)blockIdx.x * 2 + 1) * C + tid] = s1;
    }
}

// fixed-order sum of one channel's `groups` partials by ONE wave: lane l adds partials l, l + 64, ... then a shuffle tree (the same
// order on every run).  (First cut: one THREAD per channel walking 512 strided partials -- 130 us per finalize.)
__device__ __forceinline__ void wave_sum2(const float *__restrict__ part, int groups, int C, int c, double &s0, double &s1) {
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int g = lane; g < groups; g += 64) { a += part[((size_t)g * 2 + 0) * C + c]; b += part[((size_t)g * 2 + 1) * C + c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    s0 = a; s1 = b;
}

// forward finalize: mean, invstd (biased variance, eps), running statistics (momentum; unbiased variance as torch does).
// One wave per channel (256-thread workgroups: four channels each).
__global__ __launch_bounds__(256) void k_bn_fwd_finalize(const float *__restrict__ part, int groups, int C, long long P, float eps,
                                                        float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                                                        float *__restrict__ running_mean, float *__restrict__ running_var,
                                                        const int *__restrict__ p_dev) {
    P = bn_live_rows(P, p_dev);
    if (P < 1) P = 1;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s0, s1;                                          // doubles: E[y^2] - E[y]^2 over 1e5 pixels cancels badly in fp32
    wave_sum2(part, groups, C, c, s0, s1);
    if ((threadIdx.x & 63) != 0) return;
    const double m = s0 / (double)P;
    double var = s1 / (double)P - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(P > 1 ? var * (double)P / (double)(P - 1) : var);
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float *__restrict__ part, int groups, int C, float *__restrict__ dbeta,
                                                        float *__restrict__ dgamma) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s0, s1;
    wave_sum2(part, groups, C, c, s0, s1);
    if ((threadIdx.x & 63) != 0) return;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
}

// MODE 0: z = act((y - mean) * invstd * gamma + beta);  MODE 1: dy = gamma * invstd * (g - dbeta / P - xh * dgamma / P)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_bn_apply(const T *__restrict__ y, const T *__restrict__ dz, long long P, int C,
                                                 const float *__restrict__ mean, const float *__restrict__ invstd,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 const float *__restrict__ dbeta, const float *__restrict__ dgamma, int relu,
                                                 T *__restrict__ out, const int *__restrict__ p_dev) {
    P = bn_live_rows(P, p_dev);                        // rows past the live count are neither read nor written
    const int cg = C / 8;
    const long long n = P * cg;
    const float inv_p = 1.0f / (float)(P > 0 ? P : 1);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long long)gridDim.x * 256) {
        const int chg = (int)(q % cg);
        const uint4 v = reinterpret_cast<const uint4 *>(y)[q];
        const T *e = reinterpret_cast<const T *>(&v);
        uint4 gq = make_uint4(0, 0, 0, 0);
        if (MODE == 1) gq = reinterpret_cast<const uint4 *>(dz)[q];
        const T *ge = reinterpret_cast<const T *>(&gq);
        uint4 o;
        T *oe = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chg * 8 + j;
            const float xh = (t2f<T>(e[j]) - mean[c]) * invstd[c];
            const float zp = xh * gamma[c] + beta[c];
            if (MODE == 0) {
                oe[j] = f2t<T>(relu ? (zp > 0.0f ? zp : 0.0f) : zp);
            } else {
                float g = t2f<T>(ge[j]);
                if (relu && !(zp > 0.0f)) g = 0.0f;
                oe[j] = f2t<T>(gamma[c] * invstd[c] * (g - dbeta[c] * inv_p - xh * dgamma[c] * inv_p));
            }
        }
        reinterpret_cast<uint4 *>(out)[q] = o;
    }
}

// fp32 master weight [cout][cin][k][k] -> BOTH 16-bit MFMA slab images of a training step in one launch: `fwd` in the layout of
// k_conv2d_pack (packed[((tap * cin/8 + chunk) * cout + n) * 8 + e] = w[n][chunk*8 + e][tap]) and `dgrad` = the same layout of the
// flipped, transposed kernel W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx] (cin and cout swap roles): what the data gradient's
// forward-kernel launch reads.  Replaces to(dtype) + pack + flip + transpose + contiguous + to(dtype) + pack (7 launches).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv2d_pack_train(const float *__restrict__ w, int cout, int cin, int ks, T *__restrict__ fwd,
                                                             T *__restrict__ dgrad) {
    const long long total = (long long)ks * ks * cin * cout;
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    if (g < 8) {                                            // the 16-byte zero block behind each image
        fwd[total + g] = f2t<T>(0.0f);
        dgrad[total + g] = f2t<T>(0.0f);
    }
    const int e = (int)(g & 7);
    long long q = g >> 3;
    {   // forward image: n = output channel, chunk over input channels
        const int n = (int)(q % cout);
        const long long q2 = q / cout;
        const int chunk = (int)(q2 % (cin / 8)), tap = (int)(q2 / (cin / 8));
        fwd[g] = f2t<T>(w[(((size_t)n * cin + chunk * 8 + e) * ks + tap / ks) * ks + tap % ks]);
    }
    {   // dgrad image: n = INPUT channel (the dgrad conv's output), chunk over OUTPUT channels, taps mirrored
        const int n = (int)(q % cin);
        const long long q2 = q / cin;
        const int chunk = (int)(q2 % (cout / 8)), tap = (int)(q2 / (cout / 8));
        const int ky = ks - 1 - tap / ks, kx = ks - 1 - tap % ks;
        dgrad[g] = f2t<T>(w[(((size_t)(chunk * 8 + e) * cin + n) * ks + ky) * ks + kx]);
    }
}

constexpr int kBnGroups = 512;

template <typename T>
static int run_wgrad(const void *x, const void *dy, int B, int H, int W, float *dw, void *ws, size_t ws_bytes, hipStream_t st, int ntaps) {
    const long long P = (long long)B * H * W;
    const long long total_steps = (P + 63) / 64;
    int steps = ntaps == 1 ? 4 : 40;                         // ~2 workgroups per CU at batch 4 (2200 steps -> 55 runs x 9 taps; one tap: 256 runs)
    long long gx = (total_steps + steps - 1) / steps;
    if (gx > 256) { steps = (int)((total_steps + 255) / 256); gx = (total_steps + steps - 1) / steps; }
    if (gx < 1) gx = 1;
    const size_t need = (size_t)gx * ntaps * 128 * 128 * sizeof(float);
    if (ws_bytes < need) return SEC_E_WORKSPACE;
    hipLaunchKernelGGL((k_conv2d_wgrad3x3<T>), dim3((unsigned)((gx + 7) / 8 * 8 * ntaps)), dim3(256), 0, st, (const T *)x, (const T *)dy, (float *)ws, B, H, W,
                       steps, P, (int)gx, ntaps);
    hipLaunchKernelGGL(k_conv2d_wgrad_reduce, dim3(div_up(ntaps * 128 * 128, 256)), dim3(256), 0, st, (const float *)ws, (int)gx, dw, ntaps);
    return check_launch();
}

}  // namespace sec

using namespace sec;

SEC_API size_t sec_conv2d_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int ksize) {
    if (cin != 128 || cout != 128 || (ksize != 3 && ksize != 1) || batch <= 0 || h <= 0 || w <= 0) return 0;
    const long long total_steps = ((long long)batch * h * w + 63) / 64;
    const int steps0 = ksize == 1 ? 4 : 40;
    long long gx = (total_steps + steps0 - 1) / steps0;
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    return (size_t)gx * ksize * ksize * 128 * 128 * sizeof(float);
}

SEC_API int sec_conv2d_wgrad_nhwc(const void *x, const void *dy, int batch, int h, int w, int cin, int cout, int ksize, int stride,
                                  int pad, float *dweight, void *workspace, size_t workspace_bytes, int dtype, void *stream) {
    if (!x || !dy || !dweight || !workspace || batch <= 0 || h <= 0 || w <= 0) return SEC_E_INVALID;
    if (cin != 128 || cout != 128 || stride != 1 || !((ksize == 3 && pad == 1) || (ksize == 1 && pad == 0)) ||
        (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16) return run_wgrad<__hip_bfloat16>(x, dy, batch, h, w, dweight, workspace, workspace_bytes, st, ksize * ksize);
    return run_wgrad<__half>(x, dy, batch, h, w, dweight, workspace, workspace_bytes, st, ksize * ksize);
}

SEC_API int sec_conv2d_pack_weight_train(const float *weight, int cout, int cin, int ksize, int dtype, void *packed_fwd,
                                         void *packed_dgrad, void *stream) {
    if (!weight || !packed_fwd || !packed_dgrad || cout <= 0 || cin <= 0 || ksize <= 0) return SEC_E_INVALID;
    if (cin % 64 || cout % 64 || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)ksize * ksize * cin * cout;
    // (the kernel also writes the 16-byte zero block sec_conv2d_packed_weight_bytes reserves behind each image)
    if (dtype == SEC_BF16)
        hipLaunchKernelGGL((k_conv2d_pack_train<__hip_bfloat16>), dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, weight, cout, cin, ksize,
                           (__hip_bfloat16 *)packed_fwd, (__hip_bfloat16 *)packed_dgrad);
    else
        hipLaunchKernelGGL((k_conv2d_pack_train<__half>), dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, weight, cout, cin, ksize,
                           (__half *)packed_fwd, (__half *)packed_dgrad);
    return check_launch();
}

SEC_API size_t sec_bn_train_workspace_bytes(int channels) {
    if (channels <= 0 || channels % 8 || channels > 256) return 0;
    return (size_t)kBnGroups * 2 * channels * sizeof(float);
}

SEC_API int sec_bn_relu_fwd_nhwc(const void *y, long long pixels, int channels, const float *gamma, const float *beta, float eps,
                                 float momentum, float *running_mean, float *running_var, int relu, void *z, float *save_mean,
                                 float *save_invstd, void *workspace, size_t workspace_bytes, int dtype, const int *pixels_dev,
                                 void *stream) {
    if (!y || !z || !gamma || !beta || !save_mean || !save_invstd || !workspace || pixels <= 0) return SEC_E_INVALID;
    if (channels <= 0 || channels % 8 || channels > 256 || 256 % (channels / 8) || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    if (workspace_bytes < sec_bn_train_workspace_bytes(channels)) return SEC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const int blocks = (int)((pixels * (channels / 8) + 255) / 256 < 2048 ? (pixels * (channels / 8) + 255) / 256 : 2048);
#define SEC_BN_FWD(T)                                                                                                              \
    hipLaunchKernelGGL((k_bn_partial<T, 0>), dim3(kBnGroups), dim3(256), 0, st, (const T *)y, (const T *)nullptr, pixels, channels, \
                       nullptr, nullptr, nullptr, nullptr, 0, part, pixels_dev);                                                   \
    hipLaunchKernelGGL(k_bn_fwd_finalize, dim3(div_up(channels, 4)), dim3(256), 0, st, part, kBnGroups, channels, pixels, eps,    \
                       momentum, save_mean, save_invstd, running_mean, running_var, pixels_dev);                                   \
    hipLaunchKernelGGL((k_bn_apply<T, 0>), dim3(blocks), dim3(256), 0, st, (const T *)y, (const T *)nullptr, pixels, channels,      \
                       save_mean, save_invstd, gamma, beta, nullptr, nullptr, relu, (T *)z, pixels_dev);
    if (dtype == SEC_BF16) { SEC_BN_FWD(__hip_bfloat16) } else { SEC_BN_FWD(__half) }
#undef SEC_BN_FWD
    return check_launch();
}

SEC_API int sec_bn_relu_bwd_nhwc(const void *dz, const void *y, long long pixels, int channels, const float *gamma, const float *beta,
                                 const float *save_mean, const float *save_invstd, int relu, void *dy, float *dgamma, float *dbeta,
                                 void *workspace, size_t workspace_bytes, int dtype, const int *pixels_dev, void *stream) {
    if (!dz || !y || !dy || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !workspace || pixels <= 0) return SEC_E_INVALID;
    if (channels <= 0 || channels % 8 || channels > 256 || 256 % (channels / 8) || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    if (workspace_bytes < sec_bn_train_workspace_bytes(channels)) return SEC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    const int blocks = (int)((pixels * (channels / 8) + 255) / 256 < 2048 ? (pixels * (channels / 8) + 255) / 256 : 2048);
#define SEC_BN_BWD(T)                                                                                                              \
    hipLaunchKernelGGL((k_bn_partial<T, 1>), dim3(kBnGroups), dim3(256), 0, st, (const T *)y, (const T *)dz, pixels, channels,      \
                       save_mean, save_invstd, gamma, beta, relu, part, pixels_dev);                                               \
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(div_up(channels, 4)), dim3(256), 0, st, part, kBnGroups, channels, dbeta, dgamma); \
    hipLaunchKernelGGL((k_bn_apply<T, 1>), dim3(blocks), dim3(256), 0, st, (const T *)y, (const T *)dz, pixels, channels, save_mean, \
                       save_invstd, gamma, beta, dbeta, dgamma, relu, (T *)dy, pixels_dev);
    if (dtype == SEC_BF16) { SEC_BN_BWD(__hip_bfloat16) } else { SEC_BN_BWD(__half) }
#undef SEC_BN_BWD
    return check_launch();
}
