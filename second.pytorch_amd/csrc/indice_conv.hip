// indice_conv on gfx950: ONE output-stationary launch per sparse-conv layer
// (reference: spconv.ops.indice_conv / indice_subm_conv -> spconv_ops.h indiceConv, SURVEY A.5, which
//  runs 27 x (gather kernel, cuBLAS GEMM, scatter-add kernel) plus a host sync per layer).
//
//   out[o, :] = sum_k feat[nbr_out[o][k], :] @ W[k]            (fp32 accumulate)
//   y         = relu?( out * scale + shift )                    (folded BatchNorm1d / bias, optional)
//
// No atomics, no scatter read-modify-write, deterministic.  Each wave owns 32*MT output rows and walks the
// K kernel offsets; for bf16/f16 the per-offset [rows x Cin] . [Cin x Cout] product runs on
// v_mfma_f32_32x32x16_{bf16,f16}: the A fragment (8 consecutive channels of one gathered input row = one
// 16-byte load per lane) comes straight from global/L2 -- a lane pair reads a whole 32-byte sector of the
// row, four k-steps cover a 128-byte Cin=64 row -- and the B fragment is one coalesced 16-byte load per lane
// from a weight buffer pre-packed in fragment order (sec_pack_conv_weight).  Rows of the tile that have
// no neighbour at an offset contribute zeros (the rulebook is ~35 % dense); an offset with no neighbour in
// the whole wave is skipped.  The kernel is bounded by the gather (HBM/L2 bytes), not by MFMA:
// see DESIGN.md for the roofline arithmetic.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace sec {

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to(float v) { return v; }
    static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<__half> {
    static __device__ __forceinline__ float to(__half v) { return __half2float(v); }
    static __device__ __forceinline__ __half from(float v) { return __float2half_rn(v); }
};
template <> struct Cvt<__hip_bfloat16> {
    static __device__ __forceinline__ float to(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ __hip_bfloat16 from(float v) { return __float2bfloat16(v); }
};

__device__ __forceinline__ float epilogue(float v, const float *scale, const float *shift, int c, int relu) {
    if (scale) v = __fmul_rn(v, scale[c]);
    if (shift) v = __fadd_rn(v, shift[c]);
    if (relu) v = v > 0.0f ? v : 0.0f;
    return v;
}

// The same epilogue for FOUR consecutive channels c0..c0+3 with the scale / shift entries fetched as one 16-byte load each.
// (Per-element `epilogue(...)` calls compile into one dependent global load + s_waitcnt vmcnt(0) per channel: 64 serialised
// round trips = 11 500 clocks in the row-split kernel's tail before this form replaced them.)  scale / shift: 16-byte aligned.
struct Affine4 { float sc[4], sh[4]; };
__device__ __forceinline__ Affine4 load_affine4(const float *scale, const float *shift, int c0) {
    Affine4 a;
    float4 s4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f), h4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (scale) s4 = *reinterpret_cast<const float4 *>(scale + c0);
    if (shift) h4 = *reinterpret_cast<const float4 *>(shift + c0);
    a.sc[0] = s4.x; a.sc[1] = s4.y; a.sc[2] = s4.z; a.sc[3] = s4.w;
    a.sh[0] = h4.x; a.sh[1] = h4.y; a.sh[2] = h4.z; a.sh[3] = h4.w;
    return a;
}
__device__ __forceinline__ float epilogue_v(float v, float sc, float sh, bool has_scale, bool has_shift, int relu) {
    if (has_scale) v = __fmul_rn(v, sc);
    if (has_shift) v = __fadd_rn(v, sh);
    if (relu) v = v > 0.0f ? v : 0.0f;
    return v;
}

// four consecutive output channels of one row in one store (8 bytes for 16-bit outputs, 16 for fp32)
template <typename OT> __device__ __forceinline__ void store4(OT *p, float a, float b, float c, float d) {
    OT v[4] = {Cvt<OT>::from(a), Cvt<OT>::from(b), Cvt<OT>::from(c), Cvt<OT>::from(d)};
    if constexpr (sizeof(OT) == 2) {
        uint2 u;
        __builtin_memcpy(&u, v, 8);
        *reinterpret_cast<uint2 *>(p) = u;
    } else {
        float4 u;
        __builtin_memcpy(&u, v, 16);
        *reinterpret_cast<float4 *>(p) = u;
    }
}

// ------------------------------------------------------------------ generic VALU path (any Cin/Cout/dtype)
// one thread per (output row, output channel); fp32 fmaf chain in (k, ci) order.
template <typename T, typename OT>
__global__ __launch_bounds__(kBlock) void k_conv_generic(const T *__restrict__ feat, const T *__restrict__ w,
                                                        const int *__restrict__ nbr, int n_out,
                                                        const int *__restrict__ num_out_dev, int cin, int cout,
                                                        int kvol, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, int relu,
                                                        OT *__restrict__ out) {
    if (num_out_dev) n_out = *num_out_dev;
    long long total = (long long)n_out * cout;
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < total; g += (long long)gridDim.x * kBlock) {
        int o = (int)(g / cout), co = (int)(g % cout);
        const int *row = nbr + (size_t)o * kvol;
        float acc = 0.0f;
        for (int k = 0; k < kvol; ++k) {
            int idx = row[k];
            if (idx < 0) continue;
            const T *f = feat + (size_t)idx * cin;
            const T *wk = w + (size_t)k * cin * cout + co;
            for (int ci = 0; ci < cin; ++ci) acc = fmaf(Cvt<T>::to(f[ci]), Cvt<T>::to(wk[(size_t)ci * cout]), acc);
        }
        out[g] = Cvt<OT>::from(epilogue(acc, scale, shift, co, relu));
    }
}

// ------------------------------------------------------------------ MFMA path (bf16 / f16)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mfma;
template <> struct Mfma<__hip_bfloat16> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<__half> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// packed weight element ((((k*KS + s)*NT + t)*64 + lane)*8 + e) = W[k][s*16 + (lane>>5)*8 + e][t*32 + (lane&31)]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_weight(const T *__restrict__ w, int kvol, int cin, int cout,
                                                       T *__restrict__ packed) {
    int ks = cin / 16, nt = (cout + 31) / 32;
    long long total = (long long)kvol * ks * nt * 64 * 8;
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    int e = (int)(g & 7), lane = (int)((g >> 3) & 63);
    long long q = g >> 9;
    int t = (int)(q % nt);
    q /= nt;
    int s = (int)(q % ks);
    int k = (int)(q / ks);
    int ci = s * 16 + (lane >> 5) * 8 + e, co = t * 32 + (lane & 31);
    packed[g] = co < cout ? w[((size_t)k * cin + ci) * cout + co] : Cvt<T>::from(0.0f);
}

template <typename T, typename OT, int CIN, int COUT, int MT>
__global__ __launch_bounds__(kBlock) void k_conv_mfma(const T *__restrict__ feat, const T *__restrict__ packed,
                                                     const int *__restrict__ nbr, int n_out,
                                                     const int *__restrict__ num_out_dev, int kvol,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     int relu, OT *__restrict__ out) {
    constexpr int KS = CIN / 16, NT = (COUT + 31) / 32;
    if (num_out_dev) n_out = *num_out_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long base = ((long long)blockIdx.x * (kBlock / 64) + w) * (32 * MT);
    if (base >= n_out) return;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.0f;

    const int *nrow[MT];
    bool valid[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        long long row = base + m * 32 + r;
        valid[m] = row < n_out;
        nrow[m] = nbr + (size_t)(valid[m] ? row : 0) * kvol;
    }
    const uint4 *wp = reinterpret_cast<const uint4 *>(packed) + lane;

    int idx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) idx[m] = valid[m] ? nrow[m][0] : -1;

    for (int k = 0; k < kvol; ++k) {
        int cur[MT];
        bool any = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            cur[m] = idx[m];
            any |= cur[m] >= 0;
            if (k + 1 < kvol) idx[m] = valid[m] ? nrow[m][k + 1] : -1;  // prefetch next offset's rows
        }
        if (__ballot(any) == 0ull) continue;  // wave-uniform: nobody has a neighbour at this offset
        uint4 a[MT][KS];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint4 *src = reinterpret_cast<const uint4 *>(feat + (size_t)(cur[m] >= 0 ? cur[m] : 0) * CIN) + h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (cur[m] >= 0) v = src[s * 2];  // 16 channels per k-step = two 16-byte pieces (h = 0, 1)
                a[m][s] = v;
            }
        }
        const uint4 *wk = wp + (size_t)k * KS * NT * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint4 b = wk[(s * NT + t) * 64];
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][t] = Mfma<T>::run(a[m][s], b, acc[m][t]);
            }
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int col = t * 32 + r;
            if (col >= COUT) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                long long row = base + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < n_out) out[(size_t)row * COUT + col] = Cvt<OT>::from(epilogue(acc[m][t][i], scale, shift, col, relu));
            }
        }
}

// Split-K variant: the NW waves of a workgroup share ONE 32-row output tile and split the kernel offsets
// (wave w takes offsets w, w+NW, ...).  Versus one-wave-per-tile this puts NW times more waves in flight and
// shortens every wave's serial gather->MFMA chain NW-fold (the layer is latency-bound: ~440 workgroups of a
// batch-8 subm2 layer leave a 256-CU chip at < 2 waves/SIMD otherwise).  Partial accumulators are reduced
// through LDS in the MFMA register layout ([wave][reg][lane]: conflict-free), each wave finishing and
// storing 1/NW of the tile's registers.
// XCD-aware tile order (guide T1): workgroup b runs on XCD b % 8, so give XCD x the CONTIGUOUS tile range
// [x * n/8, (x+1) * n/8): with spatially ordered rows every XCD's L2 then holds one slab of the feature matrix.
__device__ __forceinline__ int xcd_tile(int b, int n, int swz) {
    if (!swz) return b;
    int per = (n + 7) / 8;
    int t = (b % 8) * per + b / 8;
    return t;   // may be >= n for the padded tail: callers bounds-check rows
}

#ifdef SEC_CONV_TIMELINE   // profiling builds only (tools/conv_microbench.py --timeline)
__device__ long long *g_timeline = nullptr;
#define SEC_TL_DECL long long *tl = g_timeline; long long t0 = 0, t1 = 0, t2 = 0, t3 = 0
#define SEC_TL_STAMP(x) do { if (tl) x = clock64(); } while (0)
#else
#define SEC_TL_DECL
#define SEC_TL_STAMP(x) do {} while (0)
#endif

static int conv_swizzle() { return 1; }      // XCD-aware tile order (round-1 A/B settled: on)
static int rows_xcd_order() {                // the same for k_conv_rows_buf (round 6): SEC_CONV_ROWS_XCD=0 keeps the plain order (A/B)
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SEC_CONV_ROWS_XCD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}
  // profiling aid (tools/conv_microbench.py --timeline); null in production

#ifndef SEC_SK_MIN_WAVES
#define SEC_SK_MIN_WAVES 5
#endif
template <typename T, typename OT, int CIN, int COUT, int NW>
__global__ __launch_bounds__(NW * 64, (CIN * COUT > 64 * 64) ? 2 : SEC_SK_MIN_WAVES) void k_conv_mfma_sk(   // wide shapes: registers over occupancy (no spills)
    const T *__restrict__ feat, const T *__restrict__ packed,
                                                        const int *__restrict__ nbr, int n_out,
                                                        const int *__restrict__ num_out_dev, int kvol,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        int relu, OT *__restrict__ out) {
    SEC_TL_DECL;
    SEC_TL_STAMP(t0);
    constexpr int KS = CIN / 16, NT = (COUT + 31) / 32, NREG = NT * 16;
    static_assert(NREG % NW == 0, "registers must split evenly over the waves");
    __shared__ float red[2][NREG][64];
    if (num_out_dev) n_out = *num_out_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long base = (long long)xcd_tile(blockIdx.x, gridDim.x, relu & 0x10000) * 32;
    if (base >= n_out) return;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const long long row = base + r;
    const bool valid = row < n_out;
    const int *nrow = nbr + (size_t)(valid ? row : 0) * kvol;
    const uint4 *wp = reinterpret_cast<const uint4 *>(packed) + lane;

    SEC_TL_STAMP(t1);
    int idx = (valid && w < kvol) ? nrow[w] : -1;
    for (int k = w; k < kvol; k += NW) {
        int cur = idx;
        if (k + NW < kvol) idx = valid ? nrow[k + NW] : -1;  // prefetch this wave's next offset
        if (__ballot(cur >= 0) == 0ull) continue;
        uint4 a[KS];
        const uint4 *src = reinterpret_cast<const uint4 *>(feat + (size_t)(cur >= 0 ? cur : 0) * CIN) + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (cur >= 0) v = src[s * 2];
            a[s] = v;
        }
        const uint4 *wk = wp + (size_t)k * KS * NT * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = Mfma<T>::run(wk[(s * NT + t) * 64], a[s], acc[t]);   // D^T: lane = one row
    }
    relu &= 0xff;
    SEC_TL_STAMP(t2);
    // pairwise LDS reduction in the MFMA register layout (2 slots = 16 KB instead of NW slots: LDS no longer caps
    // the occupancy): waves 1,3 -> 0,2 ; wave 2 -> 0 ; then waves 0..NW-1 each finish 1/NW of the registers
    static_assert(NW == 4, "pairwise reduction is written for 4 waves");
    if (w & 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[w >> 1][t * 16 + i][lane] = acc[t][i];
    }
    __syncthreads();
    if (!(w & 1)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] += red[w >> 1][t * 16 + i][lane];
    }
    __syncthreads();
    if (w == 0 || w == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[w >> 1][t * 16 + i][lane] = acc[t][i];
    }
    __syncthreads();
    SEC_TL_STAMP(t3);
    // transposed accumulators (weights were the first MFMA operand): a lane owns output row base + r and, per group of
    // four registers, four CONSECUTIVE channels t*32 + 8g + 4h + (0..3): one 8-byte store instead of four 2-byte scatters
    constexpr int PER = NREG / NW;
    static_assert(PER % 4 == 0, "registers are finished in groups of four");
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        const int reg0 = w * PER + q * 4;
        const int t = reg0 >> 4, g = (reg0 & 15) >> 2;
        const int c0 = t * 32 + 8 * g + 4 * h;
        if (c0 < COUT) {
            float v[4];
            const Affine4 af = load_affine4(scale, shift, c0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = epilogue_v(red[0][reg0 + j][lane] + red[1][reg0 + j][lane], af.sc[j], af.sh[j], scale != nullptr, shift != nullptr, relu);
            if (valid) store4<OT>(out + (size_t)row * COUT + c0, v[0], v[1], v[2], v[3]);
        }
    }
#ifdef SEC_CONV_TIMELINE
    if (tl && lane == 0) {
        long long *rec = tl + ((size_t)blockIdx.x * NW + w) * 6;
        rec[0] = t0; rec[1] = t1; rec[2] = t2; rec[3] = t3; rec[4] = clock64();
        rec[5] = __builtin_amdgcn_s_getreg((4 << 11) | 20);  // HW_REG_XCC_ID etc. (informative only)
    }
#endif
}

// Split-K + cout-sliced variant: grid.y selects a 32-column slice of the output, so a wave carries 16 accumulator
// registers and 4*KS weight registers per offset instead of NT times that -- more waves per SIMD (the layer is
// latency bound) at the price of gathering the rows once per slice.
template <typename T, typename OT, int CIN, int COUT, int NW>
__global__ __launch_bounds__(NW * 64) void k_conv_mfma_sks(const T *__restrict__ feat, const T *__restrict__ packed,
                                                         const int *__restrict__ nbr, int n_out,
                                                         const int *__restrict__ num_out_dev, int kvol,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         int relu, OT *__restrict__ out) {
    constexpr int KS = CIN / 16, NT = (COUT + 31) / 32;
    __shared__ float red[NW][16][64];
    if (num_out_dev) n_out = *num_out_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.y;
    const long long base = (long long)xcd_tile(blockIdx.x, gridDim.x, 1) * 32;
    if (base >= n_out) return;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const long long row = base + r;
    const bool valid = row < n_out;
    const int *nrow = nbr + (size_t)(valid ? row : 0) * kvol;
    const uint4 *wp = reinterpret_cast<const uint4 *>(packed) + lane;
    int idx = (valid && w < kvol) ? nrow[w] : -1;
    for (int k = w; k < kvol; k += NW) {
        const int cur = idx;
        if (k + NW < kvol) idx = valid ? nrow[k + NW] : -1;
        if (__ballot(cur >= 0) == 0ull) continue;
        uint4 a[KS];
        const uint4 *src = reinterpret_cast<const uint4 *>(feat + (size_t)(cur >= 0 ? cur : 0) * CIN) + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = cur >= 0 ? src[s * 2] : make_uint4(0, 0, 0, 0);
        const uint4 *wk = wp + (size_t)k * KS * NT * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = Mfma<T>::run(wk[(s * NT + t) * 64], a[s], acc);   // D^T: lane = one row
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[i];
    __syncthreads();
    // wave w finishes register group g = w: channels t*32 + 8w + 4h + (0..3) of row base + r, one 8-byte store
    static_assert(NW == 4, "one four-register group per wave");
    {
        const int c0 = t * 32 + 8 * w + 4 * h;
        if (c0 < COUT) {
            float v[4];
            const Affine4 af = load_affine4(scale, shift, c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.0f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) sum += red[ww][w * 4 + j][lane];
                v[j] = epilogue_v(sum, af.sc[j], af.sh[j], scale != nullptr, shift != nullptr, relu);
            }
            if (valid) store4<OT>(out + (size_t)row * COUT + c0, v[0], v[1], v[2], v[3]);
        }
    }
}

// First layer of SpMiddleFHD (Cin = 4, middle.py:146): one thread per output row, all COUT channels in
// registers, the 27 x 4 x COUT weight block broadcast from LDS, 8-byte (4 x bf16) gathers.
template <typename T, typename OT, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_c4(const T *__restrict__ feat, const T *__restrict__ w,
                                                   const int *__restrict__ nbr, int n_out,
                                                   const int *__restrict__ num_out_dev, int kvol,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   int relu, OT *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [kvol][4][COUT]
    if (num_out_dev) n_out = *num_out_dev;
    for (int e = threadIdx.x; e < kvol * 4 * COUT; e += kBlock) wl[e] = Cvt<T>::to(w[e]);
    __syncthreads();
    long long o = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (o >= n_out) return;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
    const int *row = nbr + (size_t)o * kvol;
    for (int k = 0; k < kvol; ++k) {
        int idx = row[k];
        if (idx < 0) continue;
        float f[4];
        if (sizeof(T) == 2) {
            uint2 v = *reinterpret_cast<const uint2 *>(feat + (size_t)idx * 4);
            const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
            for (int c = 0; c < 4; ++c) f[c] = Cvt<T>::to(pv[c]);
        } else {
            float4 v = *reinterpret_cast<const float4 *>(feat + (size_t)idx * 4);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        }
        const float *wk = wl + k * 4 * COUT;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(f[ci], wk[ci * COUT + c], acc[c]);
    }
    OT *dst = out + (size_t)o * COUT;
#pragma unroll
    for (int c = 0; c < COUT; ++c) dst[c] = Cvt<OT>::from(epilogue(acc[c], scale, shift, c, relu));
}

#ifdef SEC_CONV_EXPERIMENTS
#include "../../tools/kernel_experiments/indice_conv_experiments.inc"
#endif



// ------------------------------------------------------------------------------------------------------------------
// Row-split kernels: shared configuration and epilogue.  (The first form, with LDS-staged operands -- SEC_CONV_VARIANT=9 -- now lives in
// tools/kernel_experiments/indice_conv_rows_ab.inc; its design notes follow because k_conv_rows_buf keeps the work split.)
// The split-K kernel above moves 5x more weight bytes than feature bytes through the vector L1 (every 32-row tile
// re-fetches all kvol weight blocks: 380 MB of B against 76 MB of gathered A for the 64->64 SubM layer) and gathers
// rows as 32-byte fragments of 32 different cache lines per instruction.  Here a workgroup owns 128 output rows
// (one 32-row tile per wave), all waves walk the kernel offsets together so ONE copy of W[k] serves 128 rows
// (LDS-DMA, ring of 3), and each wave gathers its neighbour rows with LDS-DMA in full 128-byte lines (8 lanes per
// row, 8 rows per instruction) into a private ring of 3 -- operands are fetched two offsets ahead of their use.
// Ordering is explicit (counted s_waitcnt vmcnt + one LDS barrier per offset); the step body takes __restrict__ LDS
// pointers so the compiler does not put vmcnt(0) in front of every ds_read (see k_conv2d_halo_pipe in dense.hip).
typedef __attribute__((address_space(3))) void *lds_ptr_c;
typedef const __attribute__((address_space(1))) void *glb_ptr_c;
template <int N> __device__ __forceinline__ void cwait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void clds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int CIN, int COUT>
struct RowsCfg {
    static constexpr int KS = CIN / 16, NT = (COUT + 31) / 32;   // COUT = 16: one half-used 32-column tile (packed weights are padded)
    static constexpr int LPR = CIN / 8;            // 16-byte chunks (= DMA lanes) per feature row
    static constexpr int RPI = 64 / LPR;           // rows per DMA instruction
    static constexpr int ND = 32 / RPI;            // DMA instructions per 32-row tile
    static constexpr int BPIECES = KS * NT;        // 1 KB pieces of one packed W[k]
    static constexpr int NBW = BPIECES >= 4 ? BPIECES / 4 : 1;   // pieces copied by each wave
    static constexpr int L = ND + NBW + 1;         // vector-memory ops per wave per offset (A DMAs, B DMAs, index load)
    static constexpr int ASLOT = 32 * LPR, BSLOT = BPIECES * 64;   // uint4 entries
    static constexpr int RING = 3;
};

// two fp32 -> one dword of two 16-bit values, one conversion per PAIR (v_cvt_pk_bf16_f32; Cvt<T>::from per value compiles
// into a convert plus a merge each); round to nearest even like Cvt<T>::from
typedef float f32x2p __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ unsigned pack2_16(float a, float b);
template <> __device__ __forceinline__ unsigned pack2_16<__hip_bfloat16>(float a, float b) {
    typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
    const f32x2p v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2p));
}
template <> __device__ __forceinline__ unsigned pack2_16<__half>(float a, float b) {
    typedef _Float16 f16x2p __attribute__((ext_vector_type(2)));
    const f32x2p v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2p));
}

// epilogue of the row-per-lane (D^T) accumulator layout: lane owns row `r`, channels t*32 + 8g + 4h + (0..3); half-waves
// swap a register group so every lane stores 16-byte runs; fused scale / shift / ReLU
// `aff` = LDS copy of scale[COUT] | shift[COUT] made by rows_stage_affine at kernel start (16-byte reads, all in flight together)
template <typename T, int COUT, bool FUSED>
__device__ __forceinline__ void rows_store_impl(f32x16 (&acc)[(COUT + 31) / 32], T *__restrict__ out, long long row, bool valid, int h,
                                                const float *__restrict__ aff, bool has_scale, bool has_shift, int relu) {
    if (FUSED) { has_scale = has_shift = true; relu = 1; }   // the inference layers: straight-line multiply, add, max
    constexpr int NT = (COUT + 31) / 32, NG = COUT >= 32 ? 4 : COUT / 8;   // register groups of 4 that hold real channels
    static_assert(COUT % 16 == 0, "whole 16-byte runs per half-wave pair");
    T *orow = out + (size_t)row * COUT;
    float4 sc4[NT][NG], sh4[NT][NG];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = t * 32 + 8 * g + 4 * h;
            if (has_scale) sc4[t][g] = *reinterpret_cast<const float4 *>(aff + c);
            if (has_shift) sh4[t][g] = *reinterpret_cast<const float4 *>(aff + COUT + c);
        }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        unsigned lo[NG], hi[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float sc[4] = {sc4[t][g].x, sc4[t][g].y, sc4[t][g].z, sc4[t][g].w};
            const float sh[4] = {sh4[t][g].x, sh4[t][g].y, sh4[t][g].z, sh4[t][g].w};
            float v4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = epilogue_v(acc[t][4 * g + j], sc[j], sh[j], has_scale, has_shift, relu);
            lo[g] = pack2_16<T>(v4[0], v4[1]);
            hi[g] = pack2_16<T>(v4[2], v4[3]);
        }
        // half-wave exchange = v_permlane32_swap of the two group registers: afterwards the low half-wave holds both halves of
        // 16-byte run 2pr, the high half-wave of run 2pr + 1 (ds_bpermute + selects before)
#pragma unroll
        for (int pr = 0; pr < NG / 2; ++pr) {
            const auto sx = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
            if (valid) *reinterpret_cast<uint4 *>(orow + t * 32 + 8 * (2 * pr + h)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    }
}
template <typename T, int COUT>
__device__ __forceinline__ void rows_store(f32x16 (&acc)[(COUT + 31) / 32], T *__restrict__ out, long long row, bool valid, int h,
                                           const float *__restrict__ aff, bool has_scale, bool has_shift, int relu) {
    if (has_scale && has_shift && relu) rows_store_impl<T, COUT, true>(acc, out, row, valid, h, aff, true, true, 1);
    else rows_store_impl<T, COUT, false>(acc, out, row, valid, h, aff, has_scale, has_shift, relu);
}

// scale / shift -> LDS, first thing in the kernel (published by the first barrier of the offset loop)
template <int COUT>
__device__ __forceinline__ void rows_stage_affine(float *aff, const float *__restrict__ scale, const float *__restrict__ shift) {
    const int c = threadIdx.x;
    if (c < COUT) {
        if (scale) aff[c] = scale[c];
        if (shift) aff[COUT + c] = shift[c];
    }
}

#ifdef SEC_CONV_TIMELINE
#define SEC_RTL(...) __VA_ARGS__
#else
#define SEC_RTL(...)
#endif

#ifdef SEC_CONV_EXPERIMENTS   // the LDS-DMA and register-direct row-split forms (SEC_CONV_VARIANT 9-15): measured, superseded
#include "../../tools/kernel_experiments/indice_conv_rows_ab.inc"
#endif

// ------------------------------------------------------------------------------------------------------------------
// Row-split kernel, BUFFER-load form (SEC_CONV_VARIANT 16..19).  What the per-wave timeline of the forms above showed
// (profiles/r02_a_timeline_conv_rows.txt): the waves hardly wait -- 7 % of the offset loop -- they are busy ISSUING: ~100
// instructions per offset, a quarter of them v_cndmask that zero the fragments of rows without a neighbour, plus 64-bit
// address arithmetic, and (LDS-DMA forms) an M0 hand-off per DMA.  Here the gathers are raw buffer loads over the feature
// matrix: a row without a neighbour gets an OUT-OF-RANGE offset and the hardware returns zeros for it -- no select, no
// dummy row, no memory access -- and the per-offset byte offsets (32-bit, precomputed once) replace the 64-bit pointers.
// Per offset a wave issues 4 gather loads, its share of W[k] (global -> VGPR -> ds_write, at the same distance as the
// gathers because vmcnt retires in order), 8 ds_read_b128 and 8 MFMAs: ~35 instructions.  WAVES = 8 makes the workgroup 256
// rows, so one copy of W[k] per CU and step instead of two.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBalWgs = 256;                                 // one workgroup per CU of an MI355X (FL bit 11, below)
// FL bit 0: the tile's 32 x KVOL neighbour table arrives as four coalesced 16-byte loads per lane staged through LDS (instead
//           of KVOL strided dword loads per lane, each a 32-cache-line gather: 27 x 8 waves of them kept the CU's address
//           pipe busy for ~5000 clocks before the first offset);
//    bit 1: the B fragments of offset k+1 are read from LDS while offset k's MFMAs run (register double buffer), so a wave's
//           step no longer starts with an exposed LDS round trip.
// Packed fp32 VALU instructions: the whole library is built without them (build.py: they returned wrong results in the rotated-NMS
// clipper while another wave of the CU ran the RPN conv's MFMA loop; the cause was never isolated).  Until round 5 this kernel alone
// was exempted (its only packed operations are the 32 scale / shift instructions of the epilogue, it lost ~10 % without them in
// round 2, and 9000+ stressed replays never differed); with the round-3/4 loop forms the exemption is worth nothing any more
// (profiles/r05_b_packed_fp32_exemption_ab.txt: 191-194 us for the fourteen layers either way), so it is gone.
// -DSEC_PACKED_F32_EXEMPTION restores it for A/B builds.
#if defined(__HIP_DEVICE_COMPILE__) && defined(SEC_PACKED_F32_EXEMPTION)
#define SEC_PACKED_F32_OK __attribute__((target("packed-fp32-ops")))
#else
#define SEC_PACKED_F32_OK
#endif
template <typename T, int CIN, int COUT, int KVOL, int DIST, int WAVES, int MINW, int FL>
SEC_PACKED_F32_OK __global__ __launch_bounds__(WAVES * 64, MINW) void k_conv_rows_buf(const T *__restrict__ feat, long long feat_bytes,
                                                                   const T *__restrict__ packed, const int *__restrict__ nbr,
                                                                   int n_out, const int *__restrict__ num_out_dev,
                                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                                   int relu, T *__restrict__ out) {
    using C = RowsCfg<T, CIN, COUT>;
    constexpr int NBW = (C::BPIECES + WAVES - 1) / WAVES;   // 1 KB weight pieces per wave and offset (narrow layers: duplicated)
    constexpr int ROWB = CIN * (int)sizeof(T);           // bytes per feature row
    constexpr bool STAGE = (FL & 1) != 0, PIPE = (FL & 2) != 0;
    // FL bit 6 (ALLW, narrow layers): W[0..KVOL) of a 16- or 32-channel layer is 14-55 KB -- the whole tensor is copied into LDS
    // once per workgroup and the offset loop runs WITHOUT the per-offset workgroup barrier of the three-slot ring: the eight
    // waves drift apart and hide each other's gather latency.  The staged neighbour tables alias the same LDS (prologue only).
    constexpr bool ALLW = (FL & 64) != 0;
    // FL bit 7 (LAZY): the 27 neighbour offsets of a lane are not held in VGPRs for the whole kernel but re-read from the staged table in
    // LDS when the gather of that offset is issued (one ds_read_b32 + 3 VALU per offset): -25 VGPRs, which is what lets the 64 -> 64
    // kernel fit 168 registers -- three waves per SIMD, or one RPN-conv workgroup beside it on the CU while steps are in flight.
    constexpr bool LAZY = (FL & 128) != 0 && STAGE && !ALLW;
    // FL bit 5 (SKEW, 8-wave workgroups): every step ends in a workgroup barrier, so the two waves of a SIMD leave it together,
    // want the MFMA pipe together, and the loser's later instructions (its next gathers) sit behind its queued MFMAs.  Waves
    // 4..7 therefore issue their gathers BEFORE their MFMAs (one offset later than waves 0..3 would): while one wave of the
    // SIMD multiplies, the other one issues loads, and vice versa.
    constexpr bool SKEW = (FL & 32) != 0 && WAVES == 8 && PIPE;
    const bool early = SKEW && __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) != 0;
    constexpr int TBL16 = 32 * KVOL / 4;                 // 16-byte pieces of one tile's neighbour table
    static_assert(NBW >= 1 && NBW <= 2, "one or two weight pieces per wave");
    static_assert(C::KS <= 4, "up to four 16-channel k-steps per row");
    static_assert(!STAGE || (32 * KVOL) % 4 == 0, "tile table in whole 16-byte pieces");
    static_assert(!ALLW || (STAGE && PIPE), "the all-weights form stages its tables and double-buffers its B fragments");
    // FL bit 9 (WIN3): a six-slot weight ring and ONE workgroup barrier per THREE offsets -- inside a window the waves of a SIMD drift
    // apart, so one's B reads run under the other's MFMAs instead of both leaving every barrier together (the ablations put 18 of the
    // launch's 24 us on that lock step).  W of window g + 1 is loaded when window g starts and stored into the other half of the ring
    // when it ends; every wave has left that half at the barrier that opened window g.
    constexpr bool WIN3 = (FL & 512) != 0;
    static_assert(!WIN3 || (STAGE && !PIPE && !ALLW && KVOL % 3 == 0 && DIST <= 3), "window form: non-pipelined, 3 offsets per window");
    constexpr int LBUF = ALLW ? (KVOL * C::BSLOT > WAVES * TBL16 ? KVOL * C::BSLOT : WAVES * TBL16) : 1;
    __shared__ __attribute__((aligned(16))) uint4 lbuf[LBUF];
    __shared__ __attribute__((aligned(16))) uint4 bring[ALLW ? 1 : (WIN3 ? 6 : 3)][ALLW ? 1 : C::BSLOT];
    __shared__ __attribute__((aligned(16))) float aff[2 * COUT];
    __shared__ __attribute__((aligned(16))) u32x4_t stage[(STAGE && !ALLW) ? WAVES : 1][(STAGE && !ALLW) ? TBL16 : 1];
    rows_stage_affine<COUT>(aff, scale, shift);
    const int n_cap = n_out;                             // rows the table holds (>= the live count of a static-capacity launch)
    if (num_out_dev) n_out = *num_out_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // FL bit 11 (BAL): rows per wave chosen so that the launch fills every CU once.  The subm2 stage of car.fhd at batch 8 is 56 298 rows
    // = 220 workgroups of 256 rows for 256 CUs -- 36 CUs idle while the others each push their 590 KB through a ~13 B/clk/CU fill path,
    // which is what the launch time is made of (round 3: a software-pipelined one-wave-per-SIMD form of the same loop, k_conv_rows_m2,
    // lands on the same 21-22 us).  With 28 rows per wave instead of 32 the same rows make 252 workgroups: every CU busy, 12 % less per
    // CU; the 23 k-row stage goes from 179 four-wave workgroups to 238 of 96 rows.  Lanes r >= rw of the 32-row MFMA tile gather
    // nothing (out-of-range offsets) and store nothing.  rw is a multiple of 4 (16-byte table pieces) computed from the DEVICE-side row
    // count; the host launches max(ceil(capacity / (32 * WAVES)), kBalWgs) workgroups, the surplus ones exit here.
    constexpr bool BAL = (FL & 2048) != 0;
    int rw = 32;
    if constexpr (BAL) {
        const int per_wg = (n_out + kBalWgs - 1) / kBalWgs;
        rw = ((per_wg + WAVES - 1) / WAVES + 3) & ~3;
        rw = rw < 4 ? 4 : (rw > 32 ? 32 : rw);
    }
    // XCD-aware tile order (relu bit 16): workgroup b runs on XCD b % 8, and the rows of a batch are concatenated frame by frame, so
    // giving XCD x the CONTIGUOUS range of tiles [x * per, (x + 1) * per) keeps a frame's feature rows, its slice of the gather table
    // and its output rows in ONE XCD's 4 MB L2 -- with the plain order every XCD pulls every frame's rows through its own L2 (the
    // 16-channel layers fetched 1.6-1.9x their algorithmic bytes from the fabric, L2 hit rate 0.5-0.7: profiles/r05_w_pmc.txt).
    // The host launches a multiple of eight workgroups >= the live tile count; surplus workgroups exit.
    int bid = blockIdx.x;
    if (relu & 0x10000) {
        const int nact = (int)(((long long)n_out + rw * WAVES - 1) / (rw * WAVES));
        const int per = (nact + 7) >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if ((int)(blockIdx.x >> 3) >= per) return;
    }
    relu &= 1;
    if ((long long)bid * (rw * WAVES) >= n_out) return;
    const long long row = (long long)bid * (rw * WAVES) + w * rw + r;
    const bool valid = row < n_out && r < rw;
    SEC_RTL(long long *tl = g_timeline; long long tl0 = 0, tl1 = 0, tl2 = 0; if (tl) tl0 = clock64();)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(feat), 0, (int)feat_bytes, 0x00020000);
    const int piece0 = (w * NBW) % C::BPIECES;           // waves beyond the last piece re-copy one (identical bytes, same slot)
    const u32x4_t *wpv = reinterpret_cast<const u32x4_t *>(packed) + (size_t)piece0 * 64 + lane;
    // byte offset of this lane's first 16-byte chunk of every neighbour row; no neighbour -> beyond the buffer -> zeros
    unsigned off[KVOL];
    if constexpr (STAGE) {
        const long long tile_row0 = (long long)bid * (rw * WAVES) + w * rw;   // (32 rows are staged whatever rw is)
        const long long tbl_bytes = (long long)n_cap * KVOL * 4;
        const int *tile = nbr + tile_row0 * KVOL;
        // bytes of the table from this tile on: <= 0 for the waves of the last workgroup that start beyond the table (a table
        // of exactly n_out rows) -- an empty resource then, never a negative record count (= unbounded reads past the table)
        const long long left = tbl_bytes - tile_row0 * KVOL * 4;
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int *>(left > 0 ? tile : nbr), 0, (int)(left <= 0 ? 0 : (left < 32 * KVOL * 4 ? left : 32 * KVOL * 4)), 0x00020000);
        u32x4_t *stg = ALLW ? reinterpret_cast<u32x4_t *>(lbuf) + w * TBL16 : &stage[w][0];
#pragma unroll
        for (int i = 0; i < (TBL16 + 63) / 64; ++i) {
            const int p16 = i * 64 + lane;
            if (p16 < TBL16) stg[p16] = __builtin_amdgcn_raw_buffer_load_b128(trs, p16 * 16, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        const int *mine = reinterpret_cast<const int *>(stg) + r * KVOL;
        if constexpr (!LAZY) {
#pragma unroll
            for (int k = 0; k < KVOL; ++k) {
                const int t = valid ? mine[k] : -1;
                off[k] = t >= 0 ? (unsigned)t * ROWB + h * 16 : 0x80000000u;
            }
        }
    } else {
        const int *nrow = nbr + (size_t)(valid ? row : 0) * KVOL;
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            const int t = valid ? nrow[k] : -1;
            off[k] = t >= 0 ? (unsigned)t * ROWB + h * 16 : 0x80000000u;
        }
    }
#ifdef SEC_CONV_ABLATIONS   // timing-only forms (wrong results): where does the offset loop's time go?
    if constexpr ((FL & 4) != 0) {          // no gather touches memory
#pragma unroll
        for (int k = 0; k < KVOL; ++k) off[k] = 0x80000000u;
    }
    if constexpr ((FL & 8) != 0) {          // every gather instruction touches 8 cache lines instead of 32
#pragma unroll
        for (int k = 0; k < KVOL; ++k) off[k] = (unsigned)__shfl((int)off[k], lane & ~3, 64);
    }
    if constexpr ((FL & 16) != 0) {         // all gathers hit one line
#pragma unroll
        for (int k = 0; k < KVOL; ++k) off[k] = h * 16;
    }
#endif
    SEC_RTL(if (tl) { cwait_vmcnt<0>(); tl1 = clock64(); })
    f32x16 acc[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    u32x4_t areg[DIST][C::KS];
    if constexpr (ALLW) {
#define SEC_FETCH_A(k)                                                                                                \
    {                                                                                                                 \
        areg[(k) % DIST][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[k], 0, 0);                              \
        if (C::KS > 1) areg[(k) % DIST][1 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[k] + 32, 0, 0);  \
        if (C::KS > 2) areg[(k) % DIST][2 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[k] + 64, 0, 0);  \
        if (C::KS > 3) areg[(k) % DIST][3 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[k] + 96, 0, 0);  \
    }
#pragma unroll
        for (int k = 0; k < DIST && k < KVOL; ++k) SEC_FETCH_A(k)
        __syncthreads();                                     // every wave has turned its staged table into offsets: the LDS is free
        {
            const uint4 *pk = reinterpret_cast<const uint4 *>(packed);
            for (int i = threadIdx.x; i < KVOL * C::BSLOT; i += WAVES * 64) lbuf[i] = pk[i];
        }
        __syncthreads();
        uint4 bf[2][C::KS * C::NT];
#pragma unroll
        for (int i = 0; i < C::KS * C::NT; ++i) bf[0][i] = lbuf[i * 64 + lane];
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            if (k + 1 < KVOL) {
#pragma unroll
                for (int i = 0; i < C::KS * C::NT; ++i) bf[(k + 1) & 1][i] = lbuf[(k + 1) * C::BSLOT + i * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const uint4 a = __builtin_bit_cast(uint4, areg[k % DIST][s]);
#pragma unroll
                for (int t = 0; t < C::NT; ++t) acc[t] = Mfma<T>::run(bf[k & 1][s * C::NT + t], a, acc[t]);   // D^T
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k + DIST < KVOL) SEC_FETCH_A(k + DIST)
        }
#undef SEC_FETCH_A
        SEC_RTL(if (tl) tl2 = clock64();)
        rows_store<T, COUT>(acc, out, row, valid, h, aff, scale != nullptr, shift != nullptr, relu);
        return;
    }
    u32x4_t wr0[DIST], wr1[DIST];
    u32x4_t *bslot = reinterpret_cast<u32x4_t *>(&bring[0][piece0 * 64 + lane]);
    const int *lazy_tbl = reinterpret_cast<const int *>(&stage[(STAGE && !ALLW) ? w : 0][0]) + r * KVOL;
    auto off_of = [&](int k) -> unsigned {
        if constexpr (LAZY) {
            const int t = valid ? lazy_tbl[k] : -1;
#ifdef SEC_CONV_ABLATIONS
            if constexpr ((FL & 4) != 0) return 0x80000000u;                                  // no gather touches memory
            if constexpr ((FL & 16) != 0) return t >= 0 ? (unsigned)(h * 16) : 0x80000000u;   // all gathers hit one row
#endif
            return t >= 0 ? (unsigned)t * ROWB + h * 16 : 0x80000000u;
        } else {
            return off[k];
        }
    };
#define SEC_FETCH(k)                                                                                                  \
    {                                                                                                                 \
        const unsigned o_ = off_of(k);                                                                                \
        wr0[(k) % DIST] = wpv[(size_t)(k) * C::BSLOT];                                                                \
        if (NBW > 1) wr1[(k) % DIST] = wpv[(size_t)(k) * C::BSLOT + 64];                                              \
        areg[(k) % DIST][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_, 0, 0);                                  \
        if (C::KS > 1) areg[(k) % DIST][1 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 32, 0, 0);      \
        if (C::KS > 2) areg[(k) % DIST][2 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 64, 0, 0);      \
        if (C::KS > 3) areg[(k) % DIST][3 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 96, 0, 0);      \
    }
#define SEC_WPUT(k)                                                                                                   \
    {                                                                                                                 \
        bslot[((k) % 3) * C::BSLOT] = wr0[(k) % DIST];                                                                \
        if (NBW > 1) bslot[((k) % 3) * C::BSLOT + 64] = wr1[(k) % DIST];                                              \
    }
#pragma unroll
    for (int k = 0; k < DIST && k < KVOL; ++k) SEC_FETCH(k)
    SEC_RTL(long long ts_wait = 0, ts_comp = 0, ts_issue = 0;)
    if constexpr (WIN3) {
        u32x4_t ww0[3], ww1[3];
#define SEC_GFETCH(k)                                                                                                 \
    {                                                                                                                 \
        const unsigned o_ = off_of(k);                                                                                \
        areg[(k) % DIST][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_, 0, 0);                                  \
        if (C::KS > 1) areg[(k) % DIST][1 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 32, 0, 0);      \
        if (C::KS > 2) areg[(k) % DIST][2 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 64, 0, 0);      \
        if (C::KS > 3) areg[(k) % DIST][3 % C::KS] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_ + 96, 0, 0);      \
    }
#ifdef SEC_CONV_ABLATIONS   // FL bit 12: the weight slices are not fetched (the ring holds whatever the registers do): what does the W stream cost?
#define SEC_WLOAD(g)                                                                                                  \
    {                                                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) {                                                            \
            if constexpr ((FL & 4096) != 0) { ww0[j_] = u32x4_t{(unsigned)lane, 0u, 0u, 0u}; if (NBW > 1) ww1[j_] = ww0[j_]; }  \
            else {                                                                                                    \
            ww0[j_] = wpv[(size_t)(3 * (g) + j_) * C::BSLOT];                                                         \
            if (NBW > 1) ww1[j_] = wpv[(size_t)(3 * (g) + j_) * C::BSLOT + 64];                                       \
            }                                                                                                         \
        }                                                                                                             \
    }
#else
#define SEC_WLOAD(g)                                                                                                  \
    {                                                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) {                                                            \
            ww0[j_] = wpv[(size_t)(3 * (g) + j_) * C::BSLOT];                                                         \
            if (NBW > 1) ww1[j_] = wpv[(size_t)(3 * (g) + j_) * C::BSLOT + 64];                                       \
        }                                                                                                             \
    }
#endif
#define SEC_WSTORE(g)                                                                                                 \
    {                                                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) {                                                            \
            bslot[((((g) & 1) * 3) + j_) * C::BSLOT] = ww0[j_];                                                       \
            if (NBW > 1) bslot[((((g) & 1) * 3) + j_) * C::BSLOT + 64] = ww1[j_];                                     \
        }                                                                                                             \
    }
        SEC_WLOAD(0)
#pragma unroll
        for (int k = 0; k < DIST && k < KVOL; ++k) SEC_GFETCH(k)
        SEC_WSTORE(0)
        if (KVOL > 3) SEC_WLOAD(1)
#pragma unroll
        for (int g = 0; g < KVOL / 3; ++g) {
            __syncthreads();                                 // W of window g is visible; every wave has left the other half of the ring
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = 3 * g + j;
                uint4 bf[C::KS * C::NT];
#pragma unroll
                for (int i = 0; i < C::KS * C::NT; ++i) bf[i] = bring[(g & 1) * 3 + j][i * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < C::KS; ++s2) {
                    const uint4 a = __builtin_bit_cast(uint4, areg[k % DIST][s2]);
#pragma unroll
                    for (int t = 0; t < C::NT; ++t) acc[t] = Mfma<T>::run(bf[s2 * C::NT + t], a, acc[t]);   // D^T
                }
                __builtin_amdgcn_sched_barrier(0);
                if (k + DIST < KVOL) SEC_GFETCH(k + DIST)
            }
            if (g + 1 < KVOL / 3) {
                SEC_WSTORE(g + 1)
                if (g + 2 < KVOL / 3) SEC_WLOAD(g + 2)
            }
        }
#undef SEC_GFETCH
#undef SEC_WLOAD
#undef SEC_WSTORE
    } else if constexpr (!PIPE) {
        SEC_WPUT(0)
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            SEC_RTL(long long s0 = 0, s1 = 0, s2 = 0; if (tl) s0 = clock64();)
            if (k + 1 < KVOL) SEC_WPUT(k + 1)                // W[k+1] leaves the register ring; its LDS slot was last read two barriers ago
            __syncthreads();                                 // W[k] (stored during step k-1) is visible to every wave
            SEC_RTL(if (tl) s1 = clock64();)                 // profiling builds: W store + barrier (the slowest wave's gather wait included)
            uint4 bf[C::KS * C::NT];
#pragma unroll
            for (int i = 0; i < C::KS * C::NT; ++i) bf[i] = bring[k % 3][i * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);               // all B fragments in flight together, then the MFMAs back to back
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const uint4 a = __builtin_bit_cast(uint4, areg[k % DIST][s]);
#pragma unroll
                for (int t = 0; t < C::NT; ++t) acc[t] = Mfma<T>::run(bf[s * C::NT + t], a, acc[t]);   // D^T
            }
            __builtin_amdgcn_sched_barrier(0);
            SEC_RTL(if (tl) s2 = clock64();)                 // ... B reads, this wave's own gather wait, MFMA issue
            if (k + DIST < KVOL) SEC_FETCH(k + DIST)         // into the registers this step just consumed
            SEC_RTL(if (tl) { const long long s3 = clock64(); ts_wait += s1 - s0; ts_comp += s2 - s1; ts_issue += s3 - s2; })
        }
    } else {
        static_assert(!PIPE || DIST >= 3, "W[k+2] must have left the ring before its registers are refilled");
        // B fragments one offset ahead: step k stores W[k+2], the barrier publishes W[k+1], whose fragments are then read
        // while the MFMAs of offset k run on the fragments read during step k-1
        uint4 bf[2][C::KS * C::NT];
        SEC_WPUT(0)
        if (KVOL > 1) SEC_WPUT(1)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C::KS * C::NT; ++i) bf[0][i] = bring[0][i * 64 + lane];
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            if (k + 2 < KVOL) SEC_WPUT(k + 2)                // slot (k+2)%3 held W[k-1]: its reads were issued before the last barrier
            if (k + 1 < KVOL) {
                __syncthreads();                             // W[k+1] (stored during step k-1) is visible
#pragma unroll
                for (int i = 0; i < C::KS * C::NT; ++i) bf[(k + 1) & 1][i] = bring[(k + 1) % 3][i * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SKEW) {                            // waves 4..7 share their SIMDs with waves 0..3: see below
                if (early && k >= 1 && k - 1 + DIST < KVOL) SEC_FETCH(k - 1 + DIST)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const uint4 a = __builtin_bit_cast(uint4, areg[k % DIST][s]);
#pragma unroll
                for (int t = 0; t < C::NT; ++t) acc[t] = Mfma<T>::run(bf[k & 1][s * C::NT + t], a, acc[t]);   // D^T
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(SKEW && early) && k + DIST < KVOL) SEC_FETCH(k + DIST)
        }
    }
#undef SEC_FETCH
#undef SEC_WPUT
    SEC_RTL(if (tl) tl2 = clock64();)
    rows_store<T, COUT>(acc, out, row, valid, h, aff, scale != nullptr, shift != nullptr, relu);
#ifdef SEC_CONV_TIMELINE
    if (tl && lane == 0) {
        long long *rec = tl + ((size_t)blockIdx.x * WAVES + w) * 8;
        rec[0] = tl0; rec[1] = tl1; rec[2] = tl2; rec[3] = clock64(); rec[4] = ts_wait; rec[5] = ts_issue; rec[6] = ts_comp;   // per-step sums: non-pipelined forms only
        rec[7] = __builtin_amdgcn_s_getreg((4 << 11) | 20);
    }
#endif
}

template <typename T, int CIN, int COUT, int DIST, int WAVES, int MINW, int FL, int KVOL = 27>
static void launch_rows_buf(const void *feat, long long n_feat, const void *packed, const int *nbr, int n_out, const int *num_out_dev,
                            const float *scale, const float *shift, int relu, void *out, hipStream_t st) {
    set_last_kernel("k_conv_rows_buf<%s, %d, %d, %d, %d, %d, %d, %d>", dtype_name<T>(), CIN, COUT, KVOL, DIST, WAVES, MINW, FL);
    int blocks = div_up(n_out, 32 * WAVES);
    if ((FL & 2048) != 0 && blocks < kBalWgs) blocks = kBalWgs;      // BAL: the kernel spreads the rows over up to kBalWgs workgroups
    const int xcd = rows_xcd_order();
    if (xcd) blocks = (blocks + 7) / 8 * 8 + 8;                      // the XCD-contiguous order needs ceil(live tiles / 8) workgroups per XCD
    hipLaunchKernelGGL((k_conv_rows_buf<T, CIN, COUT, KVOL, DIST, WAVES, MINW, FL>), dim3(blocks), dim3(WAVES * 64), 0, st,
                       (const T *)feat, n_feat * CIN * (long long)sizeof(T), (const T *)packed, nbr, n_out, num_out_dev, scale, shift,
                       (relu & 1) | (xcd << 16), (T *)out);
}

#ifdef SEC_CONV_EXPERIMENTS   // round-6 A/B form: the offsets of a row tile split over three wave groups (k_conv_rows_ks): measured slower
#include "../../tools/kernel_experiments/indice_conv_rows_r06.inc"
#endif

#ifdef SEC_CONV_EXPERIMENTS   // round-3 A/B forms: two row tiles per wave (k_conv_rows_m2), input planes in LDS windows (k_conv_rows_lds)
#include "../../tools/kernel_experiments/indice_conv_rows_r03.inc"
#endif

// ------------------------------------------------------------------------------------------------------------------
// First layer of SpMiddleFHD (Cin = 4 -> 16, 3x3x3; middle.py:146) on the matrix cores.  The whole receptive field of a
// row is ONE K dimension: 27 neighbours x 4 channels = 108, padded to 112 = seven 16-deep MFMA steps.  Lane (r, h) feeds step s
// with the 8-byte rows of neighbours 4s + 2h and 4s + 2h + 1 of output row r (raw buffer loads: no neighbour -> out-of-range
// offset -> zeros); the 7 weight fragments (7 KB, packed by k_pack_weight_c4 in the matching K order) come straight from L2.
// No LDS operands, no barrier, 14 gathers + 7 MFMAs per 32 rows (the thread-per-row VALU kernel: 1728 FMAs per row).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_weight_c4(const T *__restrict__ w, int cout, T *__restrict__ packed) {
    // element ((s * 64 + lane) * 8 + e) = W[kk][ci][c]:  K = s*16 + (lane>>5)*8 + e, kk = K / 4, ci = K % 4, c = lane & 31
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= 7 * 64 * 8) return;
    const int e = g & 7, lane = (g >> 3) & 63, s_ = g >> 9;
    const int K = s_ * 16 + (lane >> 5) * 8 + e, kk = K >> 2, ci = K & 3, c = lane & 31;
    packed[g] = (kk < 27 && c < cout) ? w[((size_t)kk * 4 + ci) * cout + c] : Cvt<T>::from(0.0f);
}

template <typename T, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_c4_mfma(const T *__restrict__ feat, long long feat_bytes, const T *__restrict__ packed,
                                                        const int *__restrict__ nbr, int n_out, const int *__restrict__ num_out_dev,
                                                        const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                        T *__restrict__ out) {
    constexpr int KVOL = 27, TBL16 = 32 * KVOL / 4;
    __shared__ __attribute__((aligned(16))) u32x4_t stage[4][TBL16];
    __shared__ __attribute__((aligned(16))) float aff[2 * COUT];
    rows_stage_affine<COUT>(aff, scale, shift);
    const int n_cap = n_out;
    if (num_out_dev) n_out = *num_out_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    if (relu & 0x10000) {                                // XCD-contiguous tile order, as in k_conv_rows_buf: a frame's rows stay in one XCD's L2
        const int per = (int)((((long long)n_out + 127) / 128 + 7) >> 3);
        bid = (int)(blockIdx.x >> 3) < per ? (bid & 7) * per + (bid >> 3) : 0x3fffff;      // surplus workgroups: beyond every row
    }
    relu &= 1;
    const long long tile_row0 = (long long)bid * 128 + w * 32;
    const long long row = tile_row0 + r;
    const bool valid = row < n_out;
    if (tile_row0 < n_out) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(feat), 0, (int)feat_bytes, 0x00020000);
        const long long left = ((long long)n_cap - tile_row0) * KVOL * 4;
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int *>(nbr + tile_row0 * KVOL), 0, (int)(left < 32 * KVOL * 4 ? left : 32 * KVOL * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < (TBL16 + 63) / 64; ++i) {
            const int p16 = i * 64 + lane;
            if (p16 < TBL16) stage[w][p16] = __builtin_amdgcn_raw_buffer_load_b128(trs, p16 * 16, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        const int *mine = reinterpret_cast<const int *>(&stage[w][0]) + r * KVOL;
        const u32x4_t *wp = reinterpret_cast<const u32x4_t *>(packed) + lane;
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        u32x4_t bfr[7];
        u32x2_t lo[7], hi[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            bfr[s] = wp[s * 64];
            const int k0 = 4 * s + 2 * h, k1 = k0 + 1;
            const int t0 = valid ? mine[k0] : -1;
            const int t1 = (valid && k1 < KVOL) ? mine[k1 < KVOL ? k1 : 0] : -1;
            lo[s] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, t0 >= 0 ? (unsigned)t0 * 8u : 0x80000000u, 0, 0);
            hi[s] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, t1 >= 0 ? (unsigned)t1 * 8u : 0x80000000u, 0, 0);
        }
        f32x16 acc[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const uint4 a = make_uint4(lo[s].x, lo[s].y, hi[s].x, hi[s].y);
            acc[0] = Mfma<T>::run(__builtin_bit_cast(uint4, bfr[s]), a, acc[0]);   // D^T: lane owns row r
        }
        __syncthreads();                                     // the affine vectors staged above are visible
        rows_store<T, COUT>(acc, out, row, valid, h, aff, scale != nullptr, shift != nullptr, relu);
    } else {
        __syncthreads();
    }
}

static int g_variant_override = -1;     // sec_indice_conv_set_variant (A/B runs and the parity tests of every shipped kernel)
// sec_set_fp32_mode: arithmetic of the fp32 sparse convolutions (forward and data gradient).  0 = SEC_FP32_SPLIT16: operands split
// into bf16 (hi, lo) pairs on the bf16 matrix pipe, 16 significant bits per operand, fp32 accumulation (fast).  1 = SEC_FP32_EXACT:
// v_mfma_f32_32x32x2_f32 / VALU fma -- IEEE fp32 products and accumulation, the arithmetic of the reference's default precision.
static int g_fp32_mode = 0;
static int conv_variant() {
    if (g_variant_override >= 0) return g_variant_override;
    return 1;  // 1 = automatic choice; 0 = one wave per 32-row tile; 8 / 9 / 10.. force one kernel family (sec_indice_conv_set_variant)
}

// kernel ids reported by sec_indice_conv_fwd_plan
enum { PLAN_GENERIC = 0, PLAN_TILED = 1, PLAN_C4 = 2, PLAN_MFMA_WAVE = 3, PLAN_MFMA_SK = 4, PLAN_MFMA_SKS = 5, PLAN_ROWS = 6,
       PLAN_ROWS_COMPACT = 7, PLAN_ROWS_TOUCH = 8, PLAN_ROWS_COMPACT_TOUCH = 9, PLAN_ROWS_REG = 10, PLAN_ROWS_BUF = 11, PLAN_C4_MFMA = 12, PLAN_ROWS_M2 = 13, PLAN_ROWS_LDS = 14, PLAN_ROWS_KS = 15, PLAN_EXPERIMENT = 99 };

// Row count from which the buffer-load row-split kernel replaces split-K in the automatic choice (the
// row-split chain of 27 offsets needs enough workgroups to fill the chip, split-K has a 4x shorter chain per wave)
static int rows_min() {
    return 40000;   // car.fhd batch 8: layers 1-8 (>= 56k rows) gain 25-30 %, the 23k-row layers lose
}
constexpr int kRowsMinSmall = 8192;
// A/B switch of the 64 -> 64 row-split kernel: 0 = default (small footprint + one barrier per three offsets, 142 VGPRs), 1 = the 194-VGPR
// form (prefetch distance 4, double-buffered B fragments, a barrier per offset), 3 = small footprint with a barrier per offset (134 VGPRs),
// 4 = the 128-row workgroups of the mid-size layers with a barrier per offset (198 VGPRs)
static int rows_footprint() {
    return 0;       // rounds 2-3 A/B settled: form 0 (numbers in DESIGN_APPENDIX.md)
}
// 1: the automatic choice would take the two-tiles-per-wave kernel (k_conv_rows_m2, experiment builds) for the 64 -> 64 layers: measured slower
static int m2_auto() { return 0; }
static bool ks_auto() {                      // SEC_CONV_KS=1 (experiment builds): k_conv_rows_ks for the mid-size 64 -> 64 layers (A/B)
    static int v = -1;
    if (v < 0) { const char *e = getenv("SEC_CONV_KS"); v = e ? atoi(e) != 0 : 0; }
    return v != 0;
}
static bool rows_balance() { return true; }   // rows per wave chosen on the device so that a launch fills every CU once (round 3: kept)
static bool buf_shape(int cin, int cout, int kvol) {
    if (kvol == 3) return cin == 64 && cout == 64;
    if (kvol != 27) return false;
    return (cin == 16 && (cout == 16 || cout == 32)) || (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && cout == 64);
}

// which row-split kernel (0 = none) the 16-bit MFMA path takes for this shape under the current variant setting
static int rows_plan(int cin, int cout, int kvol, int n_out, bool same_dtype) {
    if (!same_dtype) return 0;
    const int v = conv_variant();
    if (buf_shape(cin, cout, kvol)) {
#ifdef SEC_CONV_EXPERIMENTS
        if (cin == 64 && cout == 64 && kvol == 27 && v == 46) return PLAN_ROWS_LDS;          // input planes staged in LDS (round 3)
        if (cin == 64 && cout == 64 && kvol == 27 && (v == 41 || v == 42 || v == 43 || (v == 1 && m2_auto() && n_out >= rows_min())))
            return PLAN_ROWS_M2;                                                           // two row tiles per wave (round 3)
#endif
        // offsets split over wave groups (k_conv_rows_ks, round 6, experiment builds): 50 forces it for any row count, SEC_CONV_KS=1 makes
        // it the automatic choice of the mid-size 64 -> 64 layers (measured slower than the four-wave form of k_conv_rows_buf)
#ifdef SEC_CONV_EXPERIMENTS
        if (cin == 64 && cout == 64 && kvol == 27 && (v == 50 || (v >= 71 && v <= 73) || (v == 1 && n_out >= kRowsMinSmall && n_out < rows_min() && ks_auto()))) return PLAN_ROWS_KS;
#endif
        if (v == 22 || (((v >= 16 && v <= 28) || v == 44 || v == 45 || (v >= 60 && v <= 67)) && cin == 64 && cout == 64 && kvol == 27)) return PLAN_ROWS_BUF;
        if (v >= 36 && v <= 40) return PLAN_ROWS_BUF;
        if (v == 1 && n_out >= rows_min()) return PLAN_ROWS_BUF;
        // 64 -> 64, 27 offsets, 8 k .. 40 k rows (the 23 k-row stage of car.fhd at batch 8): four-wave workgroups (128 rows) fill the
        // chip where the eight-wave form leaves CUs idle: 13.2 us vs 14.9 us split-K
        if (v == 1 && cin == 64 && cout == 64 && kvol == 27 && n_out >= kRowsMinSmall) return PLAN_ROWS_BUF;
    }
#ifdef SEC_CONV_EXPERIMENTS
    if (kvol != 27 || !(cin == 64 || cin == 32) || !(cout == 64 || cout == 32)) return 0;
    if (cin == 64) {
        if (v == 10) return PLAN_ROWS_COMPACT;
        if (v == 11) return PLAN_ROWS_TOUCH;
        if (v == 12) return PLAN_ROWS_COMPACT_TOUCH;
        if (v >= 13 && v <= 15 && cout == 64) return PLAN_ROWS_REG;
    }
    if (v == 9) return PLAN_ROWS;
#endif
    return 0;
}

#ifdef SEC_CONV_EXPERIMENTS
template <typename T>
static void launch_rows_lds(const void *feat, long long n_feat, const void *packed, const int *nbr, int n_out, const int *num_out_dev,
                            const float *scale, const float *shift, int relu, void *out, hipStream_t st) {
    set_last_kernel("k_conv_rows_lds<%s>", dtype_name<T>());
    hipLaunchKernelGGL((k_conv_rows_lds<T>), dim3(div_up(n_out, 256)), dim3(256), 0, st, (const T *)feat,
                       n_feat * 64 * (long long)sizeof(T), (const T *)packed, nbr, n_out, num_out_dev, scale, shift, relu, (T *)out);
}

#endif
template <typename T, typename OT, int CIN, int COUT>
static void launch_mfma(const void *feat, long long n_feat, const void *packed, const int *nbr, int n_out, const int *num_out_dev,
                        int kvol, const float *scale, const float *shift, int relu, void *out, hipStream_t st) {
    constexpr int MT = 1;
    if constexpr (std::is_same<T, OT>::value && CIN <= 64 && COUT <= 64 && (COUT >= CIN) ) {
        const int rp = feat ? rows_plan(CIN, COUT, kvol, n_out, true) : 0;
#ifdef SEC_CONV_EXPERIMENTS
        if constexpr (CIN == 64 && COUT == 64) {
            if (rp == PLAN_ROWS_M2 && n_feat * CIN * (long long)sizeof(T) < 0x7fffffffll) {
                launch_rows_m2<T>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                return;
            }
            if (rp == PLAN_ROWS_LDS && n_feat * CIN * (long long)sizeof(T) < 0x7fffffffll) {
                launch_rows_lds<T>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                return;
            }
        }
#endif
#ifdef SEC_CONV_EXPERIMENTS
        if constexpr (CIN == 64 && COUT == 64) {
            if (rp == PLAN_ROWS_KS && n_feat * CIN * (long long)sizeof(T) < 0x7fffffffll) {
#ifdef SEC_CONV_ABLATIONS
                if (conv_variant() == 71) { launch_rows_ks<T, 4, 3, 2>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return; }
                if (conv_variant() == 72) { launch_rows_ks<T, 4, 3, 1>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return; }
                if (conv_variant() == 73) { launch_rows_ks<T, 4, 3, 3>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return; }
#endif
                launch_rows_ks<T, 4, 3>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                return;
            }
        }
#endif
        if (rp == PLAN_ROWS_BUF && n_feat * CIN * (long long)sizeof(T) < 0x7fffffffll) {
#define SEC_BUF(D, W, FLG, KV) launch_rows_buf<T, CIN, COUT, D, W, 2, FLG, KV>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st)
            if (kvol == 3) {
                if constexpr (CIN == 64 && COUT == 64) { SEC_BUF(3, 8, 3, 3); return; }
            } else {
#ifdef SEC_CONV_EXPERIMENTS   // A/B forms of the buffer-load kernel: prefetch distance, 4- or 8-wave workgroups, staging / pipelining off, skew
                if constexpr (CIN <= 32) {     // narrow layers: a gather is 1-2 registers per offset, deeper prefetch is nearly free
                    switch (conv_variant()) {
                    case 36: SEC_BUF(6, 8, 3, 27); return;
                    case 37: SEC_BUF(8, 8, 3, 27); return;
                    case 38: SEC_BUF(12, 8, 3, 27); return;
                    case 39: if constexpr (COUT <= 32) { SEC_BUF(4, 8, 3 + 64, 27); return; } break;
                    case 40: if constexpr (COUT <= 32) { SEC_BUF(6, 8, 3 + 64, 27); return; } break;
                    default: break;
                    }
                }
                if constexpr (CIN == 64 && COUT == 64) {
                    switch (conv_variant()) {
                    case 16: SEC_BUF(4, 4, 0, 27); return;
                    case 17: SEC_BUF(4, 4, 1, 27); return;
                    case 18: SEC_BUF(4, 4, 2, 27); return;
                    case 19: SEC_BUF(4, 4, 3, 27); return;
                    case 20: SEC_BUF(5, 4, 3, 27); return;
                    case 21: SEC_BUF(5, 8, 3, 27); return;
                    case 23: SEC_BUF(6, 4, 3, 27); return;
                    case 27: SEC_BUF(5, 8, 3 + 32, 27); return;
                    case 28: SEC_BUF(6, 8, 3 + 32, 27); return;
#ifdef SEC_CONV_ABLATIONS
                    case 24: SEC_BUF(4, 8, 3 + 4, 27); return;
                    case 25: SEC_BUF(4, 8, 3 + 8, 27); return;
                    case 26: SEC_BUF(4, 8, 3 + 16, 27); return;
                    // 60-65 (round 6): the shipped four-wave (23 k rows) and eight-wave (56 k rows) forms without the W stream (FL 4096), without
                    // gathers that touch memory (FL 4), without both
                    case 60: launch_rows_buf<T, CIN, COUT, 3, 4, 2, 1 + 128 + 512 + 2048 + 4096, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 61: launch_rows_buf<T, CIN, COUT, 3, 4, 2, 1 + 128 + 512 + 2048 + 4, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 62: launch_rows_buf<T, CIN, COUT, 3, 4, 2, 1 + 128 + 512 + 2048 + 4096 + 4, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 63: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 2048 + 4096, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 64: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 2048 + 4, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 65: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 2048 + 4096 + 4, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 66: launch_rows_buf<T, CIN, COUT, 3, 4, 2, 1 + 128 + 512 + 2048, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;   // the plain four-wave form, any row count
                    case 67: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 2048, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;   // the plain eight-wave form, any row count
                    case 44: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 4, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
                    case 45: launch_rows_buf<T, CIN, COUT, 3, 8, 3, 1 + 128 + 512 + 16, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st); return;
#endif
                    default: break;
                    }
                }
#endif
                // FL + 2048 (BAL): rows per wave chosen on the device so that the launch fills every CU once (SEC_CONV_BAL=0: fixed 32)
                const bool bal = rows_balance();
                // 16- and 32-channel layers: the whole weight tensor lives in LDS, no per-offset barrier (-13 .. -24 % per layer)
                if constexpr (CIN <= 32 && COUT <= 32) { if (bal) { SEC_BUF(6, 8, 3 + 64 + 2048, 27); } else { SEC_BUF(6, 8, 3 + 64, 27); } }
                else if constexpr (CIN == 64 && COUT == 64) {
#define SEC_BUFM(D, W, M, FLG) launch_rows_buf<T, CIN, COUT, D, W, M, FLG, 27>(feat, n_feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st)
                    // prefetch distance 3, B fragments not double-buffered, neighbour offsets re-read from the staged table (FL 1 + 128):
                    // 134 VGPRs instead of 194 at the same stand-alone time (23.3 vs 23.6 us) -- and +3.9 % frames/s with three steps in
                    // flight, where the kernel's footprint decides what else fits on the CU beside it (SEC_CONV_FOOTPRINT=1: the old form)
                    // (the same form for the 128-row workgroups of the mid-size layers: 13.4 vs 13.2 us stand-alone, no gain in flight)
                    // + one barrier per three offsets (FL 512, six-slot weight ring): 23.5 -> 22.1 us stand-alone
                    if (n_out < rows_min() && conv_variant() == 1) {      // mid-size layers: 128-row workgroups
                        if (rows_footprint() == 4) { SEC_BUF(4, 4, 3, 27); }          // the form with a barrier per offset (13.3 vs 12.65 us)
                        // (two waves per SIMD asked for, not three: 63 KB of LDS per four-wave workgroup allow two workgroups per CU whatever
                        // the registers do -- with MINW = 3 hipcc squeezed the kernel into 122 VGPRs + 32 AGPR spill slots for nothing and
                        // warned that it could not meet the occupancy)
                        else if (bal) { SEC_BUFM(3, 4, 2, 1 + 128 + 512 + 2048); }
                        else { SEC_BUFM(3, 4, 2, 1 + 128 + 512); }
                    } else if (rows_footprint() == 1) { SEC_BUF(4, 8, 3, 27); }
                    else if (rows_footprint() == 3) { SEC_BUFM(3, 8, 3, 1 + 128); }
                    else if (bal) { SEC_BUFM(3, 8, 3, 1 + 128 + 512 + 2048); }
                    else { SEC_BUFM(3, 8, 3, 1 + 128 + 512); }
#undef SEC_BUFM
                } else if (bal) { SEC_BUF(4, 8, 3 + 2048, 27); }
                else { SEC_BUF(4, 8, 3, 27); }
                return;
            }
#undef SEC_BUF
        }
    }
#ifdef SEC_CONV_EXPERIMENTS
    if constexpr (std::is_same<T, OT>::value && (CIN == 64 || CIN == 32) && (COUT == 64 || COUT == 32)) {
        const int rp = feat ? rows_plan(CIN, COUT, kvol, n_out, true) : 0;
        if constexpr (CIN == 64) {
            if (rp == PLAN_ROWS_COMPACT) { launch_rows<T, CIN, COUT, 32>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); return; }
            if (rp == PLAN_ROWS_TOUCH) { launch_rows<T, CIN, COUT, 64>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); return; }
            if (rp == PLAN_ROWS_COMPACT_TOUCH) { launch_rows<T, CIN, COUT, 96>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); return; }
            if constexpr (COUT == 64) {
                if (rp == PLAN_ROWS_REG) {
                    const int v = conv_variant();
                    if (v == 13) launch_rows_reg<T, CIN, COUT, 4, 2>(feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                    else if (v == 14) launch_rows_reg<T, CIN, COUT, 2, 3>(feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                    else launch_rows_reg<T, CIN, COUT, 5, 2>(feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
                    return;
                }
            }
        }
        if (rp == PLAN_ROWS) {
            launch_rows<T, CIN, COUT, 0>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st);
            return;
        }
#ifdef SEC_CONV_ABLATIONS
        if (conv_variant() >= 91 && conv_variant() <= 96 && kvol == 27 && feat && CIN == 64 && COUT == 64) {
            switch (conv_variant()) {
            case 91: launch_rows<T, CIN, COUT, 1>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            case 92: launch_rows<T, CIN, COUT, 2>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            case 93: launch_rows<T, CIN, COUT, 4>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            case 94: launch_rows<T, CIN, COUT, 8>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            case 95: launch_rows<T, CIN, COUT, 16>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            default: launch_rows<T, CIN, COUT, 3>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); break;
            }
            return;
        }
#endif
    }
#endif
#ifdef SEC_CONV_EXPERIMENTS   // measured dead ends, compiled only for A/B builds
    if (conv_variant() == 2 && kvol == 27) {
        hipLaunchKernelGGL((k_conv_mfma_lds<T, OT, CIN, COUT, 27>), dim3(div_up(n_out, 128)), dim3(kBlock), 0, st,
                           (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, scale, shift, relu, (OT *)out);
        return;
    }
    if (conv_variant() == 7 && kvol == 27 && CIN <= 64 && CIN >= 32) {
        launch_wlds_fl<T, OT, CIN, COUT>(feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
        return;
    }
    if ((conv_variant() == 6 || conv_variant() == 7) && kvol == 27 && CIN <= 64) {
        launch_wlds<T, OT, CIN, COUT>(feat, packed, nbr, n_out, num_out_dev, scale, shift, relu, out, st);
        return;
    }
#endif
    if (conv_variant() == 8 || ((conv_variant() == 1 || conv_variant() == 29) && COUT <= 32)) {   // single 32-column slice: leaner split-K kernel (29 = the automatic choice without the row-split kernels)
        hipLaunchKernelGGL((k_conv_mfma_sks<T, OT, CIN, COUT, 4>), dim3((div_up(n_out, 32) + 7) / 8 * 8, (COUT + 31) / 32),
                           dim3(256), 0, st, (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift,
                           relu, (OT *)out);
        return;
    }
#ifdef SEC_CONV_EXPERIMENTS
    if (conv_variant() == 3 && kvol == 27) {
        hipLaunchKernelGGL((k_conv_mfma_lds2<T, OT, CIN, COUT, 27>), dim3(div_up(n_out, 128)), dim3(kBlock), 0, st,
                           (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, scale, shift, relu, (OT *)out);
        return;
    }
    if (conv_variant() == 4 || conv_variant() == 5) {
        if (conv_variant() == 4)
            hipLaunchKernelGGL((k_conv_mfma_skm<T, OT, CIN, COUT, 2>), dim3(div_up(n_out, 64)), dim3(kBlock), 0, st,
                               (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, (OT *)out);
        else
            hipLaunchKernelGGL((k_conv_mfma_skm<T, OT, CIN, COUT, 1>), dim3(div_up(n_out, 32)), dim3(kBlock), 0, st,
                               (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, (OT *)out);
        return;
    }
#endif
    if (conv_variant() >= 1) {
        constexpr int NW = 4;
        hipLaunchKernelGGL((k_conv_mfma_sk<T, OT, CIN, COUT, NW>), dim3((div_up(n_out, 32) + 7) / 8 * 8), dim3(NW * 64), 0, st,
                           (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift,
                           relu | (conv_swizzle() << 16), (OT *)out);
        return;
    }
    int rows_per_block = (kBlock / 64) * 32 * MT;
    hipLaunchKernelGGL((k_conv_mfma<T, OT, CIN, COUT, MT>), dim3(div_up(n_out, rows_per_block)), dim3(kBlock), 0, st,
                       (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, (OT *)out);
}

template <typename T, typename OT>
static bool dispatch_mfma(int cin, int cout, const void *feat, long long n_feat, const void *packed, const int *nbr, int n_out,
                          const int *num_out_dev, int kvol, const float *scale, const float *shift, int relu, void *out,
                          hipStream_t st) {
#define SEC_CASE(CI, CO)                                                                                         \
    if (cin == CI && cout == CO) {                                                                               \
        launch_mfma<T, OT, CI, CO>(feat, n_feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st); \
        return true;                                                                                             \
    }
    SEC_CASE(16, 16) SEC_CASE(16, 32) SEC_CASE(32, 32) SEC_CASE(32, 64) SEC_CASE(64, 64) SEC_CASE(64, 128)
    SEC_CASE(128, 128) SEC_CASE(16, 64) SEC_CASE(64, 32) SEC_CASE(32, 16) SEC_CASE(128, 64)
#undef SEC_CASE
    return false;
}

// fp32 features on the BF16 matrix pipe (round 5): every fp32 operand is split in registers into two bf16 values, v = hi + lo with
// hi = bf16(v), lo = bf16(v - hi) (16 significant bits), and the product is x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 --
// the form sec_conv2d_nhwc_x3 uses for the RPN; error <= 3 * 2^-18 per product, within the 1e-4 feature tolerance of fp32 networks.
// The fp32 MFMA (v_mfma_f32_32x32x2_f32: 256 FLOP / clk / CU) makes k_conv_mfma_f32 matrix bound -- 79 us of dense-per-offset work on
// the 64 -> 64 layer at 56 k rows, 105 us measured; three bf16 MFMAs per product term cost a fifth of that, and the launch becomes what
// the 16-bit kernels are: bound by the bytes a CU gathers.  Same interface as k_conv_mfma_f32 (fp32 rows in, fp32 rows out, fp32
// weights [k][ci][co] or their transpose / mirror for the data gradient): the split happens on the way into the MFMA operands --
// W[k] -> registers -> (hi | lo) B fragments in LDS (double buffered, one barrier per offset; a thread owns 8 input channels of one
// output channel = one 16-byte fragment piece of each plane), gathered rows -> registers -> (hi, lo) A fragments.  WAVES x 32 rows
// per workgroup (8 waves: one copy of W[k] per 256 rows; 4 waves for launches that would leave CUs idle).
template <int CIN, int COUT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_conv_rows_x3_f32(const float *__restrict__ feat, const float *__restrict__ w,
                                                                const int *__restrict__ nbr, int n_out, const int *__restrict__ num_out_dev,
                                                                int kvol, int w_t, int mirror, const float *__restrict__ scale,
                                                                const float *__restrict__ shift, int relu, float *__restrict__ out) {
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "16-channel k-steps, 32-column MFMA tiles");
    constexpr int KS = CIN / 16, NT = COUT / 32, ROWS = WAVES * 32, NTH = WAVES * 64;
    constexpr int PIECES = KS * NT * 64;                 // 16-byte fragment pieces of one plane of W[k]: piece = (s, t, lane)
    constexpr int PPT = (PIECES + NTH - 1) / NTH;        // pieces staged per thread and offset
    __shared__ __attribute__((aligned(16))) uint4 sB[2][2][PIECES];     // [buffer][hi | lo][piece]
    __shared__ int s_nbr[ROWS * 27];
    if (num_out_dev) n_out = *num_out_dev;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long long base = (long long)blockIdx.x * ROWS;
    if (base >= n_out) return;
    {   // the workgroup's slice of the gather table (contiguous), -1 behind the last row
        const long long lim = ((long long)n_out - base) * kvol;
        const int *src = nbr + base * kvol;
        for (int e = tid; e < ROWS * kvol; e += NTH) s_nbr[e] = e < lim ? src[e] : -1;
    }
    // piece p = (s * NT + t) * 64 + l holds W[ci = s * 16 + (l >> 5) * 8 + e][co = t * 32 + (l & 31)], e = 0..7
    float wreg[PPT][8];
    auto load_w = [&](int k) {
        const float *wk = w + (size_t)(mirror ? kvol - 1 - k : k) * CIN * COUT;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int p = q * NTH + tid;
            if (PIECES % NTH != 0 && p >= PIECES) break;
            const int l = p & 63, st = p >> 6, t = st % NT, sidx = st / NT;
            const int ci = sidx * 16 + (l >> 5) * 8, co = t * 32 + (l & 31);
            if (w_t) {
                const float4 a = *reinterpret_cast<const float4 *>(wk + (size_t)co * CIN + ci);
                const float4 b = *reinterpret_cast<const float4 *>(wk + (size_t)co * CIN + ci + 4);
                wreg[q][0] = a.x; wreg[q][1] = a.y; wreg[q][2] = a.z; wreg[q][3] = a.w;
                wreg[q][4] = b.x; wreg[q][5] = b.y; wreg[q][6] = b.z; wreg[q][7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wreg[q][e] = wk[(size_t)(ci + e) * COUT + co];
            }
        }
    };
    auto split8 = [](const float (&v)[8], uint4 &hi, uint4 &lo) {
        const unsigned h0 = pack2_16<__hip_bfloat16>(v[0], v[1]), h1 = pack2_16<__hip_bfloat16>(v[2], v[3]);
        const unsigned h2 = pack2_16<__hip_bfloat16>(v[4], v[5]), h3 = pack2_16<__hip_bfloat16>(v[6], v[7]);
        hi = make_uint4(h0, h1, h2, h3);
        lo = make_uint4(pack2_16<__hip_bfloat16>(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(v[4] - __uint_as_float(h2 << 16), v[5] - __uint_as_float(h2 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(v[6] - __uint_as_float(h3 << 16), v[7] - __uint_as_float(h3 & 0xffff0000u)));
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int p = q * NTH + tid;
            if (PIECES % NTH != 0 && p >= PIECES) break;
            uint4 hi, lo;
            split8(wreg[q], hi, lo);
            sB[buf][0][p] = hi;
            sB[buf][1][p] = lo;
        }
    };
    load_w(0);
    store_w(0);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    // gathers of offset k + 1 are issued before the MFMAs of offset k
    float4 a[2 * KS], an[2 * KS];
    auto gather = [&](int k, float4 (&dst)[2 * KS]) -> bool {
        const int idx = s_nbr[(wave * 32 + r) * kvol + k];
        const bool any = __ballot(idx >= 0) != 0ull;
        if (any) {
            const float4 *row = reinterpret_cast<const float4 *>(feat + (size_t)(idx >= 0 ? idx : 0) * CIN) + 2 * h;
#pragma unroll
            for (int sidx = 0; sidx < KS; ++sidx) {
                dst[2 * sidx] = idx >= 0 ? row[4 * sidx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                dst[2 * sidx + 1] = idx >= 0 ? row[4 * sidx + 1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        return any;
    };
    __syncthreads();                                 // s_nbr and W[0] are visible
    bool any = gather(0, a);
    for (int k = 0; k < kvol; ++k) {
        bool any_next = false;
        if (k + 1 < kvol) { load_w(k + 1); any_next = gather(k + 1, an); }
        if (any) {
            const uint4 *bh = &sB[k & 1][0][lane], *bl = &sB[k & 1][1][lane];
#pragma unroll
            for (int sidx = 0; sidx < KS; ++sidx) {
                const float v[8] = {a[2 * sidx].x, a[2 * sidx].y, a[2 * sidx].z, a[2 * sidx].w,
                                    a[2 * sidx + 1].x, a[2 * sidx + 1].y, a[2 * sidx + 1].z, a[2 * sidx + 1].w};
                uint4 ahi, alo;
                split8(v, ahi, alo);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint4 whi = bh[(sidx * NT + t) * 64], wlo = bl[(sidx * NT + t) * 64];
                    acc[t] = Mfma<__hip_bfloat16>::run(whi, ahi, acc[t]);       // D^T: lane owns row r
                    acc[t] = Mfma<__hip_bfloat16>::run(wlo, ahi, acc[t]);
                    acc[t] = Mfma<__hip_bfloat16>::run(whi, alo, acc[t]);
                }
            }
        }
        if (k + 1 < kvol) store_w((k + 1) & 1);      // the other buffer: every wave left it at the barrier that opened offset k
#pragma unroll
        for (int j = 0; j < 2 * KS; ++j) a[j] = an[j];
        any = any_next;
        __syncthreads();
    }
    // D^T layout: lane owns row r, acc[t][4 g + j] = channel t * 32 + 8 g + 4 h + j
    const long long row = base + wave * 32 + r;
    if (row < n_out) {
        float *orow = out + (size_t)row * COUT;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = t * 32 + 8 * g + 4 * h;
                float v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v4[j] = epilogue_v(acc[t][4 * g + j], scale ? scale[c + j] : 1.0f, shift ? shift[c + j] : 0.0f, scale != nullptr, shift != nullptr, relu);
                *reinterpret_cast<float4 *>(orow + c) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
    }
}

// The same arithmetic for INFERENCE (weights constant): W is pre-split and pre-packed once per layer -- `wpk` = [k][hi | lo][piece]
// 16-byte B-fragment pieces (ops.pack_weight on an fp32 weight) -- and the loop is software pipelined DIST offsets deep, because with
// three cheap MFMA sets per term an offset's arithmetic (~0.3 us) no longer hides a load: the form above, which fetches offset k + 1
// while offset k multiplies, spent ~1.5 us per offset waiting (55 us for the 23 k-row layers).  Here the gathers and the weight
// pieces of offset k + DIST are issued while offset k runs (register rings), W[k + 1] goes to the LDS buffer the barrier just freed.
template <int CIN, int COUT, int WAVES, int KVOL>
__global__ __launch_bounds__(WAVES * 64) void k_conv_rows_x3p_f32(const float *__restrict__ feat, const uint4 *__restrict__ wpk,
                                                                 const int *__restrict__ nbr, int n_out, const int *__restrict__ num_out_dev,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                                 float *__restrict__ out) {
    constexpr int KS = CIN / 16, NT = (COUT + 31) / 32, ROWS = WAVES * 32, NTH = WAVES * 64, DIST = 3;   // COUT = 16: one half-used tile (the packed image is zero padded)
    constexpr int PIECES = KS * NT * 64, PPT = (PIECES + NTH - 1) / NTH;
    constexpr bool PART = PIECES % NTH != 0;             // fewer pieces than threads: the first PIECES threads stage
    __shared__ __attribute__((aligned(16))) uint4 sB[2][2][PIECES];
    __shared__ int s_nbr[ROWS * KVOL];
    if (num_out_dev) n_out = *num_out_dev;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long long base = (long long)blockIdx.x * ROWS;
    if (base >= n_out) return;
    {
        const long long lim = ((long long)n_out - base) * KVOL;
        const int *src = nbr + base * KVOL;
        for (int e = tid; e < ROWS * KVOL; e += NTH) s_nbr[e] = e < lim ? src[e] : -1;
    }
    // threads beyond the last piece (narrow layers) re-stage piece p % PIECES: identical bytes into the same slot, no branch
    typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));      // (an ext-vector ring stays in registers; a ring of HIP's uint4 structs went to scratch)
    u32x4w wq[DIST][PPT][2];
    const u32x4w *wpv = reinterpret_cast<const u32x4w *>(wpk);
    u32x4w *sBv = reinterpret_cast<u32x4w *>(&sB[0][0][0]);
#define SEC_LOADW(k_, slot_)                                                                                          \
    {                                                                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < PPT; ++q_) {                                                          \
            const int p_ = PART ? (q_ * NTH + tid) % PIECES : q_ * NTH + tid;                                         \
            wq[slot_][q_][0] = wpv[((size_t)(k_) * 2 + 0) * PIECES + p_];                                             \
            wq[slot_][q_][1] = wpv[((size_t)(k_) * 2 + 1) * PIECES + p_];                                             \
        }                                                                                                             \
    }
#define SEC_STOREW(buf_, slot_)                                                                                       \
    {                                                                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < PPT; ++q_) {                                                          \
            const int p_ = PART ? (q_ * NTH + tid) % PIECES : q_ * NTH + tid;                                         \
            sBv[((buf_) * 2 + 0) * PIECES + p_] = wq[slot_][q_][0];                                                   \
            sBv[((buf_) * 2 + 1) * PIECES + p_] = wq[slot_][q_][1];                                                   \
        }                                                                                                             \
    }
    auto split8 = [](const float4 &x, const float4 &y, uint4 &hi, uint4 &lo) {
        const unsigned h0 = pack2_16<__hip_bfloat16>(x.x, x.y), h1 = pack2_16<__hip_bfloat16>(x.z, x.w);
        const unsigned h2 = pack2_16<__hip_bfloat16>(y.x, y.y), h3 = pack2_16<__hip_bfloat16>(y.z, y.w);
        hi = make_uint4(h0, h1, h2, h3);
        lo = make_uint4(pack2_16<__hip_bfloat16>(x.x - __uint_as_float(h0 << 16), x.y - __uint_as_float(h0 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(x.z - __uint_as_float(h1 << 16), x.w - __uint_as_float(h1 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(y.x - __uint_as_float(h2 << 16), y.y - __uint_as_float(h2 & 0xffff0000u)),
                        pack2_16<__hip_bfloat16>(y.z - __uint_as_float(h3 << 16), y.w - __uint_as_float(h3 & 0xffff0000u)));
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    // rows without a neighbour read zeros through the buffer's bounds check (no select, no memory access)
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, 0x7ffffffc, 0x00020000);
    typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
    u32x4g areg[DIST][2 * KS];
    __syncthreads();                                 // s_nbr is visible
    const int *mine = &s_nbr[(wave * 32 + r) * KVOL];
#define SEC_GATHER(k_, slot_)                                                                                         \
    {                                                                                                                 \
        const int idx_ = mine[k_];                                                                                    \
        const unsigned off_ = idx_ >= 0 ? (unsigned)idx_ * (CIN * 4u) + h * 32u : 0x80000000u;                        \
        _Pragma("unroll") for (int s_ = 0; s_ < KS; ++s_) {                                                           \
            areg[slot_][2 * s_] = __builtin_amdgcn_raw_buffer_load_b128(frs, off_ + s_ * 64, 0, 0);                   \
            areg[slot_][2 * s_ + 1] = __builtin_amdgcn_raw_buffer_load_b128(frs, off_ + s_ * 64 + 16, 0, 0);          \
        }                                                                                                             \
    }
#pragma unroll
    for (int k = 0; k < DIST && k < KVOL; ++k) { SEC_LOADW(k, k % DIST) SEC_GATHER(k, k % DIST) }
    SEC_STOREW(0, 0)
    if (DIST < KVOL) SEC_LOADW(DIST, 0)              // (ring slot 0 is free again: W[0] sits in LDS)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KVOL; ++k) {
        // W[k + 1] -> the LDS buffer every wave left at the last barrier; its loads were issued DIST - 1 offsets ago
        if (k + 1 < KVOL) SEC_STOREW((k + 1) & 1, (k + 1) % DIST)
        if (k + 1 + DIST < KVOL) SEC_LOADW(k + 1 + DIST, (k + 1) % DIST)
        const uint4 *bh = &sB[k & 1][0][lane], *bl = &sB[k & 1][1][lane];
#pragma unroll
        for (int sidx = 0; sidx < KS; ++sidx) {
            uint4 ahi, alo;
            split8(__builtin_bit_cast(float4, areg[k % DIST][2 * sidx]), __builtin_bit_cast(float4, areg[k % DIST][2 * sidx + 1]), ahi, alo);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4 whi = bh[(sidx * NT + t) * 64], wlo = bl[(sidx * NT + t) * 64];
                acc[t] = Mfma<__hip_bfloat16>::run(whi, ahi, acc[t]);           // D^T: lane owns row r
                acc[t] = Mfma<__hip_bfloat16>::run(wlo, ahi, acc[t]);
                acc[t] = Mfma<__hip_bfloat16>::run(whi, alo, acc[t]);
            }
        }
        if (k + DIST < KVOL) SEC_GATHER(k + DIST, k % DIST)
        __syncthreads();
    }
#undef SEC_LOADW
#undef SEC_STOREW
#undef SEC_GATHER
    const long long row = base + wave * 32 + r;
    if (row < n_out) {
        float *orow = out + (size_t)row * COUT;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = t * 32 + 8 * g + 4 * h;
                if (c >= COUT) continue;
                float v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v4[j] = epilogue_v(acc[t][4 * g + j], scale ? scale[c + j] : 1.0f, shift ? shift[c + j] : 0.0f, scale != nullptr, shift != nullptr, relu);
                *reinterpret_cast<float4 *>(orow + c) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
    }
}

// packed (hi | lo) weights of k_conv_rows_x3p_f32: bytes, and whether a shape has an instantiation
static bool x3p_shape(int cin, int cout, int kvol) {
    const bool sh = (cin == 16 && cout == 16) || (cin == 16 && cout == 32) || (cin == 32 && cout == 32) || (cin == 32 && cout == 64) ||
                    (cin == 64 && cout == 32) || (cin == 64 && cout == 64);
    return sh && (kvol == 27 || (kvol == 3 && cin == 64 && cout == 64));
}

static bool launch_x3p(const void *feat, long long n_in, const void *wpk, const int *nbr, int n_out, const int *num_out_dev, int cin, int cout,
                       int kvol, const float *scale, const float *shift, int relu, void *out, hipStream_t st) {
    if (!x3p_shape(cin, cout, kvol) || n_in * cin * 4 >= 0x7fffffffll) return false;
#define SEC_X3P(CI, CO, KV)                                                                                                   \
    if (cin == CI && cout == CO && kvol == KV) {                                                                              \
        if (n_out >= 40000) {                                                                                                 \
            set_last_kernel("void sec::k_conv_rows_x3p_f32<%d, %d, 8, %d>", CI, CO, KV);                                      \
            hipLaunchKernelGGL((k_conv_rows_x3p_f32<CI, CO, 8, KV>), dim3(div_up(n_out, 256)), dim3(512), 0, st, (const float *)feat, \
                               (const uint4 *)wpk, nbr, n_out, num_out_dev, scale, shift, relu, (float *)out);                \
        } else {                                                                                                              \
            set_last_kernel("void sec::k_conv_rows_x3p_f32<%d, %d, 4, %d>", CI, CO, KV);                                      \
            hipLaunchKernelGGL((k_conv_rows_x3p_f32<CI, CO, 4, KV>), dim3(div_up(n_out, 128)), dim3(256), 0, st, (const float *)feat, \
                               (const uint4 *)wpk, nbr, n_out, num_out_dev, scale, shift, relu, (float *)out);                \
        }                                                                                                                     \
        return true;                                                                                                          \
    }
    SEC_X3P(16, 16, 27) SEC_X3P(16, 32, 27) SEC_X3P(32, 32, 27) SEC_X3P(32, 64, 27) SEC_X3P(64, 32, 27) SEC_X3P(64, 64, 27) SEC_X3P(64, 64, 3)
#undef SEC_X3P
    return false;
}

template <typename T, typename OT>
static bool launch_tiled(const void *feat, const void *w, const int *nbr, int n_out, const int *num_out_dev, int cin, int cout,
                         int kvol, int w_t, int mirror, const float *scale, const float *shift, int relu, void *out,
                         hipStream_t st);

template <typename T, typename OT>
static void launch_generic(const void *feat, const void *w, const int *nbr, int n_out, const int *num_out_dev, int cin,
                           int cout, int kvol, const float *scale, const float *shift, int relu, void *out,
                           hipStream_t st) {
    if (cin == 4 && cout == 16 && kvol * 4 * 16 * sizeof(float) <= 48 * 1024) {
        hipLaunchKernelGGL((k_conv_c4<T, OT, 16>), dim3(div_up(n_out, kBlock)), dim3(kBlock), kvol * 4 * 16 * sizeof(float), st,
                           (const T *)feat, (const T *)w, nbr, n_out, num_out_dev, kvol, scale, shift, relu, (OT *)out);
        return;
    }
    if (launch_tiled<T, OT>(feat, w, nbr, n_out, num_out_dev, cin, cout, kvol, 0, 0, scale, shift, relu, out, st)) return;
    long long total = (long long)n_out * cout;
    int blocks = div_up(total, kBlock);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((k_conv_generic<T, OT>), dim3(blocks), dim3(kBlock), 0, st, (const T *)feat, (const T *)w, nbr,
                       n_out, num_out_dev, cin, cout, kvol, scale, shift, relu, (OT *)out);
}


// ------------------------------------------------------------------ register-tiled VALU path (fp32 training / inference)
// Output stationary like the MFMA kernels, for dtypes without a matrix-core path here (fp32) and as the dgrad of fp32
// training: a workgroup owns ROWS output rows x all COUT channels, walks the offsets, stages the gathered input rows
// (transposed: [ci][row], fp32) and W[k] ([ci][co]) in LDS, and every thread keeps a 4 row x 4 channel block in registers:
// two 16-byte LDS reads per 16 FMAs, versus two global loads per FMA in k_conv_generic (27x faster on the subm2 layer).
// The per-element accumulation order (offsets outer, input channels inner, fmaf) is that of k_conv_generic and of the
// oracle, so fp32 results are unchanged.  w_t: read W[k] transposed (dgrad: Cin/Cout already swapped by the caller);
// mirror: use offset K-1-k's weights (SubM dgrad through nbr_out).
template <typename T, typename OT, int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_tiled(const T *__restrict__ feat, const T *__restrict__ w,
                                                      const int *__restrict__ nbr, int n_out,
                                                      const int *__restrict__ num_out_dev, int kvol, int w_t, int mirror,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      int relu, OT *__restrict__ out) {
    constexpr int CT = COUT / 4;                  // threads across the channels
    constexpr int ROWS = (kBlock / CT) * 4;       // 64 rows for COUT = 64, 128 for 32, 256 for 16
    __shared__ float sA[CIN][ROWS + 4];           // gathered inputs, transposed
    __shared__ float sW[CIN][COUT + 4];
    __shared__ int s_idx[ROWS];
    if (num_out_dev) n_out = *num_out_dev;
    const int tid = threadIdx.x;
    const long long base = (long long)blockIdx.x * ROWS;
    if (base >= n_out) return;
    const int r0 = (tid / CT) * 4, c0 = (tid % CT) * 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    for (int k = 0; k < kvol; ++k) {
        for (int r = tid; r < ROWS; r += kBlock) s_idx[r] = base + r < n_out ? nbr[(size_t)(base + r) * kvol + k] : -1;
        __syncthreads();
        {   // W[k] (or its transpose / mirror) -> sW[ci][co]
            const T *wk = w + (size_t)(mirror ? kvol - 1 - k : k) * CIN * COUT;
            for (int e = tid; e < CIN * COUT; e += kBlock) {
                const int ci = e / COUT, co = e - ci * COUT;
                sW[ci][co] = Cvt<T>::to(w_t ? wk[(size_t)co * CIN + ci] : wk[e]);   // w_t: stored [co][ci] per offset
            }
        }
        // gathered rows -> sA[ci][row]; rows without a neighbour contribute exact zeros
        for (int e = tid; e < ROWS * (CIN / 4); e += kBlock) {
            const int r = e / (CIN / 4), c4 = (e - r * (CIN / 4)) * 4;
            const int idx = s_idx[r];
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (idx >= 0) {
                const T *src = feat + (size_t)idx * CIN + c4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = Cvt<T>::to(src[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sA[c4 + j][r] = v[j];
        }
        __syncthreads();
        const bool any = s_idx[r0] >= 0 || s_idx[r0 + 1] >= 0 || s_idx[r0 + 2] >= 0 || s_idx[r0 + 3] >= 0;
        if (any) {
#pragma unroll 8
            for (int ci = 0; ci < CIN; ++ci) {
                const float4 a4 = *reinterpret_cast<const float4 *>(&sA[ci][r0]);
                const float4 w4 = *reinterpret_cast<const float4 *>(&sW[ci][c0]);
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(aa[a], ww[b], acc[a][b]);
            }
        }
        __syncthreads();
    }
    const Affine4 af = load_affine4(scale, shift, c0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long long row = base + r0 + a;
        if (row < n_out)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                out[(size_t)row * COUT + c0 + b] = Cvt<OT>::from(epilogue_v(acc[a][b], af.sc[b], af.sh[b], scale != nullptr, shift != nullptr, relu));
    }
}


// ------------------------------------------------------------------ fp32 on the matrix cores (v_mfma_f32_32x32x2_f32)
// The reference's default precision is fp32; k_conv_tiled (VALU, 223 us on the 64 -> 64 layer at 56 k rows) left it ten times
// slower than the 16-bit path.  Output stationary like every kernel here: a workgroup owns 128 output rows (four waves, a 32-row
// tile each) x all COUT channels; per kernel offset W[k] sits in LDS as fp32 (double buffered: W[k+1] travels global -> registers
// -> LDS while offset k computes, one barrier per offset) and a lane gathers 16 bytes of its row per 8 input channels.
// Operand layout of v_mfma_f32_32x32x2_f32: A[m = lane % 32][k = lane / 32], B[k = lane / 32][n = lane % 32], so with h = lane / 32
// a lane's float4 of channels 8j + 4h .. + 3 feeds four MFMAs (s = 0..3) whose B operand is W[8j + 4h + s][n]: the contraction
// index runs in the order (j, s, h) instead of ascending -- any order is a valid dot product -- and no lane shuffles are needed.
// Rows without a neighbour contribute exact zeros; an offset no row of the wave uses is skipped (wave-uniform ballot).
// Dense per offset: the MFMAs run for all 32 rows of a tile whatever share of them has a neighbour (39 % on car.fhd's subm2), so
// the matrix-pipe floor of that layer is 12.4 GFLOP / 157 TFLOP/s = 79 us; pair compaction (LDS accumulators) would be needed to
// go below it.  Accumulation order differs from the oracle's (offsets outer, channels ascending): results agree to fp32 rounding
// (tests: 1e-4 of the range, as for the 16-bit kernels), not bit for bit.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_mfma_f32(const float *__restrict__ feat, const float *__restrict__ w,
                                                         const int *__restrict__ nbr, int n_out, const int *__restrict__ num_out_dev,
                                                         int kvol, int w_t, int mirror, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, int relu, float *__restrict__ out) {
    static_assert(CIN % 8 == 0 && COUT % 32 == 0, "8-channel gathers, 32-column MFMA tiles");
    constexpr int LDW = COUT + 8, CT = COUT / 32, NJ = CIN / 8, ROWS = 128;
    constexpr int WPT = CIN * COUT / kBlock;     // weights per thread and offset
    static_assert(CIN * COUT % kBlock == 0, "whole weights per thread");
    __shared__ float sW[2][CIN * LDW];
    __shared__ int s_nbr[ROWS * 27];
    if (num_out_dev) n_out = *num_out_dev;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    const long long base = (long long)blockIdx.x * ROWS;
    if (base >= n_out) return;
    {   // the workgroup's slice of the gather table (contiguous), -1 behind the last row
        const long long lim = ((long long)n_out - base) * kvol;
        const int *src = nbr + base * kvol;
        for (int e = tid; e < ROWS * kvol; e += kBlock) s_nbr[e] = e < lim ? src[e] : -1;
    }
    // W[k] (or its transpose / mirror: the data gradient) -> registers -> LDS, element e = ci * COUT + co.  (A 16-byte form of this
    // staging, unconditional gathers with a select, and an explicit double buffer of the B operands were each measured on the
    // 64 -> 64 layer: 151-155 us against 138 us for this form -- the register count, 206 against 172, costs more than they save.)
    float wreg[WPT];
    auto load_w = [&](int k) {
        const float *wk = w + (size_t)(mirror ? kvol - 1 - k : k) * CIN * COUT;
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int e = q * kBlock + tid, ci = e / COUT, co = e - ci * COUT;
            wreg[q] = w_t ? wk[(size_t)co * CIN + ci] : wk[e];
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int e = q * kBlock + tid, ci = e / COUT, co = e - ci * COUT;
            sW[buf][ci * LDW + co] = wreg[q];
        }
    };
    load_w(0);
    store_w(0);
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    __syncthreads();
    // the gathers of offset k + 1 are issued before the MFMAs of offset k (a wave is often alone on its SIMD here: 137 VGPRs, and a
    // 23 k-row layer is fewer workgroups than CUs -- without the prefetch every offset paid a full gather latency: 3.4 us per offset)
    float4 a[NJ], an[NJ];
    auto gather = [&](int k, float4 (&dst)[NJ]) -> bool {
        const int idx = s_nbr[(wave * 32 + m) * kvol + k];
        const bool any = __ballot(idx >= 0) != 0ull;
        if (any) {
            const float4 *row = reinterpret_cast<const float4 *>(feat + (size_t)(idx >= 0 ? idx : 0) * CIN) + h;
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[j] = idx >= 0 ? row[2 * j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        return any;
    };
    bool any = gather(0, a);
    for (int k = 0; k < kvol; ++k) {
        bool any_next = false;
        if (k + 1 < kvol) { load_w(k + 1); any_next = gather(k + 1, an); }
        if (any) {
            const float *wb = &sW[k & 1][(4 * h) * LDW + m];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float as[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[t], wb[(8 * j + t) * LDW + c * 32], acc[c], 0, 0, 0);
            }
        }
        if (k + 1 < kvol) store_w((k + 1) & 1);      // the other buffer: every wave left it at the barrier that opened offset k
#pragma unroll
        for (int j = 0; j < NJ; ++j) a[j] = an[j];
        any = any_next;
        __syncthreads();
    }
    // D[mm][n]: lane holds column n = lane % 32 and rows mm = 8 * (i / 4) + 4 * h + i % 4
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = c * 32 + m;
        const float sc = scale ? scale[col] : 1.0f, sh = shift ? shift[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long r = base + wave * 32 + 8 * (i / 4) + 4 * h + (i % 4);
            if (r < n_out) out[(size_t)r * COUT + col] = epilogue_v(acc[c][i], sc, sh, scale != nullptr, shift != nullptr, relu);
        }
    }
}

template <typename T, typename OT>
static bool launch_tiled(const void *feat, const void *w, const int *nbr, int n_out, const int *num_out_dev, int cin, int cout,
                         int kvol, int w_t, int mirror, const float *scale, const float *shift, int relu, void *out,
                         hipStream_t st) {
    if constexpr (std::is_same<T, float>::value && std::is_same<OT, float>::value) {
        // fp32 on the matrix cores for the channel plans of SECOND's layers (and their data gradients); g_variant_override 30 keeps
        // the VALU form (parity tests compare the two)
        // round 5: the split-operand form on the bf16 pipe (k_conv_rows_x3_f32); variant 31 keeps the fp32-MFMA form, 30 the VALU form
        if (kvol <= 27 && conv_variant() != 30 && conv_variant() != 31 && g_fp32_mode == 0) {
#define SEC_X3(CI, CO)                                                                                                       \
            if (cin == CI && cout == CO) {                                                                                   \
                if (n_out >= 40000) {                                                                                        \
                    set_last_kernel("void sec::k_conv_rows_x3_f32<%d, %d, 8>", CI, CO);                                      \
                    hipLaunchKernelGGL((k_conv_rows_x3_f32<CI, CO, 8>), dim3(div_up(n_out, 256)), dim3(512), 0, st, (const float *)feat, \
                                       (const float *)w, nbr, n_out, num_out_dev, kvol, w_t, mirror, scale, shift, relu, (float *)out); \
                } else {                                                                                                     \
                    set_last_kernel("void sec::k_conv_rows_x3_f32<%d, %d, 4>", CI, CO);                                      \
                    hipLaunchKernelGGL((k_conv_rows_x3_f32<CI, CO, 4>), dim3(div_up(n_out, 128)), dim3(256), 0, st, (const float *)feat, \
                                       (const float *)w, nbr, n_out, num_out_dev, kvol, w_t, mirror, scale, shift, relu, (float *)out); \
                }                                                                                                            \
                return true;                                                                                                 \
            }
            SEC_X3(16, 32) SEC_X3(32, 32) SEC_X3(32, 64) SEC_X3(64, 32) SEC_X3(64, 64)
#undef SEC_X3
        }
        if (kvol <= 27 && conv_variant() != 30) {
#define SEC_MF(CI, CO)                                                                                                       \
            if (cin == CI && cout == CO) {                                                                                   \
                set_last_kernel("void sec::k_conv_mfma_f32<%d, %d>", CI, CO);                                                \
                hipLaunchKernelGGL((k_conv_mfma_f32<CI, CO>), dim3(div_up(n_out, 128)), dim3(kBlock), 0, st, (const float *)feat, \
                                   (const float *)w, nbr, n_out, num_out_dev, kvol, w_t, mirror, scale, shift, relu, (float *)out); \
                return true;                                                                                                 \
            }
            SEC_MF(16, 32) SEC_MF(32, 32) SEC_MF(32, 64) SEC_MF(64, 32) SEC_MF(64, 64)
#undef SEC_MF
        }
    }
#define SEC_TL(CI, CO)                                                                                                   \
    if (cin == CI && cout == CO) {                                                                                       \
        constexpr int ROWS = (kBlock / (CO / 4)) * 4;                                                                    \
        hipLaunchKernelGGL((k_conv_tiled<T, OT, CI, CO>), dim3(div_up(n_out, ROWS)), dim3(kBlock), 0, st, (const T *)feat,   \
                           (const T *)w, nbr, n_out, num_out_dev, kvol, w_t, mirror, scale, shift, relu, (OT *)out);      \
        return true;                                                                                                     \
    }
    SEC_TL(16, 16) SEC_TL(16, 32) SEC_TL(32, 16) SEC_TL(32, 32) SEC_TL(32, 64) SEC_TL(64, 32) SEC_TL(64, 64)
#undef SEC_TL
    return false;
}

// ------------------------------------------------------------------ backward: generic fallbacks
// dfeat[j][ci] = sum_k sum_co dout[tbl[j][col(k)]][co] * W[k][ci][co]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv_dgrad(const T *__restrict__ dout, const T *__restrict__ w,
                                                      const int *__restrict__ tbl, int mirror, int n_in, int cin,
                                                      int cout, int kvol, T *__restrict__ dfeat) {
    long long total = (long long)n_in * cin;
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < total; g += (long long)gridDim.x * kBlock) {
        int j = (int)(g / cin), ci = (int)(g % cin);
        const int *row = tbl + (size_t)j * kvol;
        float acc = 0.0f;
        for (int k = 0; k < kvol; ++k) {
            int o = row[mirror ? kvol - 1 - k : k];
            if (o < 0) continue;
            const T *d = dout + (size_t)o * cout;
            const T *wk = w + ((size_t)k * cin + ci) * cout;
            for (int co = 0; co < cout; ++co) acc = fmaf(Cvt<T>::to(d[co]), Cvt<T>::to(wk[co]), acc);
        }
        dfeat[g] = Cvt<T>::from(acc);
    }
}

// dW[k][ci][co] += sum_{o in chunk} feat[nbr_out[o][k]][ci] * dout[o][co]   (fp32 atomics across chunks)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv_wgrad(const T *__restrict__ feat, const T *__restrict__ dout,
                                                      const int *__restrict__ nbr, int n_out, int cin, int cout,
                                                      int kvol, int rows_per_chunk, float *__restrict__ dw) {
    int k = blockIdx.y;
    int o0 = blockIdx.x * rows_per_chunk, o1 = o0 + rows_per_chunk;
    if (o1 > n_out) o1 = n_out;
    for (int e = threadIdx.x; e < cin * cout; e += kBlock) {
        int ci = e / cout, co = e % cout;
        float acc = 0.0f;
        for (int o = o0; o < o1; ++o) {
            int idx = nbr[(size_t)o * kvol + k];
            if (idx < 0) continue;
            acc = fmaf(Cvt<T>::to(feat[(size_t)idx * cin + ci]), Cvt<T>::to(dout[(size_t)o * cout + co]), acc);
        }
        if (acc != 0.0f) atomicAdd(&dw[((size_t)k * cin + ci) * cout + co], acc);
    }
}


// ---- backward ----------------------------------------------------------------------------------------------------------
// dgrad is the forward operator with the roles swapped: dfeat[j,:] = sum_k dout[tbl[j][k], :] @ W[k]^T with tbl = nbr_in
// (strided conv) or, for SubM, nbr_out itself read through mirrored offsets (nbr_in[j][k] == nbr_out[j][K-1-k]).  So for
// 16-bit dtypes the weights are re-packed transposed (and offset-mirrored for SubM) into the caller's workspace and the
// MFMA forward kernels run unchanged with Cin <-> Cout swapped.
// packed element ((((k*KS + s)*NT + t)*64 + lane)*8 + e) = Wt[k][s*16 + (lane>>5)*8 + e][t*32 + (lane&31)],
// Wt[k][co][ci] = W[mirror ? K-1-k : k][ci][co]      (KS = Cout/16, NT = ceil(Cin/32))
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_weight_t(const T *__restrict__ w, int kvol, int cin, int cout, int mirror,
                                                         T *__restrict__ packed) {
    int ks = cout / 16, nt = (cin + 31) / 32;
    long long total = (long long)kvol * ks * nt * 64 * 8;
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    int e = (int)(g & 7), lane = (int)((g >> 3) & 63);
    long long q = g >> 9;
    int t = (int)(q % nt);
    q /= nt;
    int sidx = (int)(q % ks);
    int k = (int)(q / ks);
    int co = sidx * 16 + (lane >> 5) * 8 + e, ci = t * 32 + (lane & 31);
    int km = mirror ? kvol - 1 - k : k;
    packed[g] = ci < cin ? w[((size_t)km * cin + ci) * cout + co] : Cvt<T>::from(0.0f);
}

// wgrad: dW[k][ci][co] = sum over the pairs (i, o) of offset k of feat[i][ci] * dout[o][co].
// One workgroup = one offset x one chunk of output rows.  The chunk's pairs are compacted 32 at a time into LDS (fp32,
// gathered feature row next to its dout row); every thread keeps a 4 x 4 (ci x co) block of the Cin x Cout result in
// registers and consumes the staged rows with two 16-byte LDS reads per 16 FMAs; chunks combine with fp32 atomics.
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_wgrad_tiled(const T *__restrict__ feat, const T *__restrict__ dout,
                                                            const int *__restrict__ nbr, int n_out, int kvol,
                                                            int rows_per_chunk, float *__restrict__ dw) {
    constexpr int SUB = 32;                                   // pairs staged per step
    constexpr int TILES = (CIN / 4) * (COUT / 4);             // 4x4 register blocks
    constexpr int TPT = (TILES + kBlock - 1) / kBlock;        // blocks per thread (1 for 64x64)
    __shared__ float sF[SUB][CIN + 4], sD[SUB][COUT + 4];
    __shared__ int s_src[SUB], s_dst[SUB], s_n;
    const int k = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int o0 = blockIdx.x * rows_per_chunk;
    const int o1 = o0 + rows_per_chunk < n_out ? o0 + rows_per_chunk : n_out;
    float acc[TPT][4][4];
#pragma unroll
    for (int u = 0; u < TPT; ++u)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[u][a][b] = 0.0f;
    for (int ob = o0; ob < o1; ob += SUB) {
        if (tid < 64) {                                       // wave 0 compacts the valid pairs of these SUB rows
            const int o = ob + lane;
            const int idx = (lane < SUB && o < o1) ? nbr[(size_t)o * kvol + k] : -1;
            const unsigned long long m = __ballot(idx >= 0);
            if (idx >= 0) {
                const int pos = __popcll(m & ((1ull << lane) - 1ull));
                s_src[pos] = idx;
                s_dst[pos] = o;
            }
            if (lane == 0) s_n = __popcll(m);
        }
        __syncthreads();
        const int np = s_n;
        for (int e = tid; e < np * (CIN / 4); e += kBlock) {      // 4 channels per thread per step
            const int p = e / (CIN / 4), c4 = (e - p * (CIN / 4)) * 4;
            const T *src = feat + (size_t)s_src[p] * CIN + c4;
#pragma unroll
            for (int j = 0; j < 4; ++j) sF[p][c4 + j] = Cvt<T>::to(src[j]);
        }
        for (int e = tid; e < np * (COUT / 4); e += kBlock) {
            const int p = e / (COUT / 4), c4 = (e - p * (COUT / 4)) * 4;
            const T *src = dout + (size_t)s_dst[p] * COUT + c4;
#pragma unroll
            for (int j = 0; j < 4; ++j) sD[p][c4 + j] = Cvt<T>::to(src[j]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int tile = tid + u * kBlock;
            if (tile < TILES) {
                const int ci0 = (tile / (COUT / 4)) * 4, co0 = (tile % (COUT / 4)) * 4;
                for (int p = 0; p < np; ++p) {
                    const float4 f = *reinterpret_cast<const float4 *>(&sF[p][ci0]);
                    const float4 d = *reinterpret_cast<const float4 *>(&sD[p][co0]);
                    const float fa[4] = {f.x, f.y, f.z, f.w}, da[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[u][a][b] = fmaf(fa[a], da[b], acc[u][a][b]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int tile = tid + u * kBlock;
        if (tile < TILES) {
            const int ci0 = (tile / (COUT / 4)) * 4, co0 = (tile % (COUT / 4)) * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (acc[u][a][b] != 0.0f) atomicAdd(&dw[((size_t)k * CIN + ci0 + a) * COUT + co0 + b], acc[u][a][b]);
        }
    }
}


// MFMA wgrad for 16-bit dtypes.  dW[k] = F^T D with F = gathered feature rows [pairs x Cin], D = dout rows [pairs x Cout]:
// the contraction runs over the PAIRS, so both MFMA operands need 8 consecutive pairs per lane for a fixed channel -- the
// transpose of how rows lie in memory.  The compacted pairs of a 64-row step are therefore staged TRANSPOSED in LDS
// (sFt[channel][pair], sDt[channel][pair], 16-bit), from where a fragment is one ds_read_b128.  The (Cin/32) x (Cout/32)
// result tiles are spread over the 4 waves; with fewer than 4 tiles the waves split the 16-pair k-steps between them and
// are summed by the final fp32 atomics.  Channel counts below 32 are zero padded in LDS.
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_wgrad_mfma(const T *__restrict__ feat, const T *__restrict__ dout,
                                                           const int *__restrict__ nbr, int n_out, int kvol,
                                                           int rows_per_chunk, float *__restrict__ dw) {
    constexpr int SUB = 64;                                    // rows scanned per step -> up to four 16-pair MFMA k-steps
    constexpr int CI_T = (CIN + 31) / 32, CO_T = (COUT + 31) / 32, TILES = CI_T * CO_T;
    static_assert(TILES <= 4 && 4 % TILES == 0, "tiles must divide the four waves");
    constexpr int WPT = 4 / TILES;                             // waves sharing one tile (they split the k-steps)
    constexpr int LD = SUB + 8;                                // row pitch in elements: 80 bytes, 16-byte aligned fragments
    __shared__ __attribute__((aligned(16))) T sFt[CI_T * 32][LD], sDt[CO_T * 32][LD];
    __shared__ int s_src[SUB], s_dst[SUB], s_n;
    const int k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = wv / WPT, ksel = wv % WPT;
    const int ti = tile / CO_T, tj = tile % CO_T;
    const int o0 = blockIdx.x * rows_per_chunk;
    const int o1 = o0 + rows_per_chunk < n_out ? o0 + rows_per_chunk : n_out;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int e = tid; e < CI_T * 32 * LD; e += kBlock) (&sFt[0][0])[e] = Cvt<T>::from(0.0f);   // padding rows / columns stay zero
    for (int e = tid; e < CO_T * 32 * LD; e += kBlock) (&sDt[0][0])[e] = Cvt<T>::from(0.0f);
    int idx_next = (tid < 64 && o0 + lane < o1) ? nbr[(size_t)(o0 + lane) * kvol + k] : -1;
    for (int ob = o0; ob < o1; ob += SUB) {
        if (tid < 64) {                                        // wave 0 compacts the valid pairs of these SUB rows
            const int o = ob + lane;
            const int idx = idx_next;
            const int on = o + SUB;                            // next step's neighbour indices: in flight during this step
            idx_next = on < o1 ? nbr[(size_t)on * kvol + k] : -1;
            const unsigned long long m = __ballot(idx >= 0);
            if (idx >= 0) {
                const int pos = __popcll(m & ((1ull << lane) - 1ull));
                s_src[pos] = idx;
                s_dst[pos] = o;
            }
            if (lane == 0) s_n = __popcll(m);
        }
        __syncthreads();
        const int np = s_n;
        if (np > 0) {
            // transposed staging, 8 channels (one 16-byte load) per thread-step; pairs beyond np are written as zeros
            for (int e = tid; e < SUB * (CIN / 8 > 0 ? CIN / 8 : 1); e += kBlock) {
                const int p = e % SUB, c8 = (e / SUB) * 8;
                T v[8];
                if (p < np) {
                    if constexpr (CIN % 8 == 0) {
                        *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(feat + (size_t)s_src[p] * CIN + c8);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = c8 + j < CIN ? feat[(size_t)s_src[p] * CIN + c8 + j] : Cvt<T>::from(0.0f);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = Cvt<T>::from(0.0f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c8 + j < CIN) sFt[c8 + j][p] = v[j];
            }
            for (int e = tid; e < SUB * (COUT / 8); e += kBlock) {
                const int p = e % SUB, c8 = (e / SUB) * 8;
                T v[8];
                if (p < np) *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(dout + (size_t)s_dst[p] * COUT + c8);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = Cvt<T>::from(0.0f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sDt[c8 + j][p] = v[j];
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < SUB / 16; ++ks) {
                if (ks % WPT == ksel && ks * 16 < np) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(&sFt[ti * 32 + r][ks * 16 + h * 8]);   // A[ci][8 pairs]
                    const uint4 b = *reinterpret_cast<const uint4 *>(&sDt[tj * 32 + r][ks * 16 + h * 8]);   // B[8 pairs][co]
                    acc = Mfma<T>::run(a, b, acc);
                }
            }
        }
        __syncthreads();
    }
    // D layout: column (co) = lane & 31, rows (ci) = (i & 3) + 8 (i >> 2) + 4 h
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ci = ti * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, co = tj * 32 + r;
        if (ci < CIN && co < COUT && acc[i] != 0.0f) atomicAdd(&dw[((size_t)k * CIN + ci) * COUT + co], acc[i]);
    }
}

// The same contraction (dW[k] += X[nbr[o][k]]^T dOut[o] over a chunk of output rows o, one offset k per workgroup) with the staging of
// the dense weight gradient (dense_train.hip, round 5): every 64-row step loads FULL feature / gradient rows with unconditional buffer
// loads (a row without a neighbour at this offset reads zeros through the out-of-range offset -- no pair compaction, no branch around a
// load), stores them untransposed ([row][channel], ds_write_b128; 64-channel rows swap their 64-byte halves on rows 2, 3 mod 4 so that
// the four rows of a read group fall into four bank windows) and lets ds_read_b64_tr_b16 hand each lane eight ROWS of one channel: the
// MFMA operands of a contraction over rows.  The loads of step s + 1 are in flight under the MFMAs of step s (the neighbour indices one
// step further), LDS is double-buffered, one barrier per step.  k_conv_wgrad_mfma compacted the pairs through wave 0 and LDS, loaded
// under `if (p < np)` (a branch and s_waitcnt vmcnt(0) per load) and transposed with eight 2-byte LDS stores per load: three barriers
// and ~2 us per step -- 26.6 -> 17.9 us for 64 -> 64 at 28 k rows (32 -> 64: 31 -> 26; 32 -> 32 unchanged at 36: one MFMA per wave and step).
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 wg_tr16_b64(const char *p) {
    const ws16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws16x4 __attribute__((address_space(3))) *)p);
    return __builtin_bit_cast(uint2, v);
}
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void k_conv_wgrad_tr(const T *__restrict__ feat, const T *__restrict__ dout, const int *__restrict__ nbr,
                                                         int n_out, int kvol, int rows_per_chunk, float *__restrict__ dw) {
    static_assert((CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "full 32-channel tiles");
    constexpr int SUB = 64, FP = CIN * 2, DP = COUT * 2;                     // rows per step; row pitches in bytes
    constexpr int CPR = CIN / 8, DPR = COUT / 8;                             // 16-byte chunks per row
    constexpr int FL = SUB * CPR / kBlock, DL = SUB * DPR / kBlock;          // loads per thread and step (1 or 2)
    constexpr int CI_T = CIN / 32, CO_T = COUT / 32, TILES = CI_T * CO_T, WPT = 4 / TILES;
    __shared__ __attribute__((aligned(16))) char sF[2][SUB * FP];
    __shared__ __attribute__((aligned(16))) char sD[2][SUB * DP];
    const int k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = wv / WPT, ksel = wv % WPT;
    const int ti = tile / CO_T, tj = tile % CO_T;
    const int o0 = blockIdx.x * rows_per_chunk;
    const int o1 = o0 + rows_per_chunk < n_out ? o0 + rows_per_chunk : n_out;
    const int nsteps = (o1 - o0 + SUB - 1) / SUB;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(feat), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dout), 0, 0x7ffffff0, 0x00020000);
    // this thread's staging slots: chunk fc[i] of step-row fr[i] (features), dc / dr (gradient rows)
    int fr[FL], fc[FL], dr[DL], dc[DL], fst[FL], dst_[DL];
#pragma unroll
    for (int i = 0; i < FL; ++i) {
        const int e = tid + kBlock * i;
        fr[i] = e / CPR; fc[i] = e % CPR;
        fst[i] = fr[i] * FP + ((fc[i] ^ (CIN == 64 ? ((fr[i] >> 1) & 1) << 2 : 0)) << 4);
    }
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const int e = tid + kBlock * i;
        dr[i] = e / DPR; dc[i] = e % DPR;
        dst_[i] = dr[i] * DP + ((dc[i] ^ (COUT == 64 ? ((dr[i] >> 1) & 1) << 2 : 0)) << 4);
    }
    int idx[FL], didx[DL];                                  // neighbour rows of the NEXT step to load (and, for the gradient rows, whether one exists)
    auto load_idx = [&](int step) {
#pragma unroll
        for (int i = 0; i < FL; ++i) {
            const int o = o0 + step * SUB + fr[i];
            const bool ok = o < o1;
            const int v = nbr[(size_t)(ok ? o : o0) * kvol + k];       // unconditional load, clamped row
            idx[i] = ok ? v : -1;
        }
#pragma unroll
        for (int i = 0; i < DL; ++i) {                       // a gradient row without a neighbour is NOT read: with static capacities the rows
            const int o = o0 + step * SUB + dr[i];           // behind the live count hold whatever was there (0 x NaN would poison the sum)
            const bool ok = o < o1;
            const int v = nbr[(size_t)(ok ? o : o0) * kvol + k];
            didx[i] = ok ? v : -1;
        }
    };
    wu32x4 rf[FL], rd[DL];
    auto fetch = [&](int step) {
#pragma unroll
        for (int i = 0; i < FL; ++i)
            rf[i] = __builtin_amdgcn_raw_buffer_load_b128(frs, idx[i] >= 0 ? (unsigned)idx[i] * (unsigned)FP + fc[i] * 16u : 0xfffffff0u, 0, 0);
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            const int o = o0 + step * SUB + dr[i];
            rd[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, didx[i] >= 0 ? (unsigned)o * (unsigned)DP + dc[i] * 16u : 0xfffffff0u, 0, 0);
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < FL; ++i) *reinterpret_cast<wu32x4 *>(&sF[buf][fst[i]]) = rf[i];
#pragma unroll
        for (int i = 0; i < DL; ++i) *reinterpret_cast<wu32x4 *>(&sD[buf][dst_[i]]) = rd[i];
    };
    // operand reads: 16-lane group g, lane li: rows ks * 16 + (g / 2) * 8 + t * 4 + li / 4 (t = 0, 1), channels tile * 32 + (g & 1) * 16
    // + (li & 3) * 4 .. + 3; the lane receives channel tile * 32 + r
    const int g = lane >> 4, li = lane & 15;
    const int prow = (g >> 1) * 8 + (li >> 2), swz = ((li >> 3) & 1) << 2;     // (row >> 1) & 1 of every row this lane addresses
    const int qa = ti * 4 + (g & 1) * 2 + ((li & 3) >> 1), qb = tj * 4 + (g & 1) * 2 + ((li & 3) >> 1);
    const int a_off = prow * FP + ((qa ^ (CIN == 64 ? swz : 0)) << 4) + (li & 1) * 8;
    const int b_off = prow * DP + ((qb ^ (COUT == 64 ? swz : 0)) << 4) + (li & 1) * 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    if (nsteps > 0) {
        load_idx(0);
        fetch(0);
        load_idx(1);
        put(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
            fetch(s + 1);                                   // past the chunk: indices -1 / rows >= o1 -> zeros, no memory traffic
            load_idx(s + 2);
#pragma unroll
            for (int ks = 0; ks < SUB / 16; ++ks) {
                if (ks % WPT == ksel) {
                    const char *fa = &sF[buf][ks * 16 * FP + a_off], *fb = &sD[buf][ks * 16 * DP + b_off];
                    const uint2 a0 = wg_tr16_b64(fa), a1 = wg_tr16_b64(fa + 4 * FP);
                    const uint2 b0 = wg_tr16_b64(fb), b1 = wg_tr16_b64(fb + 4 * DP);
                    acc = Mfma<T>::run(make_uint4(a0.x, a0.y, a1.x, a1.y), make_uint4(b0.x, b0.y, b1.x, b1.y), acc);
                }
            }
            put(buf ^ 1);                                   // that buffer was last read before the previous barrier
            __syncthreads();
        }
    }
    // D layout: column (co) = lane & 31, rows (ci) = (i & 3) + 8 (i >> 2) + 4 h
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ci = ti * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, co = tj * 32 + r;
        if (acc[i] != 0.0f) atomicAdd(&dw[((size_t)k * CIN + ci) * COUT + co], acc[i]);
    }
}

template <typename T>
static bool launch_wgrad_tiled(const void *feat, const void *dout, const int *nbr, int n_out, int cin, int cout, int kvol,
                               float *dw, hipStream_t st) {
    // large chunks keep the fp32 atomics few, but every workgroup walks its chunk as a serial chain of gather steps:
    // ~5 workgroups per CU measured best on subm2 (chunk 512/1024/2048/4096: 81/67/78/128 us bf16, 222/169/228/417 us fp32)
    // (A workgroup that owns a chunk of rows AND a group of nine offsets -- dout staged once per 64-row block for all of them, no
    // pair compaction, accumulators of the nine offsets in registers -- was measured 3-5 x SLOWER, 77 vs 26 us on subm2: its serial
    // chain is 9 x longer and the kernel is bound by that chain, not by the staging.  Capping the chunk at 1024 rows on the large
    // nuScenes layers -- 6 600 workgroups instead of 1 700 -- was slower too: 378 vs 207 us for 32 -> 32 at 250 k rows.)
    int chunk = 8192;
    while (chunk > 512 && (long long)div_up(n_out, chunk) * kvol < 1400) chunk >>= 1;      // (700 / 2800 with k_conv_wgrad_tr: no difference)
    dim3 grid(div_up(n_out, chunk), kvol);
    if constexpr (!std::is_same<T, float>::value) {            // 16-bit dtypes: matrix cores
#define SEC_WM(CI, CO)                                                                                                    \
        if (cin == CI && cout == CO) {                                                                                    \
            hipLaunchKernelGGL((k_conv_wgrad_mfma<T, CI, CO>), grid, dim3(kBlock), 0, st, (const T *)feat, (const T *)dout, nbr, \
                               n_out, kvol, chunk, dw);                                                                   \
            return true;                                                                                                  \
        }
#define SEC_WT(CI, CO)                                                                                                    \
        if (cin == CI && cout == CO && conv_variant() != 41) {                                                            \
            hipLaunchKernelGGL((k_conv_wgrad_tr<T, CI, CO>), grid, dim3(kBlock), 0, st, (const T *)feat, (const T *)dout, nbr,   \
                               n_out, kvol, chunk, dw);                                                                   \
            return true;                                                                                                  \
        }
        SEC_WT(32, 32) SEC_WT(32, 64) SEC_WT(64, 32) SEC_WT(64, 64)        // (SEC_CONV_VARIANT=41: the compacting kernel below, for A/B)
#undef SEC_WT
        SEC_WM(4, 16) SEC_WM(16, 16) SEC_WM(16, 32) SEC_WM(32, 32) SEC_WM(32, 64) SEC_WM(64, 64)
#undef SEC_WM
    }
#define SEC_WG(CI, CO)                                                                                                    \
    if (cin == CI && cout == CO) {                                                                                        \
        hipLaunchKernelGGL((k_conv_wgrad_tiled<T, CI, CO>), grid, dim3(kBlock), 0, st, (const T *)feat, (const T *)dout, nbr,  \
                           n_out, kvol, chunk, dw);                                                                       \
        return true;                                                                                                      \
    }
    SEC_WG(4, 16) SEC_WG(16, 16) SEC_WG(16, 32) SEC_WG(32, 32) SEC_WG(32, 64) SEC_WG(64, 64) SEC_WG(64, 128) SEC_WG(128, 128)
#undef SEC_WG
    return false;
}

static size_t elt_size(int dtype) { return dtype == SEC_F32 ? 4 : 2; }

}  // namespace sec

using namespace sec;

#ifdef SEC_CONV_TIMELINE
extern "C" __attribute__((visibility("default"))) int sec__debug_timeline(long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sec::g_timeline), &buf, sizeof(buf)) == hipSuccess ? 0 : -4;
}
#endif

SEC_API size_t sec_packed_weight_bytes(int kvol, int cin, int cout, int dtype) {
    if (dtype != SEC_F32 && cin == 4 && cout == 16 && kvol == 27) return (size_t)7 * 64 * 8 * elt_size(dtype);   // k_conv_c4_mfma
    if (dtype == SEC_F32 || cin % 16 != 0 || kvol <= 0 || cout <= 0) return 0;
    return (size_t)kvol * cin * ((cout + 31) / 32) * 32 * elt_size(dtype);
}

SEC_API int sec_pack_conv_weight(const void *weight, int kvol, int cin, int cout, int dtype, void *packed, void *stream) {
    if (!weight || !packed || sec_packed_weight_bytes(kvol, cin, cout, dtype) == 0) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 4) {
        if (dtype == SEC_BF16)
            hipLaunchKernelGGL(k_pack_weight_c4<__hip_bfloat16>, dim3(div_up(7 * 64 * 8, kBlock)), dim3(kBlock), 0, st,
                               (const __hip_bfloat16 *)weight, cout, (__hip_bfloat16 *)packed);
        else
            hipLaunchKernelGGL(k_pack_weight_c4<__half>, dim3(div_up(7 * 64 * 8, kBlock)), dim3(kBlock), 0, st, (const __half *)weight, cout,
                               (__half *)packed);
        return check_launch();
    }
    long long total = (long long)kvol * cin * ((cout + 31) / 32) * 32;
    if (dtype == SEC_BF16)
        hipLaunchKernelGGL(k_pack_weight<__hip_bfloat16>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st,
                           (const __hip_bfloat16 *)weight, kvol, cin, cout, (__hip_bfloat16 *)packed);
    else
        hipLaunchKernelGGL(k_pack_weight<__half>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, (const __half *)weight,
                           kvol, cin, cout, (__half *)packed);
    return check_launch();
}

SEC_API int sec_set_fp32_mode(int mode) {
    if (mode != 0 && mode != 1) return SEC_E_INVALID;
    g_fp32_mode = mode;
    return SEC_OK;
}
SEC_API int sec_get_fp32_mode(void) { return g_fp32_mode; }

SEC_API int sec_indice_conv_set_variant(int variant) {
    g_variant_override = variant;      // < 0: back to SEC_CONV_VARIANT / the automatic choice
    return SEC_OK;
}

SEC_API int sec_indice_conv_fwd_plan(int cin, int cout, int kvol, int n_out, int dtype, int out_dtype, int has_packed) {
    static const int mfma_shapes[][2] = {{16, 16}, {16, 32}, {32, 32}, {32, 64}, {64, 64}, {64, 128}, {128, 128}, {16, 64},
                                         {64, 32}, {32, 16}, {128, 64}};
    bool mfma = false;
    for (auto &sh : mfma_shapes) mfma |= sh[0] == cin && sh[1] == cout;
    if (has_packed && dtype != SEC_F32 && mfma) {
        const int rp = rows_plan(cin, cout, kvol, n_out, dtype == out_dtype);
        if (rp) return rp;
        const int v = conv_variant();
#ifdef SEC_CONV_EXPERIMENTS
        if (v >= 2 && v <= 7) return PLAN_EXPERIMENT;
#endif
        if (v == 8 || ((v == 1 || v == 29) && cout <= 32)) return PLAN_MFMA_SKS;
        return v >= 1 ? PLAN_MFMA_SK : PLAN_MFMA_WAVE;
    }
    if (has_packed && dtype != SEC_F32 && cin == 4 && cout == 16 && kvol == 27 && out_dtype == dtype && conv_variant() != 29) return PLAN_C4_MFMA;
    if (cin == 4 && cout == 16 && (size_t)kvol * 4 * 16 * sizeof(float) <= 48 * 1024) return PLAN_C4;
    static const int tiled_shapes[][2] = {{16, 16}, {16, 32}, {32, 16}, {32, 32}, {32, 64}, {64, 32}, {64, 64}};
    for (auto &sh : tiled_shapes)
        if (sh[0] == cin && sh[1] == cout) return PLAN_TILED;
    return PLAN_GENERIC;
}

SEC_API size_t sec_packed_weight_x3_bytes(int kvol, int cin, int cout) {
    return x3p_shape(cin, cout, kvol) ? (size_t)kvol * 2 * cin * ((cout + 31) / 32 * 32) * 2 : 0;
}

SEC_API int sec_indice_conv_fwd(const void *features, int n_in, int cin, const void *weight, const void *packed_weight,
                                int kvol, int cout, const int *nbr_out, int n_out, const int *num_out_dev,
                                const float *scale, const float *shift, int relu, void *out, int dtype, int out_dtype,
                                void *stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || !weight) return SEC_E_INVALID;
    if (dtype < 0 || dtype > 2 || (out_dtype != dtype && out_dtype != SEC_F32)) return SEC_E_UNSUPPORTED;
    if (n_out == 0) return SEC_OK;                 // an empty batch: empty tensors carry null pointers
    if (!nbr_out || !out || (n_in > 0 && !features)) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    if (packed_weight && dtype != SEC_F32 && cin == 4 && cout == 16 && kvol == 27 && out_dtype == dtype && conv_variant() != 29 &&
        (long long)n_in * 8 < 0x7fffffffll) {
        const long long fb = (long long)n_in * 8;
        const int c4_xcd = rows_xcd_order();
        const int c4_blocks = c4_xcd ? (div_up(n_out, 128) + 7) / 8 * 8 : div_up(n_out, 128);
        const int c4_relu = (relu & 1) | (c4_xcd << 16);
        if (dtype == SEC_BF16)
            hipLaunchKernelGGL((k_conv_c4_mfma<__hip_bfloat16, 16>), dim3(c4_blocks), dim3(kBlock), 0, st, (const __hip_bfloat16 *)features,
                               fb, (const __hip_bfloat16 *)packed_weight, nbr_out, n_out, num_out_dev, scale, shift, c4_relu, (__hip_bfloat16 *)out);
        else
            hipLaunchKernelGGL((k_conv_c4_mfma<__half, 16>), dim3(c4_blocks), dim3(kBlock), 0, st, (const __half *)features, fb,
                               (const __half *)packed_weight, nbr_out, n_out, num_out_dev, scale, shift, c4_relu, (__half *)out);
        return check_launch();
    }
    if (packed_weight && dtype != SEC_F32 && cin % 16 == 0) {
        if (dtype == SEC_BF16) {
            done = out_dtype == SEC_F32
                       ? dispatch_mfma<__hip_bfloat16, float>(cin, cout, features, n_in, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st)
                       : dispatch_mfma<__hip_bfloat16, __hip_bfloat16>(cin, cout, features, n_in, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st);
        } else {
            done = out_dtype == SEC_F32
                       ? dispatch_mfma<__half, float>(cin, cout, features, n_in, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st)
                       : dispatch_mfma<__half, __half>(cin, cout, features, n_in, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st);
        }
    }
    // fp32 features with an x3-packed weight (ops.pack_weight of an fp32 weight: [k][hi | lo] bf16 B-fragment pieces): the pipelined
    // split-operand kernel; variants 30 / 31 (VALU / fp32-MFMA forms) and 32 (the unpacked split form) ignore the packed image
    if (!done && packed_weight && dtype == SEC_F32 && out_dtype == SEC_F32 && conv_variant() != 30 && conv_variant() != 31 && conv_variant() != 32 &&
        g_fp32_mode == 0)
        done = launch_x3p(features, n_in, packed_weight, nbr_out, n_out, num_out_dev, cin, cout, kvol, scale, shift, relu, out, st);
    if (!done) {
#define SEC_GEN(T, OT) launch_generic<T, OT>(features, weight, nbr_out, n_out, num_out_dev, cin, cout, kvol, scale, shift, relu, out, st)
        if (dtype == SEC_F32) SEC_GEN(float, float);
        else if (dtype == SEC_BF16) { if (out_dtype == SEC_F32) SEC_GEN(__hip_bfloat16, float); else SEC_GEN(__hip_bfloat16, __hip_bfloat16); }
        else { if (out_dtype == SEC_F32) SEC_GEN(__half, float); else SEC_GEN(__half, __half); }
#undef SEC_GEN
    }
    return check_launch();
}

// fp32 master weight -> the three 16-bit images a mixed-precision training step reads, in ONE launch: the plain [k][ci][co]
// rounding (the weight gradient kernels' shape reference and the VALU fall-backs), the forward MFMA image (k_pack_weight, or
// k_pack_weight_c4 for the 4-channel first layer) and the data-gradient image (k_pack_weight_t).  Replaces to(dtype) + pack in the
// forward and the transposed pack in the backward: three launches per layer and step.
template <typename T>
__device__ __forceinline__ void pack_weight_train_at(long long g, const float *__restrict__ w, int kvol, int cin, int cout, int mirror,
                                                     T *__restrict__ w16, T *__restrict__ packed, long long total_fwd,
                                                     T *__restrict__ packed_t, long long total_t, float *__restrict__ zero_dw) {
    const long long total0 = (long long)kvol * cin * cout;
    if (g < total0) w16[g] = Cvt<T>::from(w[g]);
    if (zero_dw && g < total0) zero_dw[g] = 0.0f;              // the accumulator of this step's weight gradient (sec_indice_conv_bwd, dweight_zeroed)
    const int e = (int)(g & 7), lane = (int)((g >> 3) & 63);
    if (packed && g < total_fwd) {
        if (cin == 4) {                                         // k_pack_weight_c4
            const int s_ = (int)(g >> 9);
            const int K = s_ * 16 + (lane >> 5) * 8 + e, kk = K >> 2, ci = K & 3, c = lane & 31;
            packed[g] = (kk < 27 && c < cout) ? Cvt<T>::from(w[((size_t)kk * 4 + ci) * cout + c]) : Cvt<T>::from(0.0f);
        } else {                                                // k_pack_weight
            const int ks = cin / 16, nt = (cout + 31) / 32;
            long long q = g >> 9;
            const int t = (int)(q % nt);
            q /= nt;
            const int sidx = (int)(q % ks), k = (int)(q / ks);
            const int ci = sidx * 16 + (lane >> 5) * 8 + e, co = t * 32 + (lane & 31);
            packed[g] = co < cout ? Cvt<T>::from(w[((size_t)k * cin + ci) * cout + co]) : Cvt<T>::from(0.0f);
        }
    }
    if (packed_t && g < total_t) {                              // k_pack_weight_t
        const int ks = cout / 16, nt = (cin + 31) / 32;
        long long q = g >> 9;
        const int t = (int)(q % nt);
        q /= nt;
        const int sidx = (int)(q % ks), k = (int)(q / ks);
        const int co = sidx * 16 + (lane >> 5) * 8 + e, ci = t * 32 + (lane & 31);
        const int km = mirror ? kvol - 1 - k : k;
        packed_t[g] = ci < cin ? Cvt<T>::from(w[((size_t)km * cin + ci) * cout + co]) : Cvt<T>::from(0.0f);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_weight_train(const float *__restrict__ w, int kvol, int cin, int cout, int mirror,
                                                             T *__restrict__ w16, T *__restrict__ packed, long long total_fwd,
                                                             T *__restrict__ packed_t, long long total_t, float *__restrict__ zero_dw) {
    pack_weight_train_at<T>((long long)blockIdx.x * kBlock + threadIdx.x, w, kvol, cin, cout, mirror, w16, packed, total_fwd, packed_t, total_t, zero_dw);
}

// every layer of a network in ONE launch: the descriptors travel as kernel arguments (no device table to fill -- capturable as is)
constexpr int kPackMulti = 16;
struct PackSparseDesc {
    const float *w;
    void *w16, *packed, *packed_t;
    float *zero_dw;
    long long total_fwd, total_t;
    int kvol, cin, cout, mirror, blk0;
};
struct PackSparseArgs {
    PackSparseDesc d[kPackMulti];
    int n;
};
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_weight_train_multi(PackSparseArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.d[i + 1].blk0) ++i;      // wave-uniform: blockIdx only
    const PackSparseDesc &d = a.d[i];
    pack_weight_train_at<T>((long long)(blockIdx.x - d.blk0) * kBlock + threadIdx.x, d.w, d.kvol, d.cin, d.cout, d.mirror, (T *)d.w16, (T *)d.packed,
                            d.total_fwd, (T *)d.packed_t, d.total_t, d.zero_dw);
}

template <typename T>
static int run_bwd(const void *features, int n_in, int cin, const void *weight, int kvol, int cout, const int *nbr_out,
                   const int *nbr_in, int n_out, const void *dout, void *dfeat, float *dweight, void *workspace,
                   size_t workspace_bytes, int dtype, hipStream_t st, const void *packed_dgrad = nullptr, bool dweight_zeroed = false) {
    int rc;
    if (dfeat && n_in > 0) {
        const int *tbl = nbr_in ? nbr_in : nbr_out;  // SubM: nbr_in is the mirror image of nbr_out
        bool done = false;
        if constexpr (!std::is_same<T, float>::value) {
            // MFMA path: forward kernels on (dout, Wt) with Cin <-> Cout swapped
            const size_t need = sec_packed_weight_bytes(kvol, cout, cin, dtype);
            if (need > 0 && packed_dgrad && n_out > 0) {   // the caller packed it already (sec_pack_conv_weight_train)
                done = dispatch_mfma<T, T>(cout, cin, dout, n_out, packed_dgrad, tbl, n_in, nullptr, kvol, nullptr, nullptr, 0, dfeat, st);
            } else if (need > 0 && workspace && workspace_bytes >= need && n_out > 0) {
                long long total = (long long)kvol * cout * ((cin + 31) / 32) * 32;
                hipLaunchKernelGGL(k_pack_weight_t<T>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, (const T *)weight, kvol, cin,
                                   cout, nbr_in ? 0 : 1, (T *)workspace);
                done = dispatch_mfma<T, T>(cout, cin, dout, n_out, workspace, tbl, n_in, nullptr, kvol, nullptr, nullptr, 0, dfeat, st);
            }
        }
        if (!done)   // register-tiled VALU forward on (dout, W^T): Cin <-> Cout swapped, weights read transposed in place
            done = launch_tiled<T, T>(dout, weight, tbl, n_in, nullptr, cout, cin, kvol, 1, nbr_in ? 0 : 1, nullptr, nullptr, 0,
                                      dfeat, st);
        if (!done) {
            long long total = (long long)n_in * cin;
            int blocks = div_up(total, kBlock);
            if (blocks > 256 * 64) blocks = 256 * 64;
            hipLaunchKernelGGL(k_conv_dgrad<T>, dim3(blocks), dim3(kBlock), 0, st, (const T *)dout, (const T *)weight, tbl,
                               nbr_in ? 0 : 1, n_in, cin, cout, kvol, (T *)dfeat);
        }
    }
    if (dweight) {
        // (dweight_zeroed: the forward's sec_pack_conv_weight_train zeroed it -- one memset node per layer and step less)
        if (!dweight_zeroed && (rc = fill_words(dweight, (size_t)kvol * cin * cout * sizeof(float), 0u, st))) return rc;
        if (n_out > 0 && !launch_wgrad_tiled<T>(features, dout, nbr_out, n_out, cin, cout, kvol, dweight, st)) {
            int rows_per_chunk = 512;
            hipLaunchKernelGGL(k_conv_wgrad<T>, dim3(div_up(n_out, rows_per_chunk), kvol), dim3(kBlock), 0, st,
                               (const T *)features, (const T *)dout, nbr_out, n_out, cin, cout, kvol, rows_per_chunk, dweight);
        }
    }
    return check_launch();
}

SEC_API size_t sec_indice_conv_bwd_workspace_bytes(int kvol, int cin, int cout, int dtype) {
    return sec_packed_weight_bytes(kvol, cout, cin, dtype);   // the transposed packed weights of the MFMA dgrad (0 for fp32)
}

SEC_API int sec_indice_conv_bwd(const void *features, int n_in, int cin, const void *weight, int kvol, int cout,
                                const int *nbr_out, const int *nbr_in, int n_out, const void *dout, void *dfeat,
                                float *dweight, int dtype, void *workspace, size_t workspace_bytes, const void *packed_dgrad,
                                int dweight_zeroed, void *stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || !weight || !nbr_out || !dout) return SEC_E_INVALID;
    if (!nbr_in && n_in != n_out) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32) return run_bwd<float>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, workspace, workspace_bytes, dtype, st, nullptr, dweight_zeroed != 0);
    if (dtype == SEC_F16) return run_bwd<__half>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, workspace, workspace_bytes, dtype, st, packed_dgrad, dweight_zeroed != 0);
    if (dtype == SEC_BF16) return run_bwd<__hip_bfloat16>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, workspace, workspace_bytes, dtype, st, packed_dgrad, dweight_zeroed != 0);
    return SEC_E_UNSUPPORTED;
}

SEC_API int sec_pack_conv_weight_train(const float *weight, int kvol, int cin, int cout, int subm, int dtype, void *weight16,
                                       void *packed_fwd, void *packed_dgrad, float *zero_dweight, void *stream) {
    if (!weight || !weight16 || kvol <= 0 || cin <= 0 || cout <= 0 || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_INVALID;
    const long long total0 = (long long)kvol * cin * cout;
    const long long total_fwd = packed_fwd ? (long long)(sec_packed_weight_bytes(kvol, cin, cout, dtype) / 2) : 0;
    const long long total_t = packed_dgrad ? (long long)(sec_packed_weight_bytes(kvol, cout, cin, dtype) / 2) : 0;
    if ((packed_fwd && total_fwd == 0) || (packed_dgrad && (total_t == 0 || cout % 16))) return SEC_E_UNSUPPORTED;
    long long total = total0 > total_fwd ? total0 : total_fwd;
    if (total_t > total) total = total_t;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16)
        hipLaunchKernelGGL(k_pack_weight_train<__hip_bfloat16>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, weight, kvol, cin, cout, subm ? 1 : 0,
                           (__hip_bfloat16 *)weight16, (__hip_bfloat16 *)packed_fwd, total_fwd, (__hip_bfloat16 *)packed_dgrad, total_t, zero_dweight);
    else
        hipLaunchKernelGGL(k_pack_weight_train<__half>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, weight, kvol, cin, cout, subm ? 1 : 0,
                           (__half *)weight16, (__half *)packed_fwd, total_fwd, (__half *)packed_dgrad, total_t, zero_dweight);
    return check_launch();
}

SEC_API int sec_pack_conv_weight_train_multi(int n, const float *const *weights, const int *kvol, const int *cin, const int *cout, const int *subm,
                                             int dtype, void *const *weight16, void *const *packed_fwd, void *const *packed_dgrad,
                                             float *const *zero_dweight, void *stream) {
    if (n <= 0 || !weights || !kvol || !cin || !cout || !subm || !weight16 || !packed_fwd || !packed_dgrad || !zero_dweight ||
        (dtype != SEC_BF16 && dtype != SEC_F16))
        return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += kPackMulti) {
        PackSparseArgs a;
        a.n = n - i0 < kPackMulti ? n - i0 : kPackMulti;
        long long blocks = 0;
        for (int j = 0; j < a.n; ++j) {
            const int i = i0 + j;
            if (!weights[i] || !weight16[i] || kvol[i] <= 0 || cin[i] <= 0 || cout[i] <= 0) return SEC_E_INVALID;
            const long long total0 = (long long)kvol[i] * cin[i] * cout[i];
            const long long total_fwd = packed_fwd[i] ? (long long)(sec_packed_weight_bytes(kvol[i], cin[i], cout[i], dtype) / 2) : 0;
            const long long total_t = packed_dgrad[i] ? (long long)(sec_packed_weight_bytes(kvol[i], cout[i], cin[i], dtype) / 2) : 0;
            if ((packed_fwd[i] && total_fwd == 0) || (packed_dgrad[i] && (total_t == 0 || cout[i] % 16))) return SEC_E_UNSUPPORTED;
            long long total = total0 > total_fwd ? total0 : total_fwd;
            if (total_t > total) total = total_t;
            PackSparseDesc &d = a.d[j];
            d.w = weights[i]; d.w16 = weight16[i]; d.packed = packed_fwd[i]; d.packed_t = packed_dgrad[i]; d.zero_dw = zero_dweight[i];
            d.total_fwd = total_fwd; d.total_t = total_t;
            d.kvol = kvol[i]; d.cin = cin[i]; d.cout = cout[i]; d.mirror = subm[i] ? 1 : 0; d.blk0 = (int)blocks;
            blocks += div_up(total, kBlock);
        }
        if (dtype == SEC_BF16) hipLaunchKernelGGL(k_pack_weight_train_multi<__hip_bfloat16>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL(k_pack_weight_train_multi<__half>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
    }
    return check_launch();
}
