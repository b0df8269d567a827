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

template <typename T, typename OT, int CIN, int COUT>
static void launch_mfma(const void *feat, const void *packed, const int *nbr, int n_out, const int *num_out_dev,
                        int kvol, const float *scale, const float *shift, int relu, void *out, hipStream_t st) {
    constexpr int MT = 1;
    int rows_per_block = (kBlock / 64) * 32 * MT;
    hipLaunchKernelGGL((k_conv_mfma<T, OT, CIN, COUT, MT>), dim3(div_up(n_out, rows_per_block)), dim3(kBlock), 0, st,
                       (const T *)feat, (const T *)packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, (OT *)out);
}

template <typename T, typename OT>
static bool dispatch_mfma(int cin, int cout, const void *feat, const void *packed, const int *nbr, int n_out,
                          const int *num_out_dev, int kvol, const float *scale, const float *shift, int relu, void *out,
                          hipStream_t st) {
#define SEC_CASE(CI, CO)                                                                                         \
    if (cin == CI && cout == CO) {                                                                               \
        launch_mfma<T, OT, CI, CO>(feat, packed, nbr, n_out, num_out_dev, kvol, scale, shift, relu, out, st);     \
        return true;                                                                                             \
    }
    SEC_CASE(16, 16) SEC_CASE(16, 32) SEC_CASE(32, 32) SEC_CASE(32, 64) SEC_CASE(64, 64) SEC_CASE(64, 128)
    SEC_CASE(128, 128) SEC_CASE(16, 64) SEC_CASE(64, 32) SEC_CASE(32, 16) SEC_CASE(128, 64)
#undef SEC_CASE
    return false;
}

template <typename T, typename OT>
static void launch_generic(const void *feat, const void *w, const int *nbr, int n_out, const int *num_out_dev, int cin,
                           int cout, int kvol, const float *scale, const float *shift, int relu, void *out,
                           hipStream_t st) {
    long long total = (long long)n_out * cout;
    int blocks = div_up(total, kBlock);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((k_conv_generic<T, OT>), dim3(blocks), dim3(kBlock), 0, st, (const T *)feat, (const T *)w, nbr,
                       n_out, num_out_dev, cin, cout, kvol, scale, shift, relu, (OT *)out);
}

// ------------------------------------------------------------------ backward (correctness-first VALU kernels)
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

static size_t elt_size(int dtype) { return dtype == SEC_F32 ? 4 : 2; }

}  // namespace sec

using namespace sec;

SEC_API size_t sec_packed_weight_bytes(int kvol, int cin, int cout, int dtype) {
    if (dtype == SEC_F32 || cin % 16 != 0 || kvol <= 0 || cout <= 0) return 0;
    return (size_t)kvol * cin * ((cout + 31) / 32) * 32 * elt_size(dtype);
}

SEC_API int sec_pack_conv_weight(const void *weight, int kvol, int cin, int cout, int dtype, void *packed, void *stream) {
    if (!weight || !packed || sec_packed_weight_bytes(kvol, cin, cout, dtype) == 0) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    long long total = (long long)kvol * cin * ((cout + 31) / 32) * 32;
    if (dtype == SEC_BF16)
        hipLaunchKernelGGL(k_pack_weight<__hip_bfloat16>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st,
                           (const __hip_bfloat16 *)weight, kvol, cin, cout, (__hip_bfloat16 *)packed);
    else
        hipLaunchKernelGGL(k_pack_weight<__half>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, (const __half *)weight,
                           kvol, cin, cout, (__half *)packed);
    return check_launch();
}

SEC_API int sec_indice_conv_fwd(const void *features, int n_in, int cin, const void *weight, const void *packed_weight,
                                int kvol, int cout, const int *nbr_out, int n_out, const int *num_out_dev,
                                const float *scale, const float *shift, int relu, void *out, int dtype, int out_dtype,
                                void *stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || !weight || !nbr_out || !out ||
        (n_in > 0 && !features))
        return SEC_E_INVALID;
    if (dtype < 0 || dtype > 2 || (out_dtype != dtype && out_dtype != SEC_F32)) return SEC_E_UNSUPPORTED;
    if (n_out == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    if (packed_weight && dtype != SEC_F32) {
        if (dtype == SEC_BF16) {
            done = out_dtype == SEC_F32
                       ? dispatch_mfma<__hip_bfloat16, float>(cin, cout, features, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st)
                       : dispatch_mfma<__hip_bfloat16, __hip_bfloat16>(cin, cout, features, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st);
        } else {
            done = out_dtype == SEC_F32
                       ? dispatch_mfma<__half, float>(cin, cout, features, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st)
                       : dispatch_mfma<__half, __half>(cin, cout, features, packed_weight, nbr_out, n_out, num_out_dev, kvol, scale, shift, relu, out, st);
        }
    }
    if (!done) {
#define SEC_GEN(T, OT) launch_generic<T, OT>(features, weight, nbr_out, n_out, num_out_dev, cin, cout, kvol, scale, shift, relu, out, st)
        if (dtype == SEC_F32) SEC_GEN(float, float);
        else if (dtype == SEC_BF16) { if (out_dtype == SEC_F32) SEC_GEN(__hip_bfloat16, float); else SEC_GEN(__hip_bfloat16, __hip_bfloat16); }
        else { if (out_dtype == SEC_F32) SEC_GEN(__half, float); else SEC_GEN(__half, __half); }
#undef SEC_GEN
    }
    return check_launch();
}

template <typename T>
static int run_bwd(const void *features, int n_in, int cin, const void *weight, int kvol, int cout, const int *nbr_out,
                   const int *nbr_in, int n_out, const void *dout, void *dfeat, float *dweight, hipStream_t st) {
    int rc;
    if (dfeat && n_in > 0) {
        long long total = (long long)n_in * cin;
        int blocks = div_up(total, kBlock);
        if (blocks > 256 * 64) blocks = 256 * 64;
        const int *tbl = nbr_in ? nbr_in : nbr_out;  // SubM: nbr_in is the mirror image of nbr_out
        hipLaunchKernelGGL(k_conv_dgrad<T>, dim3(blocks), dim3(kBlock), 0, st, (const T *)dout, (const T *)weight, tbl,
                           nbr_in ? 0 : 1, n_in, cin, cout, kvol, (T *)dfeat);
    }
    if (dweight) {
        if ((rc = hip_ok(hipMemsetAsync(dweight, 0, (size_t)kvol * cin * cout * sizeof(float), st)))) return rc;
        if (n_out > 0) {
            int rows_per_chunk = 512;
            hipLaunchKernelGGL(k_conv_wgrad<T>, dim3(div_up(n_out, rows_per_chunk), kvol), dim3(kBlock), 0, st,
                               (const T *)features, (const T *)dout, nbr_out, n_out, cin, cout, kvol, rows_per_chunk, dweight);
        }
    }
    return check_launch();
}

SEC_API int sec_indice_conv_bwd(const void *features, int n_in, int cin, const void *weight, int kvol, int cout,
                                const int *nbr_out, const int *nbr_in, int n_out, const void *dout, void *dfeat,
                                float *dweight, int dtype, void *stream) {
    if (n_in < 0 || n_out < 0 || cin <= 0 || cout <= 0 || kvol <= 0 || !weight || !nbr_out || !dout) return SEC_E_INVALID;
    if (!nbr_in && n_in != n_out) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32) return run_bwd<float>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, st);
    if (dtype == SEC_F16) return run_bwd<__half>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, st);
    if (dtype == SEC_BF16) return run_bwd<__hip_bfloat16>(features, n_in, cin, weight, kvol, cout, nbr_out, nbr_in, n_out, dout, dfeat, dweight, st);
    return SEC_E_UNSUPPORTED;
}
