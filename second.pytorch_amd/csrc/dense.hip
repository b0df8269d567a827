// Dense-side helpers for the RPN (second/pytorch/models/rpn.py:468-497): the convolutions themselves run
// through MIOpen in phase 1; the per-channel bias (folded BatchNorm2d) + ReLU that follows every conv is ONE
// in-place pass here instead of MIOpen's separate bias tensor-op plus a ReLU kernel (3 passes -> 1).
#include "common.hpp"
#include <stdlib.h>

namespace sec {

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ __hip_bfloat16 from_f(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half from_f(float v) { return __float2half_rn(v); }

// x: [pixels, C] channels-last, 16-bit; 8 channels (16 bytes) per thread; C % 8 == 0
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bias_act16(T *__restrict__ x, const float *__restrict__ bias, long long n_vec,
                                                      int c_vec, int relu) {
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < n_vec; g += (long long)gridDim.x * kBlock) {
        uint4 v = reinterpret_cast<uint4 *>(x)[g];
        T *e = reinterpret_cast<T *>(&v);
        const float *b = bias + (size_t)(g % c_vec) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f = to_f<T>(e[i]) + b[i];
            if (relu) f = f > 0.0f ? f : 0.0f;
            e[i] = from_f<T>(f);
        }
        reinterpret_cast<uint4 *>(x)[g] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_bias_act32(float *__restrict__ x, const float *__restrict__ bias, long long n_vec,
                                                      int c_vec, int relu) {
    for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < n_vec; g += (long long)gridDim.x * kBlock) {
        float4 v = reinterpret_cast<float4 *>(x)[g];
        const float *b = bias + (size_t)(g % c_vec) * 4;
        v.x += b[0]; v.y += b[1]; v.z += b[2]; v.w += b[3];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4 *>(x)[g] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Dense conv2d for the RPN (second/pytorch/models/rpn.py:468-497, 386-420): NHWC bf16/f16 implicit GEMM on MFMA
// with the folded-BatchNorm bias and ReLU fused into the epilogue (MIOpen needs a zero-fill tensor op + the
// conv + a separate bias/ReLU pass per layer).
//   M = output pixels, N = Cout, K = taps x Cin.  Workgroup tile 128 pixels x BN couts, 4 waves as 2 x 2, each
//   wave (64 x BN/2) = 2 x (BN/64) MFMA 32x32x16 tiles.  K is walked one (tap, 64-channel slab) at a time:
//   the A slab [128 px][64 ch] (16 KB, halo / padding resolved per row, whole 128-byte lines per 8 lanes) and the
//   B slab [8 chunks][BN][8] (pre-packed so it is a straight copy) are register-staged one slab ahead and
//   double-buffered in LDS (XOR swizzle on the A chunk index => conflict-free ds_read_b128).
typedef float f32x16d __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
template <typename T> struct MfmaD;
template <> struct MfmaD<__hip_bfloat16> {
    static __device__ __forceinline__ f32x16d run(uint4 a, uint4 b, f32x16d c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8d, a), __builtin_bit_cast(bf16x8d, b), c, 0, 0, 0);
    }
};
template <> struct MfmaD<__half> {
    static __device__ __forceinline__ f32x16d run(uint4 a, uint4 b, f32x16d c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8d, a), __builtin_bit_cast(f16x8d, b), c, 0, 0, 0);
    }
};

struct Conv2dParams {
    int batch, h, w, cin, cout, ho, wo, ksize, stride, pad, relu;
    long long m;  // batch * ho * wo
};

// packed[((tap * cin/8 + chunk) * cout + n) * 8 + e] = w[n][chunk*8 + e][tap / ks][tap % ks]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv2d_pack(const T *__restrict__ w, int cout, int cin, int ks, T *__restrict__ packed) {
    long long total = (long long)ks * ks * cin * cout;
    long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (g >= total) return;
    int e = (int)(g & 7);
    long long q = g >> 3;
    int n = (int)(q % cout);
    q /= cout;
    int chunk = (int)(q % (cin / 8));
    int tap = (int)(q / (cin / 8));
    int ci = chunk * 8 + e;
    packed[g] = w[(((size_t)n * cin + ci) * ks + tap / ks) * ks + tap % ks];
}

template <typename T, int BN>
__global__ __launch_bounds__(kBlock) void k_conv2d_nhwc(const T *__restrict__ x, const T *__restrict__ wpk,
                                                       const float *__restrict__ bias, T *__restrict__ y, Conv2dParams p) {
    constexpr int BM = 128, NTW = BN / 64;          // n-tiles per wave
    constexpr int PERB = BN * 8 / kBlock;           // B uint4 per thread per slab
    __shared__ uint4 sA[2][BM * 8];
    __shared__ uint4 sB[2][8 * BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wv & 1, wn = wv >> 1;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int cin8 = p.cin / 8, CC = p.cin / 64, NIT = p.ksize * p.ksize * CC;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w4 = reinterpret_cast<const uint4 *>(wpk);

    // A rows owned by this thread: pixels tid/8 + 32 j (j = 0..3), 16-byte chunk tid % 8
    const int chunk = tid & 7;
    long long abase[4];
    int iy0[4], ix0[4];
    bool pval[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long pix = m0 + (tid >> 3) + 32 * j;
        pval[j] = pix < p.m;
        long long q = pval[j] ? pix : 0;
        int ox = (int)(q % p.wo);
        q /= p.wo;
        int oy = (int)(q % p.ho);
        int b = (int)(q / p.ho);
        iy0[j] = oy * p.stride - p.pad;
        ix0[j] = ox * p.stride - p.pad;
        abase[j] = ((long long)b * p.h + iy0[j]) * p.w + ix0[j];
    }
    uint4 ra[4], rb[PERB];
    auto load = [&](int it) {
        const int tap = it / CC, cc = it - tap * CC;
        const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + dy, ix = ix0[j] + dx;
            const bool ok = pval[j] && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = x4[(abase[j] + (long long)dy * p.w + dx) * cin8 + cc * 8 + chunk];
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < PERB; ++j) {
            const int e = tid + j * kBlock, ch = e / BN, n = e - ch * BN;
            rb[j] = w4[((size_t)tap * cin8 + cc * 8 + ch) * p.cout + n0 + n];
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pl = (tid >> 3) + 32 * j;
            sA[buf][pl * 8 + (chunk ^ (pl & 7))] = ra[j];
        }
#pragma unroll
        for (int j = 0; j < PERB; ++j) sB[buf][tid + j * kBlock] = rb[j];
    };

    f32x16d acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    load(0);
    store(0);
    __syncthreads();
    for (int it = 0; it < NIT; ++it) {
        const int buf = it & 1;
        if (it + 1 < NIT) load(it + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bf[NTW];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = sA[buf][(wm * 64 + mt * 32 + r) * 8 + ((s * 2 + hh) ^ (r & 7))];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bf[nt] = sB[buf][(s * 2 + hh) * BN + wn * (BN / 2) + nt * 32 + r];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MfmaD<T>::run(af[mt], bf[nt], acc[mt][nt]);
        }
        if (it + 1 < NIT) store(buf ^ 1);
        __syncthreads();
    }
    // epilogue: C/D layout col = lane & 31 (cout), row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5) (pixel)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + wn * (BN / 2) + nt * 32 + r;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long pix = m0 + wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (pix < p.m) {
                    float v = acc[mt][nt][i] + bv;
                    if (p.relu) v = v > 0.0f ? v : 0.0f;
                    y[(size_t)pix * p.cout + co] = from_f<T>(v);
                }
            }
    }
}

// Same tiling, but both slabs travel global -> LDS with the asynchronous LDS-DMA (global_load_lds_dwordx4): no
// staging VGPRs, no ds_write pass.  The DMA writes LDS linearly in lane order, so the XOR swizzle of the A slab is
// applied to the SOURCE chunk index (guide rule 21); rows that fall into the zero padding read a 16-byte zero
// block appended to the packed weights (an exec-masked lane would leave stale LDS bytes).
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

template <typename T, int BN>
__global__ __launch_bounds__(kBlock) void k_conv2d_nhwc_dma(const T *__restrict__ x, const T *__restrict__ wpk,
                                                           const float *__restrict__ bias, T *__restrict__ y, Conv2dParams p) {
    constexpr int BM = 128, NTW = BN / 64;
    constexpr int PERB = BN * 8 / kBlock;
    __shared__ uint4 sA[2][BM * 8];
    __shared__ uint4 sB[2][8 * BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wv & 1, wn = wv >> 1;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD x owns the contiguous tile range
    // [x * per, (x+1) * per), i.e. a band of image rows, so the 3x3 halo re-reads hit that XCD's own L2
    const int per = gridDim.x / 8;
    const long long m0 = (long long)((blockIdx.x % 8) * per + blockIdx.x / 8) * BM;
    if (m0 >= p.m) return;
    const int n0 = blockIdx.y * BN;
    const int cin8 = p.cin / 8, CC = p.cin / 64, NIT = p.ksize * p.ksize * CC;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w4 = reinterpret_cast<const uint4 *>(wpk);
    const uint4 *zero16 = w4 + (size_t)p.ksize * p.ksize * cin8 * p.cout;   // appended by sec_conv2d_pack_weight

    // DMA instruction j of wave wv fills LDS entries [(j*4 + wv)*64, +64): pixel (j*4+wv)*8 + lane/8, slot lane%8
    const int slot = lane & 7;
    long long abase[4];
    int iy0[4], ix0[4], srcchunk[4];
    bool pval[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pl = (j * 4 + wv) * 8 + (lane >> 3);
        srcchunk[j] = slot ^ (pl & 7);
        long long pix = m0 + pl;
        pval[j] = pix < p.m;
        long long q = pval[j] ? pix : 0;
        int ox = (int)(q % p.wo);
        q /= p.wo;
        int oy = (int)(q % p.ho);
        int b = (int)(q / p.ho);
        iy0[j] = oy * p.stride - p.pad;
        ix0[j] = ox * p.stride - p.pad;
        abase[j] = ((long long)b * p.h + iy0[j]) * p.w + ix0[j];
    }
    auto issue = [&](int it, int buf) {
        const int tap = it / CC, cc = it - tap * CC;
        const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + dy, ix = ix0[j] + dx;
            const bool ok = pval[j] && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const uint4 *src = ok ? x4 + (abase[j] + (long long)dy * p.w + dx) * cin8 + cc * 8 + srcchunk[j] : zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sA[buf][(j * 4 + wv) * 64], 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < PERB; ++j) {
            const int e = (j * 4 + wv) * 64 + lane, ch = e / BN, n = e - ch * BN;
            const uint4 *src = w4 + ((size_t)tap * cin8 + cc * 8 + ch) * p.cout + n0 + n;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sB[buf][(j * 4 + wv) * 64], 16, 0, 0);
        }
    };

    f32x16d acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    issue(0, 0);
    __syncthreads();   // hipcc drains the DMA (vmcnt(0)) before the barrier
    for (int it = 0; it < NIT; ++it) {
        const int buf = it & 1;
        if (it + 1 < NIT) issue(it + 1, buf ^ 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bf[NTW];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = sA[buf][(wm * 64 + mt * 32 + r) * 8 + ((s * 2 + hh) ^ (r & 7))];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bf[nt] = sB[buf][(s * 2 + hh) * BN + wn * (BN / 2) + nt * 32 + r];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MfmaD<T>::run(af[mt], bf[nt], acc[mt][nt]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + wn * (BN / 2) + nt * 32 + r;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long long pix = m0 + wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (pix < p.m) {
                    float v = acc[mt][nt][i] + bv;
                    if (p.relu) v = v > 0.0f ? v : 0.0f;
                    y[(size_t)pix * p.cout + co] = from_f<T>(v);
                }
            }
    }
}

// Halo-tile variant for the 3x3 / stride 1 / pad 1 layers (all six 128->128 layers of the car.fhd RPN): the
// implicit-GEMM kernels above re-fetch every input pixel once per tap (9x); here a workgroup owns a 16 x 16 output
// tile, brings the (16+2)^2 input halo into LDS ONCE (all Cin channels: 18*18*Cin*2 B = 83 KB for Cin = 128 --
// CDNA4's 160 KB LDS makes this possible) and then only streams the nine 3x3 weight slabs (double-buffered LDS-DMA).
// 8 waves as 4 (pixel quarters) x 2 (cout halves), each 64 px x 64 cout on MFMA 32x32x16.
template <typename T, int CIN, int TH, int NQ>
__global__ __launch_bounds__(TH * 16 * NQ) void k_conv2d_halo(const T *__restrict__ x, const T *__restrict__ wpk,
                                                    const float *__restrict__ bias, T *__restrict__ y, Conv2dParams p,
                                                    int tiles_y, int tiles_x) {
    constexpr int TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * (TW + 2);   // TH = 16: 324 halo pixels, 8 waves; TH = 8: 180, 4 waves
    constexpr int PQ = TH / 4, NWV = PQ * NQ;      // pixel groups (64 px = 4 tile rows each) x NQ cout groups = waves
    constexpr int NTW = 128 / (NQ * 32);           // 32-wide cout tiles per wave
    constexpr int CH = CIN / 8;                    // 16-byte chunks per pixel (16 for Cin = 128)
    constexpr int HENT = HPIX * CH;                // uint4 entries of the halo
    constexpr int BN = 128, CC = CIN / 64, NIT = 9 * CC;
    extern __shared__ __attribute__((aligned(16))) uint4 halo_smem[];
    uint4 *hal = halo_smem;                        // [HPIX][CH], chunk index XOR-swizzled with (pixel & (CH-1))
    uint4 *sB = halo_smem + HENT;                  // [2][8 * BN]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wv % PQ, wn = wv / PQ;
    // XCD-aware tile order
    const int per = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int ntile = p.batch * tiles_y * tiles_x;
    if (tile >= ntile) return;
    const int b = tile / (tiles_y * tiles_x);
    const int trem = tile - b * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int n0 = blockIdx.y * BN;
    const int cin8 = CIN / 8;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w4 = reinterpret_cast<const uint4 *>(wpk);
    const uint4 *zero16 = w4 + (size_t)9 * cin8 * p.cout;

    // halo DMA: instruction i fills entries [i*64, i*64+64); HENT is a multiple of 64 for CIN in {64, 128}
    for (int i = wv; i < (HENT + 63) / 64; i += NWV) {
        const int e = i * 64 + lane;
        if (e >= HENT) break;
        const int hp = e / CH, slot = e - hp * CH;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        const uint4 *src = ok ? x4 + (((long long)b * p.h + iy) * p.w + ix) * cin8 + (slot ^ (hp & (CH - 1))) : zero16;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&hal[i * 64], 16, 0, 0);
    }
    auto issue_b = [&](int it, int buf) {
        const int tap = it / CC, cc = it - tap * CC;
#pragma unroll
        for (int j = 0; j < 16 / NWV; ++j) {
            const int e = (j * NWV + wv) * 64 + lane, ch = e / BN, n = e - ch * BN;
            const uint4 *src = w4 + ((size_t)tap * cin8 + cc * 8 + ch) * p.cout + n0 + n;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sB[buf * (8 * BN) + (j * NWV + wv) * 64], 16, 0, 0);
        }
    };
    f32x16d acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NTW; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.0f;
    // halo pixel (top-left tap) of this lane's two A rows: wave quarter wm = 4 tile rows, m-tile mt = 2 rows
    int hp0[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int q = mt * 32 + r;
        hp0[mt] = (wm * 4 + (q >> 4)) * HW_ + (q & 15);
    }
    issue_b(0, 0);
    __syncthreads();
    for (int it = 0; it < NIT; ++it) {
        const int buf = it & 1;
        if (it + 1 < NIT) issue_b(it + 1, buf ^ 1);
        const int tap = it / CC, cc = it - tap * CC;
        const int dy = tap / 3, dx = tap - dy * 3;
        const uint4 *bb = sB + buf * (8 * BN);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bf[NTW];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int hp = hp0[mt] + dy * HW_ + dx;
                af[mt] = hal[hp * CH + ((cc * 8 + s * 2 + hh) ^ (hp & (CH - 1)))];
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bf[nt] = bb[(s * 2 + hh) * BN + wn * (BN / NQ) + nt * 32 + r];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MfmaD<T>::run(af[mt], bf[nt], acc[mt][nt]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + wn * (BN / NQ) + nt * 32 + r;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;       // pixel inside the wave's 64
                const int oy = y0 + wm * 4 + (q >> 4), ox = x0 + (q & 15);
                if (oy < p.h && ox < p.w) {
                    float v = acc[mt][nt][i] + bv;
                    if (p.relu) v = v > 0.0f ? v : 0.0f;
                    y[(((size_t)b * p.h + oy) * p.w + ox) * p.cout + co] = from_f<T>(v);
                }
            }
    }
}

template <typename T, int CIN, int TH, int NQ>
static int launch_conv2d_halo(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, hipStream_t st) {
    constexpr size_t lds = ((size_t)(TH + 2) * 18 * (CIN / 8) + 2 * 8 * 128) * 16;
    static bool configured = false;
    auto fn = k_conv2d_halo<T, CIN, TH, NQ>;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ty = div_up(p.h, TH), tx = div_up(p.w, 16);
    const int gx = (p.batch * ty * tx + 7) / 8 * 8;
    hipLaunchKernelGGL(fn, dim3(gx, p.cout / 128), dim3(TH * 16 * NQ), lds, st, (const T *)x, (const T *)wpk, bias, (T *)y, p, ty, tx);
    return check_launch();
}

// 1x1 convolutions (the ConvTranspose2d(k=1) "deconv" and the merged heads of the RPN, rpn.py:275-285,386-391) are
// memory bound: the generic implicit-GEMM kernel re-fetches the whole [Cin x BN] weight block for every 128-pixel
// tile (as many bytes as the activations it reads).  Here the weight block (Cin = 128: two 16 KB slabs) stays
// RESIDENT in LDS while the workgroup streams TPW consecutive pixel tiles through a double-buffered A slab.
template <typename T, int BN, int TPW>
__global__ __launch_bounds__(kBlock) void k_conv1x1_nhwc(const T *__restrict__ x, const T *__restrict__ wpk,
                                                        const float *__restrict__ bias, T *__restrict__ y, Conv2dParams p) {
    constexpr int BM = 128, NTW = BN / 64, CC = 2;   // Cin == 128
    constexpr int PERB = BN * 8 / kBlock;
    __shared__ uint4 sA[2][BM * 8];
    __shared__ uint4 sB[CC][8 * BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wv & 1, wn = wv >> 1;
    const int n0 = blockIdx.y * BN;
    const int cin8 = p.cin / 8;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w4 = reinterpret_cast<const uint4 *>(wpk);
    const uint4 *zero16 = w4 + (size_t)cin8 * p.cout;
    const long long tile0 = (long long)blockIdx.x * TPW;
    const long long ntiles = (p.m + BM - 1) / BM;
    if (tile0 >= ntiles) return;
    const int nt_local = (int)((ntiles - tile0 < TPW) ? ntiles - tile0 : TPW);
    const int slot = lane & 7;
    auto issue_a = [&](int it, int buf) {           // it = local_tile * CC + cc ; 1x1: pixel p reads input pixel p
        const long long m0 = (tile0 + it / CC) * BM;
        const int cc = it % CC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pl = (j * 4 + wv) * 8 + (lane >> 3);
            const long long pix = m0 + pl;
            const uint4 *src = pix < p.m ? x4 + pix * cin8 + cc * 8 + (slot ^ (pl & 7)) : zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sA[buf][(j * 4 + wv) * 64], 16, 0, 0);
        }
    };
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int j = 0; j < PERB; ++j) {
            const int e = (j * 4 + wv) * 64 + lane, ch = e / BN, n = e - ch * BN;
            const uint4 *src = w4 + ((size_t)cc * 8 + ch) * p.cout + n0 + n;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sB[cc][(j * 4 + wv) * 64], 16, 0, 0);
        }
    issue_a(0, 0);
    __syncthreads();
    f32x16d acc[2][NTW];
    const int NIT = nt_local * CC;
    for (int it = 0; it < NIT; ++it) {
        const int buf = it & 1, cc = it % CC;
        if (cc == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NTW; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
        }
        if (it + 1 < NIT) issue_a(it + 1, buf ^ 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bf[NTW];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = sA[buf][(wm * 64 + mt * 32 + r) * 8 + ((s * 2 + hh) ^ (r & 7))];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bf[nt] = sB[cc][(s * 2 + hh) * BN + wn * (BN / 2) + nt * 32 + r];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MfmaD<T>::run(af[mt], bf[nt], acc[mt][nt]);
        }
        if (cc == CC - 1) {
            const long long m0 = (tile0 + it / CC) * BM;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int co = n0 + wn * (BN / 2) + nt * 32 + r;
                const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const long long pix = m0 + wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (pix < p.m) {
                            float v = acc[mt][nt][i] + bv;
                            if (p.relu) v = v > 0.0f ? v : 0.0f;
                            y[(size_t)pix * p.cout + co] = from_f<T>(v);
                        }
                    }
            }
        }
        __syncthreads();
    }
}

static int conv2d_variant() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("SEC_CONV2D_VARIANT"); v = e ? atoi(e) : 4; }  // 0 register staged, 1 LDS-DMA implicit GEMM, 2/3/4 halo tile 16x16 / 8x16 / 8x16 with 8 waves
    return v;
}

template <typename T>
static int launch_conv2d(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, hipStream_t st) {
    dim3 block(kBlock);
    if (conv2d_variant() >= 2 && conv2d_variant() <= 4 && p.ksize == 3 && p.stride == 1 && p.pad == 1 &&
        p.cout % 128 == 0 && (p.cin == 128 || p.cin == 64)) {
        if (conv2d_variant() == 2)
            return p.cin == 128 ? launch_conv2d_halo<T, 128, 16, 2>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 16, 2>(x, wpk, bias, y, p, st);
        if (conv2d_variant() == 4)   // 8 waves on a 16x8 tile: 64 px x 32 cout per wave, 4 waves / SIMD at 2 workgroups per CU
            return p.cin == 128 ? launch_conv2d_halo<T, 128, 8, 4>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 8, 4>(x, wpk, bias, y, p, st);
        return p.cin == 128 ? launch_conv2d_halo<T, 128, 8, 2>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 8, 2>(x, wpk, bias, y, p, st);
    }
    if (conv2d_variant() >= 1 && p.ksize == 1 && p.stride == 1 && p.pad == 0 && p.cin == 128) {
        constexpr int TPW = 4;
        const int gx1 = div_up(div_up(p.m, 128), TPW);
        if (p.cout % 128 == 0)
            hipLaunchKernelGGL((k_conv1x1_nhwc<T, 128, TPW>), dim3(gx1, p.cout / 128), block, 0, st, (const T *)x, (const T *)wpk,
                               bias, (T *)y, p);
        else
            hipLaunchKernelGGL((k_conv1x1_nhwc<T, 64, TPW>), dim3(gx1, p.cout / 64), block, 0, st, (const T *)x, (const T *)wpk,
                               bias, (T *)y, p);
        return check_launch();
    }
    if (conv2d_variant() >= 1) {
        const int gx = (div_up(p.m, 128) + 7) / 8 * 8;   // multiple of 8 for the XCD-aware tile order
        if (p.cout % 128 == 0)
            hipLaunchKernelGGL((k_conv2d_nhwc_dma<T, 128>), dim3(gx, p.cout / 128), block, 0, st, (const T *)x,
                               (const T *)wpk, bias, (T *)y, p);
        else
            hipLaunchKernelGGL((k_conv2d_nhwc_dma<T, 64>), dim3(gx, p.cout / 64), block, 0, st, (const T *)x,
                               (const T *)wpk, bias, (T *)y, p);
        return check_launch();
    }
    if (p.cout % 128 == 0) {
        hipLaunchKernelGGL((k_conv2d_nhwc<T, 128>), dim3(div_up(p.m, 128), p.cout / 128), block, 0, st, (const T *)x,
                           (const T *)wpk, bias, (T *)y, p);
    } else {
        hipLaunchKernelGGL((k_conv2d_nhwc<T, 64>), dim3(div_up(p.m, 128), p.cout / 64), block, 0, st, (const T *)x,
                           (const T *)wpk, bias, (T *)y, p);
    }
    return check_launch();
}

}  // namespace sec

using namespace sec;

SEC_API size_t sec_conv2d_packed_weight_bytes(int cout, int cin, int ksize, int dtype) {
    if (dtype == SEC_F32 || cout <= 0 || cin <= 0 || cin % 64 || cout % 64 || ksize <= 0) return 0;
    return (size_t)ksize * ksize * cin * cout * 2 + 16;   // + one 16-byte zero block (padding source of the LDS-DMA path)
}

SEC_API int sec_conv2d_pack_weight(const void *weight, int cout, int cin, int ksize, int dtype, void *packed, void *stream) {
    if (!weight || !packed || sec_conv2d_packed_weight_bytes(cout, cin, ksize, dtype) == 0) return SEC_E_INVALID;
    long long total = (long long)ksize * ksize * cin * cout;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync((char *)packed + total * 2, 0, 16, st) != hipSuccess) return SEC_E_LAUNCH;
    if (dtype == SEC_BF16)
        hipLaunchKernelGGL(k_conv2d_pack<__hip_bfloat16>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st,
                           (const __hip_bfloat16 *)weight, cout, cin, ksize, (__hip_bfloat16 *)packed);
    else
        hipLaunchKernelGGL(k_conv2d_pack<__half>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, (const __half *)weight,
                           cout, cin, ksize, (__half *)packed);
    return check_launch();
}

SEC_API int sec_conv2d_nhwc(const void *x, int batch, int h, int w, int cin, const void *packed_weight, const float *bias,
                            int cout, int ksize, int stride, int pad, int relu, void *y, int dtype, void *stream) {
    if (!x || !packed_weight || !y || batch <= 0 || h <= 0 || w <= 0 || ksize <= 0 || stride <= 0 || pad < 0) return SEC_E_INVALID;
    if (cin % 64 || cout % 64 || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = cin; p.cout = cout; p.ksize = ksize; p.stride = stride; p.pad = pad; p.relu = relu;
    p.ho = (h + 2 * pad - ksize) / stride + 1;
    p.wo = (w + 2 * pad - ksize) / stride + 1;
    if (p.ho <= 0 || p.wo <= 0) return SEC_E_INVALID;
    p.m = (long long)batch * p.ho * p.wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16) return launch_conv2d<__hip_bfloat16>(x, packed_weight, bias, y, p, st);
    return launch_conv2d<__half>(x, packed_weight, bias, y, p, st);
}

SEC_API int sec_bias_act_nhwc(void *x, const float *bias, size_t pixels, int channels, int relu, int dtype, void *stream) {
    if (!x || !bias || channels <= 0) return SEC_E_INVALID;
    if (pixels == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_F32) {
        if (channels % 4) return SEC_E_UNSUPPORTED;
        long long n = (long long)pixels * channels / 4;
        int blocks = div_up(n, kBlock); if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_bias_act32, dim3(blocks), dim3(kBlock), 0, st, (float *)x, bias, n, channels / 4, relu);
    } else {
        if (channels % 8) return SEC_E_UNSUPPORTED;
        long long n = (long long)pixels * channels / 8;
        int blocks = div_up(n, kBlock); if (blocks > 256 * 16) blocks = 256 * 16;
        if (dtype == SEC_BF16)
            hipLaunchKernelGGL(k_bias_act16<__hip_bfloat16>, dim3(blocks), dim3(kBlock), 0, st, (__hip_bfloat16 *)x, bias, n, channels / 8, relu);
        else if (dtype == SEC_F16)
            hipLaunchKernelGGL(k_bias_act16<__half>, dim3(blocks), dim3(kBlock), 0, st, (__half *)x, bias, n, channels / 8, relu);
        else return SEC_E_UNSUPPORTED;
    }
    return check_launch();
}
