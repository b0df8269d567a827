// Dense-side helpers for the RPN (second/pytorch/models/rpn.py:468-497): the convolutions themselves run
// through MIOpen in phase 1; the per-channel bias (folded BatchNorm2d) + ReLU that follows every conv is ONE
// in-place pass here instead of MIOpen's separate bias tensor-op plus a ReLU kernel (3 passes -> 1).
#include "common.hpp"
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
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


// Two fp32 -> one dword of two 16-bit values with ONE conversion instruction pair (v_cvt_pk_bf16_f32 on gfx950; the scalar
// from_f<T> form compiles to one convert + one merge per VALUE).  Same rounding as from_f<T> (round to nearest even).
typedef float f32x2d __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<__hip_bfloat16>(float a, float b) {
    typedef __bf16 bf16x2d __attribute__((ext_vector_type(2)));
    const f32x2d v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2d));
}
template <> __device__ __forceinline__ unsigned pack2<__half>(float a, float b) {
    typedef _Float16 f16x2d __attribute__((ext_vector_type(2)));
    const f32x2d v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2d));
}

// Transposed-accumulator epilogue.  With the MFMA operands swapped (weights as the first operand) the 32x32 result
// tile is D^T: a lane owns ONE pixel (lane & 31) and, per group g = i >> 2, four CONSECUTIVE output channels
// 8g + 4*(lane >> 5) + (i & 3).  The two half-waves exchange one group each so that every lane ends up with eight
// consecutive channels of its pixel: two 16-byte stores per tile instead of sixteen 2-byte scatters.
template <typename T> __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    T t[4] = {from_f<T>(a), from_f<T>(b), from_f<T>(c), from_f<T>(d)};
    uint2 u;
    __builtin_memcpy(&u, t, 8);
    return u;
}
template <typename T, bool HAS_BIAS>
__device__ __forceinline__ void store_tile_tb(const f32x16d &acc, const float *__restrict__ bias, int c0, int relu,
                                              T *__restrict__ ypix, bool ok, int hh) {
    // (`relu ? (v > 0 ? v : 0) : v` per value compiled into a BRANCH per value, and from_f<T> into one convert + merge per
    // value: ~230 instructions per tile.  One uniform branch, max, pair converts and v_permlane32_swap for the half-wave
    // exchange -- see the epilogue of k_conv2d_halo_reg -- are ~60.)
    unsigned lo[4], hi[4];
    auto cvt = [&](auto relu_tag) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = c0 + 8 * g + 4 * hh;
            float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (HAS_BIAS) bv = *reinterpret_cast<const float4 *>(bias + c);
            float v[4] = {acc[4 * g] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
            if (decltype(relu_tag)::value) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.0f);
            }
            lo[g] = pack2<T>(v[0], v[1]);
            hi[g] = pack2<T>(v[2], v[3]);
        }
    };
    if (relu) cvt(std::true_type{});
    else cvt(std::false_type{});
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const auto sx = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
        if (ok) *reinterpret_cast<uint4 *>(ypix + c0 + 8 * (2 * pr + hh)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
}
template <typename T>
__device__ __forceinline__ void store_tile_t(const f32x16d &acc, const float *__restrict__ bias, int c0, int relu,
                                             T *__restrict__ ypix, bool ok, int hh) {
    if (bias) store_tile_tb<T, true>(acc, bias, c0, relu, ypix, ok, hh);
    else store_tile_tb<T, false>(acc, bias, c0, relu, ypix, ok, hh);
}

struct Conv2dParams {
    int batch, h, w, cin, cout, ho, wo, ksize, stride, pad, relu;
    int zskip;    // input known to be mostly zero (scattered sparse voxels): all-zero halo tiles skip the MFMA loop
    long long m;  // batch * ho * wo
    int stagger;  // profiling builds (-DSEC_CONV_TIMELINE): start delay in clocks per resident-slot index
};

// k_conv2d_halo_reg<..., TAIL>: the fused 1x1 tail (deblock 128 -> 128 + merged heads 128 -> 64, k_conv1x1_chain's two GEMMs) run on the
// conv's output tile while it is still in LDS
struct ConvTailArgs {
    const void *w1, *w2;      // packed 1x1 weights ([cin8][cout] uint4, sec_conv2d_pack_weight with ksize 1)
    const float *b1, *b2;
    void *y;                  // [batch][h][w][64]
    int relu1;
};

// Live share (of 256) up to which a conv / the fused tail follows its live-tile list; above it every tile is taken in the plain order
// (computing a background tile is always correct).  SEC_RPN_LIST_MAX_LIVE=<percent> overrides (read once, before the first launch).
// 225 / 256 = 88 % since round 6 (192 = 75 % before): on the bench's dense seeded scene, whose last two convs have 76-78 % of their tiles
// live, 75 % gave 8 000 frames/s and 82 / 88 / 94 / 100 % 8 130-8 180 (gpurun r06_x, two runs each); with every tile live the lists
// cost what they cannot save (480 instead of 427 us for the six convs + tail, round 4), so the switch stays below 100 %.
__device__ int g_list_max_live_q8 = 225;

static void apply_list_threshold_env() {
    static bool done = false;
    if (done) return;
    done = true;
    const char *e = getenv("SEC_RPN_LIST_MAX_LIVE");
    if (!e || !*e) return;
    int q8 = atoi(e) * 256 / 100;
    if (q8 < 0) q8 = 0;
    if (q8 > 256) q8 = 256;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_list_max_live_q8), &q8, sizeof(int));
}

#ifdef SEC_CONV_TIMELINE
__device__ long long *g_timeline2 = nullptr;
__device__ int g_cu_resident[8 * 4096];     // profiling: workgroups currently resident per (XCC, HW_ID cu/sh/se)
#endif

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

    // DMA instruction j of wave wv fills LDS entries [(j*4 + wv)*64, +64): pixel (j*4+wv)*8 + lane/8, slot lane%8.
    // All per-lane address arithmetic happens once, here: inside the loop a source address is a per-lane pointer plus
    // a wave-uniform (scalar) offset that advances with (tap, channel slab) -- the first version re-derived tap / dy / dx by
    // division and multiplied 64-bit pixel indices per DMA, ~100 quarter-rate multiplies per 16 MFMAs.
    const int slot = lane & 7;
    const uint4 *aptr[4];
    int iy0[4], ix0[4];
    bool pval[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pl = (j * 4 + wv) * 8 + (lane >> 3);
        long long pix = m0 + pl;
        pval[j] = pix < p.m;
        long long q = pval[j] ? pix : 0;
        int ox = (int)(q % p.wo);
        q /= p.wo;
        int oy = (int)(q % p.ho);
        int b = (int)(q / p.ho);
        iy0[j] = oy * p.stride - p.pad;
        ix0[j] = ox * p.stride - p.pad;
        aptr[j] = x4 + (((long long)b * p.h + iy0[j]) * p.w + ix0[j]) * cin8 + (slot ^ (pl & 7));
    }
    const uint4 *bptr[PERB];
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
        const int e = (j * 4 + wv) * 64 + lane, ch = e / BN, n = e - ch * BN;
        bptr[j] = w4 + (size_t)ch * p.cout + n0 + n;
    }
    // wave-uniform state of the NEXT slab to issue: tap (dy, dx), channel slab cc and the two scalar offsets
    int n_dy = 0, n_dx = 0, n_cc = 0;
    long long n_aoff = 0, n_boff = 0;          // (dy * w + dx) * cin8 + cc * 8   and   (tap * cin8 + cc * 8) * cout
    const long long b_step = (long long)8 * p.cout;
    auto issue = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + n_dy, ix = ix0[j] + n_dx;
            const bool ok = pval[j] && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const uint4 *src = ok ? aptr[j] + n_aoff : zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&sA[buf][(j * 4 + wv) * 64], 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < PERB; ++j)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bptr[j] + n_boff), (lds_ptr_t)&sB[buf][(j * 4 + wv) * 64], 16, 0, 0);
        n_boff += b_step;                       // packed weights are [tap][cin8][cout]: consecutive slabs are contiguous
        if (++n_cc == CC) {
            n_cc = 0;
            if (++n_dx == p.ksize) { n_dx = 0; ++n_dy; }
            n_aoff = ((long long)n_dy * p.w + n_dx) * cin8;
        } else {
            n_aoff += 8;
        }
    };

    f32x16d acc[2][NTW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    issue(0);
    __syncthreads();   // hipcc drains the DMA (vmcnt(0)) before the barrier
    for (int it = 0; it < NIT; ++it) {
        const int buf = it & 1;
        if (it + 1 < NIT) issue(buf ^ 1);
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
    const size_t ldc = (size_t)p.cout;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + wn * (BN / 2) + nt * 32 + r;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long long pix0 = m0 + wm * 64 + mt * 32 + 4 * hh;
            T *yp = y + (size_t)pix0 * ldc + co;            // one 64-bit multiply per tile; the row offsets below are scalar
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ro = (i & 3) + 8 * (i >> 2);
                if (pix0 + ro < p.m) {
                    float v = acc[mt][nt][i] + bv;
                    if (p.relu) v = v > 0.0f ? v : 0.0f;
                    yp[(size_t)ro * ldc] = from_f<T>(v);
                }
            }
        }
    }
}

// ---- software-pipelined halo kernel -------------------------------------------------------------------------
// k_conv2d_halo above double-buffers the weight slabs, but the compiler cannot tell the LDS-DMA destination from
// the slab being read (one dynamic LDS array) and puts s_waitcnt vmcnt(0) in front of the first ds_read of every
// iteration: the "prefetch" is waited for before the MFMAs it was meant to hide behind.  Here the iteration body is
// a function whose LDS pointers are __restrict__ (inlining turns that into alias scopes, which the waitcnt pass
// honours), the slabs form an NB-deep ring filled DIST = NB-1 iterations ahead, and the only waits are the counted
// s_waitcnt vmcnt(n) + LDS-only barrier written out below.  KS = input channels per slab (64: 16 KB, 32: 8 KB).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XOR key of a halo pixel's 16-byte chunks.  ds_read_b128 is served in groups of 16 lanes that must hit 16 distinct
// 16-B slots of the 256-B bank row; a group's lanes are 8 + 8 pixels of two ADJACENT tile rows at complementary
// columns ({0-3,12-15} and {4-11}), so keying on the halo COLUMN makes them distinct (keying on the linear halo index,
// as k_conv2d_halo does, shifts the second row by HW_ - 16 = 2 and collides two slots).  Cin = 64 keeps the old key.
template <int CH, int HW_, bool COLKEY> __device__ __forceinline__ int halo_key(int hp) {
    return (COLKEY && CH == 16) ? ((hp % HW_) & 15) : (hp & (CH - 1));
}

#ifdef SEC_CONV2D_EXPERIMENTS   // superseded 3x3 kernels (register-staged implicit GEMM, LDS weight slabs / rings): A/B builds only
#include "../../tools/kernel_experiments/dense_conv2d_ab.inc"
#endif

// ---- halo kernel with register-resident weights ----------------------------------------------------------------
// PMC on the LDS-slab kernels (profiles/r01_g_pmc_conv2d.txt): MFMA pipe 38 % busy, waves parked 48 % of their
// cycles (one s_barrier per 16 KB weight slab keeps all eight waves in lock-step) and only 12 of 16 wave slots
// filled on average (2200 tiles over 512 workgroup slots = 4.3 rounds).  Here the weights never touch LDS: a
// wave owns ALL 128 pixels of the 8 x 16 tile for 32 output channels, so no two waves of a workgroup need the same
// B fragment and each streams its own from L2 (1 KB coalesced per k-step, prefetched one iteration ahead in VGPRs).
// LDS holds only the halo (46 KB -> 3 workgroups per CU, 2200 / 768 = 2.9 rounds), the main loop has no barrier,
// and LDS traffic drops to one conflict-free ds_read_b128 per MFMA.
// Cin = 128 runs the ROLL == 2 main loop (below): m-tiles pair output rows (mt, mt + 4) so that SIX halo fragments per (dx, k-step)
// feed the twelve MFMAs of the three kernel rows -- half the LDS reads of one fragment per MFMA (ROLL == 1, kept for A/B as
// SEC_CONV2D_VARIANT=15): 79.9 -> 74.0 us on one box, 76.7 -> 71.8 us on another (batch 8, 200 x 176, 128 -> 128).  Setting wave
// priorities (prologue / epilogue above the loop, and the reverse) and a 12-deep B ring were measured on that loop: +1 ... +5 % slower.
// Both ROLL loops: B fragments in an 8-deep ring loaded 7 fragments ahead, and halo addresses built
// from a per-lane base, a dx-only swizzle key and ds_read immediates.  The first version recomputed `hp % HW_` per m-tile and
// tap -- ~125 VALU instructions per tap, a third of them quarter-rate 32-bit multiplies, against 32 MFMAs: 93 us -> 84 us
// (1000 TFLOP/s) once they were gone.  Bounding runs on that shape: halos served from L2-resident tiles -2 us, epilogue
// stores removed -5 us, MFMA loop removed 37 us (26 us of it the halo DMA at ~12 B/clk/CU when all CUs burst at once):
// what is left is prologue / epilogue of the three lock-stepped workgroup rounds, not the loop.
// With the ROLL loop re-measured: persistent workgroups (96 per XCD striding through the tiles, next halo DMA issued before the
// epilogue stores, B ring wrapping into the next tile; 159 VGPRs, no spills) 83.7 vs 83.2 us -- no gain; 12 x 16 tiles 89 us,
// 4 x 16 tiles at 5 workgroups per CU 94 us.
// GATHER: the input is not a dense image but a sparse tensor -- `x` = its feature rows [rows][CIN / 2] (the two z planes of a
// site column are two rows), `site_map` = [batch][2][h][w] row + 1 (sec_sparse_site_map).  The halo pieces are gathered from
// the rows (input channel z * 64 + c: the weights are packed in that order), sites without a row read zeros through the buffer
// bounds check, and a tile whose 2 x 180 map entries are all empty skips the DMA, the LDS sweep and the MFMA loop: what the
// zero fill + scatter + zero-tile test of the dense form computed, without the 72 MB image.
// X3 (sec_conv2d_nhwc_x3; fp32 networks, the reference's default precision, rpn.py:468-497 under train.py:232-235): the operands
// are fp32 values carried as TWO bf16 planes each, v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits), and the
// product is x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 -- the dropped x_lo w_lo term and the two representation errors
// are each <= 2^-17 relative.  Three passes of the SAME loop over the same accumulators: (x_hi, w_hi), (x_hi, w_lo), then the
// halo is replaced by x_lo's and (x_lo, w_hi) runs; the LDS footprint and the three workgroups per CU stay.  The epilogue splits
// the fp32 result (bias + ReLU applied in fp32) into the two planes of the next layer's input.  18.4x the fp32 MFMA rate per
// product / 3 passes: the matrix pipe's fp32 instruction (v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU) is what MIOpen's fp32
// convolution runs at ~80 % of (0.65 ms per layer at batch 8).
template <typename T, int CIN, int TH, int ROLL = 0, bool GATHER = false, int NSPLIT = 1, bool X3 = false, bool TAIL = false>
__global__ __launch_bounds__(256, CIN == 128 ? 3 : 2) void k_conv2d_halo_reg(const T *__restrict__ x, const T *__restrict__ wpk,
                                                            const float *__restrict__ bias, T *__restrict__ y,
                                                            Conv2dParams p, int tiles_y, int tiles_x, int per_xcd,
                                                            const int *__restrict__ site_map, unsigned feat_bytes,
                                                            const unsigned short *__restrict__ tile_order = nullptr,
                                                            const int *__restrict__ live_counts = nullptr,
                                                            const T *__restrict__ background = nullptr,
                                                            const unsigned short *__restrict__ nbr_masks = nullptr,
                                                            const T *__restrict__ bg_in = nullptr,
                                                            const T *__restrict__ x_lo = nullptr, T *__restrict__ y_lo = nullptr,
                                                            const T *__restrict__ background_lo = nullptr,
                                                            const T *__restrict__ bg_in_lo = nullptr, ConvTailArgs tail = ConvTailArgs{}) {
    static_assert(!GATHER || (ROLL == 2 && CIN == 128), "gather prologue: the shared-row loop on two 64-channel planes");
    static_assert(!TAIL || (ROLL == 2 && CIN == 128 && TH == 8 && !GATHER && NSPLIT == 1 && !X3), "fused 1x1 tail: the lazy list form of the 128-channel conv");
    static_assert(!X3 || (ROLL == 2 && CIN == 128 && TH == 8 && !GATHER && NSPLIT == 1 && std::is_same<T, __hip_bfloat16>::value),
                  "three-pass split-fp32 form: the shared-row bf16 loop");
    constexpr int TW = 16, HW_ = TW + 2, HPIX = (TH + 2) * (TW + 2);
    // NSPLIT == 2: 64 output channels per workgroup -- the waves split the tile's pixels two ways and the channels two ways (the
    // 64 -> 64 layers of the PointPillars RPN); NSPLIT == 1: a wave owns all pixels for 32 of 128 channels
    static_assert(NSPLIT == 1 || (NSPLIT == 2 && ROLL == 0 && !GATHER), "pixel split: the two-stage loop only");
    constexpr int MT = TH * TW / 32 / NSPLIT;      // 32-pixel m-tiles per wave (4 for an 8 x 16 tile)
    constexpr int CH = CIN / 8, HENT = HPIX * CH;
    constexpr int KC = CIN / 64, NIT = 9 * KC;
    extern __shared__ __attribute__((aligned(16))) uint4 halo_smem[];
    uint4 *hal = halo_smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ntile = p.batch * tiles_y * tiles_x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 and takes tile b / 8 of that XCD's contiguous range.
    // (A persistent form -- 3 resident workgroups per CU striding through the range, next halo DMA overlapped with
    // the epilogue -- was measured 5 % SLOWER: its extra live state spills at the 168-VGPR budget.)
    // (A 12 x 16 tile -- B fragments reused over 6 m-tiles, 202 VGPRs, 2 workgroups per CU -- measured 3 % slower, 96 vs 93 us;
    // a 6 x 16 tile -- 3 m-tiles, 128 VGPRs, 4 workgroups per CU -- 6 % slower, 97 vs 92 us: 8 x 16 at 3 per CU is the optimum
    // between weight-stream reuse and resident waves.)
    const int xcd = blockIdx.x % 8, local = blockIdx.x / 8;
    const int n0 = blockIdx.y * (128 / NSPLIT) + (wv % (4 / NSPLIT)) * 32;     // this wave's 32 output channels
    const int mtb = (wv / (4 / NSPLIT)) * MT;      // ... and its first m-tile of the tile's pixels
    constexpr int cin8 = CIN / 8;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w4 = reinterpret_cast<const uint4 *>(wpk);
    const uint4 *zero16 = w4 + (size_t)9 * cin8 * p.cout;
    auto issue_halo = [&](int tile) {
        const int b = tile / (tiles_y * tiles_x);
        const int trem = tile - b * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        for (int i = wv; i < (HENT + 63) / 64; i += 4) {
            const int e = i * 64 + lane;
            if (e >= HENT) break;
            const int hp = e / CH, slot = e - hp * CH;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const uint4 *src = ok ? x4 + (((long long)b * p.h + iy) * p.w + ix) * cin8 + (slot ^ halo_key<CH, HW_, true>(hp)) : zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&hal[i * 64], 16, 0, 0);
        }
    };
    // The same halo by buffer LDS-DMA with incrementally built offsets (ROLL == 2).  The loop above costs ~40 instructions per
    // 1 KB piece (two divisions by 18, 64-bit address multiplies, two divergent branches) and, issued beside two other waves'
    // MFMA loops at one slot per ~18 clocks, took 8 900 of the prologue's 9 850 clocks (timeline) -- the DMA latency was the small
    // part.  Here a piece's four halo pixels advance by 16 per step (hx += 16, wrapping at 18 into the next row), the image is a
    // buffer resource whose bounds check zero-fills the rows above and below it, and only the left / right border columns need
    // a select: ~14 VALU per piece, no branches.
    auto issue_halo2 = [&](int tile, const T *xsrc) {
        const int b = tile / (tiles_y * tiles_x);
        const int trem = tile - b * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const unsigned img_bytes = (unsigned)p.h * (unsigned)p.w * (CIN * 2u);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(xsrc) + (size_t)b * p.h * p.w * CIN, 0, (int)img_bytes, 0x00020000);
        const int wvs = __builtin_amdgcn_readfirstlane(wv);
        const unsigned slot = lane & 15;
        const unsigned row_pitch = (unsigned)p.w * (CIN * 2u);
        int hx = wvs * 4 + (lane >> 4);                                                   // halo pixel hp = i * 4 + lane / 16 -> (hy, hx), hy = 0 here
        unsigned rowoff = (unsigned)((y0 - 1) * p.w + (x0 - 1)) * (CIN * 2u);            // may wrap below zero: out of bounds, zero fill
#pragma unroll
        for (int t = 0; t < (HENT / 64 + 3) / 4; ++t) {
            const int i = wvs + 4 * t;
            if (i < HENT / 64) {
                const unsigned key = (slot ^ ((unsigned)hx & 15u)) << 4;
                unsigned off = rowoff + ((unsigned)hx << 8) + key;
                const unsigned ix = (unsigned)(x0 - 1 + hx);
                off = ix < (unsigned)p.w ? off : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)&hal[i * 64], 16, off, 0, 0, 0);
            }
            hx += 16;
            const bool wrap = hx >= HW_;
            hx = wrap ? hx - HW_ : hx;
            rowoff = wrap ? rowoff + row_pitch : rowoff;
        }
    };
    // LAZY-BACKGROUND form of issue_halo2 (sec_conv2d_nhwc_tiles_lazy): the layer that produced `x` wrote only ITS live tiles; the
    // others were never materialised.  `nmask` bit (ry * 3 + rx) says whether the tile holding the halo pixels of row class ry
    // (0: the row above the tile, 1: its own rows, 2: the row below) and column class rx (left column / own columns / right column)
    // was live there; pixels of a tile that was not come from `bg_in` = that layer's output for an EMPTY frame ([h][w][CIN]) at the
    // same position -- exactly what the producer's copy would have put into `x`.  Both sources share the byte
    // offsets; a lane issues ONE of the two DMAs (the lanes are the LDS slots, an inactive lane writes nothing).
    auto issue_halo2_lazy = [&](int tile, unsigned nmask, const T *xsrc, const T *bgsrc) {
        const int b = tile / (tiles_y * tiles_x);
        const int trem = tile - b * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const unsigned img_bytes = (unsigned)p.h * (unsigned)p.w * (CIN * 2u);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(xsrc) + (size_t)b * p.h * p.w * CIN, 0, (int)img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(bgsrc), 0, (int)img_bytes, 0x00020000);
        const int wvs = __builtin_amdgcn_readfirstlane(wv);
        const unsigned slot = lane & 15;
        const unsigned row_pitch = (unsigned)p.w * (CIN * 2u);
        int hx = wvs * 4 + (lane >> 4), hy = 0;
        unsigned rowoff = (unsigned)((y0 - 1) * p.w + (x0 - 1)) * (CIN * 2u);
#pragma unroll
        for (int t = 0; t < (HENT / 64 + 3) / 4; ++t) {
            const int i = wvs + 4 * t;
            if (i < HENT / 64) {
                const unsigned key = (slot ^ ((unsigned)hx & 15u)) << 4;
                unsigned off = rowoff + ((unsigned)hx << 8) + key;
                const unsigned ix = (unsigned)(x0 - 1 + hx);
                off = ix < (unsigned)p.w ? off : 0xfffffff0u;
                const int ry = hy == 0 ? 0 : (hy == TH + 1 ? 2 : 1), rx = hx == 0 ? 0 : (hx == HW_ - 1 ? 2 : 1);
                if ((nmask >> (ry * 3 + rx)) & 1u) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)&hal[i * 64], 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(ers, (lds_ptr_t)&hal[i * 64], 16, off, 0, 0, 0);
            }
            hx += 16;
            const bool wrap = hx >= HW_;
            hx = wrap ? hx - HW_ : hx;
            hy = wrap ? hy + 1 : hy;
            rowoff = wrap ? rowoff + row_pitch : rowoff;
        }
    };
    // GATHER form of the same pieces: lane (pixel, slot) fetches chunk c = slot ^ key of its halo pixel = 16 bytes at
    // (c & 7) * 16 of the row of plane z = c >> 3; the row comes from the site map (0 = none -> out-of-range offset -> zeros).
    // Returns whether any of this lane's map entries names a row.
    auto issue_halo_gather = [&](int tile) -> unsigned {
        constexpr int NP = (HENT / 64 + 3) / 4;
        const int b = tile / (tiles_y * tiles_x);
        const int trem = tile - b * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const unsigned plane = (unsigned)p.h * (unsigned)p.w;
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(site_map) + (size_t)b * 2 * plane, 0, (int)(plane * 8u), 0x00020000);
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x), 0, (int)feat_bytes, 0x00020000);
        const int wvs = __builtin_amdgcn_readfirstlane(wv);
        const unsigned slot = lane & 15;
        int hx = wvs * 4 + (lane >> 4), iy = y0 - 1;
        unsigned rowp1[NP], coff[NP], any = 0;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int i = wvs + 4 * t;
            rowp1[t] = 0;
            coff[t] = 0;
            if (i < HENT / 64) {
                const unsigned c = slot ^ ((unsigned)hx & 15u);
                const int ix = x0 - 1 + hx;
                const bool ok = (unsigned)ix < (unsigned)p.w && (unsigned)iy < (unsigned)p.h;
                const unsigned moff = ((c >> 3 ? plane : 0u) + (unsigned)(iy * p.w + ix)) * 4u;
                rowp1[t] = __builtin_amdgcn_raw_buffer_load_b32(mrs, ok ? moff : 0xfffffffcu, 0, 0);
                coff[t] = (c & 7u) << 4;
            }
            hx += 16;
            const bool wrap = hx >= HW_;
            hx = wrap ? hx - HW_ : hx;
            iy = wrap ? iy + 1 : iy;
        }
#pragma unroll
        for (int t = 0; t < NP; ++t) any |= rowp1[t];
        // workgroup-wide "any": one word per wave, ONE barrier (__syncthreads_or compiles to three)
        __shared__ unsigned wave_any[4];
        const unsigned long long bal = __ballot(any != 0);
        if (lane == 0) wave_any[wv] = bal != 0ull;
        __syncthreads();
        if ((wave_any[0] | wave_any[1] | wave_any[2] | wave_any[3]) == 0u) return 0u;          // empty tile: nothing to fetch
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int i = wvs + 4 * t;
            if (i < HENT / 64) {
                const unsigned off = rowp1[t] ? (rowp1[t] - 1u) * (CIN * 1u) + coff[t] : 0xfffffff0u;   // a row = CIN / 2 16-bit channels = CIN bytes
                __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lds_ptr_t)&hal[i * 64], 16, off, 0, 0, 0);
            }
        }
        return 1u;
    };
    static_assert(HENT % 64 == 0 || ROLL < 2, "whole 1 KB pieces");
    // B fragment of k-step s of slab `it` = (tap, kc): packed weights are [tap][cin8][cout] uint4
    const uint4 *wlane = w4 + (size_t)hh * p.cout + n0 + r;
    auto load_b = [&](int it, uint4 (&dst)[4]) {
        const uint4 *src = wlane + (size_t)it * 8 * p.cout;     // it * 8 chunks == (tap * cin8 + kc * 8) because KC * 8 == cin8
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[s] = src[(size_t)s * 2 * p.cout];
    };
    int hp0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (mtb + mt) * 32 + r;
        hp0[mt] = (q >> 4) * HW_ + (q & 15);
    }
    auto load_a = [&](int tap_, int kc_, int s, uint4 (&dst)[MT]) {
        const int dy = tap_ / 3, dx = tap_ - dy * 3;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int hp = hp0[mt] + dy * HW_ + dx;
            dst[mt] = hal[hp * CH + ((kc_ * 8 + s * 2 + hh) ^ halo_key<CH, HW_, true>(hp))];
        }
    };
    int tile = xcd * per_xcd + local;
    if (local >= per_xcd) return;
    // BACKGROUND tiles (sec_conv2d_nhwc_tiles / _gather with tile lists): the caller knows (sec_rpn_tile_live: the BEV site map,
    // dilated once per conv layer) that no site lies within the receptive field of such a tile's outputs, so they equal what this
    // very kernel computes for an EMPTY frame at the same position -- `background` = that image [h][w][cout], computed once per
    // network and map size -- and are copied from it: no halo, no LDS, no MFMA.  (Interior tiles of it hold one channel vector, the
    // tiles along the image border the zero padding's imprint.)
    // The LIVE tiles of all frames form one list (frame-major) that is cut into eight equal contiguous runs, one per XCD, so that a
    // dense frame does not leave its XCD working while the others idle; the workgroups behind them copy the background tiles.
    unsigned nmask = 0x1ffu;
    bool listed = false;
    {
        int n_live = 0;
        if (tile_order)
            for (int f = 0; f < p.batch; ++f) n_live += live_counts[f];
        // A scene whose sites reach (almost) every tile gains nothing from the lists and would pay their dependent lookups in every
        // workgroup's prologue (all tiles live: 480 instead of 427 us for the six convs + tail): above g_list_max_live_q8 / 256 live, every
        // tile is convolved in the plain order -- computing a background tile is always correct.
        if (tile_order && n_live * 256 <= ntile * g_list_max_live_q8) {
            listed = true;
            const int tpf = tiles_y * tiles_x;
            const int per_live = (n_live + 7) >> 3;
            int item;
            bool is_live = false;
            if (local < per_live) {
                item = xcd * per_live + local;
                is_live = item < n_live;
                if (!is_live) item -= n_live;
            } else {
                item = 8 * per_live - n_live + (local - per_live) * 8 + xcd;
            }
            if (is_live) {
                int f = 0;
                while (item >= live_counts[f]) item -= live_counts[f++];
                tile = f * tpf + tile_order[f * tpf + item];
                if (nbr_masks) nmask = nbr_masks[f * tpf + item];        // rank-indexed half: loaded beside the tile index
            } else {
                if (!background) return;                                  // lazy consumers: background tiles are never materialised
                // A copying workgroup takes kCopyTiles background tiles, two at a time with all their loads in flight: one tile per
                // workgroup left ~1 500 short-lived workgroups queueing for the ~100 slots the live tiles' first round leaves free
                // (a slot costs this kernel's 46 KB of LDS whatever the workgroup does) and the launch ended with them, not with
                // the convolution: 31.8 instead of 26 us at 650 live tiles.
                constexpr int kCopyTiles = 4;
                const int n_bg = ntile - n_live;
                const int px = tid >> 4, ch = tid & 15;
#pragma unroll 1
              for (int plane = 0; plane < (X3 ? 2 : 1); ++plane) {       // X3: the residual plane of the empty frame's map too
                const uint4 *e4 = reinterpret_cast<const uint4 *>(plane ? background_lo : background);
                uint4 *y4 = reinterpret_cast<uint4 *>(plane ? y_lo : y);
                auto bg_tile = [&](int it) -> int {              // background item -> tile index over the batch, -1 past the end
                    if (it >= n_bg) return -1;
                    int f = 0;
                    while (it >= tpf - live_counts[f]) it -= tpf - live_counts[f++];
                    return f * tpf + tile_order[f * tpf + tpf - 1 - it];
                };
#pragma unroll 1
                for (int q = 0; q < kCopyTiles; q += 2) {
                    const int t0 = bg_tile(item * kCopyTiles + q), t1 = bg_tile(item * kCopyTiles + q + 1);
                    if (t0 < 0) break;
                    uint4 v[2][TH];
                    size_t dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int tt = u ? t1 : t0;
                        const int f = tt / tpf, trem = tt - f * tpf;
                        const int y0 = (trem / tiles_x) * TH, ox = (trem % tiles_x) * TW + px;
                        const bool okx = tt >= 0 && ox < p.w;
                        dst[u] = (((size_t)f * p.h + y0) * p.w + ox) * (p.cout / 8) + blockIdx.y * 16 + ch;
#pragma unroll
                        for (int ty_ = 0; ty_ < TH; ++ty_)
                            v[u][ty_] = (okx && y0 + ty_ < p.h) ? e4[((size_t)(y0 + ty_) * p.w + ox) * (p.cout / 8) + blockIdx.y * 16 + ch]
                                                                 : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int tt = u ? t1 : t0;
                        const int f = tt / tpf, trem = tt - f * tpf;
                        const int y0 = (trem / tiles_x) * TH, ox = (trem % tiles_x) * TW + px;
                        if (tt < 0 || ox >= p.w) continue;
#pragma unroll
                        for (int ty_ = 0; ty_ < TH; ++ty_)
                            if (y0 + ty_ < p.h) y4[dst[u] + (size_t)ty_ * p.w * (p.cout / 8)] = v[u][ty_];
                    }
                }
              }
                return;
            }
        }
    }
    if (tile >= ntile) return;
    if (nbr_masks && !listed) nmask = nbr_masks[ntile + tile];            // plain order: the tile-indexed half of the masks
#ifdef SEC_CONV_TIMELINE
    long long *tl = g_timeline2;
    long long tl0 = 0, tl1 = 0, tl2 = 0;
    if (p.stagger == -1) return;                                   // experiment: dispatch cost of the grid alone
    if (p.stagger == -2) { issue_halo(tile); __syncthreads(); return; }   // ... plus the halo DMA
    if (p.stagger > 0) {       // experiment: desynchronise the three workgroups a CU holds
        const int slot = (blockIdx.x / 256) % 3;
        const long long until = clock64() + (long long)slot * p.stagger;
        while (clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    long long wl0 = 0;
    if (tl) { tl0 = clock64(); wl0 = wall_clock64(); }
    int cu_key = 0, resident_at_start = 0;
    if (tl && tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID[3:0]
        cu_key = (int)((xcc & 7) * 4096 + ((hw >> 8) & 0xff));
        resident_at_start = atomicAdd(&g_cu_resident[cu_key], 1);
    }
#endif
    unsigned gather_live = 1;
    if constexpr (GATHER) {
        gather_live = issue_halo_gather(tile);
        if (!gather_live) {
            // Empty tile (four of five on the KITTI-like clouds): every pixel is act(0 + bias).  Written straight from the bias -- no
            // weight prefetch, no LDS, no further barrier; bit-identical to the full epilogue on zero accumulators.
            const int b = tile / (tiles_y * tiles_x);
            const int trem = tile - b * tiles_y * tiles_x;
            const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
            const int px = tid >> 4, ch = tid & 15;
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bv[j] = 0.0f + (bias ? bias[blockIdx.y * 128 + ch * 8 + j] : 0.0f);
                if (p.relu) bv[j] = __builtin_fmaxf(bv[j], 0.0f);
            }
            const uint4 v = make_uint4(pack2<T>(bv[0], bv[1]), pack2<T>(bv[2], bv[3]), pack2<T>(bv[4], bv[5]), pack2<T>(bv[6], bv[7]));
            uint4 *y4 = reinterpret_cast<uint4 *>(y);
            const int ox = x0 + px;
#pragma unroll
            for (int ty_ = 0; ty_ < TH; ++ty_) {
                const int oy = y0 + ty_;
                if (oy < p.h && ox < p.w) y4[(((size_t)b * p.h + oy) * p.w + ox) * (p.cout / 8) + blockIdx.y * 16 + ch] = v;
            }
            return;
        }
    } else if constexpr (ROLL >= 2) {
        if (nbr_masks) issue_halo2_lazy(tile, nmask, x, bg_in);
        else issue_halo2(tile, x);
    } else issue_halo(tile);
#ifdef SEC_CONV_TIMELINE
    long long tl_issue = 0, tl_eb1 = 0, tl_eb2 = 0;
    if (tl) tl_issue = clock64();
#endif
    {
        const int b = tile / (tiles_y * tiles_x);
        const int trem = tile - b * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        uint4 bq[2][4];
        constexpr int RD = 8;
        uint4 br[RD];                               // ROLL: ring of B fragments, each loaded RD - 1 k-steps ahead of its use
        // ROLL == 2: B fragments by buffer loads -- one lane-offset VGPR + a SCALAR chunk offset per fragment (24 global pointers
        // per dx cost 48 VGPRs and spilled)
        typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(wpk), 0, (int)(((X3 ? 18 : 9) * cin8 + 1) * p.cout * 16), 0x00020000);
        const unsigned wvoff = (unsigned)(hh * p.cout + n0 + r) * 16u;
        const unsigned wstep = (unsigned)p.cout * 16u;            // bytes per chunk row of the packed weights
        auto ld_b = [&](unsigned chunk) {
            return __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, chunk * wstep, 0));
        };
        if constexpr (ROLL >= 2) {
#pragma unroll
            for (int f = 0; f < RD - 1; ++f) br[f] = ld_b((f % 3) * 48 + (f / 3) * 2);
        } else if constexpr (ROLL) {
#pragma unroll
            for (int f = 0; f < RD - 1; ++f) br[f] = wlane[(size_t)f * 2 * p.cout];
        } else {
            load_b(0, bq[0]);
        }
        f32x16d acc[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
        __syncthreads();                            // halo landed
#ifdef SEC_CONV_TIMELINE
        if (tl) tl1 = clock64();
#endif
        // First RPN layer: its input is the scattered sparse-middle output, most 10 x 18 halos hold nothing but zeros and
        // the result is act(bias) exactly (0 * w accumulates to 0) -- one LDS sweep + a barrier decides, uniformly.
        bool live = true;
        if constexpr (GATHER) live = gather_live != 0;   // uniform: decided from the site map before any DMA
        else if (p.zskip) {
            unsigned nz = 0;
            for (int e = tid; e < HENT; e += 256) {
                const uint4 v = hal[e];
                nz |= v.x | v.y | v.z | v.w;
            }
            live = __syncthreads_or((int)(nz != 0)) != 0;
        }
        // A fragments run one whole k-step (MT MFMAs = 128 cycles of matrix pipe) ahead of their use, across iterations
        uint4 af[2][MT];
        int tap = 0, kc = 0;
        if (live && !ROLL) load_a(0, 0, 0, af[0]);
        if constexpr (ROLL >= 2) {
            // Shared-row fragments.  An m-tile pairs output rows (mt, mt + 4), so its A fragment for kernel row dy is the pair of
            // halo rows (mt + dy, mt + dy + 4) = F[mt + dy]: for one (dx, k-step) SIX fragments F[0..5] feed all 3 x 4 = 12 MFMAs
            // of the three kernel rows -- half the ds_read_b128 traffic of the ROLL == 1 loop (12 reads per 12 MFMAs), same B
            // stream (one fragment per four MFMAs), same accumulators.  Rolled over dx, unrolled over 8 k-steps x 3 kernel rows;
            // B fragment j = (k-step, dy) of this dx sits at chunk dy * 48 + dx * 16 + 2 * k-step of the packed [tap][cin8][cout].
            static_assert(ROLL < 2 || (KC == 2 && CH == 16 && TH == 8), "8 k-steps per tap, rows (mt, mt + 4)");
            const char *halb = reinterpret_cast<const char *>(hal);
            const int colr = r & 15;
            const unsigned base0 = (unsigned)((r >> 4) * 4 * HW_ + colr) * (CH * 16);
            uint4 fr[2][6];
            auto load_f2 = [&](unsigned bdx, unsigned key, int cidx, int j0, uint4 (&dst)[6]) {
                const unsigned a = bdx + (((unsigned)cidx << 4) ^ key);
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j) dst[j] = *reinterpret_cast<const uint4 *>(halb + a + j * HW_ * (CH * 16));
            };
            auto frag_off = [](int j) { return (j % 3) * 48 + (j / 3) * 2; };
            if (live) {
                constexpr int NPASS = X3 ? 3 : 1;
#pragma unroll 1
                for (int pass = 0; pass < NPASS; ++pass) {
                    // X3: pass 1 = the same halo (x_hi) against w_lo (the second half of the packed weights); pass 2 = x_lo's halo
                    // against w_hi.  The B ring is re-primed per pass (its tail prefetched fragments of the pass that just ended).
                    const unsigned wp = (X3 && pass == 1) ? 9u * cin8 : 0u;
                    if (X3 && pass > 0) {
                        if (pass == 2) {
                            __syncthreads();                    // every wave is done reading x_hi's halo
                            if (nbr_masks) issue_halo2_lazy(tile, nmask, x_lo, bg_in_lo);
                            else issue_halo2(tile, x_lo);
                        }
#pragma unroll
                        for (int f = 0; f < RD - 1; ++f) br[f] = ld_b(wp + (f % 3) * 48 + (f / 3) * 2);
                        if (pass == 2) __syncthreads();         // x_lo's halo landed
                    }
                    {
                        const unsigned key0 = (unsigned)(hh ^ (colr & 15)) << 4;
#pragma unroll
                        for (int h = 0; h < 3; ++h) load_f2(base0, key0, 0, 2 * h, fr[0]);
                    }
#pragma unroll 1
                    for (int dx = 0; dx < 3; ++dx) {
                        const int dxn = dx < 2 ? dx + 1 : 2;
                        const unsigned bdx = base0 + dx * (CH * 16), bdn = base0 + dxn * (CH * 16);
                        const unsigned key = (unsigned)(hh ^ ((colr + dx) & 15)) << 4, keyn = (unsigned)(hh ^ ((colr + dxn) & 15)) << 4;
                        const unsigned ct = wp + dx * 16, cn = wp + dxn * 16;
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int j = ks * 3 + dy;
                                br[(j + RD - 1) % RD] = j + RD - 1 < 24 ? ld_b(ct + frag_off(j + RD - 1)) : ld_b(cn + frag_off(j + RD - 1 - 24));
                                // a third of the next k-step's six fragments per kernel row
                                if (ks + 1 < 8) load_f2(bdx, key, (ks + 1) * 2, 2 * dy, fr[(ks + 1) & 1]);
                                else load_f2(bdn, keyn, 0, 2 * dy, fr[0]);
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) acc[mt] = MfmaD<T>::run(br[j % RD], fr[ks & 1][mt + dy], acc[mt]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
            }
            live = false;
        } else if constexpr (ROLL) {
            // Rolled over the kernel ROW dy, unrolled over its 3 taps x 8 k-steps (three turns of the 8-deep B ring; fragment g
            // of the packed weights [tap][cin8][cout] sits at chunk 2 g).  The halo address of (pixel, tap, chunk) splits into
            //   lane base + dy * row pitch            (one VGPR, updated per dy)
            //   ^ chunk swizzle                        (key = halo column & 15 = ((r & 15) + dx) & 15: depends on dx only)
            //   + (mt * 2 * HW_ + dx) * 256            (ds_read immediate)
            // so a k-step costs two VALU address instructions instead of a `% HW_` per m-tile (quarter-rate multiplies that
            // competed with the MFMA issue slots).
            static_assert(!ROLL || (KC == 2 && CH == 16), "ring of 8 == k-steps per tap");
            const char *halb = reinterpret_cast<const char *>(hal);
            const int colr = r & 15;
            unsigned kk[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) kk[dx] = (unsigned)(hh ^ ((colr + dx) & 15)) << 4;
            const unsigned base0 = (unsigned)((r >> 4) * HW_ + colr) * (CH * 16);
            auto load_a2 = [&](unsigned bdy, int dx, int cidx, uint4 (&dst)[MT]) {
                const unsigned a = bdy + (((unsigned)cidx << 4) ^ kk[dx]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) dst[mt] = *reinterpret_cast<const uint4 *>(halb + a + (mt * 2 * HW_ + dx) * (CH * 16));
            };
            if (live) {
                load_a2(base0, 0, 0, af[0]);
#pragma unroll 1
                for (int dy = 0; dy < 3; ++dy) {
                    // unconditional prefetches (the last row re-reads its own first fragments): static wait counts, no branches
                    const int dyn = dy < 2 ? dy + 1 : 2;
                    const unsigned bdy = base0 + dy * (HW_ * CH * 16), bdn = base0 + dyn * (HW_ * CH * 16);
                    const uint4 *wt = wlane + (size_t)dy * 48 * p.cout, *wn = wlane + (size_t)dyn * 48 * p.cout;
#pragma unroll
                    for (int j = 0; j < 24; ++j) {
#if defined(SEC_CONV2D_ABL) && SEC_CONV2D_ABL == 3
                        br[(j + 7) & 7] = wlane[(size_t)((j + 7) & 1) * 2 * p.cout];   // ablation build: B fragments always from two hot lines
#else
                        br[(j + 7) & 7] = j + 7 < 24 ? wt[(size_t)(j + 7) * 2 * p.cout] : wn[(size_t)(j + 7 - 24) * 2 * p.cout];
#endif
                        if (j + 1 < 24) load_a2(bdy, (j + 1) / 8, ((j + 1) % 8) * 2, af[(j + 1) & 1]);
                        else load_a2(bdn, 0, 0, af[0]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = MfmaD<T>::run(br[j & 7], af[j & 1][mt], acc[mt]);
                        // pins [B prefetch, 4 A reads, 4 MFMAs] per k-step: free scheduling sinks the loads towards their use
                        // (126 VGPRs, 96 us instead of 84 us); a 3-deep A ring measured no gain
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            live = false;                           // skip the two-stage loop below
        }
#pragma unroll 2
        for (int it = 0; live && it < NIT; ++it) {
            const int cur = it & 1;
            if (it + 1 < NIT) load_b(it + 1, bq[cur ^ 1]);
            int ntap = tap, nkc = kc + 1;
            if (nkc == KC) { nkc = 0; ++ntap; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) load_a(tap, kc, s + 1, af[(s + 1) & 1]);
                else if (it + 1 < NIT) load_a(ntap, nkc, 0, af[0]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MfmaD<T>::run(bq[cur][s], af[s & 1][mt], acc[mt]);   // D^T: see store_tile_t
            }
            tap = ntap;
            kc = nkc;
        }
#ifdef SEC_CONV_TIMELINE
        if (tl) tl2 = clock64();
#endif
        if constexpr (CIN == 128 && TH == 8) {
            // Epilogue through LDS (the halo buffer is free now): the direct form stores 32-byte pieces of 32 different pixel rows
            // per instruction -- every 256-byte row is completed by eight instructions of four waves -- and the per-workgroup
            // timeline showed the store tail as long as the MFMA loop (13 500 of 37 400 clocks).  Here the waves drop their
            // bias + ReLU'd 16-bit results into a [pixel][channel] tile in LDS (row pitch 272 B: conflict-free 16-byte writes),
            // and after one barrier every instruction of the workgroup stores ONE 4 KB tile row (16 pixels x 256 B, contiguous).
            constexpr int PITCH = 17;                       // uint4 per pixel row: 16 + 1 pad
            __syncthreads();                                // every wave is done reading the halo
#ifdef SEC_CONV_TIMELINE
            if (tl) tl_eb1 = clock64();
#endif
            uint4 *ot = halo_smem;
            float4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = bias ? *reinterpret_cast<const float4 *>(bias + n0 + 8 * g + 4 * hh) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            // This phase runs beside two other waves' MFMA loops on the SIMD and gets an issue slot every ~18 clocks (timeline:
            // 7 300 clocks for the ~900 instructions of the first form -- a branch per value for the ReLU, one convert + merge per
            // value, ds_bpermute + selects for the half-wave exchange).  Here: add, max, one convert per PAIR, and
            // v_permlane32_swap, which IS the exchange: after swap(pk[2pr], pk[2pr+1]) the low half-wave holds both halves of
            // chunk 2pr and the high half-wave both halves of chunk 2pr+1 -- no selects.  ~200 instructions.
            auto put_tile = [&](auto relu_tag, auto lo_tag) {
                constexpr bool RELU = decltype(relu_tag)::value;
                constexpr bool LOW = decltype(lo_tag)::value;     // X3: the residual plane v - bf16(v) instead of bf16(v)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    unsigned lo[4], hi[4];                  // channels 8g+4hh+(0,1) and +(2,3) of this lane's pixel
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[mt][4 * g] + bv[g].x, acc[mt][4 * g + 1] + bv[g].y, acc[mt][4 * g + 2] + bv[g].z, acc[mt][4 * g + 3] + bv[g].w};
                        if (RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.0f);
                        }
                        lo[g] = pack2<T>(v[0], v[1]);
                        hi[g] = pack2<T>(v[2], v[3]);
                        if constexpr (LOW) {
                            lo[g] = pack2<T>(v[0] - __uint_as_float(lo[g] << 16), v[1] - __uint_as_float(lo[g] & 0xffff0000u));
                            hi[g] = pack2<T>(v[2] - __uint_as_float(hi[g] << 16), v[3] - __uint_as_float(hi[g] & 0xffff0000u));
                        }
                    }
                    const int q = ROLL >= 2 ? (mt + 4 * (r >> 4)) * 16 + (r & 15) : mt * 32 + r;   // tile pixel of (m-tile, lane)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const auto sx = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
                        ot[q * PITCH + wv * 4 + 2 * pr + hh] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    }
                }
            };
            // TAIL: first-GEMM weights of the fused tail (this wave: all 128 pixels x mid channels [32 wv, 32 wv + 32)), in flight over the tile exchange
            uint4 tq[TAIL ? 8 : 1];
            if constexpr (TAIL) {
                const uint4 *wl = reinterpret_cast<const uint4 *>(tail.w1) + (size_t)hh * 128 + wv * 32 + r;
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) tq[s_] = wl[(size_t)s_ * 2 * 128];
            }
            if (p.relu) put_tile(std::true_type{}, std::false_type{});
            else put_tile(std::false_type{}, std::false_type{});
            __syncthreads();
#ifdef SEC_CONV_TIMELINE
            if (tl) tl_eb2 = clock64();
#endif
            if constexpr (TAIL) {
                // The conv's 16-bit output tile [pixel q = ty * 16 + px][16 chunks of 8 channels] (pitch 17 uint4: the 16 lanes a ds_read_b128
                // serves together hit 16 different 16-byte slots) is exactly k_conv1x1_chain's x tile: y = W2 relu(W1 x + b1) + b2 on it --
                // same fragments, same k order, same roundings as the separate launch (bit-identical heads); the conv's output never
                // reaches memory (28 MB written + 28 MB read per batch of 8) and the tail's launch, list lookups and lock-stepped round go.
                f32x16d a1[4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int i = 0; i < 16; ++i) a1[a][i] = 0.0f;
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a1[mt] = MfmaD<T>::run(tq[s_], ot[(mt * 32 + r) * PITCH + s_ * 2 + hh], a1[mt]);
                // second-GEMM weights: wave = 64 pixels (wm) x 32 of the 64 head channels (wn)
                const int wm = wv & 1, wn = wv >> 1;
                uint4 cq[8];
                {
                    const uint4 *wl = reinterpret_cast<const uint4 *>(tail.w2) + (size_t)hh * 64 + wn * 32 + r;
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) cq[s_] = wl[(size_t)s_ * 2 * 64];
                }
                lds_barrier();                              // every wave has read all of the conv tile
                {
                    unsigned char *tb = reinterpret_cast<unsigned char *>(ot);
                    auto put_mid = [&](auto relu_tag) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 b1v = *reinterpret_cast<const float4 *>(tail.b1 + wv * 32 + 8 * g + 4 * hh);
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                                const int px = mt * 32 + r;
                                float v[4] = {a1[mt][4 * g] + b1v.x, a1[mt][4 * g + 1] + b1v.y, a1[mt][4 * g + 2] + b1v.z, a1[mt][4 * g + 3] + b1v.w};
                                if (decltype(relu_tag)::value) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.0f);
                                }
                                *reinterpret_cast<uint2 *>(tb + ((size_t)px * PITCH + (wv * 4 + g)) * 16 + hh * 8) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                            }
                        }
                    };
                    if (tail.relu1 & 1) put_mid(std::true_type{});
                    else put_mid(std::false_type{});
                }
                lds_barrier();
                f32x16d a2[2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < 16; ++i) a2[a][i] = 0.0f;
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) a2[mt] = MfmaD<T>::run(cq[s_], ot[(wm * 64 + mt * 32 + r) * PITCH + s_ * 2 + hh], a2[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int px = wm * 64 + mt * 32 + r;
                    const int oy = y0 + (px >> 4), ox = x0 + (px & 15);
                    const bool ok = oy < p.h && ox < p.w;
                    T *ypix = reinterpret_cast<T *>(tail.y) + (ok ? (((size_t)b * p.h + oy) * p.w + ox) * 64 : 0);
                    store_tile_t<T>(a2[mt], tail.b2, wn * 32, 0, ypix, ok, hh);
                }
#ifdef SEC_CONV_TIMELINE
                if (tl && tid == 0) atomicSub(&g_cu_resident[cu_key], 1);     // (no timeline record for the tail form; keep the residency count right)
#endif
                return;
            }
            uint4 *y4 = reinterpret_cast<uint4 *>(y);
#pragma unroll
            for (int ty_ = 0; ty_ < TH; ++ty_) {            // one tile row (16 pixels x 16 chunks) per instruction
                const int px = tid >> 4, ch = tid & 15;
                const int oy = y0 + ty_, ox = x0 + px;
                const uint4 v = ot[(ty_ * 16 + px) * PITCH + ch];
#if defined(SEC_CONV2D_ABL) && SEC_CONV2D_ABL == 4
                if (v.x != 0x12345678u) continue;                          // ablation build: no output stores
#endif
                // (non-temporal stores here: 76.0 vs 76.5 us, within noise)
                if (oy < p.h && ox < p.w) y4[(((size_t)b * p.h + oy) * p.w + ox) * (p.cout / 8) + blockIdx.y * 16 + ch] = v;
            }
            if constexpr (X3) {                             // the residual plane, through the same LDS tile
                __syncthreads();
                if (p.relu) put_tile(std::true_type{}, std::true_type{});
                else put_tile(std::false_type{}, std::true_type{});
                __syncthreads();
                uint4 *yl4 = reinterpret_cast<uint4 *>(y_lo);
#pragma unroll
                for (int ty_ = 0; ty_ < TH; ++ty_) {
                    const int px = tid >> 4, ch = tid & 15;
                    const int oy = y0 + ty_, ox = x0 + px;
                    const uint4 v = ot[(ty_ * 16 + px) * PITCH + ch];
                    if (oy < p.h && ox < p.w) yl4[(((size_t)b * p.h + oy) * p.w + ox) * (p.cout / 8) + blockIdx.y * 16 + ch] = v;
                }
            }
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int q = (mtb + mt) * 32 + r;
            const int oy = y0 + (q >> 4), ox = x0 + (q & 15);
            const bool ok = oy < p.h && ox < p.w;
            T *ypix = y + (((size_t)b * p.h + oy) * p.w + ox) * p.cout;
            store_tile_t<T>(acc[mt], bias, n0, p.relu, ypix, ok, hh);
        }
        }
#ifdef SEC_CONV_TIMELINE
        if (tl && tid == 0 && blockIdx.y == 0) {
            long long *rec = tl + (size_t)blockIdx.x * 8;
            rec[0] = tl0; rec[1] = tl1; rec[2] = tl2 + ((long long)resident_at_start << 56); rec[3] = clock64();
            rec[7] = ((tl_issue - tl0) >> 4 & 0xffff) | ((tl_eb1 - tl2) >> 4 & 0xffff) << 16 | ((tl_eb2 - tl2) >> 4 & 0xffff) << 32;
            rec[4] = wl0; rec[5] = wall_clock64(); rec[6] = cu_key;       // constant 100 MHz counter: real time, comparable across CUs
            atomicSub(&g_cu_resident[cu_key], 1);
        }
#endif
    }
}

#ifdef SEC_CONV_TIMELINE
// LDS canary (debug builds): a workgroup fills 40 KB of LDS with a pattern, idles for `spin` clocks and counts the words that
// changed -- run beside another kernel to see whether that kernel writes outside its own LDS allocation.
__global__ __launch_bounds__(256) void k_lds_canary(int *errors, int spin, int *first_bad) {
    __shared__ unsigned canary[10240];
    for (int i = threadIdx.x; i < 10240; i += 256) canary[i] = 0xC0DE0000u + i;
    __syncthreads();
    const long long until = clock64() + spin;
    while (clock64() < until) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < 10240; i += 256)
        if (canary[i] != 0xC0DE0000u + i) { ++bad; atomicMin(first_bad, i); }
    if (bad) atomicAdd(errors, bad);
}
extern "C" __attribute__((visibility("default"))) int sec__debug_lds_canary(int *errors, int *first_bad, int blocks, int spin, void *stream) {
    hipLaunchKernelGGL(k_lds_canary, dim3(blocks), dim3(256), 0, (hipStream_t)stream, errors, spin, first_bad);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" __attribute__((visibility("default"))) int sec__debug_timeline2(long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline2), &buf, sizeof(buf)) == hipSuccess ? 0 : -4;
}
#endif

template <typename T, int CIN, int TH, int ROLL = 0, bool GATHER = false, int NSPLIT = 1, bool X3 = false, bool TAIL = false>
static int launch_conv2d_halo_reg(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, hipStream_t st,
                                  const int *site_map = nullptr, unsigned feat_bytes = 0, const unsigned short *tile_order = nullptr,
                                  const int *live_counts = nullptr, const void *background = nullptr,
                                  const unsigned short *nbr_masks = nullptr, const void *bg_in = nullptr,
                                  const void *x_lo = nullptr, void *y_lo = nullptr, const void *background_lo = nullptr,
                                  const void *bg_in_lo = nullptr, const ConvTailArgs &tail = ConvTailArgs{}) {
    constexpr size_t lds_tile = (size_t)(TH + 2) * 18 * (CIN / 8) * 16;
    // (padding the dynamic LDS to hold 2 instead of 3 workgroups per CU was measured in round 3: slower in every combination)
    const long lds_pad = 0;
    const size_t lds = lds_tile + (size_t)lds_pad;
    static bool configured = false;
    auto fn = k_conv2d_halo_reg<T, CIN, TH, ROLL, GATHER, NSPLIT, X3, TAIL>;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ty = div_up(p.h, TH), tx = div_up(p.w, 16);
    const int per_xcd = div_up(p.batch * ty * tx, 8);
    const int gx = per_xcd * 8;
    if (TAIL) set_last_kernel("k_conv2d_halo_reg<%s, %d, %d, %d, false, 1, false, true>", dtype_name<T>(), CIN, TH, ROLL);
    else if (X3) set_last_kernel("k_conv2d_halo_reg<%s, %d, %d, %d, false, 1, true>", dtype_name<T>(), CIN, TH, ROLL);
    else if (NSPLIT == 1) set_last_kernel("k_conv2d_halo_reg<%s, %d, %d, %d, %s>", dtype_name<T>(), CIN, TH, ROLL, GATHER ? "true" : "false");
    else set_last_kernel("k_conv2d_halo_reg<%s, %d, %d, %d, %s, %d>", dtype_name<T>(), CIN, TH, ROLL, GATHER ? "true" : "false", NSPLIT);
    hipLaunchKernelGGL(fn, dim3(gx, p.cout / (128 / NSPLIT)), dim3(256), lds, st, (const T *)x, (const T *)wpk, bias, (T *)y, p, ty, tx, per_xcd,
                       site_map, feat_bytes, tile_order, live_counts, (const T *)background, nbr_masks, (const T *)bg_in, (const T *)x_lo, (T *)y_lo,
                       (const T *)background_lo, (const T *)bg_in_lo, tail);
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
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MfmaD<T>::run(bf[nt], af[mt], acc[mt][nt]);   // D^T: see store_tile_t
        }
        if (cc == CC - 1) {
            const long long m0 = (tile0 + it / CC) * BM;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const long long pix = m0 + wm * 64 + mt * 32 + r;
                T *ypix = y + (size_t)pix * p.cout;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    store_tile_t<T>(acc[mt][nt], bias, n0 + wn * (BN / 2) + nt * 32, p.relu, ypix, pix < p.m, hh);
            }
        }
        __syncthreads();
    }
}


// ---- fused 1x1 chain: y = W2 * relu(W1 * x + b1) + b2 ------------------------------------------------------------
// The RPN tail is two back-to-back 1x1 convolutions -- the ConvTranspose2d(k=1,s=1) "deblock" (128 -> 128, BN, ReLU;
// rpn.py:275-285) and the merged box/cls/dir heads (128 -> 64 padded; rpn.py:386-391) -- and nothing else reads the
// deblock output at inference.  Run separately they move 72 MB in + 72 MB out + 72 MB in + 36 MB out; fused, the
// 128-channel intermediate lives only in LDS: 72 MB in, 36 MB out.  One workgroup = one 128-pixel tile; weights are
// never staged (each wave streams its own B fragments from L2, as in k_conv2d_halo_reg); the x tile and then the
// intermediate share one 32 KB LDS buffer, so five workgroups fit a CU's LDS and three its registers.
// TILES: the workgroup's 128 pixels are an 8 x 16 tile of the [batch][h][w] map taken from the live-tile lists of sec_rpn_tile_live
// (the last 3x3 conv's: a 1x1 conv reaches no farther), spread evenly over the XCDs; the workgroups behind them copy the other
// tiles from `background` = this kernel's output for an empty frame ([h][w][N2]) -- the scheme of k_conv2d_halo_reg.
// (64-channel heads, NT2 == 1: four workgroups per CU -- 1024 slots hold the 650-900 live tiles of a car.fhd batch of 8 in ONE round;
// at three per CU the 862-tile list of the last conv left ~100 tiles for a second round on an otherwise empty chip)
template <typename T, int NT2, bool TILES = false>    // NT2 = 32-wide cout tiles per wave in the second GEMM: cout2 = 64 * NT2
__global__ __launch_bounds__(256, NT2 == 1 ? 4 : 3) void k_conv1x1_chain(const T *__restrict__ x, const T *__restrict__ w1pk,
                                                          const float *__restrict__ b1, const T *__restrict__ w2pk,
                                                          const float *__restrict__ b2, T *__restrict__ y, long long m,
                                                          int relu1, int batch = 0, int h = 0, int w = 0, int per_xcd = 0,
                                                          const unsigned short *__restrict__ tile_order = nullptr,
                                                          const int *__restrict__ live_counts = nullptr,
                                                          const T *__restrict__ background = nullptr) {
    constexpr int BM = 128, C = 128, CH = C / 8, N2 = 64 * NT2;
    __shared__ uint4 tile[BM * CH];                 // [pixel][16-byte chunk ^ (pixel & 15)]: x, then relu(W1 x + b1)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    long long m0 = (long long)blockIdx.x * BM;
    int tf = 0, ty0 = 0, tx0 = 0;                   // TILES: frame and origin of this workgroup's tile
    if constexpr (TILES) {
        const int tiles_x = (w + 15) / 16, tpf = ((h + 7) / 8) * tiles_x, ntile = batch * tpf;
        const int xcd = blockIdx.x % 8, local = blockIdx.x / 8;
        if (local >= per_xcd) return;
        int n_live = 0;
        for (int f = 0; f < batch; ++f) n_live += live_counts[f];
        // above g_list_max_live_q8 / 256 live: every tile in the plain order (k_conv2d_halo_reg) -- unless x holds its live tiles ONLY (relu1 bit 1,
        // SEC_CHAIN_X_LIVE_ONLY: a lazy producer left the others unwritten; the lists name exactly the tiles that exist)
        const bool lists = (relu1 & 2) != 0 || n_live * 256 <= ntile * g_list_max_live_q8;
        const int per_live = (n_live + 7) >> 3;
        int item;
        bool is_live = false;
        if (!lists) {
            item = xcd * per_xcd + local;
            if (item >= ntile) return;
        } else if (local < per_live) {
            item = xcd * per_live + local;
            is_live = item < n_live;
            if (!is_live) item -= n_live;
        } else {
            item = 8 * per_live - n_live + (local - per_live) * 8 + xcd;
        }
        if (lists && !is_live) {
            if (!background) return;                // lazy consumers (sec_predict_*_lazy): background tiles are never materialised
            constexpr int kCopyTiles = 4, NC = N2 / 8, PER = BM * NC / 256;      // 16-byte chunks per pixel / per thread and tile
            const int n_bg = ntile - n_live;
            const uint4 *e4 = reinterpret_cast<const uint4 *>(background);
            uint4 *y4 = reinterpret_cast<uint4 *>(y);
#pragma unroll 1
            for (int q = 0; q < kCopyTiles; ++q) {
                int it = item * kCopyTiles + q;
                if (it >= n_bg) return;
                int f = 0;
                while (it >= tpf - live_counts[f]) it -= tpf - live_counts[f++];
                const int t = tile_order[f * tpf + tpf - 1 - it];
                const int y0 = (t / tiles_x) * 8, x0 = (t % tiles_x) * 16;
                uint4 v[PER];
#pragma unroll
                for (int c = 0; c < PER; ++c) {
                    const int e = c * 256 + tid, px = e / NC, ch = e - px * NC;
                    const int oy = y0 + (px >> 4), ox = x0 + (px & 15);
                    v[c] = (oy < h && ox < w) ? e4[((size_t)oy * w + ox) * NC + ch] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < PER; ++c) {
                    const int e = c * 256 + tid, px = e / NC, ch = e - px * NC;
                    const int oy = y0 + (px >> 4), ox = x0 + (px & 15);
                    if (oy < h && ox < w) y4[(((size_t)f * h + oy) * w + ox) * NC + ch] = v[c];
                }
            }
            return;
        }
        int t;
        if (lists) {
            while (item >= live_counts[tf]) item -= live_counts[tf++];
            t = tile_order[tf * tpf + item];
        } else {
            tf = item / tpf;
            t = item - tf * tpf;
        }
        ty0 = (t / tiles_x) * 8;
        tx0 = (t % tiles_x) * 16;
    }
    // pixel px (0 .. 127) of this workgroup -> index in the [pixels] view, -1 when it lies outside the map
    auto gpix = [&](int px) -> long long {
        if constexpr (TILES) {
            const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
            return (oy < h && ox < w) ? ((long long)tf * h + oy) * w + ox : -1ll;
        } else {
            return m0 + px < m ? m0 + px : -1ll;
        }
    };
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    const uint4 *w1 = reinterpret_cast<const uint4 *>(w1pk);
    const uint4 *w2 = reinterpret_cast<const uint4 *>(w2pk);
    const uint4 *zero16 = w1 + (size_t)CH * C;
#pragma unroll
    for (int j = 0; j < BM * CH / 64 / 4; ++j) {
        const int e = (j * 4 + wv) * 64 + lane, px = e / CH, slot = e - px * CH;
        const long long gp = gpix(px);
        const uint4 *src = gp >= 0 ? x4 + gp * CH + (slot ^ (px & 15)) : zero16;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&tile[(j * 4 + wv) * 64], 16, 0, 0);
    }
    // GEMM 1: this wave = all 128 pixels x mid channels [32 wv, 32 wv + 32)
    uint4 bq[8];
    {
        const uint4 *wl = w1 + (size_t)hh * C + wv * 32 + r;
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[s] = wl[(size_t)s * 2 * C];
    }
    f32x16d acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int px = mt * 32 + r;
            acc[mt] = MfmaD<T>::run(bq[s], tile[px * CH + ((s * 2 + hh) ^ (px & 15))], acc[mt]);   // D^T: see store_tile_t
        }
    // second-GEMM weights: wave = 64 pixels (wm) x N2 / 2 couts (wn); issued now, needed after the two barriers
    // (launch bound 4 per CU -- 128 VGPRs -- spills 8 ... 14 dwords whether these loads sit here or are staggered through the
    // first GEMM's steps)
    const int wm = wv & 1, wn = wv >> 1;
    uint4 cq[8][NT2];
    {
        const uint4 *wl = w2 + (size_t)hh * N2 + wn * (N2 / 2) + r;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) cq[s][nt] = wl[(size_t)s * 2 * N2 + nt * 32];
    }
    lds_barrier();                                  // every wave has read all of x
    {   // intermediate -> LDS (bias, ReLU, 16-bit): lane owns pixel mt*32 + r, channels 32 wv + 8 g + 4 hh + (0..3)
        unsigned char *tb = reinterpret_cast<unsigned char *>(tile);
        auto put = [&](auto relu_tag) {         // one uniform branch for the ReLU, pair converts (see store_tile_tb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(b1 + wv * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int px = mt * 32 + r;
                    float v[4] = {acc[mt][4 * g] + bv.x, acc[mt][4 * g + 1] + bv.y, acc[mt][4 * g + 2] + bv.z, acc[mt][4 * g + 3] + bv.w};
                    if (decltype(relu_tag)::value) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.0f);
                    }
                    *reinterpret_cast<uint2 *>(tb + ((size_t)px * CH + ((wv * 4 + g) ^ (px & 15))) * 16 + hh * 8) =
                        make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                }
            }
        };
        if (relu1 & 1) put(std::true_type{});
        else put(std::false_type{});
    }
    lds_barrier();
    f32x16d acc2[2][NT2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NT2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[a][c][i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int px = wm * 64 + mt * 32 + r;
            const uint4 af = tile[px * CH + ((s * 2 + hh) ^ (px & 15))];
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc2[mt][nt] = MfmaD<T>::run(cq[s][nt], af, acc2[mt][nt]);
        }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long pix = gpix(wm * 64 + mt * 32 + r);
        T *ypix = y + (size_t)(pix >= 0 ? pix : 0) * N2;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
            store_tile_t<T>(acc2[mt][nt], b2, wn * (N2 / 2) + nt * 32, 0, ypix, pix >= 0, hh);
    }
}

// The fused 1x1 tail for fp32 networks (sec_conv1x1_chain_x3): the same two GEMMs with every operand as a (hi, lo) bf16 pair and
// three MFMAs per product term (x_hi w_hi + x_hi w_lo + x_lo w_hi, fp32 accumulation; see k_conv2d_halo_reg<..., X3>).  The last
// split-operand 3x3 conv hands over its two planes; the bias + ReLU'd intermediate is split again into two LDS tiles; the heads leave
// as fp32 [pixels][64 * NT2].  What it replaces: merging the planes to fp32 (41 us) and two fp32 GEMMs / MIOpen 1x1 convolutions with
// their bias / ReLU passes (~280 us for 14 GFLOP at batch 8).
template <int NT2>
__global__ __launch_bounds__(256, 2) void k_conv1x1_chain_x3(const __hip_bfloat16 *__restrict__ x_hi, const __hip_bfloat16 *__restrict__ x_lo,
                                                             const __hip_bfloat16 *__restrict__ w1pk, const float *__restrict__ b1,
                                                             const __hip_bfloat16 *__restrict__ w2pk, const float *__restrict__ b2,
                                                             float *__restrict__ y, long long m, int relu1) {
    using T = __hip_bfloat16;
    constexpr int BM = 128, C = 128, CH = C / 8, N2 = 64 * NT2;
    __shared__ uint4 tile[2][BM * CH];              // [plane][pixel][16-byte chunk ^ (pixel & 15)]: x, then relu(W1 x + b1)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const uint4 *w1 = reinterpret_cast<const uint4 *>(w1pk);          // [hi | lo][cin8][128]
    const uint4 *w2 = reinterpret_cast<const uint4 *>(w2pk);          // [hi | lo][cin8][N2]
    const uint4 *zero16 = w1 + (size_t)2 * CH * C;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const uint4 *x4 = reinterpret_cast<const uint4 *>(pl ? x_lo : x_hi);
#pragma unroll
        for (int j = 0; j < BM * CH / 64 / 4; ++j) {
            const int e = (j * 4 + wv) * 64 + lane, px = e / CH, slot = e - px * CH;
            const uint4 *src = m0 + px < m ? x4 + (m0 + px) * CH + (slot ^ (px & 15)) : zero16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&tile[pl][(j * 4 + wv) * 64], 16, 0, 0);
        }
    }
    // GEMM 1: this wave = all 128 pixels x mid channels [32 wv, 32 wv + 32)
    uint4 bh[8], bl[8];
    {
        const uint4 *wl = w1 + (size_t)hh * C + wv * 32 + r;
#pragma unroll
        for (int s = 0; s < 8; ++s) { bh[s] = wl[(size_t)s * 2 * C]; bl[s] = wl[(size_t)CH * C + (size_t)s * 2 * C]; }
    }
    f32x16d acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int px = mt * 32 + r, at = px * CH + ((s * 2 + hh) ^ (px & 15));
            const uint4 ahi = tile[0][at], alo = tile[1][at];
            acc[mt] = MfmaD<T>::run(bh[s], ahi, acc[mt]);
            acc[mt] = MfmaD<T>::run(bl[s], ahi, acc[mt]);
            acc[mt] = MfmaD<T>::run(bh[s], alo, acc[mt]);
        }
    const int wm = wv & 1, wn = wv >> 1;
    lds_barrier();                                  // every wave has read all of x
    {   // intermediate -> the two LDS tiles (bias, ReLU in fp32, then hi / residual): lane owns pixel mt*32 + r, channels 32 wv + 8 g + 4 hh + (0..3)
        unsigned char *tbh = reinterpret_cast<unsigned char *>(tile[0]), *tbl = reinterpret_cast<unsigned char *>(tile[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4 *>(b1 + wv * 32 + 8 * g + 4 * hh);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int px = mt * 32 + r;
                float v[4] = {acc[mt][4 * g] + bv.x, acc[mt][4 * g + 1] + bv.y, acc[mt][4 * g + 2] + bv.z, acc[mt][4 * g + 3] + bv.w};
                if (relu1 & 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], 0.0f);
                }
                const unsigned h0 = pack2<T>(v[0], v[1]), h1 = pack2<T>(v[2], v[3]);
                const size_t at = ((size_t)px * CH + ((wv * 4 + g) ^ (px & 15))) * 16 + hh * 8;
                *reinterpret_cast<uint2 *>(tbh + at) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(tbl + at) =
                    make_uint2(pack2<T>(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u)),
                               pack2<T>(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u)));
            }
        }
    }
    // second-GEMM weights: wave = 64 pixels (wm) x N2 / 2 couts (wn)
    uint4 ch_[8][NT2], cl_[8][NT2];
    {
        const uint4 *wl = w2 + (size_t)hh * N2 + wn * (N2 / 2) + r;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                ch_[s][nt] = wl[(size_t)s * 2 * N2 + nt * 32];
                cl_[s][nt] = wl[(size_t)CH * N2 + (size_t)s * 2 * N2 + nt * 32];
            }
    }
    lds_barrier();
    f32x16d acc2[2][NT2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NT2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[a][c][i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int px = wm * 64 + mt * 32 + r, at = px * CH + ((s * 2 + hh) ^ (px & 15));
            const uint4 ahi = tile[0][at], alo = tile[1][at];
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                acc2[mt][nt] = MfmaD<T>::run(ch_[s][nt], ahi, acc2[mt][nt]);
                acc2[mt][nt] = MfmaD<T>::run(cl_[s][nt], ahi, acc2[mt][nt]);
                acc2[mt][nt] = MfmaD<T>::run(ch_[s][nt], alo, acc2[mt][nt]);
            }
        }
    // D^T: lane owns pixel wm*64 + mt*32 + r, per group g four consecutive channels c0 + 8 g + 4 hh + (0..3): fp32, 16-byte stores
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long pix = m0 + wm * 64 + mt * 32 + r;
        if (pix >= m) continue;
        float *ypix = y + (size_t)pix * N2;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = wn * (N2 / 2) + nt * 32 + 8 * g + 4 * hh;
                float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (b2) bv = *reinterpret_cast<const float4 *>(b2 + c);
                *reinterpret_cast<float4 *>(ypix + c) = make_float4(acc2[mt][nt][4 * g] + bv.x, acc2[mt][nt][4 * g + 1] + bv.y,
                                                                    acc2[mt][nt][4 * g + 2] + bv.z, acc2[mt][nt][4 * g + 3] + bv.w);
            }
    }
}

template <typename T>
static int launch_conv1x1_chain_tiles(const void *x, int batch, int h, int w, const void *w1, const float *b1, const void *w2, const float *b2,
                                      int cout2, int relu1, const unsigned short *tile_order, const int *live_counts, const void *background,
                                      void *y, hipStream_t st) {
    const int ntile = batch * div_up(h, 8) * div_up(w, 16), per_xcd = div_up(ntile, 8);
    const long long m = (long long)batch * h * w;
    if (cout2 == 64)
        hipLaunchKernelGGL((k_conv1x1_chain<T, 1, true>), dim3(per_xcd * 8), dim3(256), 0, st, (const T *)x, (const T *)w1, b1, (const T *)w2, b2,
                           (T *)y, m, relu1, batch, h, w, per_xcd, tile_order, live_counts, (const T *)background);
    else
        hipLaunchKernelGGL((k_conv1x1_chain<T, 2, true>), dim3(per_xcd * 8), dim3(256), 0, st, (const T *)x, (const T *)w1, b1, (const T *)w2, b2,
                           (T *)y, m, relu1, batch, h, w, per_xcd, tile_order, live_counts, (const T *)background);
    return check_launch();
}

template <typename T>
static int launch_conv1x1_chain(const void *x, long long m, const void *w1, const float *b1, const void *w2, const float *b2,
                                int cout2, int relu1, void *y, hipStream_t st) {
    const int blocks = div_up(m, 128);
    if (cout2 == 64)
        hipLaunchKernelGGL((k_conv1x1_chain<T, 1>), dim3(blocks), dim3(256), 0, st, (const T *)x, (const T *)w1, b1, (const T *)w2, b2,
                           (T *)y, m, relu1);
    else
        hipLaunchKernelGGL((k_conv1x1_chain<T, 2>), dim3(blocks), dim3(256), 0, st, (const T *)x, (const T *)w1, b1, (const T *)w2, b2,
                           (T *)y, m, relu1);
    return check_launch();
}

#ifdef SEC_CONV2D_EXPERIMENTS
constexpr bool kConv2dExperiments = true;
#else
constexpr bool kConv2dExperiments = false;   // default build: halo_reg (3x3 s1), 1x1 kernels, LDS-DMA implicit GEMM (everything else)
#endif
static int conv2d_variant() {
    static int v = -1;
    if (v < 0) v = 13;  // 0 register staged, 1 LDS-DMA implicit GEMM, 2/3/4 halo tile 16x16 / 8x16 / 8x16 with 8 waves
    return v;
}

#include "dense_patch.hpp"

// SEC_CONV2D_PATCH=0: the strided / patch layers of the PointPillars RPN on the generic implicit GEMM (A/B; read once)
static bool conv2d_patch_enabled() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SEC_CONV2D_PATCH");
        v = (e && *e == '0') ? 0 : 1;
    }
    return v != 0;
}

template <typename T>
static int launch_conv2d(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, hipStream_t st) {
    dim3 block(kBlock);
    if (conv2d_variant() == 13 && conv2d_patch_enabled()) {
        const int rc = patch::dispatch<T>(x, wpk, bias, y, p, p.cout, st);
        if (rc != patch::kNotTaken) return rc;
    }
    if (conv2d_variant() == 14 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout % 128 == 0 && p.cin == 128)
        return launch_conv2d_halo_reg<T, 128, 8>(x, wpk, bias, y, p, st);   // A/B: the two-stage loop with per-m-tile halo addressing
    if (conv2d_variant() == 13 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout % 128 == 0 && (p.cin == 128 || p.cin == 64))
        return p.cin == 128 ? launch_conv2d_halo_reg<T, 128, 8, 2>(x, wpk, bias, y, p, st) : launch_conv2d_halo_reg<T, 64, 8>(x, wpk, bias, y, p, st);
    // 256 input channels (third block of the PointPillars RPN, 50 x 50 maps): 4 x 16 tiles -- 55 KB of halo, two workgroups per CU,
    // 416 workgroups at batch 4 -- on the two-stage loop (the generic implicit GEMM ran these layers at 0.11 of the MFMA peak)
    if (conv2d_variant() == 13 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout % 128 == 0 && p.cin == 256)
        return launch_conv2d_halo_reg<T, 256, 4>(x, wpk, bias, y, p, st);
    // 64 -> 64 (first block of the PointPillars RPN, 200 x 200 maps): 64 output channels per workgroup, the waves split the pixels
    if (conv2d_variant() == 13 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout == 64 && p.cin == 64)
        return launch_conv2d_halo_reg<T, 64, 8, 0, false, 2>(x, wpk, bias, y, p, st);
    if (conv2d_variant() == 15 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout % 128 == 0 && p.cin == 128)
        return launch_conv2d_halo_reg<T, 128, 8, 1>(x, wpk, bias, y, p, st);   // A/B: one A fragment per MFMA (12 LDS reads per 12 MFMAs)
#ifdef SEC_CONV2D_EXPERIMENTS   // earlier 3x3 kernels (LDS weight slabs / rings), kept for A/B builds: DESIGN.md section 4
    if (conv2d_variant() >= 5 && conv2d_variant() <= 12 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.cout % 128 == 0 &&
        (p.cin == 128 || p.cin == 64)) {
        // 8 waves / 16 KB slabs / ring 2 with: 5 linear key   10 column key   11 fragment pipelining   12 both
        // 8: 4 waves (64 px x 64 cout per wave), column key
#define SEC_HALO_PIPE(NQ_, KS_, NB_, CK_, FP_)                                                                  \
    return p.cin == 128 ? launch_conv2d_halo_pipe<T, 128, 8, NQ_, KS_, NB_, CK_, FP_>(x, wpk, bias, y, p, st)   \
                        : launch_conv2d_halo_pipe<T, 64, 8, NQ_, KS_, NB_, CK_, FP_>(x, wpk, bias, y, p, st)
        switch (conv2d_variant()) {
        case 5: SEC_HALO_PIPE(4, 64, 2, false, false);
        case 8: SEC_HALO_PIPE(2, 64, 2, true, false);
        case 10: SEC_HALO_PIPE(4, 64, 2, true, false);
        case 11: SEC_HALO_PIPE(4, 64, 2, false, true);
        default: SEC_HALO_PIPE(4, 64, 2, true, true);
        }
#undef SEC_HALO_PIPE
    }
    if (conv2d_variant() >= 2 && conv2d_variant() <= 4 && p.ksize == 3 && p.stride == 1 && p.pad == 1 &&
        p.cout % 128 == 0 && (p.cin == 128 || p.cin == 64)) {
        if (conv2d_variant() == 2)
            return p.cin == 128 ? launch_conv2d_halo<T, 128, 16, 2>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 16, 2>(x, wpk, bias, y, p, st);
        if (conv2d_variant() == 4)   // 8 waves on a 16x8 tile: 64 px x 32 cout per wave, 4 waves / SIMD at 2 workgroups per CU
            return p.cin == 128 ? launch_conv2d_halo<T, 128, 8, 4>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 8, 4>(x, wpk, bias, y, p, st);
        return p.cin == 128 ? launch_conv2d_halo<T, 128, 8, 2>(x, wpk, bias, y, p, st) : launch_conv2d_halo<T, 64, 8, 2>(x, wpk, bias, y, p, st);
    }
#endif
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
    if (conv2d_variant() >= 1 || !kConv2dExperiments) {
        const int gx = (div_up(p.m, 128) + 7) / 8 * 8;   // multiple of 8 for the XCD-aware tile order
        // small images (the 50 x 50 third block of the PointPillars RPN: 80 pixel tiles x 2 cout tiles = 160 workgroups for 256 CUs):
        // 64-wide cout tiles double the workgroups; the input tile is re-read from L2.  (small_split = 0: always 128.
        static int small_split = -1;
        if (small_split < 0) small_split = 1;
        if (p.cout % 128 == 0 && !(small_split && (long long)gx * (p.cout / 128) < 384))
            hipLaunchKernelGGL((k_conv2d_nhwc_dma<T, 128>), dim3(gx, p.cout / 128), block, 0, st, (const T *)x,
                               (const T *)wpk, bias, (T *)y, p);
        else
            hipLaunchKernelGGL((k_conv2d_nhwc_dma<T, 64>), dim3(gx, p.cout / 64), block, 0, st, (const T *)x,
                               (const T *)wpk, bias, (T *)y, p);
        return check_launch();
    }
#ifdef SEC_CONV2D_EXPERIMENTS
    if (p.cout % 128 == 0) {
        hipLaunchKernelGGL((k_conv2d_nhwc<T, 128>), dim3(div_up(p.m, 128), p.cout / 128), block, 0, st, (const T *)x,
                           (const T *)wpk, bias, (T *)y, p);
    } else {
        hipLaunchKernelGGL((k_conv2d_nhwc<T, 64>), dim3(div_up(p.m, 128), p.cout / 64), block, 0, st, (const T *)x,
                           (const T *)wpk, bias, (T *)y, p);
    }
#endif
    return check_launch();
}

// ---- which 8 x 16 tiles of the RPN's feature maps can differ from the empty frame's -------------------------------------------
// The BEV image the RPN starts from is zero except at the sparse middle's sites, and a pixel of conv j's output (j = 0, 1, ...)
// sees the image only within j + 1 steps (Chebyshev): a tile farther than that from every site holds exactly what the same
// network computes for an EMPTY frame at that position -- for any weights.  Per tile: d = distance from the tile's rectangle to
// the nearest site (from a bitmap of the frame in LDS); the tile is live for conv j when d <= j + 1.  Per conv and frame the tile
// indices are written LIVE first (ascending), the others from the END backwards, with the live count: the conv kernel spreads the
// live list evenly over the XCDs whatever the frames' occupancies are.
constexpr int kTileLiveThreads = 1024;
// occupancy bitmap of the BEV image: bits[b][y][k] bit j = site_map[b][0 or 1][y][32 k + j] != 0.  A thread per word, spread over the
// chip (one workgroup per frame reading its 280 KB of map took 8 us: one CU's L2 -> L1 path)
__global__ __launch_bounds__(kBlock) void k_bev_bitmap(const int *__restrict__ site_map, int h, int w, int batch, unsigned *__restrict__ bits) {
    const int wr = (w + 31) >> 5, words = h * wr;
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= batch * words) return;
    const int b = g / words, i = g - b * words;
    const int yy = i / wr, k = i - yy * wr;
    const long long plane = (long long)h * w;
    const int *m0 = site_map + (long long)b * 2 * plane + (long long)yy * w, *m1 = m0 + plane;
    unsigned out = 0u;
    if ((w & 3) == 0) {           // rows are 16-byte aligned: eight independent 16-byte loads per plane
        const int4 *r0 = reinterpret_cast<const int4 *>(m0), *r1 = reinterpret_cast<const int4 *>(m1);
        int4 va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = 32 * k + 4 * j;
            va[j] = ld_sel(r0, px >> 2, px < w, make_int4(0, 0, 0, 0));
            vb[j] = ld_sel(r1, px >> 2, px < w, make_int4(0, 0, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            out |= ((unsigned)((va[j].x | vb[j].x) != 0) | (unsigned)((va[j].y | vb[j].y) != 0) << 1 |
                    (unsigned)((va[j].z | vb[j].z) != 0) << 2 | (unsigned)((va[j].w | vb[j].w) != 0) << 3) << (4 * j);
    } else {
        for (int j = 0; j < 32; ++j) {
            const int px = 32 * k + j;
            if (px < w && (m0[px] | m1[px])) out |= 1u << j;
        }
    }
    bits[g] = out;
}

constexpr int kTileLiveMaxLayers = 8;
// `masks` (optional; sec_rpn_tile_live_masks): for the LAZY consumers of a layer's output (sec_conv2d_nhwc_tiles_lazy) -- per conv
// l >= 1 and tile, nine bits over the tile's 3 x 3 neighbourhood (bit (dy + 1) * 3 + dx + 1): the neighbour was LIVE for conv l - 1
// (d <= l), i.e. conv l - 1 really wrote it; a neighbour outside the image counts as written (its pixels are zero padding either
// way).  Two copies per layer, [l][0][b][rank] in the order of the live list and [l][1][b][tile] by tile index (what a conv that
// falls back to the plain tile order reads).
__global__ __launch_bounds__(kTileLiveThreads) void k_rpn_tile_live(const unsigned *__restrict__ bits, int h, int w, int layers, int batch,
                                                                     unsigned short *__restrict__ order, int *__restrict__ counts,
                                                                     unsigned short *__restrict__ masks) {
    extern __shared__ unsigned tl_bits[];
    constexpr int NT = kTileLiveThreads, NW = NT / 64;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (w + 31) >> 5, words = h * wr;
    const int ty = (h + 7) / 8, tx = (w + 15) / 16, tiles = ty * tx;
    unsigned char *s_d = reinterpret_cast<unsigned char *>(tl_bits + words);     // distance of every tile of the frame (phase 1)
    __shared__ int s_wave[kTileLiveMaxLayers][NW], s_run[kTileLiveMaxLayers];
    for (int i = tid; i < words; i += NT) tl_bits[i] = bits[(long long)b * words + i];
    if (tid < kTileLiveMaxLayers) s_run[tid] = 0;
    __syncthreads();
    const int K = layers;                         // farthest distance that matters
    // ---- phase 1: d of every tile
    for (int t = tid; t < tiles; t += NT) {
        int d = K + 1;
        const int tyi = t / tx, txi = t - tyi * tx;
        const int y0 = tyi * 8, x0 = txi * 16;
        const int ylo = y0 - K > 0 ? y0 - K : 0, yhi = y0 + 7 + K < h - 1 ? y0 + 7 + K : h - 1;
        // window of columns [x0 - 32, x0 + 47] as 80 bits around the tile's word: lo = bits of [x0 - 32, x0 + 31], hi = [x0 + 32, x0 + 47]
        const int k0 = x0 >> 5, sh = x0 & 31;  // x0 is a multiple of 16: sh is 0 or 16
        for (int yy = ylo; yy <= yhi; ++yy) {
            const unsigned *row = tl_bits + yy * wr;
            const unsigned wm = k0 > 0 ? row[k0 - 1] : 0u, wc = row[k0], wp = k0 + 1 < wr ? row[k0 + 1] : 0u;
            // 96 bits [32 (k0 - 1), 32 (k0 + 2)); the tile's columns are bits [32 + sh, 32 + sh + 15] of it
            const unsigned long long lo = (unsigned long long)wm | (unsigned long long)wc << 32;      // bits 0..63
            const unsigned long long win = sh ? (lo >> 16) | (unsigned long long)wp << 48 : lo;      // tile columns at bits [32, 47] either way
            const unsigned long long hi16 = sh ? (unsigned long long)(wp >> 16) : (unsigned long long)(wp & 0xffffu); // columns x0 + 32 .. x0 + 47 (sh = 16: bits 16.. of wp)
            int dx = K + 1;
            if ((win >> 32) & 0xffffull) dx = 0;
            else {
                const unsigned long long left = win & 0xffffffffull;              // columns x0 - 32 .. x0 - 1 at bits 0 .. 31
                if (left) dx = 32 - (63 - __clzll((long long)left));            // nearest set bit below the tile: distance x0 - column
                const unsigned long long right = (win >> 48) | hi16 << 16;         // columns x0 + 16 .. at bits 0 ..
                if (right) {
                    const int dr = __ffsll((long long)right);                   // 1-based: distance from column x0 + 15
                    dx = dr < dx ? dr : dx;
                }
            }
            const int dy = yy < y0 ? y0 - yy : (yy > y0 + 7 ? yy - (y0 + 7) : 0);
            const int dd = dx > dy ? dx : dy;
            d = dd < d ? dd : d;
        }
        s_d[t] = (unsigned char)d;
    }
    __syncthreads();
    // ---- phase 2: lists (and masks) of all layers
    for (int base = 0; base < tiles; base += NT) {
        const int t = base + tid;
        const int d = t < tiles ? (int)s_d[t] : K + 1;
        // compaction of all layers with one barrier pair: ballots, per-wave totals, prefix
        unsigned long long bal[kTileLiveMaxLayers];
#pragma unroll
        for (int l = 0; l < kTileLiveMaxLayers; ++l) {
            bal[l] = l < layers ? __ballot(t < tiles && d <= l + 1) : 0ull;
            if (lane == 0) s_wave[l][wv] = __popcll(bal[l]);
        }
        __syncthreads();
        int nd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (masks && t < tiles) {
            const int tyi = t / tx, txi = t - tyi * tx;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int yy = tyi + q / 3 - 1, xx = txi + q % 3 - 1;
                nd[q] = (yy >= 0 && yy < ty && xx >= 0 && xx < tx) ? (int)s_d[yy * tx + xx] : 0;
            }
        }
#pragma unroll
        for (int l = 0; l < kTileLiveMaxLayers; ++l) {
            if (l >= layers) break;
            int before = s_run[l];
            for (int i2 = 0; i2 < wv; ++i2) before += s_wave[l][i2];
            const int rank = before + __popcll(bal[l] & ((1ull << lane) - 1ull));        // live tiles of this layer before this one
            if (t < tiles) {
                unsigned short *ord = order + ((long long)l * batch + b) * tiles;
                if (d <= l + 1) ord[rank] = (unsigned short)t;
                else ord[tiles - 1 - (t - rank)] = (unsigned short)t;                       // the others: from the end backwards
                if (masks && l >= 1) {
                    unsigned m = 0u;
#pragma unroll
                    for (int q = 0; q < 9; ++q) m |= (unsigned)(nd[q] <= l) << q;
                    unsigned short *mk = masks + (((long long)l * 2) * batch + b) * tiles;
                    if (d <= l + 1) mk[rank] = (unsigned short)m;
                    mk[(long long)batch * tiles + t] = (unsigned short)m;
                }
            }
        }
        __syncthreads();
        if (tid < layers) {
            int tot = 0;
            for (int i2 = 0; i2 < NW; ++i2) tot += s_wave[tid][i2];
            s_run[tid] += tot;
        }
        __syncthreads();
    }
    if (tid < layers) counts[tid * batch + b] = s_run[tid];
}

}  // namespace sec

using namespace sec;

SEC_API size_t sec_rpn_tile_live_workspace_bytes(int batch, int h, int w) {
    if (batch <= 0 || h <= 0 || w <= 0) return 0;
    return align_up((size_t)batch * h * ((w + 31) / 32) * 4);
}

static int rpn_tile_live_impl(const int *site_map, int batch, int h, int w, int layers, unsigned short *tile_order, int *live_counts,
                              unsigned short *nbr_masks, void *workspace, size_t workspace_bytes, void *stream) {
    if (!site_map || !tile_order || !live_counts || !workspace || batch <= 0 || h <= 0 || w <= 0 || layers <= 0) return SEC_E_INVALID;
    if (workspace_bytes < sec_rpn_tile_live_workspace_bytes(batch, h, w)) return SEC_E_WORKSPACE;
    const int words = h * ((w + 31) / 32);
    const long long tiles = (long long)((h + 7) / 8) * ((w + 15) / 16);
    if (tiles > 65535 || layers > kTileLiveMaxLayers) return SEC_E_UNSUPPORTED;
    const size_t lds = (size_t)words * 4 + (((size_t)tiles + 15) & ~(size_t)15);     // the frame's bitmap + one distance byte per tile
    if (lds > 120 * 1024) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rpn_tile_live), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        configured = true;
    }
    hipLaunchKernelGGL(k_bev_bitmap, dim3(div_up((long long)batch * words, kBlock)), dim3(kBlock), 0, st, site_map, h, w, batch, (unsigned *)workspace);
    hipLaunchKernelGGL(k_rpn_tile_live, dim3(batch), dim3(kTileLiveThreads), lds, st, (const unsigned *)workspace, h, w, layers, batch, tile_order,
                       live_counts, nbr_masks);
    return check_launch();
}

SEC_API int sec_rpn_tile_live(const int *site_map, int batch, int h, int w, int layers, unsigned short *tile_order, int *live_counts,
                              void *workspace, size_t workspace_bytes, void *stream) {
    return rpn_tile_live_impl(site_map, batch, h, w, layers, tile_order, live_counts, nullptr, workspace, workspace_bytes, stream);
}

SEC_API int sec_rpn_tile_live_masks(const int *site_map, int batch, int h, int w, int layers, unsigned short *tile_order, int *live_counts,
                                    unsigned short *nbr_masks, void *workspace, size_t workspace_bytes, void *stream) {
    if (!nbr_masks) return SEC_E_INVALID;
    return rpn_tile_live_impl(site_map, batch, h, w, layers, tile_order, live_counts, nbr_masks, workspace, workspace_bytes, stream);
}

static int conv2d_tiles_impl(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int cout, int relu,
                             const unsigned short *tile_order, const int *live_counts, const void *background,
                             const unsigned short *nbr_masks, const void *background_in, void *y, int dtype, void *stream) {
    if (!x || !packed_weight || !y || batch <= 0 || h <= 0 || w <= 0 || (tile_order && !live_counts)) return SEC_E_INVALID;
    if (cout % 128 || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    apply_list_threshold_env();
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = 128; p.cout = cout; p.ksize = 3; p.stride = 1; p.pad = 1;
    p.relu = relu & 1; p.zskip = 0; p.stagger = 0;
    p.ho = h; p.wo = w;
    p.m = (long long)batch * h * w;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16)
        return launch_conv2d_halo_reg<__hip_bfloat16, 128, 8, 2>(x, packed_weight, bias, y, p, st, nullptr, 0, tile_order, live_counts, background,
                                                                 nbr_masks, background_in);
    return launch_conv2d_halo_reg<__half, 128, 8, 2>(x, packed_weight, bias, y, p, st, nullptr, 0, tile_order, live_counts, background, nbr_masks,
                                                     background_in);
}

SEC_API int sec_conv2d_nhwc_tiles(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int cout,
                                  int relu, const unsigned short *tile_order, const int *live_counts, const void *background,
                                  void *y, int dtype, void *stream) {
    if (tile_order && !background) return SEC_E_INVALID;
    return conv2d_tiles_impl(x, batch, h, w, packed_weight, bias, cout, relu, tile_order, live_counts, background, nullptr, nullptr, y, dtype,
                             stream);
}

SEC_API int sec_conv2d_nhwc_tiles_lazy(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int cout,
                                       int relu, const unsigned short *tile_order, const int *live_counts, const void *background,
                                       const unsigned short *nbr_masks, const void *background_in, void *y, int dtype, void *stream) {
    if (!tile_order || !nbr_masks || !background_in) return SEC_E_INVALID;
    return conv2d_tiles_impl(x, batch, h, w, packed_weight, bias, cout, relu, tile_order, live_counts, background, nbr_masks, background_in, y,
                             dtype, stream);
}

// The last 3x3 conv of a single-block RPN with its 1x1 tail (deblock + merged heads) in the epilogue: see k_conv2d_halo_reg<..., TAIL>.
SEC_API int sec_conv2d_nhwc_tiles_tail(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int relu,
                                       const unsigned short *tile_order, const int *live_counts, const unsigned short *nbr_masks,
                                       const void *background_in, const void *packed_w1, const float *bias1, int relu1, const void *packed_w2,
                                       const float *bias2, int cout2, void *y_heads, int dtype, void *stream) {
    if (!x || !packed_weight || !tile_order || !live_counts || !nbr_masks || !background_in || !packed_w1 || !bias1 || !packed_w2 || !y_heads ||
        batch <= 0 || h <= 0 || w <= 0)
        return SEC_E_INVALID;
    if (cout2 != 64 || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    apply_list_threshold_env();
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = 128; p.cout = 128; p.ksize = 3; p.stride = 1; p.pad = 1;
    p.relu = relu & 1; p.zskip = 0; p.stagger = 0;
    p.ho = h; p.wo = w;
    p.m = (long long)batch * h * w;
    const ConvTailArgs tail{packed_w1, packed_w2, bias1, bias2, y_heads, relu1 & 1};
    hipStream_t st = (hipStream_t)stream;
    // y of the conv itself does not exist: the kernel's `y` is never dereferenced on the TAIL path (background == NULL: lazy form)
    if (dtype == SEC_BF16)
        return launch_conv2d_halo_reg<__hip_bfloat16, 128, 8, 2, false, 1, false, true>(x, packed_weight, bias, nullptr, p, st, nullptr, 0, tile_order, live_counts,
                                                                                       nullptr, nbr_masks, background_in, nullptr, nullptr, nullptr, nullptr, tail);
    return launch_conv2d_halo_reg<__half, 128, 8, 2, false, 1, false, true>(x, packed_weight, bias, nullptr, p, st, nullptr, 0, tile_order, live_counts, nullptr,
                                                                           nbr_masks, background_in, nullptr, nullptr, nullptr, nullptr, tail);
}

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
    p.batch = batch; p.h = h; p.w = w; p.cin = cin; p.cout = cout; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.relu = relu & 1;
    p.zskip = (relu >> 1) & 1;
    p.stagger = 0;
    p.ho = (h + 2 * pad - ksize) / stride + 1;
    p.wo = (w + 2 * pad - ksize) / stride + 1;
    if (p.ho <= 0 || p.wo <= 0) return SEC_E_INVALID;
    p.m = (long long)batch * p.ho * p.wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16) return launch_conv2d<__hip_bfloat16>(x, packed_weight, bias, y, p, st);
    return launch_conv2d<__half>(x, packed_weight, bias, y, p, st);
}

SEC_API int sec_conv2d_nhwc_into(const void *x, int batch, int h, int w, int cin, const void *packed_weight, const float *bias, int cout,
                                 int ksize, int stride, int pad, int relu, void *y, int y_channels, int y_channel_offset, int dtype, void *stream) {
    if (!x || !packed_weight || !y || batch <= 0 || h <= 0 || w <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || y_channel_offset < 0 ||
        y_channel_offset + cout > y_channels)
        return SEC_E_INVALID;
    if (cin % 64 || cout % 64 || (y_channel_offset % 8) || (y_channels % 8) || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = cin; p.cout = cout; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.relu = relu & 1;
    p.zskip = 0;
    p.stagger = 0;
    p.ho = (h + 2 * pad - ksize) / stride + 1;
    p.wo = (w + 2 * pad - ksize) / stride + 1;
    if (p.ho <= 0 || p.wo <= 0) return SEC_E_INVALID;
    p.m = (long long)batch * p.ho * p.wo;
    hipStream_t st = (hipStream_t)stream;
    // only k_conv2d_patch writes with a channel pitch: the shapes of patch::dispatch (the deblocks of the multi-block RPNs among them)
    const int rc = dtype == SEC_BF16
                       ? patch::dispatch<__hip_bfloat16>(x, packed_weight, bias, (__hip_bfloat16 *)y + y_channel_offset, p, y_channels, st)
                       : patch::dispatch<__half>(x, packed_weight, bias, (__half *)y + y_channel_offset, p, y_channels, st);
    return rc == patch::kNotTaken ? SEC_E_UNSUPPORTED : rc;
}

SEC_API int sec_conv2d_nhwc_rows(const void *rows, long long feature_rows, const int *site_map, int batch, int h, int w, int cin,
                                 const void *packed_weight, const float *bias, int cout, int ksize, int stride, int pad, int relu, void *y,
                                 int dtype, void *stream) {
    if (!rows || !site_map || !packed_weight || !y || batch <= 0 || h <= 0 || w <= 0 || feature_rows < 0 || ksize <= 0 || stride <= 0 || pad < 0)
        return SEC_E_INVALID;
    if (dtype != SEC_BF16 && dtype != SEC_F16) return SEC_E_UNSUPPORTED;
    if (feature_rows * cin * 2 > 0x7fffffffll || (long long)h * w * 4 > 0x7fffffffll) return SEC_E_UNSUPPORTED;     // 32-bit buffer offsets
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = cin; p.cout = cout; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.relu = relu & 1;
    p.zskip = 0;
    p.stagger = 0;
    p.ho = (h + 2 * pad - ksize) / stride + 1;
    p.wo = (w + 2 * pad - ksize) / stride + 1;
    if (p.ho <= 0 || p.wo <= 0) return SEC_E_INVALID;
    p.m = (long long)batch * p.ho * p.wo;
    hipStream_t st = (hipStream_t)stream;
    const unsigned fb = (unsigned)(feature_rows * cin * 2);
    const int rc = dtype == SEC_BF16 ? patch::dispatch_rows<__hip_bfloat16>(rows, fb, site_map, packed_weight, bias, y, p, st)
                                     : patch::dispatch_rows<__half>(rows, fb, site_map, packed_weight, bias, y, p, st);
    return rc == patch::kNotTaken ? SEC_E_UNSUPPORTED : rc;
}

// fp32 <-> two bf16 planes (hi = bf16(v), lo = bf16(v - hi)): the operand form of sec_conv2d_nhwc_x3.  Element-wise, HBM bound.
__global__ __launch_bounds__(kBlock) void k_split_bf16x2(const float4 *__restrict__ x, long long n4, uint2 *__restrict__ hi, uint2 *__restrict__ lo) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const float4 v = x[i];
        const unsigned h0 = pack2<__hip_bfloat16>(v.x, v.y), h1 = pack2<__hip_bfloat16>(v.z, v.w);
        hi[i] = make_uint2(h0, h1);
        lo[i] = make_uint2(pack2<__hip_bfloat16>(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u)),
                           pack2<__hip_bfloat16>(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u)));
    }
}

__global__ __launch_bounds__(kBlock) void k_merge_bf16x2(const uint2 *__restrict__ hi, const uint2 *__restrict__ lo, long long n4, float4 *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const uint2 h = hi[i], l = lo[i];
        y[i] = make_float4(__uint_as_float(h.x << 16) + __uint_as_float(l.x << 16), __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u),
                           __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16), __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u));
    }
}

SEC_API int sec_split_f32_bf16x2(const float *x, long long n, void *hi, void *lo, void *stream) {
    if (n < 0 || n % 4 || (n > 0 && (!x || !hi || !lo))) return SEC_E_INVALID;
    if (n == 0) return SEC_OK;
    long long blocks = div_up(n / 4, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_split_bf16x2, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)x, n / 4, (uint2 *)hi, (uint2 *)lo);
    return check_launch();
}

SEC_API int sec_merge_bf16x2_f32(const void *hi, const void *lo, long long n, float *y, void *stream) {
    if (n < 0 || n % 4 || (n > 0 && (!y || !hi || !lo))) return SEC_E_INVALID;
    if (n == 0) return SEC_OK;
    long long blocks = div_up(n / 4, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_merge_bf16x2, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (const uint2 *)hi, (const uint2 *)lo, n / 4, (float4 *)y);
    return check_launch();
}

SEC_API int sec_conv2d_nhwc_x3(const void *x_hi, const void *x_lo, int batch, int h, int w, const void *packed_weight_hi_lo, const float *bias,
                               int cout, int relu, void *y_hi, void *y_lo, void *stream) {
    if (!x_hi || !x_lo || !packed_weight_hi_lo || !y_hi || !y_lo || batch <= 0 || h <= 0 || w <= 0) return SEC_E_INVALID;
    if (cout % 128 || (long long)h * w * 256 >= (1ll << 31)) return SEC_E_UNSUPPORTED;
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = 128; p.cout = cout; p.ksize = 3; p.stride = 1; p.pad = 1;
    p.relu = relu & 1; p.zskip = (relu >> 1) & 1; p.stagger = 0;
    p.ho = h; p.wo = w;
    p.m = (long long)batch * h * w;
    return launch_conv2d_halo_reg<__hip_bfloat16, 128, 8, 2, false, 1, true>(x_hi, packed_weight_hi_lo, bias, y_hi, p, (hipStream_t)stream, nullptr, 0,
                                                                              nullptr, nullptr, nullptr, nullptr, nullptr, x_lo, y_lo);
}

SEC_API int sec_conv2d_nhwc_x3_tiles(const void *x_hi, const void *x_lo, int batch, int h, int w, const void *packed_weight_hi_lo,
                                     const float *bias, int cout, int relu, const unsigned short *tile_order, const int *live_counts,
                                     const void *background_hi, const void *background_lo, const unsigned short *nbr_masks,
                                     const void *background_in_hi, const void *background_in_lo, void *y_hi, void *y_lo, void *stream) {
    if (!x_hi || !x_lo || !packed_weight_hi_lo || !y_hi || !y_lo || !tile_order || !live_counts || batch <= 0 || h <= 0 || w <= 0) return SEC_E_INVALID;
    if ((background_hi == nullptr) != (background_lo == nullptr)) return SEC_E_INVALID;
    if (nbr_masks && (!background_in_hi || !background_in_lo)) return SEC_E_INVALID;
    if (cout % 128 || (long long)h * w * 256 >= (1ll << 31)) return SEC_E_UNSUPPORTED;
    apply_list_threshold_env();
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = 128; p.cout = cout; p.ksize = 3; p.stride = 1; p.pad = 1;
    p.relu = relu & 1; p.zskip = 0; p.stagger = 0;
    p.ho = h; p.wo = w;
    p.m = (long long)batch * h * w;
    return launch_conv2d_halo_reg<__hip_bfloat16, 128, 8, 2, false, 1, true>(x_hi, packed_weight_hi_lo, bias, y_hi, p, (hipStream_t)stream, nullptr, 0,
                                                                              tile_order, live_counts, background_hi, nbr_masks, background_in_hi,
                                                                              x_lo, y_lo, background_lo, background_in_lo);
}

SEC_API int sec_conv2d_nhwc_gather(const void *features, long long feature_rows, const int *site_map, int batch, int h, int w,
                                   const void *packed_weight, const float *bias, int cout, int relu, const unsigned short *tile_order,
                                   const int *live_counts, const void *background, void *y, int dtype, void *stream) {
    if (!site_map || !packed_weight || !y || batch <= 0 || h <= 0 || w <= 0 || feature_rows < 0 || (!features && feature_rows > 0)) return SEC_E_INVALID;
    if (tile_order && !live_counts) return SEC_E_INVALID;       // background == NULL with lists: the other tiles are left unwritten (lazy consumers)
    if (cout % 128 || (dtype != SEC_BF16 && dtype != SEC_F16) || feature_rows * 128 >= (1ll << 31) ||
        (long long)h * w * 8 >= (1ll << 31)) return SEC_E_UNSUPPORTED;
    apply_list_threshold_env();
    Conv2dParams p;
    p.batch = batch; p.h = h; p.w = w; p.cin = 128; p.cout = cout; p.ksize = 3; p.stride = 1; p.pad = 1;
    p.relu = relu & 1; p.zskip = 0; p.stagger = 0;
    p.ho = h; p.wo = w;
    p.m = (long long)batch * h * w;
    hipStream_t st = (hipStream_t)stream;
    const unsigned fb = (unsigned)(feature_rows * 128);
    if (dtype == SEC_BF16)
        return launch_conv2d_halo_reg<__hip_bfloat16, 128, 8, 2, true>(features, packed_weight, bias, y, p, st, site_map, fb, tile_order, live_counts, background);
    return launch_conv2d_halo_reg<__half, 128, 8, 2, true>(features, packed_weight, bias, y, p, st, site_map, fb, tile_order, live_counts, background);
}

SEC_API int sec_conv1x1_chain_nhwc(const void *x, long long pixels, const void *packed_w1, const float *bias1, int relu1,
                                   const void *packed_w2, const float *bias2, int cout2, void *y, int dtype, void *stream) {
    if (!x || !packed_w1 || !packed_w2 || !bias1 || !y || pixels < 0) return SEC_E_INVALID;
    if ((cout2 != 64 && cout2 != 128) || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    if (pixels == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16)
        return launch_conv1x1_chain<__hip_bfloat16>(x, pixels, packed_w1, bias1, packed_w2, bias2, cout2, relu1, y, st);
    return launch_conv1x1_chain<__half>(x, pixels, packed_w1, bias1, packed_w2, bias2, cout2, relu1, y, st);
}

SEC_API int sec_conv1x1_chain_x3(const void *x_hi, const void *x_lo, long long pixels, const void *packed_w1_hi_lo, const float *bias1, int relu1,
                                 const void *packed_w2_hi_lo, const float *bias2, int cout2, float *y, void *stream) {
    if (!x_hi || !x_lo || !packed_w1_hi_lo || !packed_w2_hi_lo || !bias1 || !y || pixels < 0) return SEC_E_INVALID;
    if (cout2 != 64 && cout2 != 128) return SEC_E_UNSUPPORTED;
    if (pixels == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)div_up(pixels, 128);
    if (cout2 == 64)
        hipLaunchKernelGGL((k_conv1x1_chain_x3<1>), dim3(blocks), dim3(256), 0, st, (const __hip_bfloat16 *)x_hi, (const __hip_bfloat16 *)x_lo,
                           (const __hip_bfloat16 *)packed_w1_hi_lo, bias1, (const __hip_bfloat16 *)packed_w2_hi_lo, bias2, y, pixels, relu1);
    else
        hipLaunchKernelGGL((k_conv1x1_chain_x3<2>), dim3(blocks), dim3(256), 0, st, (const __hip_bfloat16 *)x_hi, (const __hip_bfloat16 *)x_lo,
                           (const __hip_bfloat16 *)packed_w1_hi_lo, bias1, (const __hip_bfloat16 *)packed_w2_hi_lo, bias2, y, pixels, relu1);
    return check_launch();
}

SEC_API int sec_conv1x1_chain_nhwc_tiles(const void *x, int batch, int h, int w, const void *packed_w1, const float *bias1, int relu1,
                                         const void *packed_w2, const float *bias2, int cout2, const unsigned short *tile_order,
                                         const int *live_counts, const void *background, void *y, int dtype, void *stream) {
    // background == NULL: only with SEC_CHAIN_X_LIVE_ONLY (relu1 bit 1: the lists are used whatever the live share) -- the tiles
    // outside the list are then NOT written, for consumers that read them from the empty frame's map (sec_predict_select_lazy)
    if (!x || !packed_w1 || !packed_w2 || !bias1 || !y || !tile_order || !live_counts || (!background && !(relu1 & 2)) || batch <= 0 || h <= 0 || w <= 0)
        return SEC_E_INVALID;
    if ((cout2 != 64 && cout2 != 128) || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16)
        return launch_conv1x1_chain_tiles<__hip_bfloat16>(x, batch, h, w, packed_w1, bias1, packed_w2, bias2, cout2, relu1, tile_order, live_counts,
                                                          background, y, st);
    return launch_conv1x1_chain_tiles<__half>(x, batch, h, w, packed_w1, bias1, packed_w2, bias2, cout2, relu1, tile_order, live_counts, background, y, st);
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
