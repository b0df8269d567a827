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
#include <stdlib.h>
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
// work is split over (pixel range, tap): a workgroup = one tap of a run of `steps` 64-pixel steps.  Both MFMA operands need 8
// consecutive PIXELS per lane for one channel -- the transpose of channels-last memory.  The first form of this kernel transposed in
// registers: a thread loaded 8 channels of 4 consecutive pixels, so one wave instruction gathered sixteen 64-byte half lines; the
// texture path spent ~50 cycles on each (TA busy 63 % of the launch, MFMA busy 0.19, 88 us -- profiles/r05_h_wgrad_pmc.txt), and
// neither a deeper prefetch nor fewer LDS reads moved it.  This form loads FULL lines (sixteen lanes = the 256 bytes of one pixel),
// stores them untransposed ([pixel][channel], ds_write_b128, the 16-byte chunk index XOR-ed with 4 * (pixel & 3)) and lets the LDS
// transpose on the way out: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-pixel x 16-channel block, i.e. four
// pixels of ONE channel -- two of them are an MFMA operand.  (The order of the eight pixels inside an operand does not matter: A and B
// are read the same way.)  With the XOR the four pixel rows of a 32-lane read group fall into four different 64-byte bank windows.
// Three register stages of loads in flight (buffer loads, out-of-range offset = zero fill for the padding and the tail: a
// conditional global load compiles to a branch with s_waitcnt vmcnt(0) behind it), LDS double-buffered, one barrier per step.  A wave
// owns a 64 (ci) x 64 (co) quadrant.  Every workgroup writes its 128 x 128 fp32 partial to the workspace; k_conv2d_wgrad_reduce sums
// the partials of a tap in a fixed order (deterministic -- no float atomics) into torch's [Cout][Cin][3][3] layout.
typedef short ts16x4 __attribute__((ext_vector_type(4)));
typedef unsigned tu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16_b64(const char *p) {
    const ts16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ts16x4 __attribute__((address_space(3))) *)p);
    return __builtin_bit_cast(uint2, v);
}
template <typename T, bool NARROW>   // NARROW: maps less than 16 pixels wide (the coordinate advance of 16 pixels may wrap more than one row)
__global__ __launch_bounds__(256, 2) void k_conv2d_wgrad3x3(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part,
                                                            int B, int H, int W, int steps, long long P, int slices, int ntaps, int dyc) {
    constexpr int C = 128, PIXB = C * 2;                   // 256 bytes per pixel
    __shared__ __attribute__((aligned(16))) char sX[2][64 * PIXB];
    __shared__ __attribute__((aligned(16))) char sD[2][64 * PIXB];
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
    const unsigned tensor_bytes = (unsigned)(P * PIXB);
    // dy may hold fewer channels than the 128 x 128 tile (`dyc` = 64: the stacked 1x1 heads): its pixels are dyc * 2 bytes apart and
    // the chunks behind them read as zeros (out-of-range offset), so rows co >= dyc of the partial are zero and never leave the reduce
    const unsigned dpix = (unsigned)dyc * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x), 0, (int)tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dy), 0, (int)(P * dpix), 0x00020000);
    // staging: this thread moves chunk `cq` (16 bytes = 8 channels) of the pixels pl, pl + 16, pl + 32, pl + 48 of a step
    const int cq = tid & 15, pl = tid >> 4;
    const bool dchunk = cq * 8 < dyc;
    // running coordinates of the NEXT step to fetch (pixel pl of it); they only ever advance by 16 pixels -- no divisions in the loop
    unsigned fq;
    int fx, fy, fb;
    {
        fq = (unsigned)(step0 * 64 + pl);
        const unsigned HW = (unsigned)H * (unsigned)W;
        fb = (int)(fq / HW);
        const unsigned rem = fq - (unsigned)fb * HW;
        fy = (int)(rem / (unsigned)W);
        fx = (int)(rem - (unsigned)fy * (unsigned)W);
    }
    int fs = 0;                                             // index of the next step to fetch
    tu32x4 ra_x[4], ra_d[4], rb_x[4], rb_d[4], rc_x[4], rc_d[4];
    auto fetch = [&](tu32x4 (&rx)[4], tu32x4 (&rd)[4]) {
        const bool oks = fs < nst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool okp = oks && (long long)fq < P;
            rd[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, okp && dchunk ? fq * dpix + cq * 16u : 0xfffffff0u, 0, 0);
            const int sy = fy + ty, sx = fx + tx;
            const bool okx = okp && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            const unsigned xo = (unsigned)((fb * H + sy) * W + sx) * (unsigned)PIXB + cq * 16u;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, okx ? xo : 0xfffffff0u, 0, 0);
            fq += 16;
            fx += 16;
            if (NARROW) {
                while (fx >= W) { fx -= W; if (++fy == H) { fy = 0; ++fb; } }
            } else {                                        // selects, no branch: the scheduler may then slide this arithmetic under the MFMAs
                const bool wrap = fx >= W;
                fx = wrap ? fx - W : fx;
                fy = wrap ? fy + 1 : fy;
                const bool wrapy = fy == H;
                fy = wrapy ? 0 : fy;
                fb = wrapy ? fb + 1 : fb;
            }
        }
        ++fs;
    };
    // LDS image of a step: [pixel][16 chunks], chunk index ^ 4 * (pixel & 3); pixel = pl + 16 i keeps pixel & 3 = pl & 3
    const int st_off = pl * PIXB + ((cq ^ ((pl & 3) << 2)) << 4);
    auto put = [&](int buf, const tu32x4 (&rx)[4], const tu32x4 (&rd)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<tu32x4 *>(&sX[buf][st_off + i * 16 * PIXB]) = rx[i];
            *reinterpret_cast<tu32x4 *>(&sD[buf][st_off + i * 16 * PIXB]) = rd[i];
        }
    };
    // operand reads: 16-lane group g = lane / 16, i = lane % 16: pixels ks * 16 + (g / 2) * 8 + t * 4 + i / 4 (t = 0, 1: two reads),
    // channels tile * 32 + (g & 1) * 16 + (i & 3) * 4 .. + 3 (8 bytes); the lane receives channel tile * 32 + (g & 1) * 16 + i = tile * 32 + r
    const int g = lane >> 4, li = lane & 15;
    const int cih = (wv >> 1) * 64, coh = (wv & 1) * 64;
    int ra_off[2], rb_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int prow = (g >> 1) * 8 + (li >> 2);
        const int sw = (li >> 2) << 2;                      // = 4 * (pixel & 3)
        const int ca = (cih / 8 + t * 4 + (g & 1) * 2 + ((li & 3) >> 1)) ^ sw;
        const int cb = (coh / 8 + t * 4 + (g & 1) * 2 + ((li & 3) >> 1)) ^ sw;
        ra_off[t] = prow * PIXB + (ca << 4) + (li & 1) * 8;
        rb_off[t] = prow * PIXB + (cb << 4) + (li & 1) * 8;
    }
    tf32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t >> 1][t & 1][i] = 0.0f;
    auto frag = [&](const char *base, int off) {
        const uint2 lo = lds_tr16_b64(base + off), hi = lds_tr16_b64(base + off + 4 * PIXB);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto mfmas = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char *bx = &sX[buf][ks * 16 * PIXB], *bd = &sD[buf][ks * 16 * PIXB];
            const uint4 a0 = frag(bx, ra_off[0]), a1 = frag(bx, ra_off[1]);
            const uint4 b0 = frag(bd, rb_off[0]), b1 = frag(bd, rb_off[1]);
            acc[0][0] = MfmaT<T>::run(a0, b0, acc[0][0]);
            acc[0][1] = MfmaT<T>::run(a0, b1, acc[0][1]);
            acc[1][0] = MfmaT<T>::run(a1, b0, acc[1][0]);
            acc[1][1] = MfmaT<T>::run(a1, b1, acc[1][1]);
        }
    };
    if (nst > 0) {
        // Straight-line steps: every step issues its eight loads (a step past the end loads zeros through the out-of-range offset and
        // adds zeros), so the compiler can count the loads in flight -- `s_waitcnt vmcnt(16)` before the LDS stores of a stage; with
        // `if (s + 3 < nst) fetch(...)` in the loop it fell back to vmcnt(0) and the prefetch bought nothing.
        fetch(ra_x, ra_d);
        fetch(rb_x, rb_d);
        fetch(rc_x, rc_d);
        put(0, ra_x, ra_d);
        __syncthreads();
        // step s: LDS buffer s & 1 holds it, the registers hold s + 1 and s + 2; the stage that held s is free for s + 3
#define SEC_WGRAD_STEP(S, FX, FD, PX, PD)                                                              \
        mfmas((S) & 1);                                        /* the MFMAs first: the address arithmetic of the fetch runs under them */ \
        fetch(FX, FD);                                                                                 \
        put(((S) + 1) & 1, PX, PD);                           /* that buffer was last read before the previous barrier */ \
        __syncthreads();
        for (int s = 0; s < nst; s += 3) {
            SEC_WGRAD_STEP(s, ra_x, ra_d, rb_x, rb_d)
            SEC_WGRAD_STEP(s + 1, rb_x, rb_d, rc_x, rc_d)
            SEC_WGRAD_STEP(s + 2, rc_x, rc_d, ra_x, ra_d)
        }
#undef SEC_WGRAD_STEP
    }
    // D layout: column (co) = lane & 31, rows (ci) = (i & 3) + 8 (i >> 2) + 4 h
    float *dst = part + ((size_t)slice * ntaps + tslot) * C * C;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = cih + (t >> 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, co = coh + (t & 1) * 32 + r;
            dst[(size_t)ci * C + co] = acc[t >> 1][t & 1][i];
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The same weight gradient with a whole KERNEL ROW per workgroup (round 6; 3x3, maps at least 16 pixels wide).  In the form above a
// workgroup is one tap: every step moves 32 KB (64 pixels of x and of dy) through the L2 -> L1 path for 64 MFMAs -- 64 FLOP per byte,
// and nine workgroups read the same pixels; the launch ran at what that path delivers (~10 TB/s: 62.9 us for 41.5 GFLOP at batch 4,
// 0.26 of the matrix peak, MFMA busy 0.3).  Here a workgroup owns the three taps (ty, -1), (ty, 0), (ty, +1) of a pixel run: it loads
// the run of x ONCE (66 pixels: the 64 of the step and one on either side) and dy once, keeps THREE images of x in LDS -- image tx
// holds x[p + tx] at pixel p, zeros where p + tx leaves the row: a loaded chunk of pixel q goes to image 0 at q, to image -1 at q + 1
// unless q ends a row, to image +1 at q - 1 unless q starts one, so every position has exactly one writer -- and multiplies each of
// them with the same dy fragments: 192 MFMAs per 33 KB, 190 FLOP per byte.  Twelve 32 x 32 accumulators per wave (192 registers: one
// wave per SIMD, one workgroup per CU, 128 KB of LDS double-buffered), same partial layout [slice][tap][ci][co], same reduce.
// Taken from ~160 k pixels on (wgrad_row_form below: what it gains and why not more).
template <typename T>
__global__ __launch_bounds__(256, 1) void k_conv2d_wgrad3x3_row(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part,
                                                                int B, int H, int W, int steps, long long P, int slices) {
    constexpr int C = 128, PIXB = C * 2, IMG = 64 * PIXB;  // 256 bytes per pixel, 16 KB per 64-pixel image
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];   // [2 buffers][x(-1) | x(0) | x(+1) | dy][IMG]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // workgroup id = (group of 8 slices) * 24 + row * 8 + xcd: the three rows of a slice on ONE XCD (they share dy and two thirds of x)
    const int within = blockIdx.x % 24, slice = (blockIdx.x / 24) * 8 + within % 8;
    const int trow = within / 8, ty = trow - 1;
    if (slice >= slices) return;
    const long long step0 = (long long)slice * steps;
    const long long total_steps = (P + 63) / 64;
    const int nst = (int)(step0 + steps <= total_steps ? steps : (total_steps > step0 ? total_steps - step0 : 0));
    const unsigned tensor_bytes = (unsigned)(P * PIXB);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x), 0, (int)tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dy), 0, (int)tensor_bytes, 0x00020000);
    const int cq = tid & 15, pl = tid >> 4;                // chunk cq of the pixels pl, pl + 16, pl + 32, pl + 48 of a step
    const bool ex_lo = pl == 0, ex_hi = pl == 15;          // ... and of the pixel in front of the step / behind it
    unsigned fq;
    int fx, fy, fb;
    {
        fq = (unsigned)(step0 * 64 + pl);
        const unsigned HW = (unsigned)H * (unsigned)W;
        fb = (int)(fq / HW);
        const unsigned rem = fq - (unsigned)fb * HW;
        fy = (int)(rem / (unsigned)W);
        fx = (int)(rem - (unsigned)fy * (unsigned)W);
    }
    int fs = 0;
    struct Stage { tu32x4 x[5], d[4]; unsigned edge; };    // edge bit i: pixel i starts its row, bit 4 + i: it ends it
    Stage sa, sb, sc;
    auto fetch = [&](Stage &s) {
        const bool oks = fs < nst;
        unsigned edge = 0u, extra = 0xfffffff0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool okp = oks && (long long)fq < P;
            s.d[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, okp ? fq * (unsigned)PIXB + cq * 16u : 0xfffffff0u, 0, 0);
            const int sy = fy + ty;
            const bool okx = okp && (unsigned)sy < (unsigned)H;
            const unsigned xo = (unsigned)((fb * H + sy) * W + fx) * (unsigned)PIXB + cq * 16u;
            s.x[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, okx ? xo : 0xfffffff0u, 0, 0);
            edge |= (unsigned)(fx == 0) << i | (unsigned)(fx == W - 1) << (4 + i);
            if (i == 0 && ex_lo && okx && fx > 0) extra = xo - (unsigned)PIXB;         // x of the pixel in front of the step, same row
            if (i == 3 && ex_hi && okx && fx < W - 1) extra = xo + (unsigned)PIXB;     // ... of the pixel behind it
            fq += 16;
            fx += 16;
            const bool wrap = fx >= W;
            fx = wrap ? fx - W : fx;
            fy = wrap ? fy + 1 : fy;
            const bool wrapy = fy == H;
            fy = wrapy ? 0 : fy;
            fb = wrapy ? fb + 1 : fb;
        }
        s.x[4] = __builtin_amdgcn_raw_buffer_load_b128(xrs, extra, 0, 0);
        s.edge = edge;
        ++fs;
    };
    // LDS image: [pixel][16 chunks], chunk index ^ 4 * (pixel & 3)
    auto img_off = [&](int pix_lo2) { return (cq ^ ((pix_lo2 & 3) << 2)) << 4; };
    const int off_c = pl * PIXB + img_off(pl), off_p = (pl + 1) * PIXB + img_off(pl + 1), off_m = (pl - 1) * PIXB + img_off(pl - 1);
    const tu32x4 zero4 = {0u, 0u, 0u, 0u};
    auto put = [&](int buf, const Stage &s) {
        char *b = wg_smem + buf * 4 * IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = i * 16 * PIXB;
            *reinterpret_cast<tu32x4 *>(b + IMG + off_c + o) = s.x[i];                                               // tap 0 of this pixel
            if (!(ex_hi && i == 3)) *reinterpret_cast<tu32x4 *>(b + off_p + o) = ((s.edge >> (4 + i)) & 1u) ? zero4 : s.x[i];   // tap -1 of the next one
            if (!(ex_lo && i == 0)) *reinterpret_cast<tu32x4 *>(b + 2 * IMG + off_m + o) = ((s.edge >> i) & 1u) ? zero4 : s.x[i];  // tap +1 of the previous one
            *reinterpret_cast<tu32x4 *>(b + 3 * IMG + off_c + o) = s.d[i];
        }
        if (ex_lo) *reinterpret_cast<tu32x4 *>(b + img_off(0)) = s.x[4];                                    // tap -1 of pixel 0
        if (ex_hi) *reinterpret_cast<tu32x4 *>(b + 2 * IMG + 63 * PIXB + img_off(63)) = s.x[4];             // tap +1 of pixel 63
    };
    const int g = lane >> 4, li = lane & 15;
    const int cih = (wv >> 1) * 64, coh = (wv & 1) * 64;
    int ra_off[2], rb_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int prow = (g >> 1) * 8 + (li >> 2);
        const int sw = (li >> 2) << 2;
        const int ca = (cih / 8 + t * 4 + (g & 1) * 2 + ((li & 3) >> 1)) ^ sw;
        const int cb = (coh / 8 + t * 4 + (g & 1) * 2 + ((li & 3) >> 1)) ^ sw;
        ra_off[t] = prow * PIXB + (ca << 4) + (li & 1) * 8;
        rb_off[t] = prow * PIXB + (cb << 4) + (li & 1) * 8;
    }
    tf32x16 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t >> 2][(t >> 1) & 1][t & 1][i] = 0.0f;
    auto frag = [&](const char *base, int off) {
        const uint2 lo = lds_tr16_b64(base + off), hi = lds_tr16_b64(base + off + 4 * PIXB);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    // one wave per SIMD: nothing but this wave's own instruction order hides an LDS round trip, so the eight fragments of k-step
    // ks + 1 are read while the twelve MFMAs of k-step ks run (register double buffer; first form: reads right in front of their
    // MFMAs, s_waitcnt lgkmcnt(2) sixteen times per step)
    auto mfmas = [&](int buf) {
        const char *b = wg_smem + buf * 4 * IMG;
        uint4 fa[2][3][2], fb[2][2];
        auto read_ks = [&](int ks, int q) {
            const char *bd = b + 3 * IMG + ks * 16 * PIXB;
            fb[q][0] = frag(bd, rb_off[0]);
            fb[q][1] = frag(bd, rb_off[1]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const char *bx = b + t * IMG + ks * 16 * PIXB;
                fa[q][t][0] = frag(bx, ra_off[0]);
                fa[q][t][1] = frag(bx, ra_off[1]);
            }
        };
        read_ks(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks & 1;
            if (ks + 1 < 4) read_ks(ks + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[t][0][0] = MfmaT<T>::run(fa[q][t][0], fb[q][0], acc[t][0][0]);
                acc[t][0][1] = MfmaT<T>::run(fa[q][t][0], fb[q][1], acc[t][0][1]);
                acc[t][1][0] = MfmaT<T>::run(fa[q][t][1], fb[q][0], acc[t][1][0]);
                acc[t][1][1] = MfmaT<T>::run(fa[q][t][1], fb[q][1], acc[t][1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (nst > 0) {
        fetch(sa);
        fetch(sb);
        fetch(sc);
        put(0, sa);
        __syncthreads();
#define SEC_WGRAD_ROW_STEP(S, FS, PS)                                                                  \
        mfmas((S) & 1);                                                                                \
        fetch(FS);                                                                                     \
        put(((S) + 1) & 1, PS);                                                                        \
        __syncthreads();
        for (int s = 0; s < nst; s += 3) {
            SEC_WGRAD_ROW_STEP(s, sa, sb)
            SEC_WGRAD_ROW_STEP(s + 1, sb, sc)
            SEC_WGRAD_ROW_STEP(s + 2, sc, sa)
        }
#undef SEC_WGRAD_ROW_STEP
    }
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        float *dst = part + ((size_t)slice * 9 + trow * 3 + tt) * C * C;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = cih + (t >> 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, co = coh + (t & 1) * 32 + r;
                dst[(size_t)ci * C + co] = acc[tt][t >> 1][t & 1][i];
            }
    }
}

// dw[co][ci][tap] (torch's [Cout][Cin][3][3]) = sum_g part[g][tap][ci][co], g ascending (fixed order: run-to-run identical)
__global__ __launch_bounds__(256) void k_conv2d_wgrad_reduce(const float *__restrict__ part, int groups, float *__restrict__ dw, int ntaps, int cout) {
    constexpr int C = 128;
    const int e = blockIdx.x * 256 + threadIdx.x;           // e = (tap * C + ci) * C + co
    if (e >= ntaps * C * C) return;
    const size_t stride = (size_t)ntaps * C * C;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;       // four chains (eight loads in flight), combined in a fixed order
    int g = 0;
    for (; g + 8 <= groups; g += 8) {
        const float *p = part + (size_t)g * stride + e;
        const float v0 = p[0], v1 = p[stride], v2 = p[2 * stride], v3 = p[3 * stride];
        const float v4 = p[4 * stride], v5 = p[5 * stride], v6 = p[6 * stride], v7 = p[7 * stride];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        s0 += v4; s1 += v5; s2 += v6; s3 += v7;
    }
    for (; g < groups; ++g) s0 += part[(size_t)g * stride + e];
    const float s = (s0 + s1) + (s2 + s3);
    const int co = e % C, ci = (e / C) % C, tap = e / (C * C);
    if (co < cout) dw[((size_t)co * C + ci) * ntaps + tap] = s;      // dw is [cout][128][taps]: cout = 64 for the stacked heads
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
        // four pixels per trip with all their loads issued first (same pixels, same order of additions as one pixel per trip: the
        // sums are bit-identical): one 16-byte load per tensor in flight per thread left the pass at 3.6-3.8 TB/s on activations the
        // previous kernel had just written (19 us for the 72 MB of a backward pass at batch 4)
        constexpr int U = 4;
        const long long stride = (long long)gridDim.x * ppi;
        const uint4 *y4 = reinterpret_cast<const uint4 *>(y), *d4 = reinterpret_cast<const uint4 *>(dz);
        for (long long p0 = (long long)blockIdx.x * ppi + pl; p0 < P; p0 += U * stride) {
            uint4 v[U], gq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long p = p0 + u * stride;
                const bool ok = p < P;
                v[u] = ld_sel(y4, p * cg + chg, ok, make_uint4(0u, 0u, 0u, 0u));
                if (MODE == 1) gq[u] = ld_sel(d4, p * cg + chg, ok, make_uint4(0u, 0u, 0u, 0u));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p0 + u * stride >= P) break;
                const T *e = reinterpret_cast<const T *>(&v[u]);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = t2f<T>(e[j]);
                        a[j] += f;
                        b2[j] += f * f;
                    }
                } else {
                    const T *ge = reinterpret_cast<const T *>(&gq[u]);
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

// the element loop of both apply kernels: chunk q, q + qs, ... of 8 channels whose parameters the caller holds in registers
template <typename T, int MODE>
__device__ __forceinline__ void bn_apply_loop(const T *__restrict__ y, const T *__restrict__ dz, T *__restrict__ out, long long q0, long long qs,
                                              long long n, const float (&mu)[8], const float (&is)[8], const float (&ga)[8],
                                              const float (&be)[8], const float (&db)[8], const float (&dg)[8], float inv_p, int relu) {
    constexpr int U = 4;
    const uint4 *y4 = reinterpret_cast<const uint4 *>(y), *d4 = reinterpret_cast<const uint4 *>(dz);
    for (long long qb = q0; qb < n; qb += U * qs) {
        uint4 v[U], gq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = qb + u * qs;
            v[u] = ld_sel(y4, q, q < n, make_uint4(0u, 0u, 0u, 0u));
            if (MODE == 1) gq[u] = ld_sel(d4, q, q < n, make_uint4(0u, 0u, 0u, 0u));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = qb + u * qs;
            if (q >= n) break;
            const T *e = reinterpret_cast<const T *>(&v[u]);
            const T *ge = reinterpret_cast<const T *>(&gq[u]);
            uint4 o;
            T *oe = reinterpret_cast<T *>(&o);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (t2f<T>(e[j]) - mu[j]) * is[j];
                const float zp = xh * ga[j] + be[j];
                if (MODE == 0) {
                    oe[j] = f2t<T>(relu ? (zp > 0.0f ? zp : 0.0f) : zp);
                } else {
                    float g = t2f<T>(ge[j]);
                    if (relu && !(zp > 0.0f)) g = 0.0f;
                    oe[j] = f2t<T>(ga[j] * is[j] * (g - db[j] * inv_p - xh * dg[j] * inv_p));
                }
            }
            reinterpret_cast<uint4 *>(out)[q] = o;
        }
    }
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
    // The grid stride is a multiple of 256 and cg divides 256, so a thread's channel group never changes: its 8 channels' parameters
    // are read ONCE (the loop used to fetch six 4-byte values per channel and element from global memory: 48 loads for one 16-byte
    // chunk), and four chunks per trip are loaded before any is used.  Same arithmetic per element: bit-identical outputs.
    const long long q0 = (long long)blockIdx.x * 256 + threadIdx.x, qs = (long long)gridDim.x * 256;
    const int chg = (int)(q0 % cg);
    float mu[8], is[8], ga[8], be[8], db[8], dg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chg * 8 + j;
        mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
        db[j] = MODE == 1 ? dbeta[c] : 0.0f; dg[j] = MODE == 1 ? dgamma[c] : 0.0f;
    }
    bn_apply_loop<T, MODE>(y, dz, out, q0, qs, n, mu, is, ga, be, db, dg, inv_p, relu);
}

// Small activations (the sparse stack's BatchNorm1d rows: at most kBnSmallRows): the partial sums come from kBnSmallGroups workgroups
// only (32 were too few to pull the rows at speed: the partial pass doubled), and every workgroup of the apply pass adds them up
// itself in its prologue (256 / C threads per channel, fixed order, doubles as in k_bn_fwd_finalize) -- the finalize launch between the two passes, ~6 us of pure latency 28 times per car.fhd step, is gone.
// Workgroup 0 also stores what the finalize stored (save_mean / save_invstd + running statistics; dgamma / dbeta).
constexpr int kBnSmallGroups = 128;
constexpr long long kBnSmallRows = 1 << 17;
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_bn_apply_small(const T *__restrict__ y, const T *__restrict__ dz, long long P, int C,
                                                       const float *__restrict__ part, float eps, float momentum,
                                                       float *__restrict__ mean_io, float *__restrict__ invstd_io,
                                                       float *__restrict__ running_mean, float *__restrict__ running_var,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ dbeta_out, float *__restrict__ dgamma_out, int relu,
                                                       T *__restrict__ out, const int *__restrict__ p_dev) {
    __shared__ float s_a[256], s_b[256], s_g[256], s_be[256];     // MODE 0: mean, invstd;  MODE 1: + dbeta, dgamma in s_g2 / s_b2
    __shared__ float s_g2[256], s_b2[256];
    P = bn_live_rows(P, p_dev);                        // rows past the live count are neither read nor written
    // 256 / C threads per channel, each adds its share of the groups (ascending), thread c then adds the shares (ascending): fixed order
    __shared__ double s_ra[256], s_rb[256];
    {
        const int c1 = threadIdx.x % C, slice = threadIdx.x / C, nsl = 256 / C;
        double a = 0.0, b = 0.0;
#pragma unroll 8
        for (int g = slice; g < kBnSmallGroups; g += nsl) { a += part[((size_t)g * 2 + 0) * C + c1]; b += part[((size_t)g * 2 + 1) * C + c1]; }
        s_ra[threadIdx.x] = a; s_rb[threadIdx.x] = b;
    }
    __syncthreads();
    const int c0 = threadIdx.x;
    if (c0 < C) {
        double a = 0.0, b = 0.0;
        for (int sl = 0; sl < 256 / C; ++sl) { a += s_ra[sl * C + c0]; b += s_rb[sl * C + c0]; }
        if (MODE == 0) {
            const long long Pn = P < 1 ? 1 : P;
            const double m = a / (double)Pn;
            double var = b / (double)Pn - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
            s_a[c0] = mf; s_b[c0] = is;
            if (blockIdx.x == 0) {
                mean_io[c0] = mf; invstd_io[c0] = is;
                if (running_mean) running_mean[c0] = (1.0f - momentum) * running_mean[c0] + momentum * mf;
                if (running_var) running_var[c0] = (1.0f - momentum) * running_var[c0] + momentum * (float)(Pn > 1 ? var * (double)Pn / (double)(Pn - 1) : var);
            }
        } else {
            s_a[c0] = mean_io[c0]; s_b[c0] = invstd_io[c0];
            s_b2[c0] = (float)a; s_g2[c0] = (float)b;             // dbeta, dgamma
            if (blockIdx.x == 0) { dbeta_out[c0] = (float)a; dgamma_out[c0] = (float)b; }
        }
        s_g[c0] = gamma[c0]; s_be[c0] = beta[c0];
    }
    __syncthreads();
    const int cg = C / 8;
    const long long n = P * cg;
    const float inv_p = 1.0f / (float)(P > 0 ? P : 1);
    const long long q0 = (long long)blockIdx.x * 256 + threadIdx.x, qs = (long long)gridDim.x * 256;
    const int chg = (int)(q0 % cg);                    // (invariant per thread: see k_bn_apply)
    float mu[8], is[8], ga[8], be[8], db[8], dg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chg * 8 + j;
        mu[j] = s_a[c]; is[j] = s_b[c]; ga[j] = s_g[c]; be[j] = s_be[c];
        db[j] = MODE == 1 ? s_b2[c] : 0.0f; dg[j] = MODE == 1 ? s_g2[c] : 0.0f;
    }
    bn_apply_loop<T, MODE>(y, dz, out, q0, qs, n, mu, is, ga, be, db, dg, inv_p, relu);
}

// fp32 master weight [cout][cin][k][k] -> BOTH 16-bit MFMA slab images of a training step in one launch: `fwd` in the layout of
// k_conv2d_pack (packed[((tap * cin/8 + chunk) * cout + n) * 8 + e] = w[n][chunk*8 + e][tap]) and `dgrad` = the same layout of the
// flipped, transposed kernel W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx] (cin and cout swap roles): what the data gradient's
// forward-kernel launch reads.  Replaces to(dtype) + pack + flip + transpose + contiguous + to(dtype) + pack (7 launches).
template <typename T>
__device__ __forceinline__ void conv2d_pack_train_at(long long g, const float *__restrict__ w, int cout, int cin, int ks, T *__restrict__ fwd,
                                                     T *__restrict__ dgrad) {
    const long long total = (long long)ks * ks * cin * cout;
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

template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv2d_pack_train(const float *__restrict__ w, int cout, int cin, int ks, T *__restrict__ fwd,
                                                             T *__restrict__ dgrad) {
    conv2d_pack_train_at<T>((long long)blockIdx.x * kBlock + threadIdx.x, w, cout, cin, ks, fwd, dgrad);
}
// several layers in ONE launch (descriptors as kernel arguments: capturable as is)
constexpr int kPack2dMulti = 16;
struct Pack2dDesc {
    const float *w;
    void *fwd, *dgrad;
    int cout, cin, ks, blk0;
};
struct Pack2dArgs {
    Pack2dDesc d[kPack2dMulti];
    int n;
};
template <typename T>
__global__ __launch_bounds__(kBlock) void k_conv2d_pack_train_multi(Pack2dArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.d[i + 1].blk0) ++i;
    const Pack2dDesc &d = a.d[i];
    conv2d_pack_train_at<T>((long long)(blockIdx.x - d.blk0) * kBlock + threadIdx.x, d.w, d.cout, d.cin, d.ks, (T *)d.fwd, (T *)d.dgrad);
}

constexpr int kBnGroups = 512;                             // (2048 measured in round 6: the partial pass 10.6 -> 9.4 us, each finalize 6 -> 16 us)
// slices of the row form: x 3 kernel rows = 240 workgroups.  The three rows of a slice go to ONE XCD (they share dy and most of x), eight
// slices per group of 24 workgroups, so an XCD gets 3 * ceil(slices / 8) workgroups of 128 KB LDS each for its 32 CUs: 80 slices is the
// most that fits one round (85 gave XCDs 0 and 1 a 33rd workgroup: a second round, the launch took twice as long)
constexpr int kWgradRowSlices = 80;
// Which maps take the row form.  Measured (gpurun r06_ab / r06_ac / r06_ad, us per weight gradient + reduce, tap form -> row form):
// 4 x 200 x 176 (car.fhd training) 77 -> 81, 3 x 248 x 248 (nuScenes) 108 -> 92, 8 x 200 x 176 139 -> 123; with NO operand traffic the
// row form still takes 71 us at 4 x 200 x 176 -- one wave per SIMD walks LDS stores (64 KB of images per step at the ~80 B/clk of
// ds_write_b128), a barrier, fragment reads and 48 MFMAs in sequence -- so it wins only where the tap form's nine passes over x and dy
// cost more than that: from ~160 k pixels.  SEC_WGRAD_ROW=0 / 1 (read per call) forces a form: tests run every shape through both.
static bool wgrad_row_form(long long pixels) {
    const char *e = getenv("SEC_WGRAD_ROW");
    if (e && *e) return atoi(e) != 0;
    return pixels >= 160000;
}

template <typename T>
static int run_wgrad(const void *x, const void *dy, int B, int H, int W, float *dw, void *ws, size_t ws_bytes, hipStream_t st, int ntaps, int cout) {
    const long long P = (long long)B * H * W;
    if (P * 128 * 2 >= (1ll << 31)) return SEC_E_UNSUPPORTED;   // 32-bit buffer offsets (8.4 M pixels; the nuScenes maps hold 0.5 M)
    const long long total_steps = (P + 63) / 64;
    if (ntaps == 9 && W >= 16 && wgrad_row_form(P)) {
        // a kernel row per workgroup: one workgroup per CU (128 KB of LDS), slices x 3 rows ~ 256 workgroups
        int steps = (int)((total_steps + kWgradRowSlices - 1) / kWgradRowSlices);
        steps = (steps + 2) / 3 * 3;
        if (steps < 3) steps = 3;
        const long long gx = (total_steps + steps - 1) / steps;
        const size_t need = (size_t)gx * 9 * 128 * 128 * sizeof(float);
        if (ws_bytes < need) return SEC_E_WORKSPACE;
        constexpr int lds = 2 * 4 * 64 * 256;
        auto fn = k_conv2d_wgrad3x3_row<T>;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_set = true;
        }
        hipLaunchKernelGGL(fn, dim3((unsigned)((gx + 7) / 8 * 8 * 3)), dim3(256), lds, st, (const T *)x, (const T *)dy, (float *)ws, B, H, W, steps, P, (int)gx);
        hipLaunchKernelGGL(k_conv2d_wgrad_reduce, dim3(div_up(9 * 128 * 128, 256)), dim3(256), 0, st, (const float *)ws, (int)gx, dw, 9, cout);
        return check_launch();
    }
    int steps = ntaps == 1 ? 6 : 42;                         // multiples of 3 (the kernel's unrolled stage ring); ~2 workgroups per CU at batch 4 (2200 steps -> 53 runs x 9 taps)
    long long gx = (total_steps + steps - 1) / steps;
    if (gx > 256) { steps = (int)((total_steps + 255) / 256); steps = (steps + 2) / 3 * 3; gx = (total_steps + steps - 1) / steps; }
    if (gx < 1) gx = 1;
    const size_t need = (size_t)gx * ntaps * 128 * 128 * sizeof(float);
    if (ws_bytes < need) return SEC_E_WORKSPACE;
    if (W >= 16)
        hipLaunchKernelGGL((k_conv2d_wgrad3x3<T, false>), dim3((unsigned)((gx + 7) / 8 * 8 * ntaps)), dim3(256), 0, st, (const T *)x, (const T *)dy,
                           (float *)ws, B, H, W, steps, P, (int)gx, ntaps, cout);
    else
        hipLaunchKernelGGL((k_conv2d_wgrad3x3<T, true>), dim3((unsigned)((gx + 7) / 8 * 8 * ntaps)), dim3(256), 0, st, (const T *)x, (const T *)dy,
                           (float *)ws, B, H, W, steps, P, (int)gx, ntaps, cout);
    hipLaunchKernelGGL(k_conv2d_wgrad_reduce, dim3(div_up(ntaps * 128 * 128, 256)), dim3(256), 0, st, (const float *)ws, (int)gx, dw, ntaps, cout);
    return check_launch();
}

}  // namespace sec

using namespace sec;

SEC_API size_t sec_conv2d_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int ksize) {
    if (cin != 128 || !(cout == 128 || (cout == 64 && ksize == 1)) || (ksize != 3 && ksize != 1) || batch <= 0 || h <= 0 || w <= 0) return 0;
    const long long total_steps = ((long long)batch * h * w + 63) / 64;
    const int steps0 = ksize == 1 ? 6 : 42;
    long long gx = (total_steps + steps0 - 1) / steps0;
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    if (ksize == 3 && gx < kWgradRowSlices) gx = kWgradRowSlices;     // the row form cuts the pixels into up to kWgradRowSlices slices
    return (size_t)gx * ksize * ksize * 128 * 128 * sizeof(float);
}

SEC_API int sec_conv2d_wgrad_nhwc(const void *x, const void *dy, int batch, int h, int w, int cin, int cout, int ksize, int stride,
                                  int pad, float *dweight, void *workspace, size_t workspace_bytes, int dtype, void *stream) {
    if (!x || !dy || !dweight || !workspace || batch <= 0 || h <= 0 || w <= 0) return SEC_E_INVALID;
    if (cin != 128 || !(cout == 128 || (cout == 64 && ksize == 1)) || stride != 1 || !((ksize == 3 && pad == 1) || (ksize == 1 && pad == 0)) ||
        (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SEC_BF16) return run_wgrad<__hip_bfloat16>(x, dy, batch, h, w, dweight, workspace, workspace_bytes, st, ksize * ksize, cout);
    return run_wgrad<__half>(x, dy, batch, h, w, dweight, workspace, workspace_bytes, st, ksize * ksize, cout);
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

SEC_API int sec_conv2d_pack_weight_train_multi(int n, const float *const *weights, const int *cout, const int *cin, const int *ksize, int dtype,
                                               void *const *packed_fwd, void *const *packed_dgrad, void *stream) {
    if (n <= 0 || !weights || !cout || !cin || !ksize || !packed_fwd || !packed_dgrad || (dtype != SEC_BF16 && dtype != SEC_F16)) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += kPack2dMulti) {
        Pack2dArgs a;
        a.n = n - i0 < kPack2dMulti ? n - i0 : kPack2dMulti;
        long long blocks = 0;
        for (int j = 0; j < a.n; ++j) {
            const int i = i0 + j;
            if (!weights[i] || !packed_fwd[i] || !packed_dgrad[i] || cout[i] <= 0 || cin[i] <= 0 || ksize[i] <= 0) return SEC_E_INVALID;
            if (cin[i] % 64 || cout[i] % 64) return SEC_E_UNSUPPORTED;
            Pack2dDesc &d = a.d[j];
            d.w = weights[i]; d.fwd = packed_fwd[i]; d.dgrad = packed_dgrad[i]; d.cout = cout[i]; d.cin = cin[i]; d.ks = ksize[i]; d.blk0 = (int)blocks;
            blocks += div_up((long long)ksize[i] * ksize[i] * cin[i] * cout[i], kBlock);
        }
        if (dtype == SEC_BF16) hipLaunchKernelGGL((k_conv2d_pack_train_multi<__hip_bfloat16>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((k_conv2d_pack_train_multi<__half>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
    }
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
#define SEC_BN_FWD_SMALL(T)                                                                                                        \
    hipLaunchKernelGGL((k_bn_partial<T, 0>), dim3(kBnSmallGroups), dim3(256), 0, st, (const T *)y, (const T *)nullptr, pixels, channels, \
                       nullptr, nullptr, nullptr, nullptr, 0, part, pixels_dev);                                                   \
    hipLaunchKernelGGL((k_bn_apply_small<T, 0>), dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, (const T *)y, (const T *)nullptr, pixels, channels, \
                       part, eps, momentum, save_mean, save_invstd, running_mean, running_var, gamma, beta, nullptr, nullptr, relu, \
                       (T *)z, pixels_dev);
    if (pixels <= kBnSmallRows) {
        if (dtype == SEC_BF16) { SEC_BN_FWD_SMALL(__hip_bfloat16) } else { SEC_BN_FWD_SMALL(__half) }
        return check_launch();
    }
#undef SEC_BN_FWD_SMALL
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
#define SEC_BN_BWD_SMALL(T)                                                                                                        \
    hipLaunchKernelGGL((k_bn_partial<T, 1>), dim3(kBnSmallGroups), dim3(256), 0, st, (const T *)y, (const T *)dz, pixels, channels, \
                       save_mean, save_invstd, gamma, beta, relu, part, pixels_dev);                                               \
    hipLaunchKernelGGL((k_bn_apply_small<T, 1>), dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, (const T *)y, (const T *)dz, pixels, channels, part, \
                       0.0f, 0.0f, const_cast<float *>(save_mean), const_cast<float *>(save_invstd), nullptr, nullptr, gamma, beta, \
                       dbeta, dgamma, relu, (T *)dy, pixels_dev);
    if (pixels <= kBnSmallRows) {
        if (dtype == SEC_BF16) { SEC_BN_BWD_SMALL(__hip_bfloat16) } else { SEC_BN_BWD_SMALL(__half) }
        return check_launch();
    }
#undef SEC_BN_BWD_SMALL
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
