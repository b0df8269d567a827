// Strided / patch convolutions of the PointPillars RPN (second/pytorch/models/rpn.py:275-302: the stride-2 first conv of every block,
// `ZeroPad2d(1) + Conv2d(k3, s2)` at :484-486, and the deblocks `Conv2d(k, stride=k)` for upsample strides < 1 / `ConvTranspose2d(k1)`
// at :274-296; nuscenes/all.pp.largea.config:22-26), included by dense.hip (MfmaD, store_tile_t, Conv2dParams, the LDS-DMA pointer
// types are defined there).
//
// The generic implicit GEMM (k_conv2d_nhwc_dma) walks K one (tap, 64-channel slab) at a time with ONE slab of lookahead and a
// full barrier per slab: at batch 4 these layers launch less than one round of workgroups, so a workgroup's time is its 4 ... 16
// dependent LDS-DMA round trips (~1.1 us each; r06_pp0: 12 ... 39 us per layer at 0.02 ... 0.13 of the MFMA peak).  Here the
// whole input footprint of an output tile lands in LDS with ONE wave of DMA pieces, the weights never touch LDS (every wave streams
// the B fragments of its own 32 output channels from L2 through an 8-deep register ring, as k_conv2d_halo_reg does) and the K loop
// runs without a barrier.
//
// LDS image: the footprint is stored as ST x ST PHASE PLANES -- plane (py, px) holds the input pixels whose (row, column) offset
// inside the footprint is = (py, px) mod ST -- so that the 32 pixels of an m-tile, which are ST apart in the image, are NEIGHBOURS in
// their plane for every tap: tap (dy, dx) reads plane (dy % ST, dx % ST) at offset (dy / ST, dx / ST).  A k == stride conv has one
// plane per tap.  Inside a plane a pixel's CIN / 8 16-byte chunks are XOR-swizzled with a key of the pixel's plane (row, column)
// chosen for the lane groups ds_read_b128 is served in ({0-3, 12-15, 20-27}, ... : MI355X_MICROARCH.md, LDS table): 16-wide tiles
// put 16 different columns into a group (key = column), 8-wide tiles 8 columns x 2 row parities; 64-channel pixels are half a
// bank row, their key drops the column's low bit (the pixel's parity is the other half of the slot; plane widths are even).
namespace patch {

template <int KS, int ST, int TH, int TW> struct Geom {
    static constexpr int IH = (TH - 1) * ST + KS, IW = (TW - 1) * ST + KS;   // input footprint of a TH x TW output tile
    static constexpr int NPL = ST * ST;
    static constexpr int ph(int py) { return (IH - py + ST - 1) / ST; }
    static constexpr int pw(int px) { return (((IW - px + ST - 1) / ST) + 1) & ~1; }    // even: see the 64-channel key
    static constexpr int base(int pl) {
        int s = 0;
        for (int i = 0; i < pl; ++i) s += ph(i / ST) * pw(i % ST);
        return s;
    }
    static constexpr int NPIX = base(NPL);
    static constexpr bool uniform = KS % ST == 0;      // every plane is TH x TW
};

template <int CH, int TW> constexpr __host__ __device__ __forceinline__ unsigned key_of(int row, int col) {
    if (CH == 8) return TW == 16 ? (unsigned)((col >> 1) & 7) : (unsigned)(((col >> 1) & 3) | ((row & 1) << 2));
    return TW == 16 ? (unsigned)(col & 15) : (unsigned)((col & 7) | ((row & 1) << 3));
}

// Where LDS entry e = piece * 64 + lane of the footprint comes from: bits 0-7 input row offset, 8-15 input column offset (from the footprint's
// corner), 16-21 the pixel's source chunk (swizzle applied), bit 31 = the entry exists.  Built at compile time: decoding e in the kernel
// (plane search, two divisions by the plane widths, key) cost ~60 VALU instructions per 1 KB piece, 3x the whole K loop's VALU count on the
// first PointPillars conv (r06_pp6: SQ_INSTS_VALU 6.9 M against 2.3 M of the stride-1 layer behind it).
template <int CIN, int KS, int ST, int TH, int TW> struct PieceTable {
    using G = Geom<KS, ST, TH, TW>;
    static constexpr int CH = CIN / 8, HENT = G::NPIX * CH, N = (HENT + 63) / 64 * 64;
    struct Arr { unsigned v[N]; };
    static constexpr Arr make() {
        Arr a{};
        int bases[G::NPL + 1] = {};
        for (int q = 0; q < G::NPL; ++q) bases[q + 1] = bases[q] + G::ph(q / ST) * G::pw(q % ST);
        for (int e = 0; e < N; ++e) {
            a.v[e] = 0u;
            if (e >= HENT) continue;
            const int pp = e / CH, slot = e - pp * CH;
            int pl = 0;
            for (int q = 1; q < G::NPL; ++q)
                if (pp >= bases[q]) pl = q;
            const int loc = pp - bases[pl];
            const int py = pl / ST, px = pl - py * ST;
            const int row = loc / G::pw(px), col = loc - row * G::pw(px);
            a.v[e] = (unsigned)(row * ST + py) | (unsigned)(col * ST + px) << 8 | (unsigned)(slot ^ (int)key_of<CH, TW>(row, col)) << 16 | 1u << 31;
        }
        return a;
    }
};
template <int CIN, int KS, int ST, int TH, int TW>
__device__ constexpr typename PieceTable<CIN, KS, ST, TH, TW>::Arr g_piece_table = PieceTable<CIN, KS, ST, TH, TW>::make();

// PXS = 1: a wave owns all TH x TW pixels for 32 output channels (128 per workgroup); PXS = 2: the waves split the pixels two ways and
// the channels two ways (64 output channels per workgroup).  `y` has a channel pitch of `ldc` elements (a slice of a wider map).
// ROWS (sec_conv2d_nhwc_rows; PointPillarsScatter + the first RPN conv, pointpillars.py:444-476 -> rpn.py:484-486): the input is not an
// image but the pillar feature rows [rows][CIN] and `site_map` [batch][h][w] = row + 1 of the cell's pillar (0 = none): the footprint
// pieces are gathered from the rows (cells without a pillar and the padding read zeros through the buffer bounds check) -- what the
// zero fill + scatter + this conv computed from the 82 MB canvas of config 4, without the canvas.  A tile whose footprint holds no
// pillar skips the DMA and the K loop (its result is act(bias), exactly: 0 * w accumulates to 0).
// TABLE: the footprint entries come from g_piece_table instead of being decoded in the kernel -- for the ROWS form on launches of several
// rounds of workgroups only.  ROWS needs every entry before its first DMA anyway (map lookups first), so there the table replaces ~60 VALU
// instructions per piece by one load (first conv of config 4, 1300 workgroups: 27.4 -> 22.9 us); on a ONE-round launch the extra dependent
// load costs ~1 us, and the image form -- which decodes an entry right before issuing its piece, so its DMAs leave while the decode of the
// next ones runs -- is 6-8 us SLOWER with the table at every size (29.3 vs 35.7 us at 4 x 400 x 400, 53.5 vs 61.7 us at config 5's 128 -> 256).
template <typename T, int CIN, int KS, int ST, int TH, int TW, int PXS, bool ROWS = false, bool TABLE = false>
__global__ __launch_bounds__(256, 2) void k_conv2d_patch(const T *__restrict__ x, const T *__restrict__ wpk, const float *__restrict__ bias,
                                                         T *__restrict__ y, Conv2dParams p, int tiles_y, int tiles_x, int per_xcd, int ldc,
                                                         const int *__restrict__ site_map = nullptr, unsigned feat_bytes = 0) {
    using G = Geom<KS, ST, TH, TW>;
    static_assert(TW == 8 || TW == 16, "m-tiles of 32 pixels: 2 x 16 or 4 x 8");
    static_assert((TH * TW) % (32 * PXS) == 0 && (PXS == 1 || PXS == 2), "whole m-tiles per wave");
    constexpr int CH = CIN / 8, HENT = G::NPIX * CH, NPIECE = (HENT + 63) / 64;
    constexpr int MT = TH * TW / 32 / PXS, NWC = 4 / PXS;
    constexpr int KSTEPS = CIN / 16, NK = KS * KS * KSTEPS;
    constexpr int OFFS = (KS - 1) / ST + 1;          // distinct tap offsets inside a plane (per axis)
#ifndef SEC_PATCH_RD
#define SEC_PATCH_RD 8
#endif
    constexpr int RD = NK < SEC_PATCH_RD ? NK : SEC_PATCH_RD;      // B ring: fragments loaded RD - 1 k-steps ahead
    extern __shared__ __attribute__((aligned(16))) uint4 patch_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): an XCD owns a contiguous run of tiles = a band of image rows
    const int xcd = blockIdx.x % 8, local = blockIdx.x / 8;
    const int tile = xcd * per_xcd + local;
    if (local >= per_xcd || tile >= p.batch * tiles_y * tiles_x) return;
    const int b = tile / (tiles_y * tiles_x);
    const int trem = tile - b * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    constexpr int cin8 = CIN / 8;

    // ---- the footprint: one wave of 1 KB LDS-DMA pieces (entry e = piece * 64 + lane = (plane pixel, slot); the swizzle goes on the SOURCE chunk)
    const int iy_base = y0 * ST - p.pad, ix_base = x0 * ST - p.pad;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    constexpr int NP = (NPIECE + 3) / 4;             // pieces per wave
    // entry e -> bits 0-7 / 8-15 input row / column offset inside the footprint, 16-21 source chunk, 31 = exists (the format of g_piece_table)
    auto decode = [&](int e) -> unsigned {
        const int pp = e / CH, slot = e - pp * CH;
        int pl = 0, loc = pp;
        if constexpr (G::uniform) {
            constexpr int S = G::ph(0) * G::pw(0);
            pl = pp / S;
            loc = pp - pl * S;
        } else {
#pragma unroll
            for (int q = 1; q < G::NPL; ++q)
                if (pp >= G::base(q)) { pl = q; loc = pp - G::base(q); }
        }
        const int py = pl / ST, px = pl - py * ST;
        int row = loc / G::pw(0), col = loc - row * G::pw(0);
        if constexpr (!G::uniform) {
#pragma unroll
            for (int q = 1; q < ST; ++q)
                if (px == q && G::pw(q) != G::pw(0)) { row = loc / G::pw(q); col = loc - row * G::pw(q); }
        }
        return e < HENT ? (unsigned)(row * ST + py) | (unsigned)(col * ST + px) << 8 | (unsigned)(slot ^ (int)key_of<CH, TW>(row, col)) << 16 | 1u << 31 : 0u;
    };
    // TABLE / ROWS: every entry first (their loads in flight together); otherwise an entry is decoded right before its piece is issued, so
    // that the first DMA leaves after one decode, not after all of them (issuing behind the whole decode cost 2 us per layer: r06_pp8)
    unsigned ent[NP];
    if constexpr (TABLE || ROWS) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int i = wvs + 4 * t;
            if constexpr (TABLE) ent[t] = i < NPIECE ? g_piece_table<CIN, KS, ST, TH, TW>.v[i * 64 + lane] : 0u;
            else ent[t] = i < NPIECE ? decode(i * 64 + lane) : 0u;
        }
    }
    bool live = true;
    if constexpr (ROWS) {
        const unsigned plane = (unsigned)p.h * (unsigned)p.w;
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(site_map) + (size_t)b * plane, 0, (int)(plane * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x), 0, (int)feat_bytes, 0x00020000);
        unsigned rowp1[NP], any = 0;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int iy = iy_base + (int)(ent[t] & 255u), ix = ix_base + (int)((ent[t] >> 8) & 255u);
            const bool ok = (int)ent[t] < 0 && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
            rowp1[t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(mrs, ok ? (unsigned)(iy * p.w + ix) * 4u : 0xfffffffcu, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NP; ++t) any |= rowp1[t];
        // workgroup-wide "any": one word per wave, ONE barrier
        __shared__ unsigned wave_any[4];
        const unsigned long long bal = __ballot(any != 0);
        if (lane == 0) wave_any[wv] = bal != 0ull;
        __syncthreads();
        live = (wave_any[0] | wave_any[1] | wave_any[2] | wave_any[3]) != 0u;
        if (live) {
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const int i = wvs + 4 * t;
                if (i < NPIECE) {
                    const unsigned off = rowp1[t] ? (rowp1[t] - 1u) * (CIN * 2u) + ((ent[t] >> 12) & 0x3f0u) : 0xfffffff0u;     // no pillar / padding: out of bounds, zeros
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lds_ptr_t)&patch_smem[i * 64], 16, off, 0, 0, 0);
                }
            }
        }
    } else {
        const unsigned img_bytes = (unsigned)p.h * (unsigned)p.w * (CIN * 2u);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x) + (size_t)b * p.h * p.w * CIN, 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int i = wvs + 4 * t;
#if defined(SEC_PATCH_ABL) && SEC_PATCH_ABL == 2     // ablation build: no footprint DMA
            if (i < 0) {
#else
            if (i < NPIECE) {
#endif
                if constexpr (!TABLE) ent[t] = decode(i * 64 + lane);
                const int iy = iy_base + (int)(ent[t] & 255u), ix = ix_base + (int)((ent[t] >> 8) & 255u);
                const bool ok = (int)ent[t] < 0 && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
                const unsigned off = ok ? (unsigned)(iy * p.w + ix) * (CIN * 2u) + ((ent[t] >> 12) & 0x3f0u) : 0xfffffff0u;   // padding: out of bounds, zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)&patch_smem[i * 64], 16, off, 0, 0, 0);
            }
        }
    }

    // ---- this wave's output channels / pixels
    const int n0 = blockIdx.y * (128 / PXS) + (wv % NWC) * 32;
    const int mtb = (wv / NWC) * MT;
    const int q0 = mtb * 32 + r;
    const int ty0 = q0 / TW, tx0 = q0 % TW;                // m-tile mt: rows ty0 + mt * (32 / TW)
    typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(wpk), 0, (int)((KS * KS * cin8 + 1) * p.cout * 16), 0x00020000);
    const unsigned wvoff = (unsigned)(hh * p.cout + n0 + r) * 16u;
    const unsigned wstep = (unsigned)p.cout * 16u;           // bytes per chunk row of the packed weights [tap][cin8][cout]
    auto ld_b = [&](int kk) {                                // B fragment of k-step kk = chunk rows 2 kk + hh
        return __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (unsigned)(2 * kk) * wstep, 0));
    };
    uint4 br[RD];
    if (live) {
#pragma unroll
        for (int f = 0; f < RD - 1; ++f) br[f] = ld_b(f);
    }

    unsigned lb[ST], kx[OFFS][OFFS];
#pragma unroll
    for (int px = 0; px < ST; ++px) lb[px] = (unsigned)(ty0 * G::pw(px) + tx0) * (CH * 16);
#pragma unroll
    for (int oy = 0; oy < OFFS; ++oy)
#pragma unroll
        for (int ox = 0; ox < OFFS; ++ox) kx[oy][ox] = ((unsigned)hh ^ key_of<CH, TW>(ty0 + oy, tx0 + ox)) << 4;
    const char *smb = reinterpret_cast<const char *>(patch_smem);
    auto load_a = [&](int kk, uint4 (&dst)[MT]) {
        const int tap = kk / KSTEPS, s = kk - tap * KSTEPS;
        const int dy = tap / KS, dx = tap - dy * KS;
        const int py = dy % ST, oy = dy / ST, px = dx % ST, ox = dx / ST;
        const unsigned a = lb[px] + ((((unsigned)(2 * s)) << 4) ^ kx[oy][ox]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned cst = (unsigned)(G::base(py * ST + px) + (oy + mt * (32 / TW)) * G::pw(px) + ox) * (CH * 16);
            dst[mt] = *reinterpret_cast<const uint4 *>(smb + a + cst);
        }
    };
    f32x16d acc[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    __syncthreads();                                         // footprint landed (hipcc drains the DMA before the barrier)

    if (live) {
    uint4 af[2][MT];
    load_a(0, af[0]);
#pragma unroll
#if defined(SEC_PATCH_ABL) && SEC_PATCH_ABL == 1     // ablation build: no K loop
    for (int kk = 0; kk < 1; ++kk) {
#else
    for (int kk = 0; kk < NK; ++kk) {
#endif
        if (kk + RD - 1 < NK) br[(kk + RD - 1) % RD] = ld_b(kk + RD - 1);
        if (kk + 1 < NK) load_a(kk + 1, af[(kk + 1) & 1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = MfmaD<T>::run(br[kk % RD], af[kk & 1][mt], acc[mt]);    // D^T: see store_tile_t
        __builtin_amdgcn_sched_barrier(0);                   // pins [B prefetch, A reads, MFMAs] per k-step
    }
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (mtb + mt) * 32 + r;
        const int oy = y0 + q / TW, ox = x0 + q % TW;
#if defined(SEC_PATCH_ABL) && SEC_PATCH_ABL == 3     // ablation build: no output stores
        const bool ok = oy < p.ho && ox < p.wo && acc[mt][0] == 1.2345f;
#else
        const bool ok = oy < p.ho && ox < p.wo;
#endif
        T *ypix = y + (((size_t)b * p.ho + oy) * p.wo + ox) * (size_t)ldc;
        store_tile_t<T>(acc[mt], bias, n0, p.relu, ypix, ok, hh);
    }
}

template <typename T, int CIN, int KS, int ST, int TH, int TW, int PXS, bool ROWS = false, bool TABLE = false>
static int launch(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, int ldc, hipStream_t st,
                  const int *site_map = nullptr, unsigned feat_bytes = 0) {
    using G = Geom<KS, ST, TH, TW>;
    constexpr size_t lds = (size_t)((G::NPIX * (CIN / 8) + 63) / 64) * 1024;
    static_assert(lds <= 160 * 1024, "one workgroup per CU at least");
    auto fn = k_conv2d_patch<T, CIN, KS, ST, TH, TW, PXS, ROWS, TABLE>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    const int ty = div_up(p.ho, TH), tx = div_up(p.wo, TW);
    const int per_xcd = div_up(p.batch * ty * tx, 8);
    set_last_kernel("k_conv2d_patch<%s, %d, %d, %d, %d, %d, %d%s>", dtype_name<T>(), CIN, KS, ST, TH, TW, PXS, TABLE ? (ROWS ? ", true, true" : ", false, true") : (ROWS ? ", true" : ""));
    hipLaunchKernelGGL(fn, dim3(per_xcd * 8, p.cout / (128 / PXS)), dim3(256), lds, st, (const T *)x, (const T *)wpk, bias, (T *)y, p, ty, tx,
                       per_xcd, ldc, site_map, feat_bytes);
    return check_launch();
}

// more workgroups than two rounds of the chip's slots (two per CU): the TABLE form's case
static bool several_rounds(const Conv2dParams &p, int th, int tw, int cout_per_wg) {
    return (long long)p.batch * div_up(p.ho, th) * div_up(p.wo, tw) * (p.cout / cout_per_wg) > 2 * 512;
}

// The layers this form takes (everything else stays on the generic kernel); returns kNotTaken when the shape is not one of them.
constexpr int kNotTaken = 0x7fffffff;
template <typename T>
static int dispatch(const void *x, const void *wpk, const float *bias, void *y, const Conv2dParams &p, int ldc, hipStream_t st) {
    const bool c128 = p.cout % 128 == 0;
    if ((long long)p.h * p.w * p.cin * 2 > 0x7fffffffll) return kNotTaken;      // a frame is one buffer resource: 32-bit offsets
    if (p.ksize == 3 && p.stride == 2 && p.pad == 1) {
        // (far larger maps than any PointPillars config of the reference -- 4 x 800 x 800: 5200 workgroups -- are better off on the generic
        // implicit GEMM, 137 vs 166 us: a 74 KB footprint per 128 x 64 outputs is a lot of LDS fill when nothing hides it; r06 A/B)
        if (p.cin == 64 && p.cout == 64 && (long long)p.batch * div_up(p.ho, 8) * div_up(p.wo, 16) <= 2560)
            return launch<T, 64, 3, 2, 8, 16, 2>(x, wpk, bias, y, p, ldc, st);
        if (p.cin == 64 && c128) return launch<T, 64, 3, 2, 8, 16, 1>(x, wpk, bias, y, p, ldc, st);
        if (p.cin == 128 && c128) return launch<T, 128, 3, 2, 4, 16, 1>(x, wpk, bias, y, p, ldc, st);
    }
    // stride-1 3x3 layers of the small PointPillars maps (batch 4: one round of workgroups, so a layer's time is one workgroup's time):
    // 256 channels at 50 x 50 and 64 -> 64 at 200 x 200 run 15 % / 12 % faster here than on k_conv2d_halo_reg's two-stage loop (steady
    // 8-deep B ring instead of 4-fragment double buffering; r06_pp3 / r06_pp4: 22.3 -> 19.0 us and 22.5 -> 18.6 us); the 128-channel
    // shared-row loop of k_conv2d_halo_reg stays ahead of this form (17.0 vs 18.3 us at 100 x 100) and keeps its layers.
    if (p.ksize == 3 && p.stride == 1 && p.pad == 1) {
        // (larger maps -- several rounds of workgroups, config 5's 124 x 124 -- stay on k_conv2d_halo_reg: -0.5 % on nusc.fhd with this form, r06_nusc_ab)
        if (p.cin == 64 && p.cout == 64 && !several_rounds(p, 16, 16, 64)) return launch<T, 64, 3, 1, 16, 16, 2>(x, wpk, bias, y, p, ldc, st);
        if (p.cin == 256 && c128 && !several_rounds(p, 4, 16, 128)) return launch<T, 256, 3, 1, 4, 16, 1>(x, wpk, bias, y, p, ldc, st);
#ifdef SEC_PATCH_S1     // experiment builds: the 128-channel layers on this form too (A/B against k_conv2d_halo_reg)
        if (p.cin == 128 && c128) return launch<T, 128, 3, 1, 8, 16, 1>(x, wpk, bias, y, p, ldc, st);
#endif
    }
    if (p.ksize == 4 && p.stride == 4 && p.pad == 0 && p.cin == 64 && c128) return launch<T, 64, 4, 4, 2, 16, 1>(x, wpk, bias, y, p, ldc, st);
    if (p.ksize == 2 && p.stride == 2 && p.pad == 0 && p.cin == 128 && c128)      // config 5's deblock (248 -> 124): 64-pixel tiles, half the weight re-reads
        return several_rounds(p, 2, 16, 128) ? launch<T, 128, 2, 2, 4, 16, 1>(x, wpk, bias, y, p, ldc, st)
                                             : launch<T, 128, 2, 2, 2, 16, 1>(x, wpk, bias, y, p, ldc, st);
    if (p.ksize == 1 && p.stride == 1 && p.pad == 0 && c128) {
        if (p.cin == 256) return launch<T, 256, 1, 1, 2, 16, 1>(x, wpk, bias, y, p, ldc, st);
        if (p.cin == 384) return launch<T, 384, 1, 1, 2, 16, 1>(x, wpk, bias, y, p, ldc, st);
    }
    return kNotTaken;
}

// sec_conv2d_nhwc_rows: the shapes the row-gathering form exists for (the first conv of the PointPillars RPN: 64 pillar channels, 3x3 / s2 / p1)
template <typename T>
static int dispatch_rows(const void *rows, unsigned feat_bytes, const int *site_map, const void *wpk, const float *bias, void *y,
                         const Conv2dParams &p, hipStream_t st) {
    if (p.ksize == 3 && p.stride == 2 && p.pad == 1 && p.cin == 64 && p.cout == 64)
        return several_rounds(p, 8, 16, 64) ? launch<T, 64, 3, 2, 8, 16, 2, true, true>(rows, wpk, bias, y, p, p.cout, st, site_map, feat_bytes)
                                            : launch<T, 64, 3, 2, 8, 16, 2, true>(rows, wpk, bias, y, p, p.cout, st, site_map, feat_bytes);
    if (p.ksize == 3 && p.stride == 2 && p.pad == 1 && p.cin == 64 && p.cout % 128 == 0)
        return launch<T, 64, 3, 2, 8, 16, 1, true>(rows, wpk, bias, y, p, p.cout, st, site_map, feat_bytes);
    return kNotTaken;
}

}  // namespace patch
