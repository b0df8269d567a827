// Sparse-convolution rulebooks on gfx950 (reference: spconv.ops.get_indice_pairs, CPU semantics of
// spconv include/spconv/indice.h + geometry.h -- SURVEY.md Appendix A.4; caller side
// second/pytorch/models/middle.py:146-189).
//
// The reference's GPU kernels number outputs by a device sort and append pairs in atomic-arrival order
// (nondeterministic).  Here everything is deterministic and equal to the sequential CPU algorithm:
//   * coordinates live in an open-addressing hash table (400 KB for a KITTI frame: L2 resident) instead of
//     the reference's dense 369 MB-per-sample grid;
//   * SubM: one lookup per (site, kernel offset) writes the output-major gather table nbr_out[N][K];
//   * strided conv: every (input, offset) candidate inserts its output cell with atomicMin(token),
//     token = input*K + offset = position in the sequential loop, so "first touch" numbering is the rank
//     of the winning token, obtained with a flag + exclusive scan (no sort);
//   * spconv-format pair lists are produced by wave-ballot/prefix-sum compaction of the table columns
//     (ascending input row inside each offset == the CPU order).
#include "common.hpp"

namespace sec {

constexpr int kMaxKvol = 128;

struct RbGeom {
    int in_shape[3], out_shape[3], ksize[3], stride[3], pad[3], dil[3];
    int kvol, n_in, batch;
    int cand[3], ncand;   // strided conv: max kernel offsets per dim that can hit an output, and their product
    uint32_t mask;
};

__device__ __forceinline__ unsigned long long cell_key(int b, int z, int y, int x, const int *shape) {
    unsigned long long vol = (unsigned long long)shape[0] * shape[1] * shape[2];
    return (unsigned long long)b * vol + ((unsigned long long)z * shape[1] + y) * shape[2] + x;
}

// Every kernel takes an optional device-side row count `n_dev` (static-capacity, sync-free pipelines):
// buffers are sized for g.n_in rows, only the first *n_dev are live.
__device__ __forceinline__ int live_rows(const RbGeom &g, const int *n_dev) {
    if (!n_dev) return g.n_in;
    int n = *n_dev;
    return n < g.n_in ? n : g.n_in;
}

// one launch instead of up to three hipMemsetAsync nodes (each costs a ~5 us launch): 64-bit words a := va,
// 32-bit words b := vb, c := vc
__global__ __launch_bounds__(kBlock) void k_rb_init(unsigned long long *__restrict__ a, long long na, unsigned long long va,
                                                   int *__restrict__ b, long long nb, int vb, int *__restrict__ c,
                                                   long long nc, int vc, int *__restrict__ d, long long nd) {
    long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < na; i += stride) a[i] = va;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nb; i += stride) b[i] = vb;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nc; i += stride) c[i] = vc;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nd; i += stride) d[i] = -1;
}
static void rb_init(unsigned long long *a, long long na, unsigned long long va, int *b, long long nb, int vb, int *c,
                    long long nc, int vc, hipStream_t st, int *d = nullptr, long long nd = 0) {
    long long m = na > nb ? na : nb;
    if (nc > m) m = nc;
    if (nd > m) m = nd;
    int blocks = div_up(m > 0 ? m : 1, kBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_rb_init, dim3(blocks), dim3(kBlock), 0, st, a, na, va, b, nb, vb, c, nc, vc, d, nd);
}

__global__ __launch_bounds__(kBlock) void k_rb_hash_rows(const int *__restrict__ indices, RbGeom g,
                                                        const int *__restrict__ n_dev,
                                                        unsigned long long *__restrict__ keys,
                                                        int *__restrict__ vals) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= live_rows(g, n_dev)) return;
    int4 c = *reinterpret_cast<const int4 *>(indices + (size_t)i * 4);
    uint32_t s = hash_insert(keys, g.mask, cell_key(c.x, c.y, c.z, c.w, g.in_shape));
    vals[s] = i;
}

// nbr_out[o][k] = row of the site at coord(o) + (k - centre) * dilation, or -1
__global__ __launch_bounds__(kBlock) void k_subm_nbr(const int *__restrict__ indices, RbGeom g,
                                                    const int *__restrict__ n_dev,
                                                    const unsigned long long *__restrict__ keys,
                                                    const int *__restrict__ vals, int *__restrict__ nbr) {
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)live_rows(g, n_dev) * g.kvol) return;
    int o = (int)(t / g.kvol), k = (int)(t % g.kvol);
    int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
    int4 c = *reinterpret_cast<const int4 *>(indices + (size_t)o * 4);
    int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
    int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
    int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
    int r = -1;
    if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) {
        int s = hash_find(keys, g.mask, cell_key(c.x, z, y, x, g.in_shape));
        if (s >= 0) r = vals[s];
    }
    nbr[t] = r;
}

// Symmetric form (odd kernels, any dilation): offset k and its mirror K-1-k describe the same pair of sites, so
// only the lower half of the offsets is looked up; a hit (i, j) at k fills nbr[i][k] = j and nbr[j][K-1-k] = i.
// The table must be pre-filled with -1; the centre column is the identity.
template <bool K3>     // K3: 3x3x3 kernel (every SubM layer of the reference): constant divisors
__global__ __launch_bounds__(kBlock) void k_subm_nbr_sym(const int *__restrict__ indices, RbGeom g,
                                                        const int *__restrict__ n_dev,
                                                        const unsigned long long *__restrict__ keys,
                                                        const int *__restrict__ vals, int *__restrict__ nbr) {
    if (K3) { g.kvol = 27; g.ksize[0] = g.ksize[1] = g.ksize[2] = 3; }
    const int half = K3 ? 13 : g.kvol / 2;            // offsets 0..half-1 are looked up, `half` is the centre
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)live_rows(g, n_dev) * (half + 1)) return;
    int o = (int)(t / (half + 1)), k = (int)(t % (half + 1));
    if (k == half) { nbr[(size_t)o * g.kvol + half] = o; return; }
    int kx = K3 ? k % 3 : k % g.ksize[2], ky = K3 ? (k / 3) % 3 : (k / g.ksize[2]) % g.ksize[1],
        kz = K3 ? k / 9 : k / (g.ksize[2] * g.ksize[1]);
    int4 c = *reinterpret_cast<const int4 *>(indices + (size_t)o * 4);
    int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
    int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
    int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
    if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) {
        int s = hash_find(keys, g.mask, cell_key(c.x, z, y, x, g.in_shape));
        if (s >= 0) {
            int j = vals[s];
            if ((unsigned)j < (unsigned)g.n_in) {   // a table inherited from an overflowed strided build may rank past the capacity
                nbr[(size_t)o * g.kvol + k] = j;
                nbr[(size_t)j * g.kvol + (g.kvol - 1 - k)] = o;
            }
        }
    }
}

// ---- strided / regular sparse conv -------------------------------------------------------------
// An input row reaches an output only through the kernel offsets k_d with k_d*dil_d == i_d + pad_d (mod stride_d):
// at most cand[d] = ceil(ksize/stride) of them per dim for dil = 1 (8 of 27 for the 3x3x3 stride-2 layers).  Candidate
// slot c = (cz, cy, cx) enumerates the cz-th / cy-th / cx-th such offset, so offsets grow with c and "earlier
// in the sequential reference loop" == "smaller c" inside a row.
__device__ __forceinline__ bool cand_offset(const RbGeom &g, int d, int in, int c, int *k_out, int *o_out) {
    int seen = 0;
    for (int k = 0; k < g.ksize[d]; ++k) {
        int num = in + g.pad[d] - k * g.dil[d];
        if (num % g.stride[d] != 0) continue;
        if (seen++ == c) {
            int o = num / g.stride[d];
            *k_out = k;
            *o_out = o;
            return num >= 0 && o < g.out_shape[d];
        }
    }
    return false;
}

// Compile-time geometries of the layers SECOND actually builds (middle.py:152-188): the generic kernels spend most of
// their instructions on runtime integer divisions (t / ncand, c / cand, num % stride: ~40 VALU instructions each, several
// per thread) and cannot unroll their per-candidate loops, so every candidate's two dependent loads are a round trip of
// their own.  GEO 0 = any geometry (runtime), 1 = 3x3x3 stride 2 dilation 1, 2 = (3,1,1) stride (2,1,1) dilation 1.
template <int GEO> struct Geo { static constexpr bool fixed = false; static constexpr int NCAND = 32, KVOL = 0; };
template <> struct Geo<1> { static constexpr bool fixed = true; static constexpr int NCAND = 8, KVOL = 27, C1 = 2, C2 = 2, K1 = 3, K2 = 3; };
template <> struct Geo<2> { static constexpr bool fixed = true; static constexpr int NCAND = 2, KVOL = 3, C1 = 1, C2 = 1, K1 = 1, K2 = 1; };
static int geo_of(const RbGeom &g) {
    bool d1 = g.dil[0] == 1 && g.dil[1] == 1 && g.dil[2] == 1;
    if (d1 && g.ksize[0] == 3 && g.ksize[1] == 3 && g.ksize[2] == 3 && g.stride[0] == 2 && g.stride[1] == 2 && g.stride[2] == 2) return 1;
    if (d1 && g.ksize[0] == 3 && g.ksize[1] == 1 && g.ksize[2] == 1 && g.stride[0] == 2 && g.stride[1] == 1 && g.stride[2] == 1) return 2;
    return 0;
}
// c-th kernel offset of one dimension through which input coordinate `in` reaches an output (same enumeration order as
// cand_offset): kernel 3 / stride 2: parity even -> k in {0, 2}, odd -> k = 1; kernel 1 / stride 1: k = 0.
template <int KS, int ST>
__device__ __forceinline__ bool cand_offset_fixed(int in, int pad, int out_dim, int c, int *k_out, int *o_out) {
    if (KS == 1 && ST == 1) { *k_out = 0; *o_out = in + pad; return c == 0 && in + pad >= 0 && in + pad < out_dim; }
    const int v = in + pad;
    const int k = (v & 1) ? 1 : 2 * c;
    if ((v & 1) && c > 0) return false;
    const int num = v - k;
    *k_out = k;
    *o_out = num >> 1;
    return num >= 0 && (num >> 1) < out_dim;
}

template <int GEO>
__global__ __launch_bounds__(kBlock) void k_conv_cand(const int *__restrict__ indices, RbGeom g,
                                                     const int *__restrict__ n_dev,
                                                     unsigned long long *__restrict__ keys,
                                                     int *__restrict__ vals, int *__restrict__ cand_slot,
                                                     unsigned char *__restrict__ cand_k, int *__restrict__ overflow) {
    using G = Geo<GEO>;
    const int ncand = G::fixed ? G::NCAND : g.ncand;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.n_in * ncand) return;
    int j = (int)(t / ncand), c = (int)(t % ncand);
    if (j >= live_rows(g, n_dev)) { cand_slot[t] = -1; return; }
    int4 q = *reinterpret_cast<const int4 *>(indices + (size_t)j * 4);
    int in[3] = {q.y, q.z, q.w}, out[3], kk[3];
    bool ok = true;
    int k = 0;
    if constexpr (G::fixed) {
        const int c0 = c / (G::C1 * G::C2), c1 = (c / G::C2) % G::C1, c2 = c % G::C2;
        ok &= cand_offset_fixed<3, 2>(in[0], g.pad[0], g.out_shape[0], c0, &kk[0], &out[0]);
        ok &= cand_offset_fixed<G::K1, G::K1 == 3 ? 2 : 1>(in[1], g.pad[1], g.out_shape[1], c1, &kk[1], &out[1]);
        ok &= cand_offset_fixed<G::K2, G::K2 == 3 ? 2 : 1>(in[2], g.pad[2], g.out_shape[2], c2, &kk[2], &out[2]);
        k = (kk[0] * G::K1 + kk[1]) * G::K2 + kk[2];
    } else {
        int cc[3] = {c / (g.cand[2] * g.cand[1]), (c / g.cand[2]) % g.cand[1], c % g.cand[2]};
#pragma unroll
        for (int d = 0; d < 3; ++d) ok &= cand_offset(g, d, in[d], cc[d], &kk[d], &out[d]);
        k = (kk[0] * g.ksize[1] + kk[1]) * g.ksize[2] + kk[2];
    }
    int s = -1;
    if (ok) {
        s = hash_insert_bounded(keys, g.mask, cell_key(q.x, out[0], out[1], out[2], g.out_shape));
        if (s >= 0) atomicMin(&vals[s], j * g.kvol + k);  // token = position in the sequential reference loop (a peek before
                                                          // this non-returning atomic was measured: no gain)
        else atomicOr(overflow, 1);                        // table sized from a too-small hint: reported, not hung
        cand_k[t] = (unsigned char)k;
    }
    cand_slot[t] = s;
}

// Per input row: how many of its candidates are the FIRST touch of their output cell, and -- in the same launch -- the
// exclusive prefix sum of those counts over the rows (rank of the row's first new output) by the single-pass scan of
// common.hpp: replaces a count kernel + three scan launches.
__global__ __launch_bounds__(kBlock) void k_conv_count_scan(const int *__restrict__ cand_slot,
                                                           const unsigned char *__restrict__ cand_k,
                                                           const int *__restrict__ vals, int n_in, int kvol, int ncand,
                                                           int *__restrict__ rank, unsigned *__restrict__ first_mask,
                                                           unsigned long long *__restrict__ status,
                                                           int *__restrict__ ticket, int *__restrict__ total_out) {
    __shared__ int smem[5];
    __shared__ int s_tile;
    const int tile = scan_take_tile(ticket, &s_tile);
    const int j = tile * kBlock + threadIdx.x;
    int cnt = 0;
    unsigned m = 0;   // bit c = candidate (j, c) is a first touch (ncand <= 32; larger ones recount in k_conv_assign)
    if (j < n_in) {
        for (int c = 0; c < ncand; ++c) {
            size_t t = (size_t)j * ncand + c;
            int s = cand_slot[t];
            bool f = s >= 0 && vals[s] == j * kvol + (int)cand_k[t];
            cnt += f ? 1 : 0;
            if (f && c < 32) m |= 1u << c;
        }
        first_mask[j] = m;
    }
    const int ex = scan_lookback(cnt, tile, (int)gridDim.x, status, smem, total_out);
    if (j < n_in) rank[j] = ex;
}

// ncand <= 32: count + scan + ASSIGN in one launch.  Once the look-back has given a row the rank of its first new output,
// the row's own first-touch candidates (its mask bits) can be numbered on the spot: orank[slot] = rank + popcount of
// the lower mask bits, out_indices[rank] = the cell decoded from the hash key.  The same launch pre-fills the gather
// tables with -1 when the caller already has them (static-capacity pipelines), so a strided build is
// init -> candidates -> this kernel -> tables: four launches instead of eight.
template <int GEO>
__global__ __launch_bounds__(kBlock) void k_conv_count_scan_assign(const int *__restrict__ cand_slot,
                                                                  const unsigned char *__restrict__ cand_k,
                                                                  const int *__restrict__ vals,
                                                                  const unsigned long long *__restrict__ keys, RbGeom g,
                                                                  int *__restrict__ orank, int *__restrict__ out_indices,
                                                                  int out_cap, unsigned long long *__restrict__ status,
                                                                  int *__restrict__ ticket, int *__restrict__ num_out,
                                                                  const int *__restrict__ overflow,
                                                                  int *__restrict__ fill_a, long long fill_a_words,
                                                                  int *__restrict__ fill_b, long long fill_b_words) {
    __shared__ int smem[5];
    __shared__ int s_tile;
    {   // table pre-fill (independent of everything else in this launch)
        const long long stride = (long long)gridDim.x * kBlock;
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < fill_a_words; i += stride) fill_a[i] = -1;
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < fill_b_words; i += stride) fill_b[i] = -1;
    }
    using G = Geo<GEO>;
    const int tile = scan_take_tile(ticket, &s_tile);
    const int j = tile * kBlock + threadIdx.x;
    const int n_in = g.n_in, ncand = G::fixed ? G::NCAND : g.ncand;
    int cnt = 0;
    unsigned m = 0;
    if (j < n_in) {
        if constexpr (G::fixed) {   // all candidate slots, then all their tokens: two load rounds instead of 2 x NCAND dependent ones
            int sl[G::NCAND], tk[G::NCAND], kk[G::NCAND];
#pragma unroll
            for (int c = 0; c < G::NCAND; ++c) {
                sl[c] = cand_slot[(size_t)j * G::NCAND + c];
                kk[c] = sl[c] >= 0 ? (int)cand_k[(size_t)j * G::NCAND + c] : 0;
            }
#pragma unroll
            for (int c = 0; c < G::NCAND; ++c) tk[c] = sl[c] >= 0 ? vals[sl[c]] : -1;
#pragma unroll
            for (int c = 0; c < G::NCAND; ++c) {
                const bool f = sl[c] >= 0 && tk[c] == j * G::KVOL + kk[c];
                cnt += f ? 1 : 0;
                if (f) m |= 1u << c;
            }
        } else {
            for (int c = 0; c < ncand; ++c) {
                const size_t t = (size_t)j * ncand + c;
                const int s = cand_slot[t];
                const bool f = s >= 0 && vals[s] == j * g.kvol + (int)cand_k[t];
                cnt += f ? 1 : 0;
                if (f) m |= 1u << c;
            }
        }
    }
    int r = scan_lookback(cnt, tile, (int)gridDim.x, status, smem, num_out);
    if (tile == (int)gridDim.x - 1 && threadIdx.x == 0) {
        // num_out[0] = live outputs (clamped to the capacity), num_out[1] = raw count (overflow check)
        const int tot = num_out[0];
        num_out[1] = *overflow ? 0x7fffffff : tot;
        if (tot > out_cap) num_out[0] = out_cap;
    }
    const unsigned long long vol = (unsigned long long)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
    const bool small = (vol * (unsigned long long)g.batch) >> 32 == 0;
    while (m) {                                   // this row's first touches, in offset order
        const int c = __ffs((int)m) - 1;
        m &= m - 1u;
        const int s = cand_slot[(size_t)j * ncand + c];
        orank[s] = r;
        if (r < out_cap) {
            const unsigned long long key = keys[s];
            int4 c4;
            if (small) {                          // the whole grid indexes in 32 bits: avoid the 64-bit divisions
                const unsigned k32 = (unsigned)key, v32 = (unsigned)vol;
                const unsigned b = k32 / v32, lin = k32 - b * v32;
                const unsigned q = lin / (unsigned)g.out_shape[2];
                c4 = make_int4((int)b, (int)(q / (unsigned)g.out_shape[1]), (int)(q % (unsigned)g.out_shape[1]),
                               (int)(lin - q * (unsigned)g.out_shape[2]));
            } else {
                const int b = (int)(key / vol);
                const unsigned long long lin = key - (unsigned long long)b * vol;
                const unsigned long long q = lin / g.out_shape[2];
                c4 = make_int4(b, (int)(q / g.out_shape[1]), (int)(q % g.out_shape[1]), (int)(lin % g.out_shape[2]));
            }
            *reinterpret_cast<int4 *>(out_indices + (size_t)r * 4) = c4;
        }
        ++r;
    }
}


__global__ __launch_bounds__(kBlock) void k_conv_assign(const int *__restrict__ cand_slot,
                                                       const unsigned char *__restrict__ cand_k,
                                                       const int *__restrict__ vals,
                                                       const unsigned long long *__restrict__ keys,
                                                       const int *__restrict__ rank, RbGeom g,
                                                       int *__restrict__ orank, int *__restrict__ out_indices,
                                                       int out_cap, int *__restrict__ num_out,
                                                       const int *__restrict__ overflow,
                                                       const unsigned *__restrict__ first_mask) {
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t == 0) {  // num_out[0] = live outputs (clamped to the capacity), num_out[1] = raw count (overflow check)
        int tot = num_out[0];
        num_out[1] = *overflow ? 0x7fffffff : tot;
        if (tot > out_cap) num_out[0] = out_cap;
    }
    if (t >= (long long)g.n_in * g.ncand) return;
    int s = cand_slot[t];
    const int j = (int)(t / g.ncand), c = (int)(t % g.ncand);
    if (s < 0 || vals[s] != j * g.kvol + (int)cand_k[t]) return;
    // rank = first touches of earlier rows (scan) + first touches of this row at smaller offsets
    int r = rank[j];
    if (g.ncand <= 32) {
        r += __popc(first_mask[j] & ((1u << c) - 1u));
    } else {
        for (int c2 = 0; c2 < c; ++c2) {
            size_t t2 = (size_t)j * g.ncand + c2;
            int s2 = cand_slot[t2];
            r += (s2 >= 0 && vals[s2] == j * g.kvol + (int)cand_k[t2]) ? 1 : 0;
        }
    }
    orank[s] = r;
    if (r < out_cap) {
        unsigned long long vol = (unsigned long long)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
        unsigned long long key = keys[s];
        int b = (int)(key / vol);
        unsigned long long lin = key - (unsigned long long)b * vol;
        int x = (int)(lin % g.out_shape[2]);
        unsigned long long q = lin / g.out_shape[2];
        int4 c4 = make_int4(b, (int)(q / g.out_shape[1]), (int)(q % g.out_shape[1]), x);
        *reinterpret_cast<int4 *>(out_indices + (size_t)r * 4) = c4;
    }
}

// nbr_out[o][k] = j (pre-filled with -1); nbr_in[j][k] = o only when requested (backward / pair lists; pre-filled too)
template <int GEO>
__global__ __launch_bounds__(kBlock) void k_conv_tables(const int *__restrict__ cand_slot,
                                                       const unsigned char *__restrict__ cand_k,
                                                       const int *__restrict__ orank, long long n, int kvol, int ncand,
                                                       int *__restrict__ nbr_in, int *__restrict__ nbr_out,
                                                       int nbr_out_rows) {
    using G = Geo<GEO>;
    if (G::fixed) { kvol = G::KVOL; ncand = G::NCAND; }
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    int s = cand_slot[t];
    if (s < 0) return;
    int o = orank[s], k = cand_k[t], j = (int)(t / ncand);
    if (nbr_in) nbr_in[(size_t)j * kvol + k] = o;
    if (o < nbr_out_rows) nbr_out[(size_t)o * kvol + k] = j;
}

// ---- spconv-format pair lists by ballot/prefix compaction of table columns -----------------------
// table is input-major [n][K]; pair list k = rows j (ascending) with table[j][col(k)] >= 0,
// col(k) = K-1-k when `mirror` (SubM: nbr_in is the mirror image of nbr_out), else k.
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_pairs(const int *__restrict__ table, int n, int kvol, int mirror,
                                                 int nblocks, int *__restrict__ blk, int *__restrict__ pairs,
                                                 int *__restrict__ pair_num) {
    __shared__ int wcnt[4 * kMaxKvol];
    int j = blockIdx.x * kBlock + threadIdx.x;
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int *row = table + (size_t)j * kvol;
    if (!WRITE) {
        for (int k = 0; k < kvol; ++k) {
            int v = j < n ? row[mirror ? kvol - 1 - k : k] : -1;
            unsigned long long m = __ballot(v >= 0);
            if (lane == 0) wcnt[w * kvol + k] = __popcll(m);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < kvol; k += kBlock)
            blk[(size_t)k * nblocks + blockIdx.x] = wcnt[k] + wcnt[kvol + k] + wcnt[2 * kvol + k] + wcnt[3 * kvol + k];
    } else {
        // pass 1: per-wave counts (again), pass 2: positions
        for (int k = 0; k < kvol; ++k) {
            int v = j < n ? row[mirror ? kvol - 1 - k : k] : -1;
            unsigned long long m = __ballot(v >= 0);
            if (lane == 0) wcnt[w * kvol + k] = __popcll(m);
        }
        __syncthreads();
        for (int k = 0; k < kvol; ++k) {
            int v = j < n ? row[mirror ? kvol - 1 - k : k] : -1;
            unsigned long long m = __ballot(v >= 0);
            if (v >= 0) {
                int pos = blk[(size_t)k * nblocks + blockIdx.x] - blk[(size_t)k * nblocks];
                for (int ww = 0; ww < w; ++ww) pos += wcnt[ww * kvol + k];
                pos += __popcll(m & ((1ull << lane) - 1ull));
                pairs[((size_t)k * 2 + 0) * n + pos] = j;
                pairs[((size_t)k * 2 + 1) * n + pos] = v;
            }
        }
        if (blockIdx.x == 0)
            for (int k = threadIdx.x; k < kvol; k += kBlock)
                pair_num[k] = blk[(size_t)(k + 1) * nblocks] - blk[(size_t)k * nblocks];
    }
}

static int emit_pairs(const int *table, int n, int kvol, int mirror, int *blk, int *scan_scratch, int *pairs,
                      int *pair_num, hipStream_t st) {
    int rc;
    if (n == 0) return hip_ok(hipMemsetAsync(pair_num, 0, kvol * sizeof(int), st));
    if ((rc = hip_ok(hipMemsetAsync(pairs, 0xff, (size_t)kvol * 2 * n * sizeof(int), st)))) return rc;
    int nblocks = div_up(n, kBlock);
    hipLaunchKernelGGL(k_pairs<false>, dim3(nblocks), dim3(kBlock), 0, st, table, n, kvol, mirror, nblocks, blk,
                       (int *)nullptr, (int *)nullptr);
    long long cnt = (long long)kvol * nblocks;
    if ((rc = exclusive_scan_i32(blk, blk, cnt, blk + cnt, scan_scratch, st))) return rc;
    hipLaunchKernelGGL(k_pairs<true>, dim3(nblocks), dim3(kBlock), 0, st, table, n, kvol, mirror, nblocks, blk,
                       pairs, pair_num);
    return check_launch();
}


// ---- "sorted" output numbering: spconv's GPU path -----------------------------------------------------
// Outputs are numbered by ascending linear cell index lin = ((b * D + z) * H + y) * W + x instead of by first touch.  That
// order needs no hash table and no winner election: a bitmap over the output grid (one bit per cell: 11.8 MB for the first
// strided layer of car.fhd at batch 8) is filled, ONE single-pass scan turns the word popcounts into ranks, and a cell's
// output row is prefix[word] + popc(bits below).  Filling the bitmap:
//   * from the input rows (k_bm_set*): atomicOr per candidate cell -- device-scope atomics retire at ~20 G/s on this chip, so
//     the 3x3x3 stride-2 form handles the two x candidates of an input (adjacent cells) with one atomic;
//   * from the INPUT sites' bitmap when the inputs are themselves the outputs of a sorted build (k_bm_dilate): an output word
//     is the OR over the <= 9 (z, y) input rows of a 65-bit input window, every second bit kept -- plain loads and stores, no atomics.
// Pair order inside an offset stays ascending input row (emit_pairs), where the reference's GPU path has atomic-arrival order.
constexpr int kBmWpt = 64;                       // bitmap words per thread of the scan of a LARGE grid (few tiles -> short look-back chain)
constexpr int kBmWptSmall = 16;                  // ... of a small one (the 64-word form has an ~8 us floor of serial per-thread work)
constexpr int kBmTile = kBlock * kBmWpt;         // words per scan tile = padding unit of the bitmap
constexpr long long kBmSmallWords = 1 << 20;     // grids up to this many words take the 16-word form (<= 256 tiles)

template <int GEO>
__device__ __forceinline__ bool bm_candidate(const RbGeom &g, int4 q, int c, int *k_out, unsigned *lin_out, int *out) {
    using G = Geo<GEO>;
    int in[3] = {q.y, q.z, q.w}, kk[3];
    bool ok = true;
    int k = 0;
    if constexpr (G::fixed) {
        const int c0 = c / (G::C1 * G::C2), c1 = (c / G::C2) % G::C1, c2 = c % G::C2;
        ok &= cand_offset_fixed<3, 2>(in[0], g.pad[0], g.out_shape[0], c0, &kk[0], &out[0]);
        ok &= cand_offset_fixed<G::K1, G::K1 == 3 ? 2 : 1>(in[1], g.pad[1], g.out_shape[1], c1, &kk[1], &out[1]);
        ok &= cand_offset_fixed<G::K2, G::K2 == 3 ? 2 : 1>(in[2], g.pad[2], g.out_shape[2], c2, &kk[2], &out[2]);
        k = (kk[0] * G::K1 + kk[1]) * G::K2 + kk[2];
    } else {
        int cc[3] = {c / (g.cand[2] * g.cand[1]), (c / g.cand[2]) % g.cand[1], c % g.cand[2]};
#pragma unroll
        for (int d = 0; d < 3; ++d) ok &= cand_offset(g, d, in[d], cc[d], &kk[d], &out[d]);
        k = (kk[0] * g.ksize[1] + kk[1]) * g.ksize[2] + kk[2];
    }
    *k_out = k;
    *lin_out = ok ? (((unsigned)q.x * g.out_shape[0] + out[0]) * g.out_shape[1] + out[1]) * g.out_shape[2] + out[2] : 0u;
    return ok;
}

template <int GEO>
__global__ __launch_bounds__(kBlock) void k_bm_set(const int *__restrict__ indices, RbGeom g, const int *__restrict__ n_dev,
                                                  unsigned *__restrict__ bm) {
    using G = Geo<GEO>;
    const int ncand = G::fixed ? G::NCAND : g.ncand;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.n_in * ncand) return;
    const int j = (int)(t / ncand), c = (int)(t % ncand);
    if (j >= live_rows(g, n_dev)) return;
    const int4 q = *reinterpret_cast<const int4 *>(indices + (size_t)j * 4);
    int k, out[3];
    unsigned lin;
    if (bm_candidate<GEO>(g, q, c, &k, &lin, out)) atomicOr(&bm[lin >> 5], 1u << (lin & 31u));
}

// 3x3x3 stride 2: one thread per (input row, z candidate, y candidate); its two x candidates are adjacent cells
__global__ __launch_bounds__(kBlock) void k_bm_set_x2(const int *__restrict__ indices, RbGeom g, const int *__restrict__ n_dev,
                                                     unsigned *__restrict__ bm) {
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.n_in * 4) return;
    const int j = (int)(t >> 2), c = (int)(t & 3);
    if (j >= live_rows(g, n_dev)) return;
    const int4 q = *reinterpret_cast<const int4 *>(indices + (size_t)j * 4);
    int kz, ky, oz, oy;
    if (!cand_offset_fixed<3, 2>(q.y, g.pad[0], g.out_shape[0], c >> 1, &kz, &oz)) return;
    if (!cand_offset_fixed<3, 2>(q.z, g.pad[1], g.out_shape[1], c & 1, &ky, &oy)) return;
    int kx, ox0, ox1;
    const bool v0 = cand_offset_fixed<3, 2>(q.w, g.pad[2], g.out_shape[2], 0, &kx, &ox0);
    const bool v1 = cand_offset_fixed<3, 2>(q.w, g.pad[2], g.out_shape[2], 1, &kx, &ox1);
    const unsigned rowb = (((unsigned)q.x * g.out_shape[0] + oz) * g.out_shape[1] + oy) * g.out_shape[2];
    const unsigned l0 = rowb + ox0, l1 = rowb + ox1;
    if (v0 && v1 && (l0 >> 5) == (l1 >> 5)) {
        atomicOr(&bm[l0 >> 5], (1u << (l0 & 31u)) | (1u << (l1 & 31u)));
    } else {
        if (v0) atomicOr(&bm[l0 >> 5], 1u << (l0 & 31u));
        if (v1) atomicOr(&bm[l1 >> 5], 1u << (l1 & 31u));
    }
}

// 64 bits of a row of the INPUT bitmap starting at x = s (s may be -1: the padding column), zero beyond the row; *b64 = bit s + 64
__device__ __forceinline__ unsigned long long bm_window(const unsigned *__restrict__ bm, long long n_words, unsigned row_base,
                                                        int s, int w_in, unsigned *b64) {
    const int s0 = s < 0 ? 0 : s;                       // first real column fetched
    const unsigned p = row_base + (unsigned)s0;
    long long w0 = p >> 5;
    if (w0 >= n_words) w0 = n_words - 1;               // only when the window starts beyond the last row (masked to zero below)
    const unsigned sh = p & 31u;
    const unsigned a0 = bm[w0], a1 = bm[w0 + 1 < n_words ? w0 + 1 : w0], a2 = bm[w0 + 2 < n_words ? w0 + 2 : w0];
    unsigned long long lo = ((unsigned long long)a1 << 32) | a0;
    unsigned long long v = sh ? (lo >> sh) | ((unsigned long long)a2 << (64 - sh)) : lo;     // 64 bits from column s0
    // bit 64 from column s0: bit (sh + 64) of the 96 fetched = bit (sh + 32) of (a2:a1) -- only needed when s0 == s
    unsigned top = (unsigned)((((unsigned long long)a2 << 32) | a1) >> sh >> 31 >> 1) & 1u;      // (a2:a1) >> (sh + 32), sh + 32 < 64
    const int left = w_in - s0;                        // columns of the row from s0 on
    if (left < 64) v &= left <= 0 ? 0ull : ((1ull << left) - 1ull);
    if (left <= 64) top = 0u;
    if (s < 0) {                                       // window starts one column before the row: shift in a zero
        top = (unsigned)(v >> 63);
        v <<= 1;
    }
    *b64 = top;
    return v;
}

__device__ __forceinline__ unsigned compress_even(unsigned long long x) {      // bit o of the result = bit 2 o of x
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return (unsigned)x;
}

// output bitmap from the input sites' bitmap.  GEO 1: 3x3x3 stride 2 (any padding); GEO 2: (3,1,1) stride (2,1,1).
// One thread per output word; a word that straddles a row end is assembled from its row segments.
// The launch also does what k_rb_init did for this build (scan control := 0, gather tables := -1): the dilation writes every
// bitmap word itself, so nothing has to be cleared BEFORE it, and the tables are only read by the launches after it -- one launch
// less per strided layer on the latency chain.
struct RbFill {
    unsigned long long *a; long long na;      // := 0
    int *b; long long nb;                     // := -1
    int *c; long long nc;                     // := -1
    int *d; long long nd;                     // := -1
};
// (A form with one input row per lane and a shuffle OR across 16-lane groups was measured in round 4: 34 us instead of 9 us on
// car.fhd's second level -- 16 x the threads, each still paying the address arithmetic -- and 7.8 instead of 8.6 us on the third.)
template <int GEO>
__device__ __forceinline__ unsigned bm_dilate_word(const unsigned *__restrict__ bm_in, long long n_words_in, const RbGeom &g, long long wi) {
    const unsigned Wo = (unsigned)g.out_shape[2], Ho = (unsigned)g.out_shape[1], Do = (unsigned)g.out_shape[0];
    const unsigned total = (unsigned)g.batch * Do * Ho * Wo;
    const unsigned lin0 = (unsigned)wi << 5;
    if (lin0 >= total) return 0u;
    unsigned row = lin0 / Wo, x = lin0 - row * Wo;     // row = (b * Do + oz) * Ho + oy
    unsigned word = 0u;
    int done = 0;
    while (done < 32 && row < (unsigned)g.batch * Do * Ho) {
        const int len = min(32 - done, (int)(Wo - x));
        const unsigned plane = row / Ho, oy = row - plane * Ho, b = plane / Do, oz = plane - b * Do;
        unsigned seg = 0u;
        if constexpr (GEO == 1) {
            unsigned long long acc = 0ull;
            unsigned acc64 = 0u;
            const int s = 2 * (int)x - g.pad[2];
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
                const int iz = 2 * (int)oz - g.pad[0] + kz;
                if (iz < 0 || iz >= g.in_shape[0]) continue;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = 2 * (int)oy - g.pad[1] + ky;
                    if (iy < 0 || iy >= g.in_shape[1]) continue;
                    const unsigned rb = ((b * g.in_shape[0] + iz) * g.in_shape[1] + iy) * g.in_shape[2];
                    unsigned t64;
                    acc |= bm_window(bm_in, n_words_in, rb, s, g.in_shape[2], &t64);
                    acc64 |= t64;
                }
            }
            const unsigned e = compress_even(acc), o1 = compress_even(acc >> 1);
            seg = e | o1 | (e >> 1) | (acc64 << 31);   // out bit o = in[2o] | in[2o+1] | in[2o+2] (window-relative)
        } else {                                         // (3,1,1) / (2,1,1): same (y, x), three input planes
            unsigned long long acc = 0ull;
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
                const int iz = 2 * (int)oz - g.pad[0] + kz;
                if (iz < 0 || iz >= g.in_shape[0]) continue;
                const int iy = (int)oy - g.pad[1], s = (int)x - g.pad[2];
                if (iy < 0 || iy >= g.in_shape[1]) continue;
                const unsigned rb = ((b * g.in_shape[0] + iz) * g.in_shape[1] + iy) * g.in_shape[2];
                unsigned t64;
                acc |= bm_window(bm_in, n_words_in, rb, s, g.in_shape[2], &t64);
            }
            seg = (unsigned)acc;
        }
        if (len < 32) seg &= (1u << len) - 1u;
        word |= seg << done;
        done += len;
        x = 0;
        ++row;
    }
    return word;
}
template <int GEO>
__global__ __launch_bounds__(kBlock) void k_bm_dilate(const unsigned *__restrict__ bm_in, long long n_words_in, RbGeom g,
                                                     unsigned *__restrict__ bm_out, long long n_words_out, RbFill f) {
    const long long wi = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (wi < n_words_out) bm_out[wi] = bm_dilate_word<GEO>(bm_in, n_words_in, g, wi);
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = wi; i < f.na; i += stride) f.a[i] = 0ull;
    for (long long i = wi; i < f.nb; i += stride) f.b[i] = -1;
    for (long long i = wi; i < f.nc; i += stride) f.c[i] = -1;
    for (long long i = wi; i < f.nd; i += stride) f.d[i] = -1;
}

// ranks: exclusive scan of the bitmap popcounts (decoupled look-back over tiles of kBlock * WPT words) -> prefix8[] = set bits
// before every BLOCK of kBmBlk = 8 words (one 32-byte sector of the bitmap); a cell's rank is prefix8[block] + the popcounts of
// the block's words before its own + the bits below it (bm_rank) -- the consumer's extra reads stay inside the sector that
// holds its word.  Round 2 stored one prefix per WORD: the scan wrote (and its consumers cached) an array as large as the
// bitmap itself, with each thread walking 256 contiguous bytes (every wave load touching 64 lines): 15 us for the 11.8 MB
// bitmap of car.fhd's first strided layer.  Here a thread counts blocks i * 256 + t (coalesced 32-byte loads), the counts are
// transposed through LDS so that thread t scans blocks t * BPT .. t * BPT + BPT - 1, and 1/8 of the bytes are written.
// num_out[0] = live outputs (clamped to out_cap), num_out[1] = raw count
constexpr int kBmBlk = 8;
__device__ __forceinline__ int popc4(const uint4 &v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
// one tile of the scan: `tile` of `ntiles` (of THIS bitmap), status words of this bitmap; cnt = LDS scratch of kBlock * WPT / 8 ints
template <int WPT>
__device__ __forceinline__ int bm_scan_tile(const unsigned *__restrict__ bm, int *__restrict__ prefix8, int out_cap,
                                             unsigned long long *__restrict__ status, int tile, int ntiles,
                                             int *__restrict__ num_out, int *smem, int *cnt) {
    constexpr int BPT = WPT / kBmBlk;                 // blocks per thread
    static_assert(WPT % kBmBlk == 0 && (BPT == 2 || BPT % 4 == 0), "prefix stores are int2 / int4");
    const int t = threadIdx.x;
    const size_t blk0 = (size_t)tile * kBlock * BPT;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bm + (blk0 + (size_t)i * kBlock + t) * kBmBlk);
        cnt[i * kBlock + t] = popc4(src[0]) + popc4(src[1]);
    }
    __syncthreads();
    int mine[BPT], v = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) { mine[k] = cnt[t * BPT + k]; v += mine[k]; }
    int r = scan_lookback(v, tile, ntiles, status, smem, num_out);
    const int r_first = r;                            // set bits before this thread's blocks (blk0 + t * BPT ...)
    if (tile == ntiles - 1 && t == 0) {
        const int tot = num_out[0];
        num_out[1] = tot;
        if (tot > out_cap) num_out[0] = out_cap;
    }
    int *dst = prefix8 + blk0 + (size_t)t * BPT;
    if constexpr (BPT == 2) {
        *reinterpret_cast<int2 *>(dst) = make_int2(r, r + mine[0]);
    } else {
#pragma unroll
        for (int k = 0; k < BPT; k += 4) {
            int4 pf;
            pf.x = r; r += mine[k];
            pf.y = r; r += mine[k + 1];
            pf.z = r; r += mine[k + 2];
            pf.w = r; r += mine[k + 3];
            *reinterpret_cast<int4 *>(dst + k) = pf;
        }
    }
    return r_first;
}
template <int WPT>
__global__ __launch_bounds__(kBlock) void k_bm_scan(const unsigned *__restrict__ bm, int *__restrict__ prefix8, int out_cap,
                                                   unsigned long long *__restrict__ status, int *__restrict__ ticket,
                                                   int *__restrict__ num_out) {
    __shared__ int smem[8];
    __shared__ int s_tile;
    __shared__ int cnt[kBlock * (WPT / kBmBlk)];
    const int tile = scan_take_tile(ticket, &s_tile);
    bm_scan_tile<WPT>(bm, prefix8, out_cap, status, tile, (int)gridDim.x, num_out, smem, cnt);
}

// set bits before word `w` of the bitmap: the block's prefix + the popcounts of the block's words in front of it
__device__ __forceinline__ int bm_word_prefix(const unsigned *__restrict__ bm, const int *__restrict__ prefix8, size_t w) {
    const size_t blk = w / kBmBlk;
    const unsigned wi = (unsigned)(w % kBmBlk);
    const uint4 a = *reinterpret_cast<const uint4 *>(bm + blk * kBmBlk), b = *reinterpret_cast<const uint4 *>(bm + blk * kBmBlk + 4);
    int r = prefix8[blk];
    r += wi > 0 ? __popc(a.x) : 0;
    r += wi > 1 ? __popc(a.y) : 0;
    r += wi > 2 ? __popc(a.z) : 0;
    r += wi > 3 ? __popc(a.w) : 0;
    r += wi > 4 ? __popc(b.x) : 0;
    r += wi > 5 ? __popc(b.y) : 0;
    r += wi > 6 ? __popc(b.z) : 0;
    return r;
}

// out_indices in rank order, one thread per bitmap word (for callers that want them before the tables exist; the tables
// kernel writes them too -- every candidate knows its output's coordinates)
__global__ __launch_bounds__(kBlock) void k_bm_emit(const unsigned *__restrict__ bm, const int *__restrict__ prefix8, RbGeom g,
                                                   long long n_words, int *__restrict__ out_indices, int out_cap) {
    const long long wi = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (wi >= n_words) return;
    unsigned word = bm[wi];
    if (!word) return;
    int r = bm_word_prefix(bm, prefix8, (size_t)wi);
    const unsigned W = (unsigned)g.out_shape[2], H = (unsigned)g.out_shape[1], D = (unsigned)g.out_shape[0];
    const unsigned lin0 = (unsigned)wi << 5;
    const unsigned row = lin0 / W;                   // (b * D + z) * H + y of the word's first cell
    const unsigned x0 = lin0 - row * W;
    while (word && r < out_cap) {
        const int bit = __ffs((int)word) - 1;
        word &= word - 1u;
        unsigned x = x0 + bit, rw = row;
        while (x >= W) { x -= W; ++rw; }             // a word may straddle row ends when W is not a multiple of 32
        const unsigned plane = rw / H, y = rw - plane * H;
        const unsigned b = plane / D, z = plane - b * D;
        *reinterpret_cast<int4 *>(out_indices + (size_t)r * 4) = make_int4((int)b, (int)z, (int)y, (int)x);
        ++r;
    }
}

__device__ __forceinline__ int bm_rank(const unsigned *__restrict__ bm, const int *__restrict__ prefix8, unsigned lin) {
    const unsigned word = bm[lin >> 5], bit = 1u << (lin & 31u);
    if (!(word & bit)) return -1;
    return bm_word_prefix(bm, prefix8, (size_t)(lin >> 5)) + __popc(word & (bit - 1u));
}

// The same rank in ONE round of loads: the 32-byte block that holds the cell's word (two 16-byte loads of one sector) and the
// block's prefix are fetched together; bit test, popcounts of the words in front and of the bits below come out of registers.
// bm_rank needs two dependent rounds (word, then block + prefix): in the output-side table build of the fused chain a thread
// walks tens of (output, offset) items, each a rank lookup, and the dependent round doubled the latency of every item.
__device__ __forceinline__ int bm_rank1(const unsigned *__restrict__ bm, const int *__restrict__ prefix8, unsigned lin) {
    const size_t blk = lin >> 8;                                     // kBmBlk = 8 words = 256 cells
    const unsigned wi = (lin >> 5) & 7u, bit = 1u << (lin & 31u);
    const uint4 a = *reinterpret_cast<const uint4 *>(bm + blk * kBmBlk), b = *reinterpret_cast<const uint4 *>(bm + blk * kBmBlk + 4);
    int r = prefix8[blk];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned word = 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        word = wi == (unsigned)q ? w[q] : word;
        r += wi > (unsigned)q ? __popc(w[q]) : 0;
    }
    return (word & bit) ? r + __popc(word & (bit - 1u)) : -1;
}

// Site map of the outputs of a sorted build, straight from its bitmap: map[cell] = rank + 1 (the output row, rows are numbered
// by ascending cell index) or 0.  One launch that writes every cell -- the generic sec_sparse_site_map needs a zero fill and a
// scatter.
__global__ __launch_bounds__(kBlock) void k_bm_site_map(const unsigned *__restrict__ bm, const int *__restrict__ prefix8,
                                                       const int *__restrict__ num_dev, int rows_cap, long long cells,
                                                       int *__restrict__ map) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cells) return;
    int rows = num_dev ? *num_dev : rows_cap;
    if (rows > rows_cap) rows = rows_cap;
    const int r = bm_rank(bm, prefix8, (unsigned)i);
    map[i] = (r >= 0 && r < rows) ? r + 1 : 0;
}

template <int GEO>
__global__ __launch_bounds__(kBlock) void k_bm_tables(const int *__restrict__ indices, RbGeom g, const int *__restrict__ n_dev,
                                                     const unsigned *__restrict__ bm, const int *__restrict__ prefix,
                                                     int *__restrict__ nbr_in, int *__restrict__ nbr_out, int nbr_out_rows,
                                                     int *__restrict__ out_indices, int out_cap) {
    using G = Geo<GEO>;
    const int ncand = G::fixed ? G::NCAND : g.ncand, kvol = G::fixed ? G::KVOL : g.kvol;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.n_in * ncand) return;
    const int j = (int)(t / ncand), c = (int)(t % ncand);
    if (j >= live_rows(g, n_dev)) return;
    const int4 q = *reinterpret_cast<const int4 *>(indices + (size_t)j * 4);
    int k, out[3];
    unsigned lin;
    if (!bm_candidate<GEO>(g, q, c, &k, &lin, out)) return;
    const int o = bm_rank(bm, prefix, lin);
    if (o < 0) return;
    if (nbr_in) nbr_in[(size_t)j * kvol + k] = o;
    if (o < nbr_out_rows) nbr_out[(size_t)o * kvol + k] = j;
    // every candidate of an output writes the same four ints: no election needed
    if (out_indices && o < out_cap) *reinterpret_cast<int4 *>(out_indices + (size_t)o * 4) = make_int4(q.x, out[0], out[1], out[2]);
}

// SubM layer on the outputs of a sorted-numbering strided build: the site lookup is the bitmap rank
template <bool K3>
__global__ __launch_bounds__(kBlock) void k_subm_nbr_bm(const int *__restrict__ indices, RbGeom g, const int *__restrict__ n_dev,
                                                       const unsigned *__restrict__ bm, const int *__restrict__ prefix,
                                                       int *__restrict__ nbr) {
    if (K3) { g.kvol = 27; g.ksize[0] = g.ksize[1] = g.ksize[2] = 3; }
    const int half = K3 ? 13 : g.kvol / 2;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)live_rows(g, n_dev) * (half + 1)) return;
    int o = (int)(t / (half + 1)), k = (int)(t % (half + 1));
    if (k == half) { nbr[(size_t)o * g.kvol + half] = o; return; }
    int kx = K3 ? k % 3 : k % g.ksize[2], ky = K3 ? (k / 3) % 3 : (k / g.ksize[2]) % g.ksize[1],
        kz = K3 ? k / 9 : k / (g.ksize[2] * g.ksize[1]);
    int4 c = *reinterpret_cast<const int4 *>(indices + (size_t)o * 4);
    int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
    int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
    int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
    if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) {
        const unsigned lin = (((unsigned)c.x * g.in_shape[0] + z) * g.in_shape[1] + y) * g.in_shape[2] + x;
        const int j = bm_rank(bm, prefix, lin);
        if ((unsigned)j < (unsigned)g.n_in) {
            nbr[(size_t)o * g.kvol + k] = j;
            nbr[(size_t)j * g.kvol + (g.kvol - 1 - k)] = o;
        }
    }
}


// ---- the whole rulebook stack of a sparse middle in one call (sec_rulebook_chain_sorted) -----------------------------------------
// Eight rulebooks (four SubM, four strided) built layer by layer are ~25 dependent launches of 4-17 us each (round 3: 160 us of a
// 610 us step, 1-3 % of the HBM roofline): every launch pays the ~2 us boundary plus a near-empty grid.  But in the sorted
// numbering a level's OUTPUT SET is a pure function of the level below -- bitmap l+1 = dilate(bitmap l) -- and every gather table is
// a pure function of two adjacent bitmaps and their rank prefixes.  So the stack is built in phases, not layers:
//   front : SubM table of level 0 (probes of the voxeliser's hash table) || level-1 bitmap (atomicOr per voxel) || -1 fill of the
//           level-1 conv table -- three independent jobs in ONE launch (block ranges);
//   dilate: bitmap l from bitmap l-1 for l >= 2 (plain loads / stores);
//   scan  : rank prefixes of ALL levels in one launch (one ticket counter; the look-back chain restarts at every level);
//   tables: ALL conv and SubM tables, output coordinates and the BEV site map in one launch.  Levels >= 2 are built OUTPUT-side:
//           a workgroup takes 256 bitmap words, compacts their set bits into LDS (ranks are consecutive: first rank of the run +
//           position) and its threads walk (output, offset) items with the offset fastest -- every table entry is written, -1
//           included, in fully coalesced runs: no pre-fill, no scattered stores, no mirror trick.  Level 1 (inputs = voxels in
//           arrival order, findable only through the hash table) keeps the input-side form on a pre-filled table.
constexpr int kChainMaxLevels = 6;
struct ChainLevel {
    int shape[3];                     // D, H, W of this level's grid
    int ksize[3], pad[3];             // the strided conv that makes this level from the one below (level >= 1); stride = geo
    int geo, kvol;                    // 1: 3x3x3 stride 2, 2: (3,1,1) stride (2,1,1)
    int out_cap;                      // rows reserved for this level (level 0: capacity of the caller's rows)
    long long n_words;                // bitmap words (padded to the scan tile)
    unsigned *bm;
    int *prefix8;
    unsigned long long *status;       // scan status words of this level
    int tile0, ntiles, wpt64;         // scan tiles [tile0, tile0 + ntiles) of the scan launch; 64 or 16 words per thread
    int *nbr_out, *out_indices, *num_out, *subm_nbr;
    int sblk0, sblks, cblk0, cblks;   // tables launch: workgroups of the SubM table / the conv table of this level
    int eblk0, eblks;                 // emit launch: workgroups of this level (one thread per bitmap word)
};
struct ChainParams {
    ChainLevel lv[kChainMaxLevels + 1];
    int levels, batch;
    int n0;                           // capacity of the level-0 rows
    const int *indices0, *n0_dev;
    int *ticket;
    // tables launch: [0, cand_blks) level-1 candidates, then the word passes, then [map_blk0, ...) the site map
    int cand_blks, map_blk0;
    int *site_map;
    long long map_cells;
};

__device__ __forceinline__ int chain_live0(const ChainParams &P) {
    if (!P.n0_dev) return P.n0;
    const int n = *P.n0_dev;
    return n < P.n0 ? n : P.n0;
}

// front: [0, nb_subm) SubM level 0 through the voxel hash table (k_subm_nbr_sym<true>), [nb_subm, nb_subm + nb_set) level-1 bitmap
// (k_bm_set_x2), the rest: -1 fill of the level-1 conv table
__global__ __launch_bounds__(kBlock) void k_chain_front(const int *__restrict__ indices, RbGeom gs, RbGeom gc, const int *__restrict__ n_dev,
                                                       const unsigned long long *__restrict__ keys, const int *__restrict__ svid,
                                                       int *__restrict__ subm0, unsigned *__restrict__ bm1, int nb_subm, int nb_set,
                                                       int *__restrict__ fill, long long fill_words) {
    // roles are interleaved in groups of ten workgroups (7 SubM probes, 2 bitmap atomics, 1 fill): the three jobs lean on
    // different parts of the memory system (L2 random reads / memory-side atomics / streaming stores) and the dispatcher hands out
    // workgroups in index order -- back-to-back block ranges would run them one after the other
    const int grp = blockIdx.x / 10, role = blockIdx.x % 10;
    const int blk = role < 7 ? grp * 7 + role : role < 9 ? nb_subm + grp * 2 + (role - 7) : nb_subm + nb_set + grp;
    if (role < 7 ? blk >= nb_subm : role < 9 ? blk >= nb_subm + nb_set : false) return;
    if (blk < nb_subm) {
        const long long t = (long long)blk * kBlock + threadIdx.x;
        if (t >= (long long)live_rows(gs, n_dev) * 14) return;
        const int o = (int)(t / 14), k = (int)(t % 14);
        if (k == 13) { subm0[(size_t)o * 27 + 13] = o; return; }
        const int kx = k % 3, ky = (k / 3) % 3, kz = k / 9;
        const int4 c = *reinterpret_cast<const int4 *>(indices + (size_t)o * 4);
        const int z = c.y + kz - 1, y = c.z + ky - 1, x = c.w + kx - 1;
        if (z >= 0 && z < gs.in_shape[0] && y >= 0 && y < gs.in_shape[1] && x >= 0 && x < gs.in_shape[2]) {
            // (fetching key and row of the home slot together -- one dependent round less on a hit -- was measured: 33 -> 42 us.
            // The probes are bound by the lines they pull from the Infinity Cache, not by latency: a speculative row load is
            // another random line for the ~80 % of probes that miss.)
            const int s = hash_find(keys, gs.mask, cell_key(c.x, z, y, x, gs.in_shape));
            const int j = s >= 0 ? svid[s] : -1;
            if ((unsigned)j < (unsigned)gs.n_in) {
                subm0[(size_t)o * 27 + k] = j;
                subm0[(size_t)j * 27 + (26 - k)] = o;
            }
        }
        return;
    }
    if (blk < nb_subm + nb_set) {
        const long long t = (long long)(blk - nb_subm) * kBlock + threadIdx.x;
        if (t >= (long long)gc.n_in * 4) return;
        const int j = (int)(t >> 2), c = (int)(t & 3);
        if (j >= live_rows(gc, n_dev)) return;
        const int4 q = *reinterpret_cast<const int4 *>(indices + (size_t)j * 4);
        int kz, ky, oz, oy;
        if (!cand_offset_fixed<3, 2>(q.y, gc.pad[0], gc.out_shape[0], c >> 1, &kz, &oz)) return;
        if (!cand_offset_fixed<3, 2>(q.z, gc.pad[1], gc.out_shape[1], c & 1, &ky, &oy)) return;
        int kx, ox0, ox1;
        const bool v0 = cand_offset_fixed<3, 2>(q.w, gc.pad[2], gc.out_shape[2], 0, &kx, &ox0);
        const bool v1 = cand_offset_fixed<3, 2>(q.w, gc.pad[2], gc.out_shape[2], 1, &kx, &ox1);
        const unsigned rowb = (((unsigned)q.x * gc.out_shape[0] + oz) * gc.out_shape[1] + oy) * gc.out_shape[2];
        const unsigned l0 = rowb + ox0, l1 = rowb + ox1;
        if (v0 && v1 && (l0 >> 5) == (l1 >> 5)) {
            atomicOr(&bm1[l0 >> 5], (1u << (l0 & 31u)) | (1u << (l1 & 31u)));
        } else {
            if (v0) atomicOr(&bm1[l0 >> 5], 1u << (l0 & 31u));
            if (v1) atomicOr(&bm1[l1 >> 5], 1u << (l1 & 31u));
        }
        return;
    }
    const long long nthreads = (long long)(gridDim.x / 10) * kBlock;
    int4 *f4 = reinterpret_cast<int4 *>(fill);
    const long long n4 = fill_words >> 2;
    for (long long i = (long long)(blk - nb_subm - nb_set) * kBlock + threadIdx.x; i < n4; i += nthreads) f4[i] = make_int4(-1, -1, -1, -1);
    if (blk == nb_subm + nb_set && threadIdx.x < (int)(fill_words & 3)) fill[(n4 << 2) + threadIdx.x] = -1;
}

// prep: 64-bit words a := 0 (level-1 bitmap + scan control), 32-bit words b := -1 (SubM table of level 0: the mirror writes of the
// symmetric probe need it)
__global__ __launch_bounds__(kBlock) void k_chain_prep(uint4 *__restrict__ a, long long na16, int4 *__restrict__ b, long long nb16,
                                                      int *__restrict__ b_tail, int nb_tail) {
    const long long stride = (long long)gridDim.x * kBlock;
    const long long i0 = (long long)blockIdx.x * kBlock + threadIdx.x;
    for (long long i = i0; i < na16; i += stride) a[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = i0; i < nb16; i += stride) b[i] = make_int4(-1, -1, -1, -1);
    if (i0 < nb_tail) b_tail[i0] = -1;
}

// rank prefixes of every level in one launch (one ticket counter; the look-back chain restarts at every level)
__global__ __launch_bounds__(kBlock) void k_chain_scan(ChainParams P) {
    __shared__ int smem[8];
    __shared__ int s_tile;
    __shared__ int cnt[kBlock * (kBmWpt / kBmBlk)];
    const int g = scan_take_tile(P.ticket, &s_tile);
    int l = 1;
    while (l < P.levels && g >= P.lv[l].tile0 + P.lv[l].ntiles) ++l;
    const ChainLevel &L = P.lv[l];
    if (L.wpt64) bm_scan_tile<kBmWpt>(L.bm, L.prefix8, L.out_cap, L.status, g - L.tile0, L.ntiles, L.num_out, smem, cnt);
    else bm_scan_tile<kBmWptSmall>(L.bm, L.prefix8, L.out_cap, L.status, g - L.tile0, L.ntiles, L.num_out, smem, cnt);
}

// output coordinates of every level in rank order, one thread per bitmap word: the tables launch then runs one thread per
// (output row, offset), perfectly balanced whatever the sites' spatial clustering.  (Measured alternatives: a table pass that
// walked the set bits of fixed word runs per workgroup left most of the chip waiting for the runs on the ground plane, 117-220 us;
// emission inside the scan tiles -- ~300 workgroups, a wave alone on its SIMD looping over the densest of its 64 words -- 43 us.)
__global__ __launch_bounds__(kBlock) void k_chain_emit(ChainParams P) {
    int l = 1;
    while (l < P.levels && (int)blockIdx.x >= P.lv[l].eblk0 + P.lv[l].eblks) ++l;
    const ChainLevel &L = P.lv[l];
    const long long wi = (long long)((int)blockIdx.x - L.eblk0) * kBlock + threadIdx.x;
    unsigned word = L.bm[wi];
    if (!word) return;
    int r = bm_word_prefix(L.bm, L.prefix8, (size_t)wi);
    const unsigned W = (unsigned)L.shape[2], H = (unsigned)L.shape[1], D = (unsigned)L.shape[0];
    const unsigned lin0 = (unsigned)wi << 5;
    const unsigned rw = lin0 / W, x0 = lin0 - rw * W;
    const unsigned plane = rw / H, y0 = rw - plane * H, b0 = plane / D, z0 = plane - b0 * D;
    while (word && r < L.out_cap) {
        const int bit = __ffs((int)word) - 1;
        word &= word - 1u;
        unsigned x = x0 + bit, y = y0, z = z0, bb = b0;
        while (x >= W) {                              // a word may straddle row ends when W is not a multiple of 32
            x -= W;
            if (++y == H) { y = 0; if (++z == D) { z = 0; ++bb; } }
        }
        *reinterpret_cast<int4 *>(L.out_indices + (size_t)r * 4) = make_int4((int)bb, (int)z, (int)y, (int)x);
        ++r;
    }
}

__device__ __forceinline__ int chain_rank(const ChainLevel &L, int b, int z, int y, int x) {
    if ((unsigned)z >= (unsigned)L.shape[0] || (unsigned)y >= (unsigned)L.shape[1] || (unsigned)x >= (unsigned)L.shape[2]) return -1;
    const unsigned lin = (((unsigned)b * L.shape[0] + z) * L.shape[1] + y) * L.shape[2] + x;
    const int r = bm_rank1(L.bm, L.prefix8, lin);
    return r < L.out_cap ? r : -1;             // a rank past the capacity is not a row (overflow is reported by num_out[1])
}

// Ranks of the THREE cells (z, y, x0), (z, y, x0 + 1), (z, y, x0 + 2) of level L in one round of loads: they are consecutive bits of
// the bitmap, so one 32-byte block + its prefix answers all three (a second block when the triple straddles a 256-cell boundary:
// 2 of 256 positions).  The tables of a 3x3x3 kernel ask for exactly such triples (dx = -1, 0, +1 of one (dz, dy)): a third of the
// loads of three bm_rank1 calls, and k_chain_tables is bound by what its lookups put through the vector L1.
__device__ __forceinline__ int rank_in_block(const uint4 &a, const uint4 &b, int prefix, unsigned lin) {
    const unsigned wi = (lin >> 5) & 7u, bit = 1u << (lin & 31u);
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned word = 0u;
    int r = prefix;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        word = wi == (unsigned)q ? w[q] : word;
        r += wi > (unsigned)q ? __popc(w[q]) : 0;
    }
    return (word & bit) ? r + __popc(word & (bit - 1u)) : -1;
}
__device__ __forceinline__ void chain_rank3(const ChainLevel &L, int b, int z, int y, int x0, int (&out)[3]) {
    out[0] = out[1] = out[2] = -1;
    if ((unsigned)z >= (unsigned)L.shape[0] || (unsigned)y >= (unsigned)L.shape[1]) return;
    const int xa = x0 < 0 ? 0 : x0, xb = x0 + 2 < L.shape[2] ? x0 + 2 : L.shape[2] - 1;
    if (xa > xb) return;
    const unsigned row = (((unsigned)b * L.shape[0] + z) * L.shape[1] + y) * L.shape[2];
    const size_t blk_a = (row + (unsigned)xa) >> 8, blk_b = (row + (unsigned)xb) >> 8;
    const uint4 a0 = *reinterpret_cast<const uint4 *>(L.bm + blk_a * kBmBlk), a1 = *reinterpret_cast<const uint4 *>(L.bm + blk_a * kBmBlk + 4);
    const int pa = L.prefix8[blk_a];
    uint4 b0 = a0, b1 = a1;
    int pb = pa;
    if (blk_b != blk_a) {
        b0 = *reinterpret_cast<const uint4 *>(L.bm + blk_b * kBmBlk);
        b1 = *reinterpret_cast<const uint4 *>(L.bm + blk_b * kBmBlk + 4);
        pb = L.prefix8[blk_b];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int x = x0 + i;
        if ((unsigned)x >= (unsigned)L.shape[2]) continue;
        const unsigned lin = row + (unsigned)x;
        const bool in_a = (lin >> 8) == blk_a;
        const int r = rank_in_block(in_a ? a0 : b0, in_a ? a1 : b1, in_a ? pa : pb, lin);
        out[i] = r < L.out_cap ? r : -1;           // a rank past the capacity is not a row (overflow is reported by num_out[1])
    }
}

constexpr int kTabU = 4;                       // table items (kernel of 3 offsets) / x-triples (27 offsets) per thread of k_chain_tables
__global__ __launch_bounds__(kBlock) void k_chain_tables(ChainParams P) {
    const int blk = blockIdx.x;
    if (blk < P.cand_blks) {
        // level 1, input side: voxel row j reaches <= 8 outputs; nbr_out[rank][k] = j (table pre-filled with -1 by the front launch)
        const ChainLevel &L = P.lv[1];
        const long long t = (long long)blk * kBlock + threadIdx.x;
        const int j = (int)(t >> 3), c = (int)(t & 7);
        if (j >= chain_live0(P)) return;
        const int4 q = *reinterpret_cast<const int4 *>(P.indices0 + (size_t)j * 4);
        int kk[3], out[3];
        bool ok = cand_offset_fixed<3, 2>(q.y, L.pad[0], L.shape[0], c >> 2, &kk[0], &out[0]);
        ok &= cand_offset_fixed<3, 2>(q.z, L.pad[1], L.shape[1], (c >> 1) & 1, &kk[1], &out[1]);
        ok &= cand_offset_fixed<3, 2>(q.w, L.pad[2], L.shape[2], c & 1, &kk[2], &out[2]);
        if (!ok) return;
        const int o = chain_rank(L, q.x, out[0], out[1], out[2]);
        if (o >= 0) L.nbr_out[(size_t)o * 27 + (kk[0] * 3 + kk[1]) * 3 + kk[2]] = j;
        return;
    }
    if (blk >= P.map_blk0) {
        const ChainLevel &L = P.lv[P.levels];
        const long long i = (long long)(blk - P.map_blk0) * kBlock + threadIdx.x;
        if (i >= P.map_cells) return;
        const int r = bm_rank1(L.bm, L.prefix8, (unsigned)i);
        P.site_map[i] = (r >= 0 && r < L.out_cap) ? r + 1 : 0;
        return;
    }
    // (output row, kernel offset) items, offset fastest: every table entry is written (-1 included) in coalesced runs.  A thread
    // takes kTabU items a workgroup-width apart and issues their loads together -- one item per thread left the launch waiting on
    // two dependent memory rounds per workgroup times 18 rounds of resident workgroups (41 us for 7.5 M lookups).  The output's
    // coordinates come from out_indices (rank order, k_chain_emit); items of rows at or past the live count are skipped (the grid is
    // sized for the capacity).
    int l = 1;
    while (l < P.levels && blk >= P.lv[l].cblk0 + P.lv[l].cblks) ++l;
    const ChainLevel &L = P.lv[l];
    const int live = L.num_out[0];
    const bool subm = blk < L.sblk0 + L.sblks;
    const ChainLevel &T = subm ? L : P.lv[l - 1];                 // the level whose bitmap answers the lookups
    const int kv = subm ? 27 : L.kvol;
    int *dst = subm ? L.subm_nbr : L.nbr_out;
    const long long t0 = (long long)(blk - (subm ? L.sblk0 : L.cblk0)) * (kBlock * kTabU) + threadIdx.x;
    if (kv == 27) {
        // 27 offsets: a thread takes kTabU x-TRIPLES (row, dz, dy) -- one bitmap block + one prefix per three table entries -- and the
        // wave hands its 192 consecutive entries to memory through LDS as three full 256-byte stores (a thread's own three entries are
        // 12 bytes apart from its neighbour's)
        __shared__ int s_tri[kBlock / 64][192];
        if (t0 - threadIdx.x >= (long long)live * 9) return;     // whole workgroups past the live rows leave here
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const long long n_ent = (long long)live * 27;
        int4 c3[kTabU];
        int r3[kTabU], j3[kTabU];
#pragma unroll
        for (int u = 0; u < kTabU; ++u) {
            const long long t = t0 + (long long)u * kBlock;
            const int r = (int)(t / 9);
            r3[u] = r; j3[u] = (int)(t - (long long)r * 9);
            c3[u] = *reinterpret_cast<const int4 *>(L.out_indices + (size_t)(r < live ? r : 0) * 4);
        }
        int v3[kTabU][3];
#pragma unroll
        for (int u = 0; u < kTabU; ++u) {
            const int dz = j3[u] / 3, dy = j3[u] - dz * 3;
            int z, y, x0;
            if (subm) { z = c3[u].y + dz - 1; y = c3[u].z + dy - 1; x0 = c3[u].w - 1; }
            else { z = 2 * c3[u].y - L.pad[0] + dz; y = 2 * c3[u].z - L.pad[1] + dy; x0 = 2 * c3[u].w - L.pad[2]; }
            if (r3[u] < live) chain_rank3(T, c3[u].x, z, y, x0, v3[u]);
            else v3[u][0] = v3[u][1] = v3[u][2] = -1;
            if (subm && j3[u] == 4) v3[u][1] = r3[u];              // the centre offset is the row itself
        }
#pragma unroll
        for (int u = 0; u < kTabU; ++u) {
            const long long e0 = (t0 - lane + (long long)u * kBlock) * 3;      // first entry of this wave's 192
#pragma unroll
            for (int i = 0; i < 3; ++i) s_tri[wv][3 * lane + i] = v3[u][i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int v = s_tri[wv][i * 64 + lane];
                if (e0 + i * 64 + lane < n_ent) dst[e0 + i * 64 + lane] = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (t0 >= (long long)live * kv) return;                      // (uniform enough: whole workgroups past the live rows leave here)
    int4 c[kTabU];
    int rr[kTabU], kk[kTabU];
    bool on[kTabU];
#pragma unroll
    for (int u = 0; u < kTabU; ++u) {
        const long long t = t0 + (long long)u * kBlock;
        const int r = kv == 27 ? (int)(t / 27) : (int)(t / 3);
        rr[u] = r; kk[u] = (int)(t - (long long)r * kv);
        on[u] = r < live;
        c[u] = *reinterpret_cast<const int4 *>(L.out_indices + (size_t)(on[u] ? r : 0) * 4);
    }
    int val[kTabU];
#pragma unroll
    for (int u = 0; u < kTabU; ++u) {
        const int k = kk[u];
        int z, y, x;
        if (subm) { z = c[u].y + k / 9 - 1; y = c[u].z + (k / 3) % 3 - 1; x = c[u].w + k % 3 - 1; }
        else if (kv == 27) { z = 2 * c[u].y - L.pad[0] + k / 9; y = 2 * c[u].z - L.pad[1] + (k / 3) % 3; x = 2 * c[u].w - L.pad[2] + k % 3; }
        else { z = 2 * c[u].y - L.pad[0] + k; y = c[u].z - L.pad[1]; x = c[u].w - L.pad[2]; }
        val[u] = (subm && k == 13) ? rr[u] : chain_rank(T, c[u].x, z, y, x);
    }
#pragma unroll
    for (int u = 0; u < kTabU; ++u)
        if (on[u]) dst[t0 + (long long)u * kBlock] = val[u];
}

struct RbWorkspace {
    unsigned long long *keys;
    int *vals, *orank, *cand_slot, *rank, *scan, *blk, *scan2, *overflow, *ticket;
    unsigned long long *status;
    long long ctl_words;
    unsigned *first_mask;
    unsigned char *cand_k;
    uint32_t table;
    size_t bytes;
};

static RbWorkspace carve_rb(void *ws, size_t cap, int n_in, int kvol, int max_out_per_in) {
    RbWorkspace w;
    Arena a(ws, cap);
    size_t entries = (size_t)(n_in > 0 ? n_in : 1) * (size_t)(max_out_per_in > 0 ? max_out_per_in : 1);
    w.table = next_pow2((uint32_t)(entries * 2 > 1024 ? entries * 2 : 1024));
    w.keys = a.take<unsigned long long>(w.table);
    w.vals = a.take<int>(w.table);
    w.orank = a.take<int>(w.table);
    long long nk = (long long)n_in * kvol;
    w.cand_slot = a.take<int>(nk);
    w.rank = a.take<int>(nk);
    w.scan = a.take<int>(scan_scratch_ints(nk));
    long long nblk = (long long)kvol * div_up(n_in > 0 ? n_in : 1, kBlock);
    w.blk = a.take<int>(nblk + 1);
    w.scan2 = a.take<int>(scan_scratch_ints(nblk));
    // control block cleared by one k_rb_init region: [ticket, overflow, -, -, status[tiles] (64-bit)]
    w.ctl_words = (long long)scan_ctl_words(n_in);
    w.ticket = a.take<int>(w.ctl_words);
    w.overflow = w.ticket + 1;
    w.status = reinterpret_cast<unsigned long long *>(w.ticket + 4);
    w.first_mask = a.take<unsigned>(n_in > 0 ? n_in : 1);
    w.cand_k = a.take<unsigned char>(nk > 0 ? nk : 1);
    w.bytes = align_up(a.used);
    return w;
}

static int fill_geom(RbGeom &g, const int *in_shape, const int *out_shape, const int *ksize, const int *stride,
                     const int *pad, const int *dil, int n_in, int batch) {
    g.kvol = 1;
    for (int d = 0; d < 3; ++d) {
        g.in_shape[d] = in_shape[d];
        g.out_shape[d] = out_shape ? out_shape[d] : in_shape[d];
        g.ksize[d] = ksize[d];
        g.stride[d] = stride ? stride[d] : 1;
        g.pad[d] = pad ? pad[d] : 0;
        g.dil[d] = dil ? dil[d] : 1;
        if (g.ksize[d] <= 0 || g.stride[d] <= 0 || g.dil[d] <= 0 || g.in_shape[d] <= 0 || g.out_shape[d] <= 0)
            return SEC_E_INVALID;
        g.kvol *= ksize[d];
    }
    if (g.kvol > kMaxKvol) return SEC_E_UNSUPPORTED;
    g.n_in = n_in;
    g.batch = batch;
    g.ncand = 1;
    for (int d = 0; d < 3; ++d) {   // most kernel offsets of one residue class mod stride
        int best = 0;
        for (int r = 0; r < g.stride[d]; ++r) {
            int c = 0;
            for (int k = 0; k < g.ksize[d]; ++k) c += (k * g.dil[d]) % g.stride[d] == r;
            if (c > best) best = c;
        }
        g.cand[d] = best;
        g.ncand *= best;
    }
    return SEC_OK;
}

static int max_out_per_in(const RbGeom &g, int hint) { return (hint > 0 && hint < g.ncand) ? hint : g.ncand; }

}  // namespace sec

using namespace sec;

SEC_API size_t sec_rulebook_workspace_bytes(int n_in, int kvol, int max_out_per_in) {
    if (n_in < 0 || kvol <= 0) return 0;
    return carve_rb(nullptr, 0, n_in, kvol, max_out_per_in).bytes;
}

SEC_API void sec_conv_output_shape(const int *in_shape, const int *ksize, const int *stride, const int *pad,
                                   const int *dil, int *out_shape) {
    for (int d = 0; d < 3; ++d)
        out_shape[d] = (in_shape[d] + 2 * pad[d] - dil[d] * (ksize[d] - 1) - 1) / stride[d] + 1;
}

SEC_API int sec_rulebook_subm3d(const int *indices, int n_in, const int *n_in_dev, int batch, const int *h_shape3,
                                const int *h_ksize3, const int *h_dilation3, int *nbr_out, int *pairs, int *pair_num,
                                void *workspace, size_t workspace_bytes, void *stream) {
    if (n_in_dev && pairs) return SEC_E_UNSUPPORTED;  // pair lists are an eager-API feature
    if (n_in < 0 || batch <= 0 || !h_shape3 || !h_ksize3 || (n_in > 0 && !nbr_out) || (pairs && !pair_num)) return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_shape3, nullptr, h_ksize3, nullptr, nullptr, h_dilation3, n_in, batch);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d)
        if (g.ksize[d] % 2 == 0) return SEC_E_UNSUPPORTED;  // submanifold kernels are odd (spconv asserts the same)
    hipStream_t st = (hipStream_t)stream;
    RbWorkspace w = carve_rb(workspace, workspace_bytes, n_in, g.kvol, 1);
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    g.mask = w.table - 1;
    if (n_in == 0) {
        if (pair_num) return hip_ok(hipMemsetAsync(pair_num, 0, g.kvol * sizeof(int), st));
        return SEC_OK;
    }
    long long nk = (long long)n_in * g.kvol;
    rb_init(w.keys, w.table, kEmptyKey, nbr_out, nk, -1, nullptr, 0, 0, st);
    hipLaunchKernelGGL(k_rb_hash_rows, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.vals);
    long long nh = (long long)n_in * (g.kvol / 2 + 1);
    if (g.kvol == 27 && g.ksize[0] == 3 && g.ksize[1] == 3)
        hipLaunchKernelGGL(k_subm_nbr_sym<true>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.vals, nbr_out);
    else
        hipLaunchKernelGGL(k_subm_nbr_sym<false>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.vals, nbr_out);
    if ((rc = check_launch())) return rc;
    if (pairs) return emit_pairs(nbr_out, n_in, g.kvol, /*mirror=*/1, w.blk, w.scan2, pairs, pair_num, st);
    return SEC_OK;
}

SEC_API int sec_rulebook_subm3d_after_conv(const int *indices, int n_in, const int *n_in_dev, int batch, const int *h_shape3,
                                           const int *h_ksize3, const int *h_dilation3, int *nbr_out,
                                           const void *conv_workspace, size_t conv_workspace_bytes, int conv_n_in,
                                           const int *h_conv_ksize3, const int *h_conv_stride3, const int *h_conv_dilation3,
                                           int conv_out_per_in_hint, void *stream) {
    if (n_in < 0 || batch <= 0 || !h_shape3 || !h_ksize3 || (n_in > 0 && !nbr_out) || !conv_workspace || !h_conv_ksize3 ||
        !h_conv_stride3)
        return SEC_E_INVALID;
    RbGeom g, cg;
    int rc = fill_geom(g, h_shape3, nullptr, h_ksize3, nullptr, nullptr, h_dilation3, n_in, batch);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d)
        if (g.ksize[d] % 2 == 0) return SEC_E_UNSUPPORTED;
    const int ones[3] = {1, 1, 1};
    if ((rc = fill_geom(cg, ones, ones, h_conv_ksize3, h_conv_stride3, nullptr, h_conv_dilation3, conv_n_in, 1))) return rc;
    RbWorkspace w = carve_rb(const_cast<void *>(conv_workspace), conv_workspace_bytes, conv_n_in, cg.kvol,
                             max_out_per_in(cg, conv_out_per_in_hint));
    if (w.bytes > conv_workspace_bytes) return SEC_E_WORKSPACE;
    if (n_in == 0) return SEC_OK;
    g.mask = w.table - 1;
    hipStream_t st = (hipStream_t)stream;
    rb_init(nullptr, 0, 0, nbr_out, (long long)n_in * g.kvol, -1, nullptr, 0, 0, st);
    long long nh = (long long)n_in * (g.kvol / 2 + 1);
    // the strided build's table maps output cell -> slot and orank[slot] = output row: exactly this layer's site lookup
    if (g.kvol == 27 && g.ksize[0] == 3 && g.ksize[1] == 3)
        hipLaunchKernelGGL(k_subm_nbr_sym<true>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.orank, nbr_out);
    else
        hipLaunchKernelGGL(k_subm_nbr_sym<false>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.orank, nbr_out);
    return check_launch();
}

SEC_API int sec_rulebook_subm3d_after_voxelize(const int *indices, int n_in, const int *n_in_dev, int batch, const int *h_shape3,
                                               const int *h_ksize3, const int *h_dilation3, int *nbr_out,
                                               const void *vox_workspace, size_t vox_workspace_bytes, int vox_num_points,
                                               int vox_max_voxels, int vox_max_points, const int *h_vox_grid3_zyx, void *stream) {
    if (n_in < 0 || batch <= 0 || !h_shape3 || !h_ksize3 || (n_in > 0 && !nbr_out) || !vox_workspace || !h_vox_grid3_zyx)
        return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_shape3, nullptr, h_ksize3, nullptr, nullptr, h_dilation3, n_in, batch);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d) {
        if (g.ksize[d] % 2 == 0) return SEC_E_UNSUPPORTED;
        if (h_vox_grid3_zyx[d] <= 0 || h_vox_grid3_zyx[d] > h_shape3[d]) return SEC_E_INVALID;
        // the voxeliser keys its cells with ITS grid (SpMiddleFHD's sparse shape has one more z plane, middle.py:139); a
        // neighbour outside that grid holds no voxel, so bounding the lookups by it is exact
        g.in_shape[d] = h_vox_grid3_zyx[d];
    }
    const unsigned long long *keys;
    const int *svid;
    uint32_t mask;
    if (!vox_table_of(vox_workspace, vox_workspace_bytes, vox_num_points, batch, vox_max_voxels, vox_max_points, &keys, &svid, &mask))
        return SEC_E_WORKSPACE;
    if (n_in == 0) return SEC_OK;
    g.mask = mask;
    hipStream_t st = (hipStream_t)stream;
    rb_init(nullptr, 0, 0, nbr_out, (long long)n_in * g.kvol, -1, nullptr, 0, 0, st);
    const long long nh = (long long)n_in * (g.kvol / 2 + 1);
    if (g.kvol == 27 && g.ksize[0] == 3 && g.ksize[1] == 3)
        hipLaunchKernelGGL(k_subm_nbr_sym<true>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, keys, svid, nbr_out);
    else
        hipLaunchKernelGGL(k_subm_nbr_sym<false>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, keys, svid, nbr_out);
    return check_launch();
}

SEC_API int sec_rulebook_conv3d_build(const int *indices, int n_in, const int *n_in_dev, int batch, const int *h_in_shape3,
                                      const int *h_out_shape3, const int *h_ksize3, const int *h_stride3,
                                      const int *h_padding3, const int *h_dilation3, int *out_indices, int out_cap,
                                      int *num_out, int out_per_in_hint, int *prefill_nbr_out, int prefill_nbr_out_rows,
                                      int *prefill_nbr_in, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_in < 0 || batch <= 0 || !h_in_shape3 || !h_out_shape3 || !h_ksize3 || !h_stride3 || !h_padding3 ||
        !out_indices || !num_out || out_cap < 0 || prefill_nbr_out_rows < 0)
        return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_padding3, h_dilation3, n_in, batch);
    if (rc) return rc;
    // stride > 1 together with dilation > 1 in the same dim: upstream getValidOutPos steps the outputs by `dilation`
    // and floors the offset division there, which no SECOND config exercises -- refused rather than guessed
    for (int d = 0; d < 3; ++d)
        if (g.stride[d] > 1 && g.dil[d] > 1) return SEC_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    RbWorkspace w = carve_rb(workspace, workspace_bytes, n_in, g.kvol, max_out_per_in(g, out_per_in_hint));
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    g.mask = w.table - 1;
    long long nc = (long long)n_in * g.ncand;
    const long long fill_a = prefill_nbr_out ? (long long)prefill_nbr_out_rows * g.kvol : 0;
    const long long fill_b = prefill_nbr_in ? (long long)n_in * g.kvol : 0;
    if (nc == 0) {
        if (fill_a + fill_b > 0) rb_init(nullptr, 0, 0, prefill_nbr_out, fill_a, -1, prefill_nbr_in, fill_b, -1, st);
        return fill_words(num_out, 2 * sizeof(int), 0u, st);
    }
    rb_init(w.keys, w.table, kEmptyKey, w.vals, w.table, kEmptyI32, w.ticket, w.ctl_words, 0, st);
    int nb = div_up(nc, kBlock);
    const int geo = geo_of(g);
#define SEC_RB_BUILD(GEO)                                                                                                         \
    do {                                                                                                                          \
        hipLaunchKernelGGL(k_conv_cand<GEO>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.keys, w.vals, w.cand_slot,    \
                           w.cand_k, w.overflow);                                                                                \
        if (g.ncand <= 32) {                                                                                                      \
            hipLaunchKernelGGL(k_conv_count_scan_assign<GEO>, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, st, w.cand_slot,       \
                               w.cand_k, w.vals, w.keys, g, w.orank, out_indices, out_cap, w.status, w.ticket, num_out,           \
                               w.overflow, prefill_nbr_out, fill_a, prefill_nbr_in, fill_b);                                      \
            return check_launch();                                                                                                \
        }                                                                                                                         \
    } while (0)
    if (geo == 1) SEC_RB_BUILD(1);
    else if (geo == 2) SEC_RB_BUILD(2);
    else SEC_RB_BUILD(0);
#undef SEC_RB_BUILD
    if (fill_a + fill_b > 0) rb_init(nullptr, 0, 0, prefill_nbr_out, fill_a, -1, prefill_nbr_in, fill_b, -1, st);
    hipLaunchKernelGGL(k_conv_count_scan, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, st, w.cand_slot, w.cand_k, w.vals, n_in,
                       g.kvol, g.ncand, w.rank, w.first_mask, w.status, w.ticket, num_out);
    hipLaunchKernelGGL(k_conv_assign, dim3(nb), dim3(kBlock), 0, st, w.cand_slot, w.cand_k, w.vals, w.keys, w.rank, g,
                       w.orank, out_indices, out_cap, num_out, w.overflow, w.first_mask);
    return check_launch();
}

// ---- sorted (spconv-GPU) numbering: host side --------------------------------------------------------------------------
struct BmWorkspace {
    unsigned *bm;
    int *prefix, *ticket, *blk, *scan2;
    unsigned long long *status;
    long long n_words, ctl_words, zero_words64;   // zero_words64: 64-bit words from bm to the end of the control block
    size_t bytes;
};

static BmWorkspace carve_bm(void *ws, size_t cap, int n_in, int kvol, long long cells) {
    BmWorkspace w;
    Arena a(ws, cap);
    w.n_words = (long long)div_up(div_up(cells > 0 ? cells : 1, 32), kBmTile) * kBmTile;
    w.bm = a.take<unsigned>(w.n_words);
    // the scan's control block directly behind the bitmap: ONE zero fill covers both
    w.ctl_words = (long long)scan_ctl_words(w.n_words / kBmWptSmall);   // status words for the finer of the two tilings
    w.ticket = a.take<int>(w.ctl_words);
    w.status = reinterpret_cast<unsigned long long *>(w.ticket + 4);
    w.zero_words64 = (long long)((reinterpret_cast<char *>(w.ticket + w.ctl_words) - reinterpret_cast<char *>(w.bm)) / 8);
    w.prefix = a.take<int>(w.n_words / kBmBlk);      // one entry per 8-word block (k_bm_scan)
    long long nblk = (long long)kvol * div_up(n_in > 0 ? n_in : 1, kBlock);
    w.blk = a.take<int>(nblk + 1);
    w.scan2 = a.take<int>(scan_scratch_ints(nblk));
    w.bytes = align_up(a.used);
    return w;
}

static long long bm_cells(int batch, const int *shape3) { return (long long)batch * shape3[0] * shape3[1] * shape3[2]; }


SEC_API size_t sec_rulebook_sorted_workspace_bytes(int n_in, int kvol, int batch, const int *h_out_shape3) {
    if (n_in < 0 || kvol <= 0 || kvol > kMaxKvol || batch <= 0 || !h_out_shape3) return 0;
    return carve_bm(nullptr, 0, n_in, kvol, bm_cells(batch, h_out_shape3)).bytes;
}

SEC_API int sec_rulebook_conv3d_build_sorted(const int *indices, int n_in, const int *n_in_dev, int batch, const int *h_in_shape3,
                                             const int *h_out_shape3, const int *h_ksize3, const int *h_stride3,
                                             const int *h_padding3, const int *h_dilation3, int *out_indices, int out_cap,
                                             int *num_out, int *prefill_nbr_out, int prefill_nbr_out_rows, int *prefill_nbr_in,
                                             int *prefill_extra, long long prefill_extra_words,
                                             const void *in_sites_workspace, size_t in_sites_workspace_bytes, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    if (n_in < 0 || batch <= 0 || !h_in_shape3 || !h_out_shape3 || !h_ksize3 || !h_stride3 || !h_padding3 || !num_out ||
        out_cap < 0 || prefill_nbr_out_rows < 0 || prefill_extra_words < 0 || (prefill_extra_words > 0 && !prefill_extra))
        return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_padding3, h_dilation3, n_in, batch);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d)
        if (g.stride[d] > 1 && g.dil[d] > 1) return SEC_E_UNSUPPORTED;
    const long long cells = bm_cells(batch, h_out_shape3);
    if (cells >= (1ll << 32) - 64) return SEC_E_UNSUPPORTED;            // cell indices are 32-bit here
    hipStream_t st = (hipStream_t)stream;
    BmWorkspace w = carve_bm(workspace, workspace_bytes, n_in, g.kvol, cells);
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    const long long fill_a = prefill_nbr_out ? (long long)prefill_nbr_out_rows * g.kvol : 0;
    const long long fill_b = prefill_nbr_in ? (long long)n_in * g.kvol : 0;
    const int geo = geo_of(g);
    // the inputs are the outputs of an earlier sorted build whose bitmap is still around: no atomics at all
    const long long cells_in = bm_cells(batch, h_in_shape3);
    bool dilate = in_sites_workspace && (geo == 1 || geo == 2) && cells_in < (1ll << 32) - 64;
    for (int d = 0; d < 3; ++d) dilate = dilate && g.pad[d] >= 0 && g.pad[d] <= 1;
    BmWorkspace wi{};
    if (dilate) {
        wi = carve_bm(const_cast<void *>(in_sites_workspace), in_sites_workspace_bytes, 0, 1, cells_in);
        if (wi.bytes > in_sites_workspace_bytes) return SEC_E_WORKSPACE;
    }
    // scan control (+ the bitmap unless the dilation writes every word) := 0, both gather tables := -1, one launch
    if (!dilate)
        rb_init(reinterpret_cast<unsigned long long *>(w.bm), w.zero_words64, 0ull, prefill_nbr_out, fill_a, -1, prefill_nbr_in,
                fill_b, -1, st, prefill_extra, prefill_extra_words);
    const long long nc = (long long)n_in * g.ncand;
    if (dilate) {
        // scan control := 0 and both gather tables := -1 ride on the same launch (grid-stride tail)
        const RbFill fl{reinterpret_cast<unsigned long long *>(w.ticket), w.ctl_words / 2, prefill_nbr_out, fill_a, prefill_nbr_in, fill_b,
                        prefill_extra, prefill_extra_words};
        long long most = w.n_words;
        for (long long v : {fl.na, fl.nb, fl.nc, fl.nd}) most = v > most ? v : most;
        long long nbl = div_up(most, (long long)kBlock);
        const long long cover = div_up(w.n_words, (long long)kBlock);       // every bitmap word needs its own thread
        if (nbl > 256 * 8) nbl = 256 * 8;
        if (nbl < cover) nbl = cover;
        const unsigned nb = (unsigned)nbl;
        if (geo == 1) hipLaunchKernelGGL(k_bm_dilate<1>, dim3(nb), dim3(kBlock), 0, st, wi.bm, wi.n_words, g, w.bm, w.n_words, fl);
        else hipLaunchKernelGGL(k_bm_dilate<2>, dim3(nb), dim3(kBlock), 0, st, wi.bm, wi.n_words, g, w.bm, w.n_words, fl);
    } else if (nc > 0) {
        const int nb = div_up(nc, kBlock);
        if (geo == 1) hipLaunchKernelGGL(k_bm_set_x2, dim3(div_up((long long)n_in * 4, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm);
        else if (geo == 2) hipLaunchKernelGGL(k_bm_set<2>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm);
        else hipLaunchKernelGGL(k_bm_set<0>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm);
    }
    if (w.n_words <= kBmSmallWords)
        hipLaunchKernelGGL(k_bm_scan<kBmWptSmall>, dim3((unsigned)(w.n_words / (kBlock * kBmWptSmall))), dim3(kBlock), 0, st, w.bm,
                           w.prefix, out_cap, w.status, w.ticket, num_out);
    else
        hipLaunchKernelGGL(k_bm_scan<kBmWpt>, dim3((unsigned)(w.n_words / kBmTile)), dim3(kBlock), 0, st, w.bm, w.prefix, out_cap,
                           w.status, w.ticket, num_out);
    if (out_indices)      // NULL: the caller lets sec_rulebook_conv3d_tables_sorted write them (one launch less)
        hipLaunchKernelGGL(k_bm_emit, dim3((unsigned)div_up(w.n_words, kBlock)), dim3(kBlock), 0, st, w.bm, w.prefix, g, w.n_words,
                           out_indices, out_cap);
    return check_launch();
}

SEC_API int sec_rulebook_conv3d_tables_sorted(const int *indices, int n_in, const int *n_in_dev, int batch,
                                              const int *h_in_shape3, const int *h_out_shape3, const int *h_ksize3,
                                              const int *h_stride3, const int *h_padding3, const int *h_dilation3, int *nbr_out,
                                              int nbr_out_rows, int *nbr_in, int prefilled, int *out_indices, int out_cap,
                                              int *pairs, int *pair_num, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_in < 0 || batch <= 0 || !h_in_shape3 || !h_out_shape3 || !h_ksize3 || !h_stride3 || !h_padding3 ||
        (nbr_out_rows > 0 && !nbr_out) || nbr_out_rows < 0 || (pairs && (!pair_num || !nbr_in)))
        return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_in_shape3, h_out_shape3, h_ksize3, h_stride3, h_padding3, h_dilation3, n_in, batch);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    BmWorkspace w = carve_bm(workspace, workspace_bytes, n_in, g.kvol, bm_cells(batch, h_out_shape3));
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    const long long n_out_words = (long long)nbr_out_rows * g.kvol, n_in_words = nbr_in ? (long long)n_in * g.kvol : 0;
    if (!prefilled && n_out_words + n_in_words > 0) rb_init(nullptr, 0, 0, nbr_out, n_out_words, -1, nbr_in, n_in_words, -1, st);
    const long long nc = (long long)n_in * g.ncand;
    if (nc > 0) {
        const int nb = div_up(nc, kBlock), geo = geo_of(g);
        if (geo == 1) hipLaunchKernelGGL(k_bm_tables<1>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm, w.prefix, nbr_in, nbr_out, nbr_out_rows, out_indices, out_cap);
        else if (geo == 2) hipLaunchKernelGGL(k_bm_tables<2>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm, w.prefix, nbr_in, nbr_out, nbr_out_rows, out_indices, out_cap);
        else hipLaunchKernelGGL(k_bm_tables<0>, dim3(nb), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm, w.prefix, nbr_in, nbr_out, nbr_out_rows, out_indices, out_cap);
        if ((rc = check_launch())) return rc;
    }
    if (pairs) return emit_pairs(nbr_in, n_in, g.kvol, /*mirror=*/0, w.blk, w.scan2, pairs, pair_num, st);
    return SEC_OK;
}

SEC_API int sec_rulebook_subm3d_after_conv_sorted(const int *indices, int n_in, const int *n_in_dev, int batch,
                                                  const int *h_shape3, const int *h_ksize3, const int *h_dilation3, int *nbr_out,
                                                  int prefilled, const void *conv_workspace, size_t conv_workspace_bytes,
                                                  void *stream) {
    if (n_in < 0 || batch <= 0 || !h_shape3 || !h_ksize3 || (n_in > 0 && !nbr_out) || !conv_workspace) return SEC_E_INVALID;
    RbGeom g;
    int rc = fill_geom(g, h_shape3, nullptr, h_ksize3, nullptr, nullptr, h_dilation3, n_in, batch);
    if (rc) return rc;
    for (int d = 0; d < 3; ++d)
        if (g.ksize[d] % 2 == 0) return SEC_E_UNSUPPORTED;
    // only the bitmap and the prefix array are read: their place in the workspace depends on the grid alone
    BmWorkspace w = carve_bm(const_cast<void *>(conv_workspace), conv_workspace_bytes, 0, 1, bm_cells(batch, h_shape3));
    if (w.bytes > conv_workspace_bytes) return SEC_E_WORKSPACE;
    if (n_in == 0) return SEC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!prefilled) rb_init(nullptr, 0, 0, nbr_out, (long long)n_in * g.kvol, -1, nullptr, 0, 0, st);
    const long long nh = (long long)n_in * (g.kvol / 2 + 1);
    if (g.kvol == 27 && g.ksize[0] == 3 && g.ksize[1] == 3)
        hipLaunchKernelGGL(k_subm_nbr_bm<true>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm, w.prefix, nbr_out);
    else
        hipLaunchKernelGGL(k_subm_nbr_bm<false>, dim3(div_up(nh, kBlock)), dim3(kBlock), 0, st, indices, g, n_in_dev, w.bm, w.prefix, nbr_out);
    return check_launch();
}


SEC_API int sec_sparse_site_map_sorted(const void *conv_workspace, size_t conv_workspace_bytes, const int *num_dev, int rows_cap,
                                       int batch, int d, int h, int w, int *site_map, void *stream) {
    if (!conv_workspace || batch <= 0 || d <= 0 || h <= 0 || w <= 0 || rows_cap < 0 || !site_map) return SEC_E_INVALID;
    const int shape[3] = {d, h, w};
    const long long cells = bm_cells(batch, shape);
    if (cells >= (1ll << 32) - 64) return SEC_E_UNSUPPORTED;
    BmWorkspace wk = carve_bm(const_cast<void *>(conv_workspace), conv_workspace_bytes, 0, 1, cells);
    if (wk.bytes > conv_workspace_bytes) return SEC_E_WORKSPACE;
    hipLaunchKernelGGL(k_bm_site_map, dim3(div_up(cells, (long long)kBlock)), dim3(kBlock), 0, (hipStream_t)stream, wk.bm, wk.prefix,
                       num_dev, rows_cap, cells, site_map);
    return check_launch();
}

SEC_API int sec_rulebook_conv3d_tables(int n_in, const int *h_ksize3, const int *h_stride3, const int *h_dilation3,
                                       int out_per_in_hint, int *nbr_out, int nbr_out_rows, int *nbr_in, int prefilled,
                                       int *pairs, int *pair_num, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_in < 0 || !h_ksize3 || !h_stride3 || (nbr_out_rows > 0 && !nbr_out) || nbr_out_rows < 0 ||
        (pairs && (!pair_num || !nbr_in)))
        return SEC_E_INVALID;
    RbGeom g;
    const int ones[3] = {1, 1, 1};
    int rc = fill_geom(g, ones, ones, h_ksize3, h_stride3, nullptr, h_dilation3, n_in, 1);   // shapes unused here
    if (rc) return rc;
    const int kvol = g.kvol;
    hipStream_t st = (hipStream_t)stream;
    RbWorkspace w = carve_rb(workspace, workspace_bytes, n_in, kvol, max_out_per_in(g, out_per_in_hint));
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;
    // one fill launch for both tables (nbr_in only when the caller wants it: backward / pair lists)
    long long n_out_words = (long long)nbr_out_rows * kvol, n_in_words = nbr_in ? (long long)n_in * kvol : 0;
    if (!prefilled && n_out_words + n_in_words > 0)   // prefilled: the build call already wrote the -1s (static pipelines)
        rb_init(nullptr, 0, 0, nbr_out, n_out_words, -1, nbr_in, n_in_words, -1, st);
    long long nc = (long long)n_in * g.ncand;
    if (nc > 0) {
        const int geo = geo_of(g);
        if (geo == 1)
            hipLaunchKernelGGL(k_conv_tables<1>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, st, w.cand_slot, w.cand_k, w.orank, nc,
                               kvol, g.ncand, nbr_in, nbr_out, nbr_out_rows);
        else if (geo == 2)
            hipLaunchKernelGGL(k_conv_tables<2>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, st, w.cand_slot, w.cand_k, w.orank, nc,
                               kvol, g.ncand, nbr_in, nbr_out, nbr_out_rows);
        else
            hipLaunchKernelGGL(k_conv_tables<0>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, st, w.cand_slot, w.cand_k, w.orank, nc,
                               kvol, g.ncand, nbr_in, nbr_out, nbr_out_rows);
        if ((rc = check_launch())) return rc;
    }
    if (pairs) return emit_pairs(nbr_in, n_in, kvol, /*mirror=*/0, w.blk, w.scan2, pairs, pair_num, st);
    return SEC_OK;
}


// ---- fused chain: host side ---------------------------------------------------------------------------------------------------------
namespace sec {
struct ChainWorkspace {
    unsigned *bm[kChainMaxLevels + 1];
    int *prefix[kChainMaxLevels + 1];
    long long n_words[kChainMaxLevels + 1];
    int ntiles[kChainMaxLevels + 1], wpt64[kChainMaxLevels + 1];
    int *ticket;
    unsigned long long *status;
    long long zero16;        // 16-byte words from bm[1] to the end of the scan control block
    size_t bytes;
};
static ChainWorkspace carve_chain(void *ws, size_t cap, int batch, int levels, const int *shapes) {
    ChainWorkspace w{};
    Arena a(ws, cap);
    int tiles = 0;
    for (int l = 1; l <= levels; ++l) {
        const long long cells = bm_cells(batch, shapes + 3 * l);
        w.n_words[l] = (long long)div_up(div_up(cells > 0 ? cells : 1, 32), kBmTile) * kBmTile;
        w.wpt64[l] = w.n_words[l] > kBmSmallWords;
        w.ntiles[l] = (int)(w.n_words[l] / (kBlock * (w.wpt64[l] ? kBmWpt : kBmWptSmall)));
        tiles += w.ntiles[l];
    }
    w.bm[1] = a.take<unsigned>(w.n_words[1]);
    const long long ctl_words = 4 + 2 * (long long)tiles;          // [ticket, -, -, -, status (64-bit) ...]
    w.ticket = a.take<int>(ctl_words + 4);
    w.status = reinterpret_cast<unsigned long long *>(w.ticket + 4);
    w.zero16 = (long long)((reinterpret_cast<char *>(w.ticket + ctl_words + 4) - reinterpret_cast<char *>(w.bm[1])) / 16);
    for (int l = 2; l <= levels; ++l) w.bm[l] = a.take<unsigned>(w.n_words[l]);
    for (int l = 1; l <= levels; ++l) w.prefix[l] = a.take<int>(w.n_words[l] / kBmBlk);
    w.bytes = align_up(a.used);
    return w;
}
static bool chain_geometry_ok(int levels, const int *shapes, const int *ksize, const int *stride, const int *pad, int batch, int *geo) {
    if (levels < 1 || levels > kChainMaxLevels) return false;
    for (int l = 1; l <= levels; ++l) {
        const int *ks = ksize + 3 * (l - 1), *st = stride + 3 * (l - 1), *pd = pad + 3 * (l - 1);
        const bool g1 = ks[0] == 3 && ks[1] == 3 && ks[2] == 3 && st[0] == 2 && st[1] == 2 && st[2] == 2;
        const bool g2 = ks[0] == 3 && ks[1] == 1 && ks[2] == 1 && st[0] == 2 && st[1] == 1 && st[2] == 1;
        if (!g1 && !g2) return false;
        if (l == 1 && !g1) return false;
        geo[l] = g1 ? 1 : 2;
        for (int d = 0; d < 3; ++d) {
            if (pd[d] < 0 || pd[d] > 1) return false;
            const int in = shapes[3 * (l - 1) + d], out = shapes[3 * l + d];
            if (in <= 0 || out != (in + 2 * pd[d] - (ks[d] - 1) - 1) / st[d] + 1) return false;
        }
        if (bm_cells(batch, shapes + 3 * l) >= (1ll << 32) - 64) return false;
    }
    return true;
}
}  // namespace sec

SEC_API size_t sec_rulebook_chain_workspace_bytes(int batch, int levels, const int *h_shapes) {
    if (batch <= 0 || levels < 1 || levels > kChainMaxLevels || !h_shapes) return 0;
    return carve_chain(nullptr, 0, batch, levels, h_shapes).bytes;
}

SEC_API int sec_rulebook_chain_sorted(const int *indices0, int n0, const int *n0_dev, int batch, int levels, const int *h_shapes,
                                      const int *h_ksize, const int *h_stride, const int *h_pad, const int *h_out_cap,
                                      int *const *h_nbr_out, int *const *h_out_indices, int *const *h_num_out,
                                      int *const *h_subm_nbr, const void *vox_workspace, size_t vox_workspace_bytes,
                                      int vox_num_points, int vox_max_voxels, int vox_max_points, const int *h_vox_grid3_zyx,
                                      int *site_map, void *workspace, size_t workspace_bytes, void *stream) {
    if (!indices0 || n0 <= 0 || batch <= 0 || !h_shapes || !h_ksize || !h_stride || !h_pad || !h_out_cap || !h_nbr_out ||
        !h_num_out || !h_subm_nbr || !h_out_indices)
        return SEC_E_INVALID;
    int geo[kChainMaxLevels + 1] = {0};
    if (!chain_geometry_ok(levels, h_shapes, h_ksize, h_stride, h_pad, batch, geo)) return SEC_E_UNSUPPORTED;
    for (int l = 1; l <= levels; ++l)
        if (!h_nbr_out[l - 1] || !h_num_out[l - 1] || !h_out_indices[l - 1] || h_out_cap[l - 1] <= 0) return SEC_E_INVALID;
    if (h_subm_nbr[0] && (!vox_workspace || !h_vox_grid3_zyx)) return SEC_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    ChainWorkspace w = carve_chain(workspace, workspace_bytes, batch, levels, h_shapes);
    if (!workspace || w.bytes > workspace_bytes) return SEC_E_WORKSPACE;

    ChainParams P{};
    P.levels = levels; P.batch = batch; P.n0 = n0; P.indices0 = indices0; P.n0_dev = n0_dev; P.ticket = w.ticket;
    int tile0 = 0, blk0 = div_up((long long)n0 * 8, kBlock);
    P.cand_blks = blk0;
    for (int l = 0; l <= levels; ++l) {
        ChainLevel &L = P.lv[l];
        for (int d = 0; d < 3; ++d) L.shape[d] = h_shapes[3 * l + d];
        L.out_cap = l == 0 ? n0 : h_out_cap[l - 1];
        L.subm_nbr = h_subm_nbr[l];
        if (l == 0) continue;
        for (int d = 0; d < 3; ++d) { L.ksize[d] = h_ksize[3 * (l - 1) + d]; L.pad[d] = h_pad[3 * (l - 1) + d]; }
        L.geo = geo[l]; L.kvol = geo[l] == 1 ? 27 : 3;
        L.n_words = w.n_words[l]; L.bm = w.bm[l]; L.prefix8 = w.prefix[l];
        L.status = w.status + tile0; L.tile0 = tile0; L.ntiles = w.ntiles[l]; L.wpt64 = w.wpt64[l];
        tile0 += w.ntiles[l];
        L.nbr_out = h_nbr_out[l - 1]; L.out_indices = h_out_indices[l - 1]; L.num_out = h_num_out[l - 1];
        // tables launch: SubM table of this level, then its conv table (levels >= 2 are built output side), sized for the capacity
        // (27 offsets: a thread takes x-triples, nine per row; 3 offsets: single entries)
        L.sblk0 = blk0; L.sblks = L.subm_nbr ? div_up((long long)L.out_cap * 9, kBlock * kTabU) : 0;
        blk0 += L.sblks;
        L.cblk0 = blk0; L.cblks = l >= 2 ? div_up((long long)L.out_cap * (L.kvol == 27 ? 9 : L.kvol), kBlock * kTabU) : 0;
        blk0 += L.cblks;
    }
    P.map_blk0 = blk0;
    int eblk = 0;
    for (int l = 1; l <= levels; ++l) { P.lv[l].eblk0 = eblk; P.lv[l].eblks = (int)(w.n_words[l] / kBlock); eblk += P.lv[l].eblks; }
    P.site_map = site_map;
    P.map_cells = site_map ? bm_cells(batch, h_shapes + 3 * levels) : 0;
    const int map_blks = site_map ? div_up(P.map_cells, (long long)kBlock) : 0;

    // geometry of the front launch: SubM level 0 over the voxeliser's table, level-1 bitmap
    RbGeom gs{}, gc{};
    const int k3[3] = {3, 3, 3};
    int rc = fill_geom(gs, h_shapes, nullptr, k3, nullptr, nullptr, nullptr, n0, batch);
    if (rc) return rc;
    if ((rc = fill_geom(gc, h_shapes, h_shapes + 3, h_ksize, h_stride, h_pad, nullptr, n0, batch))) return rc;
    const unsigned long long *keys = nullptr;
    const int *svid = nullptr;
    if (h_subm_nbr[0]) {
        uint32_t mask;
        for (int d = 0; d < 3; ++d) {
            if (h_vox_grid3_zyx[d] <= 0 || h_vox_grid3_zyx[d] > h_shapes[d]) return SEC_E_INVALID;
            gs.in_shape[d] = h_vox_grid3_zyx[d];       // the voxeliser keys its cells with ITS grid (see sec_rulebook_subm3d_after_voxelize)
        }
        if (!vox_table_of(vox_workspace, vox_workspace_bytes, vox_num_points, batch, vox_max_voxels, vox_max_points, &keys, &svid, &mask))
            return SEC_E_WORKSPACE;
        gs.mask = mask;
    }
    // 1. prep: level-1 bitmap + scan control := 0, SubM table of level 0 := -1
    {
        const long long nb = h_subm_nbr[0] ? (long long)n0 * 27 : 0;
        long long most = w.zero16 > nb / 4 ? w.zero16 : nb / 4;
        int blocks = div_up(most > 0 ? most : 1, kBlock);
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_chain_prep, dim3(blocks), dim3(kBlock), 0, st, reinterpret_cast<uint4 *>(w.bm[1]), w.zero16,
                           reinterpret_cast<int4 *>(h_subm_nbr[0]), nb / 4, h_subm_nbr[0] ? h_subm_nbr[0] + (nb / 4) * 4 : nullptr,
                           (int)(nb & 3));
    }
    // 2. front: SubM level 0 || level-1 bitmap || -1 fill of the level-1 conv table
    {
        const int nb_subm = h_subm_nbr[0] ? div_up((long long)n0 * 14, kBlock) : 0;
        const int nb_set = div_up((long long)n0 * 4, kBlock);
        const long long fill_words = (long long)P.lv[1].out_cap * 27;
        int groups = div_up(nb_subm, 7);                 // groups of 7 + 2 + 1 workgroups
        if (div_up(nb_set, 2) > groups) groups = div_up(nb_set, 2);
        hipLaunchKernelGGL(k_chain_front, dim3(groups * 10), dim3(kBlock), 0, st, indices0, gs, gc, n0_dev, keys, svid,
                           h_subm_nbr[0], w.bm[1], nb_subm, nb_set, P.lv[1].nbr_out, fill_words);
    }
    // 3. bitmaps of the levels above, each from the one below
    for (int l = 2; l <= levels; ++l) {
        RbGeom g{};
        if ((rc = fill_geom(g, h_shapes + 3 * (l - 1), h_shapes + 3 * l, h_ksize + 3 * (l - 1), h_stride + 3 * (l - 1), h_pad + 3 * (l - 1),
                            nullptr, 0, batch)))
            return rc;
        const RbFill none{nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0};
        const unsigned nb = (unsigned)(w.n_words[l] / kBlock);
        if (geo[l] == 1) hipLaunchKernelGGL(k_bm_dilate<1>, dim3(nb), dim3(kBlock), 0, st, w.bm[l - 1], w.n_words[l - 1], g, w.bm[l], w.n_words[l], none);
        else hipLaunchKernelGGL(k_bm_dilate<2>, dim3(nb), dim3(kBlock), 0, st, w.bm[l - 1], w.n_words[l - 1], g, w.bm[l], w.n_words[l], none);
    }
    // 4. rank prefixes of every level, 5. every table
    hipLaunchKernelGGL(k_chain_scan, dim3(tile0), dim3(kBlock), 0, st, P);
    hipLaunchKernelGGL(k_chain_emit, dim3(eblk), dim3(kBlock), 0, st, P);
    hipLaunchKernelGGL(k_chain_tables, dim3(P.map_blk0 + map_blks), dim3(kBlock), 0, st, P);
    return check_launch();
}
