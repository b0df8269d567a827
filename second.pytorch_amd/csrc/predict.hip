// Device-resident post-processing of VoxelNet.predict (second/pytorch/models/voxelnet.py:377-645):
//   sec_predict_select   : best class per anchor, score threshold, top-k by score, sorted descending
//                          (torch.sigmoid / max / masked_select / topk of voxelnet.py:545-569 and
//                          box_torch_ops.py:497-501) -- one workgroup per frame: 8-bit radix select over
//                          order-preserving keys, ordered tie handling, bitonic sort of the k survivors in LDS;
//   sec_predict_decode   : gather + second_box_decode (box_torch_ops.py:56-101) of the selected anchors, NMS input
//                          rows (rotated (x,y,w,l,r,score) or standup (x1,y1,x2,y2,score)), direction argmax;
//   sec_predict_finalize : gather of the NMS survivors, direction fix (limit_period, voxelnet.py:598-607),
//                          post_center_range mask (:611-621).
// The reference runs ~60 tiny torch kernels plus a GPU->CPU->GPU round trip here; these three launches plus the
// two NMS launches keep everything on the device with fixed shapes (hipGraph friendly).
// Tensors are addressed as logit(b, anchor n = (a, y, x), c) = base[b*sb + a*sa + y*sy + x*sx + c*sc] so the RPN head
// output is consumed in place (no permute / contiguous copies).
#include "common.hpp"

namespace sec {

struct View5 {          // element strides of a [B, A, H, W, C] view
    long long sb, sa, sy, sx, sc;
    // LAZY heads (sec_predict_select_lazy / _decode_lazy): the producer (sec_conv1x1_chain_nhwc_tiles with background == NULL) wrote
    // only the tiles its live list names.  `live` [B][tiles] (8 x 16 tiles, row-major) holds bit 4 = "this tile was written"; for
    // every other tile the element is read from the empty frame's head map instead -- `bg` = its element offset relative to the
    // view's base pointer (same sa / sy / sx / sc, no frame stride): exactly the value the producer's copy would have put there.
    long long bg;
    const unsigned short *live;
    int tiles_x, tpf;
};
// element offset of frame b as seen from pixel (y, x): the frame itself, or the empty frame's map for a tile that was never written
__device__ __forceinline__ long long frame_off(const View5 &v, int b, int y, int x) {
    if (v.live) {
        const unsigned m = v.live[(long long)b * v.tpf + (y >> 3) * v.tiles_x + (x >> 4)];
        if (!((m >> 4) & 1u)) return v.bg;
    }
    return (long long)b * v.sb;
}
struct PredGeom {
    int batch, A, H, W, nc;   // anchors per location, feature map, classes
};

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
template <> __device__ __forceinline__ float ldf(const __half *p) { return __half2float(*p); }

__device__ __forceinline__ unsigned f2key(float f) {   // order-preserving float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// 16-bit order-preserving keys of 16-bit logits.  bf16: the high half of f2key(float(x)) (a bf16 is the high half of its float);
// fp16: the same sign trick on the half's own bits (all 16 bits are significant, none are lost).  key16_score() is the sigmoid of
// the logit a key stands for.
template <typename T> __device__ __forceinline__ unsigned key16_of(float best);
template <> __device__ __forceinline__ unsigned key16_of<__hip_bfloat16>(float best) { return f2key(best) >> 16; }
template <> __device__ __forceinline__ unsigned key16_of<__half>(float best) {
    const unsigned u = __half_as_ushort(__float2half_rn(best));      // exact: `best` came from a half
    return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
}
template <> __device__ __forceinline__ unsigned key16_of<float>(float best) { return f2key(best) >> 16; }   // unused (fp32 heads take the 32-bit path)
template <typename T> __device__ __forceinline__ float key16_logit(unsigned k);
template <> __device__ __forceinline__ float key16_logit<__hip_bfloat16>(unsigned k) {
    return key2f((k << 16) | ((k & 0x8000u) ? 0u : 0xffffu));        // low half of f2key: 0 for a non-negative bf16, 0xffff for a negative one
}
template <> __device__ __forceinline__ float key16_logit<__half>(unsigned k) {
    const unsigned u = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
    return __half2float(__ushort_as_half((unsigned short)u));
}
template <> __device__ __forceinline__ float key16_logit<float>(unsigned k) { return key16_logit<__hip_bfloat16>(k); }

template <typename T>
__device__ __forceinline__ unsigned anchor_key(const T *cls, const View5 &v, const PredGeom &g, int b, int n, int *label) {
    int x = n % g.W;
    int t = n / g.W;
    int y = t % g.H;
    int a = t / g.H;
    const T *p = cls + frame_off(v, b, y, x) + a * v.sa + y * v.sy + x * v.sx;
    float best = ldf(p);
    int lab = 0;
    for (int c = 1; c < g.nc; ++c) {
        float f = ldf(p + c * v.sc);
        if (f > best) { best = f; lab = c; }
    }
    if (label) *label = lab;
    return f2key(best);
}

constexpr int kSelThreads = 1024;
constexpr int kCandCap = 2048;      // k_predict_select_reg: keys kept in LDS once that few remain above the bisection bound

// pass 0 (whole chip): compact, coalesced key array (the RPN head is channels-last: one anchor's logit per 128-byte
// pixel row, far too scattered to be re-read by the single workgroup that owns a frame in the select kernel)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_predict_keys(const T *__restrict__ cls, View5 v, PredGeom g,
                                                        unsigned *__restrict__ keys) {
    const int N = g.A * g.H * g.W;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.batch * N) return;
    keys[t] = anchor_key(cls, v, g, (int)(t / N), (int)(t % N), nullptr);
}

// 16-bit form for 16-bit logits (the low half of such a key is implied by its sign bit): half the bytes, and two
// consecutive anchors' keys arrive in one dword load in k_predict_select_reg.  Frame stride `ns` is N rounded up to even.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_predict_keys16(const T *__restrict__ cls, View5 v, PredGeom g, int ns,
                                                          unsigned short *__restrict__ keys) {
    const int N = g.A * g.H * g.W;
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)g.batch * ns) return;
    const int b = (int)(t / ns), n = (int)(t % ns);
    keys[t] = n < N ? (unsigned short)key16_of<T>(key2f(anchor_key(cls, v, g, b, n, nullptr))) : (unsigned short)0;
}

// one workgroup per frame
template <typename T, bool KEY16>
__global__ __launch_bounds__(kSelThreads) void k_predict_select(const T *__restrict__ cls, View5 v, PredGeom g, int K,
                                                                float score_thr, const unsigned *__restrict__ keys,
                                                                int *__restrict__ top_idx,
                                                                float *__restrict__ top_score, int *__restrict__ top_label,
                                                                int *__restrict__ counts, unsigned thr_key = 0u) {
    __shared__ unsigned hist[256];
    __shared__ unsigned ckey[kSelThreads];
    __shared__ int cidx[kSelThreads];
    __shared__ int wsum[kSelThreads / 64];
    __shared__ unsigned s_prefix, s_need;
    __shared__ int s_cnt, s_eqbase;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = g.A * g.H * g.W;
    const unsigned *fk = keys + (size_t)b * N;
    if (K > kSelThreads) K = kSelThreads;
    if (K > N) K = N;
    // ---- radix select of the K-th largest key (8 bits per pass, MSB first).  bf16 logits only carry 16 key bits
    //      (the low half of the fp32 pattern is zero): two passes instead of four.
    constexpr int UNR = 8;                       // independent key loads in flight per thread
    const int first_shift = 24, last_shift = sizeof(T) == 2 && KEY16 ? 16 : 0;
    unsigned prefix = 0, mask = 0, need = K;
    // Threshold shortcut (round 5; the 16-bit paths have had it since round 3): `thr_key` is a key no larger than that of any logit
    // whose sigmoid reaches the score threshold.  A trained head leaves a few hundred such anchors per frame: when they are at most
    // K, they ARE the selection (rows behind counts[b] are unspecified by contract), and the four radix sweeps are skipped.
    bool shortcut = false;
    if (thr_key > 1u) {
        if (tid == 0) hist[0] = 0u;
        __syncthreads();
        unsigned c = 0;
        for (int n0 = 0; n0 < N; n0 += kSelThreads * UNR) {
            unsigned kk[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int n = n0 + j * kSelThreads + tid;
                kk[j] = ld_sel(fk, n, n < N, 0u);
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) c += kk[j] >= thr_key ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0 && c) atomicAdd(&hist[0], c);
        __syncthreads();
        shortcut = hist[0] <= (unsigned)K;
        __syncthreads();
        if (shortcut) { prefix = thr_key - 1u; need = 0u; }
    }
    for (int shift = first_shift; !shortcut && shift >= last_shift; shift -= 8) {
        for (int i = tid; i < 256; i += kSelThreads) hist[i] = 0;
        __syncthreads();
        for (int n0 = 0; n0 < N; n0 += kSelThreads * UNR) {
            unsigned kk[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                int n = n0 + j * kSelThreads + tid;
                kk[j] = ld_sel(fk, n, n < N, 0u);
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                int n = n0 + j * kSelThreads + tid;
                bool act = n < N && (kk[j] & mask) == prefix;
                unsigned d = (kk[j] >> shift) & 255u;
                if (shift == first_shift) {
                    // sign/exponent byte: a handful of distinct values -> wave-aggregated update (a uniform digit
                    // costs one LDS atomic per wave instead of a 64-way conflict)
                    unsigned long long todo = __ballot(act);
                    while (todo) {
                        int leader = __ffsll((long long)todo) - 1;
                        unsigned d0 = __shfl(d, leader, 64);
                        unsigned long long same = __ballot(act && d == d0) & todo;
                        if (lane == leader) atomicAdd(&hist[d0], (unsigned)__popcll(same));
                        todo &= ~same;
                    }
                } else if (act) {
                    atomicAdd(&hist[d], 1u);      // mantissa bytes are well spread and only the prefix matches are active
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0;
            int bin = 255;
            for (; bin > 0; --bin) {
                if (acc + hist[bin] >= need) break;
                acc += hist[bin];
            }
            s_prefix = prefix | ((unsigned)bin << shift);
            s_need = need - acc;     // how many still to take inside this bin
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T_key = prefix;   // K-th largest key; take all keys > T and the first `need` (by index) equal to T
    // ---- ordered compaction into LDS
    if (tid == 0) { s_cnt = 0; s_eqbase = 0; }
    ckey[tid] = 0u;
    cidx[tid] = 0x7fffffff;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += kSelThreads * UNR) {
        unsigned kk[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            int n = n0 + j * kSelThreads + tid;
            kk[j] = ld_sel(fk, n, n < N, 0u);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            int n = n0 + j * kSelThreads + tid;
            unsigned key = kk[j];
            bool gt = n < N && key > T_key, eq = n < N && key == T_key;
            unsigned long long em = __ballot(eq);
            if (__syncthreads_or(eq)) {          // chunks without a tie candidate (almost all) skip the ordered ranking
                if (lane == 0) wsum[wv] = __popcll(em);
                __syncthreads();
                int ebase = s_eqbase;
                for (int w2 = 0; w2 < wv; ++w2) ebase += wsum[w2];
                int erank = ebase + __popcll(em & ((1ull << lane) - 1ull));
                eq = eq && erank < (int)need;
                __syncthreads();
                if (tid == 0) {
                    int tot = 0;
                    for (int w2 = 0; w2 < kSelThreads / 64; ++w2) tot += wsum[w2];
                    s_eqbase += tot;
                }
                __syncthreads();
            }
            if (gt || eq) {
                int pos = atomicAdd(&s_cnt, 1);
                if (pos < kSelThreads) { ckey[pos] = key; cidx[pos] = n; }
            }
        }
    }
    __syncthreads();
    // ---- bitonic sort of the (key desc, idx asc) pairs; unused slots hold key 0 / idx INT_MAX and sink to the end
    for (int size = 2; size <= kSelThreads; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            int partner = tid ^ stride;
            if (partner > tid) {
                unsigned ka = ckey[tid], kb = ckey[partner];
                int ia = cidx[tid], ib = cidx[partner];
                bool a_first = ka > kb || (ka == kb && ia < ib);       // a should precede b in descending order
                bool up = (tid & size) == 0;                            // this run is sorted "descending-first"
                if (up ? !a_first : a_first) {
                    ckey[tid] = kb; ckey[partner] = ka;
                    cidx[tid] = ib; cidx[partner] = ia;
                }
            }
            __syncthreads();
        }
    }
    // ---- outputs
    if (tid < K) {
        int n = cidx[tid];
        float sc = sigmoidf_(key2f(ckey[tid]));
        int lab = 0;
        if (g.nc > 1 && n < N) anchor_key(cls, v, g, b, n, &lab);
        top_idx[(size_t)b * K + tid] = n < N ? n : 0;
        top_score[(size_t)b * K + tid] = sc;
        top_label[(size_t)b * K + tid] = lab;
    }
    // number of selected anchors with score >= threshold (a prefix of the sorted list)
    bool ok = tid < K && cidx[tid] < N && sigmoidf_(key2f(ckey[tid])) >= score_thr;
    unsigned long long m = __ballot(ok);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w2 = 0; w2 < kSelThreads / 64; ++w2) tot += wsum[w2];
        counts[b] = tot;
    }
}


// ---- register-resident select (frames of up to 1024 * KPT anchors: car.fhd has 70400) ----------------------------------
// k_predict_select walks the frame's key array three times (two radix passes + compaction, 9 dependent load rounds each),
// re-synchronises the workgroup once per 1024 keys to rank ties, and finishes with a 55-barrier bitonic sort.  Here every
// thread loads its KPT keys ONCE (wave-contiguous segments, so "first by index" among equal keys is a prefix over waves),
// the radix passes, tie ranking and compaction run out of registers, and the sort does its in-wave steps with shuffles
// (10 LDS exchange steps instead of 55).  Same outputs as k_predict_select.
// INDIRECT (second stage of the chunked select): `keys` are the per-chunk candidate keys written by k_predict_select_chunk
// (n_keys of them per frame, chunk-major, ascending anchor index among equal keys inside and across chunks -- which is all the
// tie ranking needs), slot_anchor their anchor ids.
template <typename T, int KP, bool INDIRECT = false>     // 16-bit logits (bf16 / fp16: key16_of); KP = key PAIRS (dwords) per thread
__global__ __launch_bounds__(kSelThreads) void k_predict_select_reg(const T *__restrict__ cls, View5 v, PredGeom g, int K,
                                                                    float score_thr, const unsigned short *__restrict__ keys,
                                                                    int ns, int *__restrict__ top_idx,
                                                                    float *__restrict__ top_score, int *__restrict__ top_label,
                                                                    int *__restrict__ counts, int n_keys = 0,
                                                                    const int *__restrict__ slot_anchor = nullptr, unsigned thr16 = 0u) {
    __shared__ unsigned ckey[kSelThreads];
    __shared__ int cidx[kSelThreads];
    __shared__ int wsum[kSelThreads / 64];
    __shared__ int wred[2][kSelThreads / 64];
    __shared__ int s_cnt;
    __shared__ unsigned short cand_key[kCandCap];
    __shared__ int cand_idx[kCandCap];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = INDIRECT ? n_keys : g.A * g.H * g.W;
    const unsigned *fk2 = reinterpret_cast<const unsigned *>(keys + (size_t)b * ns);   // ns is even: dword aligned
    if (K > kSelThreads) K = kSelThreads;
    if (K > N) K = N;
#ifdef SEC_SELECT_TIMING
    long long tstamp[8]; int tsi = 0;
#define SEC_TS() do { __syncthreads(); tstamp[tsi++] = clock64(); } while (0)
#else
#define SEC_TS() do {} while (0)
#endif
    SEC_TS();
    const int n_base = wv * (KP * 128) + lane * 2;    // pair i of this thread = anchors n_base + 128 i and + 1
    unsigned k2[KP];                                  // low half: even anchor's key, high half: the odd one's
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int n = n_base + i * 128;
        k2[i] = ld_sel(fk2, n >> 1, n < ns, 0u);
    }
    SEC_TS();
    // K-th largest 16-bit key by bisection on its bits, MSB first (no histogram, no atomics).  Counting sweeps run over
    // all registers only while more than CAND_CAP keys lie at or above the current lower bound; then those few are
    // compacted into LDS (two per thread) and the remaining bits are resolved on them.  Out-of-range slots hold key 0
    // and are never counted because every probe is >= 1.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto block_sum = [&](int c, int it) -> int {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) wred[it & 1][wv] = c;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < kSelThreads / 64; ++w2) tot += wred[it & 1][w2];
        return tot;
    };
    auto count_ge = [&](unsigned t, int it) -> int {
        // packed 16-bit arithmetic, three VALU ops per key pair: min(1, sat(k - (t - 1))) is 1 exactly when k >= t
        const us2 tm1 = {(unsigned short)(t - 1), (unsigned short)(t - 1)}, one = {1, 1};
        us2 acc2 = {0, 0};
#pragma unroll
        for (int i = 0; i < KP; ++i)
            acc2 += __builtin_elementwise_min(__builtin_elementwise_sub_sat(__builtin_bit_cast(us2, k2[i]), tm1), one);
        return block_sum((int)acc2.x + (int)acc2.y, it);
    };
    unsigned cand = 0;
    int it = 0, bit = 15, above = N;                  // above = number of keys >= cand
    // thr16 (see k_predict_select_chunk): no more than K keys at or above the threshold's key -> they are the selection, no bisection
    bool quick = false;
    if (thr16 > 0u) {
        const int c = count_ge(thr16, it++);
        if (c <= K) { quick = true; cand = thr16; above = c; bit = -1; }
    }
    for (; bit >= 0 && above > kCandCap; --bit, ++it) {
        const unsigned t = cand | (1u << bit);
        const int c = count_ge(t, it);
        if (c >= K) { cand = t; above = c; }
    }
    bool sorted_ready = false;                        // the sort buffer already holds every key >= T (ties included)
    ckey[tid] = 0u;
    cidx[tid] = 0x7fffffff;
    if (quick) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < KP; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const unsigned kj = (k2[i] >> (hf * 16)) & 0xffffu;
                const int n = n_base + i * 128 + hf;
                const bool hit = n < N && kj >= cand;
                const unsigned long long m = __ballot(hit);
                if (m) {                               // one LDS atomic per wave; the sort orders the slots
                    int p0 = 0;
                    if (lane == 0) p0 = atomicAdd(&s_cnt, __popcll(m));
                    p0 = __shfl(p0, 0, 64);
                    const int pos = p0 + __popcll(m & lt);
                    if (hit && pos < kSelThreads) { ckey[pos] = (kj << 16) | ((kj & 0x8000u) ? 0u : 0xffffu); cidx[pos] = n; }
                }
            }
        sorted_ready = true;
    } else if (bit >= 0) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KP; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const unsigned kj = (k2[i] >> (hf * 16)) & 0xffffu;
                const int n = n_base + i * 128 + hf;
                if (n < N && kj >= cand) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < kCandCap) { cand_key[pos] = (unsigned short)kj; cand_idx[pos] = n; }
                }
            }
        __syncthreads();
        const int m = s_cnt < kCandCap ? s_cnt : kCandCap;
        unsigned c2[kCandCap / kSelThreads];
        int ci[kCandCap / kSelThreads];
#pragma unroll
        for (int u = 0; u < kCandCap / kSelThreads; ++u) {
            const int p = tid + u * kSelThreads;
            c2[u] = p < m ? (unsigned)cand_key[p] : 0u;
            ci[u] = p < m ? cand_idx[p] : 0x7fffffff;
        }
        auto count_ge_c = [&](unsigned t, int it2) -> int {
            int c = 0;
#pragma unroll
            for (int u = 0; u < kCandCap / kSelThreads; ++u) c += c2[u] >= t ? 1 : 0;
            return block_sum(c, it2);
        };
        for (; bit >= 0; --bit, ++it) {
            const unsigned t = cand | (1u << bit);
            const int c = count_ge_c(t, it);
            if (c >= K) { cand = t; above = c; }
        }
        if (above <= kSelThreads && cand > 0) {       // everything at or above T fits the sort buffer: the sort ranks the ties
            if (tid == 0) s_cnt = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kCandCap / kSelThreads; ++u)
                if (c2[u] >= cand) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    ckey[pos] = (c2[u] << 16) | ((c2[u] & 0x8000u) ? 0u : 0xffffu);
                    cidx[pos] = ci[u];
                }
            sorted_ready = true;
        }
    }
    const unsigned prefix = cand;
    SEC_TS();
    if (!sorted_ready) {
    const unsigned need = cand == 0xffffu ? (unsigned)K : (unsigned)(K - count_ge(cand + 1, it));
    const unsigned T_key = prefix;   // K-th largest (16-bit) key; take all keys > T and the first `need` (by index) equal to T
    // ties: a wave's keys are a contiguous index range (pair i, lane, half = ascending anchor index), so its first tie
    // ranks after all ties of the waves before it
    int my_eq = 0;
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            my_eq += __popcll(__ballot(n_base + i * 128 + hf < N && ((k2[i] >> (hf * 16)) & 0xffffu) == T_key));
    if (tid == 0) s_cnt = 0;
    if (lane == 0) wsum[wv] = my_eq;
    __syncthreads();
    int erun = 0;
    for (int w2 = 0; w2 < wv; ++w2) erun += wsum[w2];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int n = n_base + i * 128;
        const unsigned ka = k2[i] & 0xffffu, kb = k2[i] >> 16;
        const bool eqa = n < N && ka == T_key, eqb = n + 1 < N && kb == T_key;
        const unsigned long long ma = __ballot(eqa), mb = __ballot(eqb);
        const int before = erun + __popcll(ma & lt) + __popcll(mb & lt);
        erun += __popcll(ma) + __popcll(mb);
        const bool ta = (n < N && ka > T_key) || (eqa && before < (int)need);
        const bool tb = (n + 1 < N && kb > T_key) || (eqb && before + (eqa ? 1 : 0) < (int)need);
        // full key: f2key of a 16-bit float has low half 0 (non-negative input) or 0xffff (negative input)
        if (ta) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < kSelThreads) { ckey[pos] = (ka << 16) | ((ka & 0x8000u) ? 0u : 0xffffu); cidx[pos] = n; }
        }
        if (tb) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < kSelThreads) { ckey[pos] = (kb << 16) | ((kb & 0x8000u) ? 0u : 0xffffu); cidx[pos] = n + 1; }
        }
    }
    }   // !sorted_ready
    __syncthreads();
    SEC_TS();
    // ---- bitonic sort of (key desc, idx asc); strides < 64 stay inside the wave (shuffles), the rest go through LDS
    unsigned mk = ckey[tid];
    int mi = cidx[tid];
    for (int size = 2; size <= kSelThreads; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            unsigned ok_;
            int oi;
            if (stride >= 64) {
                __syncthreads();
                ckey[tid] = mk; cidx[tid] = mi;
                __syncthreads();
                ok_ = ckey[tid ^ stride]; oi = cidx[tid ^ stride];
            } else {
                ok_ = (unsigned)__shfl_xor((int)mk, stride, 64);
                oi = __shfl_xor(mi, stride, 64);
            }
            const bool lower = (tid & stride) == 0;                    // this thread holds the first slot of the pair
            const bool mine_first = mk > ok_ || (mk == ok_ && mi < oi);   // my element precedes the partner's (descending)
            const bool up = (tid & size) == 0;                         // this run is sorted descending-first
            const bool keep_mine = (lower == up) ? mine_first : !mine_first;
            if (!keep_mine) { mk = ok_; mi = oi; }
        }
    }
    SEC_TS();
    // ---- outputs
    if (INDIRECT) {                                   // slot -> anchor id (empty candidate slots carry 0x7fffffff)
        const int a_ = (mi < N) ? slot_anchor[(size_t)b * N + mi] : 0x7fffffff;
        mi = (a_ < g.A * g.H * g.W) ? a_ : 0x7fffffff;
    }
    const int n_valid = INDIRECT ? 0x7fffffff : N;    // below: mi < n_valid == a real anchor
    if (tid < K) {
        float sc = mi < n_valid ? sigmoidf_(key16_logit<T>(mk >> 16)) : 0.0f;      // empty slots (fewer than K selected): score 0, anchor 0
        int lab = 0;
        if (g.nc > 1 && mi < n_valid) anchor_key(cls, v, g, b, mi, &lab);
        top_idx[(size_t)b * K + tid] = mi < n_valid ? mi : 0;
        top_score[(size_t)b * K + tid] = sc;
        top_label[(size_t)b * K + tid] = lab;
    }
    const bool ok = tid < K && mi < n_valid && sigmoidf_(key16_logit<T>(mk >> 16)) >= score_thr;
    const unsigned long long m = __ballot(ok);
    __syncthreads();
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w2 = 0; w2 < kSelThreads / 64; ++w2) tot += wsum[w2];
        counts[b] = tot;
    }
#ifdef SEC_SELECT_TIMING
    SEC_TS();
    if (b == 0 && tid == 0)
        printf("[select_reg] load %lld  bisect %lld  compact %lld  sort %lld  out %lld  (clock64 ticks)\n", tstamp[1] - tstamp[0],
               tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4]);
#endif
#undef SEC_TS
}

// ---- chunked select, stage 1 (whole chip) -------------------------------------------------------------------------------
// One workgroup per (frame, chunk of 8192 anchors): its top-min(K, chunk) keys -- everything above the chunk's K-th largest key
// T plus the first ties at T by anchor index -- written in ASCENDING ANCHOR ORDER (deterministic ballot-prefix slots) to
// cand_key / cand_idx [frame][chunk][1024], zero / 0x7fffffff padded.  The frame's top K is a subset of the union of its chunks'
// top K, with the same tie order, so stage 2 (k_predict_select_reg<INDIRECT>) selects and sorts ~9 k candidates instead of
// bisecting 70 k keys in one workgroup: 65 us -> two launches of ~6 and ~17 us.
constexpr int kSelChunk = 8192;                  // 16 waves x 4 key pairs x 128 (KP = 4)
// Round 3: (a) the keys are computed HERE from the head output (the separate whole-chip k_predict_keys16 launch, 12.5 us, is gone
// from this path; a chunk's 8192 anchors are 8192 scattered 2-byte reads either way); (b) `thr16` > 0 is a 16-bit key at or below
// the key of every logit whose score reaches the caller's threshold (computed conservatively on the host): when no more than K
// keys of the chunk lie at or above it -- the normal case for a trained head: a few hundred candidates per frame -- they ARE the
// chunk's contribution and the 16-step bisection is skipped.  Entries below the threshold can then be missing from the frame's
// top K; they lie behind counts[b] and were never part of the reference's result (voxelnet.py:551-570 masks by the threshold
// before its topk).
// KP = key pairs per thread: 4 (8192-anchor chunks, the usual heads) or 36 (73 728-anchor chunks: heads of > 600 k anchors per
// frame -- nuScenes all.fhd has 1.23 M -- whose 8192-anchor chunks would hand the second stage more candidate slots than it holds).
template <typename T, int KP>
__global__ __launch_bounds__(kSelThreads) void k_predict_select_chunk(const T *__restrict__ cls, View5 v, PredGeom g, int N, int K,
                                                                      int chunks, unsigned thr16, unsigned short *__restrict__ cand_key,
                                                                      int *__restrict__ cand_idx) {
    constexpr int kChunk = kSelThreads * 2 * KP;
    __shared__ int sweep_tot[20];
    __shared__ int w_eq[kSelThreads / 64], w_gt[kSelThreads / 64];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = c * kChunk;
    const int nloc = min(kChunk, N - base);                            // > 0 by construction of the grid
    const int Kc = K < nloc ? K : nloc;
    const int n_base = wv * (KP * 128) + lane * 2;
    unsigned k2[KP];
    // (a, y, x) of the thread's first anchor by division, of the others by stepping (anchor n = (a * H + y) * W + x): the runtime
    // divisions of anchor_key() were a third of this kernel's instructions
    int ax, ay, aa;
    {
        const int n0 = base + n_base;
        ax = n0 % g.W;
        const int t0 = n0 / g.W;
        ay = t0 % g.H;
        aa = t0 / g.H;
    }
    // Offsets first, then the loads of a batch of four pairs back to back (an anchor past the frame reads element 0 and its key is
    // forced to 0 afterwards): `a0 < N ? load : 0` compiles to a branch with s_waitcnt vmcnt(0) behind every load -- eight dependent
    // round trips per thread at KP = 4, most of this kernel's 18 us.
    constexpr int PB = KP < 4 ? KP : 4;
    static_assert(KP % PB == 0, "pairs per thread must be a multiple of the load batch");
#pragma unroll
    for (int i0 = 0; i0 < KP; i0 += PB) {
        long long off[2 * PB];
        bool ok[2 * PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int a0 = base + n_base + (i0 + j) * 128;
            int bx = ax + 1, by = ay, ba = aa;             // the pair's second anchor
            if (bx >= g.W) { bx = 0; if (++by >= g.H) { by = 0; ++ba; } }
            ok[2 * j] = a0 < N;
            ok[2 * j + 1] = a0 + 1 < N;                                 // slots beyond the frame hold key 0
            off[2 * j] = ok[2 * j] ? frame_off(v, b, ay, ax) + (long long)aa * v.sa + (long long)ay * v.sy + (long long)ax * v.sx : 0;
            off[2 * j + 1] = ok[2 * j + 1] ? frame_off(v, b, by, bx) + (long long)ba * v.sa + (long long)by * v.sy + (long long)bx * v.sx : 0;
            ax += 128;
            while (ax >= g.W) { ax -= g.W; if (++ay >= g.H) { ay = 0; ++aa; } }
        }
        float best[2 * PB];
#pragma unroll
        for (int j = 0; j < 2 * PB; ++j) best[j] = ldf(cls + off[j]);
        for (int c2 = 1; c2 < g.nc; ++c2) {                              // as anchor_key()
            float f[2 * PB];
#pragma unroll
            for (int j = 0; j < 2 * PB; ++j) f[j] = ldf(cls + off[j] + (ok[j] ? (long long)c2 * v.sc : 0));
#pragma unroll
            for (int j = 0; j < 2 * PB; ++j) if (f[j] > best[j]) best[j] = f[j];
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const unsigned ka = ok[2 * j] ? key16_of<T>(best[2 * j]) : 0u;
            const unsigned kb = ok[2 * j + 1] ? key16_of<T>(best[2 * j + 1]) : 0u;
            k2[i0 + j] = ka | (kb << 16);
        }
    }
    unsigned short *ok_ = cand_key + ((size_t)b * chunks + c) * kSelThreads;
    int *oi_ = cand_idx + ((size_t)b * chunks + c) * kSelThreads;
    ok_[tid] = 0;
    oi_[tid] = 0x7fffffff;
    if (tid < 20) sweep_tot[tid] = 0;
    __syncthreads();
    auto block_sum = [&](int x, int it) -> int {
        x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
        x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
        x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);    // row_half_mirror
        x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);    // row_mirror
        const int ws_ = __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
                        __builtin_amdgcn_readlane(x, 48);
        if (lane == 0) atomicAdd(&sweep_tot[it], ws_);
        __syncthreads();
        return sweep_tot[it];
    };
    auto count_ge = [&](unsigned t, int it) -> int {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < KP; ++i) cnt += ((k2[i] & 0xffffu) >= t ? 1 : 0) + ((k2[i] >> 16) >= t ? 1 : 0);
        return block_sum(cnt, it);
    };
    unsigned T_key = 0;
    int it = 0;
    bool quick = false;
    if (thr16 > 0u) {
        quick = count_ge(thr16, it++) <= Kc;
        if (quick) T_key = thr16;
    }
    if (!quick)
        for (int bit = 15; bit >= 0; --bit, ++it) {
            const unsigned t = T_key | (1u << bit);
            if (count_ge(t, it) >= Kc) T_key = t;
        }
    const int above = T_key == 0xffffu ? 0 : count_ge(T_key + 1, it);
    const int need = Kc - above;                                          // ties at T to keep, by ascending anchor index
    int my_eq = 0, my_gt = 0;
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const unsigned kj = (k2[i] >> (hf * 16)) & 0xffffu;
            my_eq += __popcll(__ballot(kj == T_key));
            my_gt += __popcll(__ballot(kj > T_key));
        }
    if (lane == 0) { w_eq[wv] = my_eq; w_gt[wv] = my_gt; }
    __syncthreads();
    int erun = 0, slot = 0;
    for (int w2 = 0; w2 < wv; ++w2) {
        const int left = need - erun;
        slot += w_gt[w2] + (left <= 0 ? 0 : (left < w_eq[w2] ? left : w_eq[w2]));
        erun += w_eq[w2];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int n = n_base + i * 128;
        const unsigned ka = k2[i] & 0xffffu, kb = k2[i] >> 16;
        const bool eqa = ka == T_key, eqb = kb == T_key;
        const unsigned long long ma = __ballot(eqa), mb = __ballot(eqb);
        const int before = erun + __popcll(ma & lt) + __popcll(mb & lt);
        erun += __popcll(ma) + __popcll(mb);
        const bool ta = ka > T_key || (eqa && before < need);
        const bool tb = kb > T_key || (eqb && before + (eqa ? 1 : 0) < need);
        const unsigned long long qa = __ballot(ta), qb = __ballot(tb);
        const unsigned long long both_lt = lt;
        // ascending anchor order inside the pair row: lane L's two anchors (2L, 2L + 1) come before lane L + 1's
        const int pa = slot + __popcll(qa & both_lt) + __popcll(qb & both_lt);
        const int pb = pa + (ta ? 1 : 0);
        slot += __popcll(qa) + __popcll(qb);
        if (ta && pa < kSelThreads) { ok_[pa] = (unsigned short)ka; oi_[pa] = base + n; }
        if (tb && pb < kSelThreads) { ok_[pb] = (unsigned short)kb; oi_[pb] = base + n + 1; }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_predict_decode(const T *__restrict__ box, View5 vb, const T *__restrict__ dir,
                                                          View5 vd, int ndir, PredGeom g, int K,
                                                          const float *__restrict__ anchors, const int *__restrict__ top_idx,
                                                          const float *__restrict__ top_score, int rotate,
                                                          float *__restrict__ dec, float *__restrict__ dets,
                                                          int *__restrict__ dir_label) {
    int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= g.batch * K) return;
    int b = t / K;
    int n = top_idx[t];
    int x = n % g.W;
    int q = n / g.W;
    int y = q % g.H;
    int a = q / g.H;
    const T *pb = box + frame_off(vb, b, y, x) + a * vb.sa + y * vb.sy + x * vb.sx;
    float e[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) e[c] = ldf(pb + c * vb.sc);
    const float *an = anchors + (size_t)n * 7;
    float xa = an[0], ya = an[1], za = an[2], wa = an[3], la = an[4], ha = an[5], ra = an[6];
    float diag = sqrtf(__fadd_rn(__fmul_rn(la, la), __fmul_rn(wa, wa)));
    float o[7];
    o[0] = __fadd_rn(__fmul_rn(e[0], diag), xa);
    o[1] = __fadd_rn(__fmul_rn(e[1], diag), ya);
    o[2] = __fadd_rn(__fmul_rn(e[2], ha), za);
    o[3] = __fmul_rn(expf(e[3]), wa);
    o[4] = __fmul_rn(expf(e[4]), la);
    o[5] = __fmul_rn(expf(e[5]), ha);
    o[6] = __fadd_rn(e[6], ra);
#pragma unroll
    for (int c = 0; c < 7; ++c) dec[(size_t)t * 7 + c] = o[c];
    float *d = dets + (size_t)t * 6;
    if (rotate) {        // boxes_for_nms = box[:, [0, 1, 3, 4, 6]] + score
        d[0] = o[0]; d[1] = o[1]; d[2] = o[3]; d[3] = o[4]; d[4] = o[6]; d[5] = top_score[t];
    } else {             // standup box of the rotated BEV rectangle (center_to_corner_box2d + corner_to_standup_nd)
        float hx = o[3] * 0.5f, hy = o[4] * 0.5f;
        float cs = fabsf(cosf(o[6])), sn = fabsf(sinf(o[6]));
        float ex = hx * cs + hy * sn, ey = hx * sn + hy * cs;
        d[0] = o[0] - ex; d[1] = o[1] - ey; d[2] = o[0] + ex; d[3] = o[1] + ey; d[4] = top_score[t]; d[5] = 0.0f;
    }
    int dl = 0;
    if (dir) {
        const T *pd = dir + frame_off(vd, b, y, x) + a * vd.sa + y * vd.sy + x * vd.sx;
        float best = ldf(pd);
        for (int c = 1; c < ndir; ++c) {
            float f = ldf(pd + c * vd.sc);
            if (f > best) { best = f; dl = c; }
        }
    }
    dir_label[t] = dl;
}

__global__ __launch_bounds__(kBlock) void k_predict_finalize(const float *__restrict__ dec, const float *__restrict__ top_score,
                                                            const int *__restrict__ top_label, const int *__restrict__ dir_label,
                                                            const int *__restrict__ keep, const int *__restrict__ num_keep,
                                                            int batch, int K, int P, int use_dir, float dir_offset,
                                                            float dir_limit_offset, float period, const float *__restrict__ range6,
                                                            float *__restrict__ boxes, float *__restrict__ scores,
                                                            int *__restrict__ labels, unsigned char *__restrict__ valid) {
    int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= batch * P) return;
    int b = t / P, j = t - b * P;
    bool ok = j < num_keep[b];
    int sel = ok ? keep[(size_t)b * K + j] : 0;
    const float *s = dec + ((size_t)b * K + sel) * 7;
    float o[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) o[c] = s[c];
    if (use_dir) {   // limit_period(rot - offset, limit_offset, period) + offset + period * dir_label
        float val = __fsub_rn(o[6], dir_offset);
        float rot = __fsub_rn(val, __fmul_rn(floorf(__fadd_rn(__fdiv_rn(val, period), dir_limit_offset)), period));
        o[6] = __fadd_rn(__fadd_rn(rot, dir_offset), __fmul_rn(period, (float)dir_label[(size_t)b * K + sel]));
    }
    if (range6)
        ok = ok && o[0] >= range6[0] && o[1] >= range6[1] && o[2] >= range6[2] && o[0] <= range6[3] && o[1] <= range6[4] &&
             o[2] <= range6[5];
#pragma unroll
    for (int c = 0; c < 7; ++c) boxes[(size_t)t * 7 + c] = o[c];
    scores[t] = top_score[(size_t)b * K + sel];
    labels[t] = top_label[(size_t)b * K + sel];
    valid[t] = ok ? 1 : 0;
}

}  // namespace sec

using namespace sec;

static View5 mkview(const int64_t *s) { return View5{s[0], s[1], s[2], s[3], s[4], 0, nullptr, 0, 0}; }
// the lazy form of a view: `base` is the view's pointer, `background` the same element of the empty frame's map
static View5 mkview_lazy(const int64_t *s, const void *base, const void *background, const unsigned short *tile_live, int h, int w, int elt) {
    View5 v = mkview(s);
    if (tile_live && background) {
        v.bg = (long long)((const char *)background - (const char *)base) / elt;
        v.live = tile_live;
        v.tiles_x = (w + 15) / 16;
        v.tpf = ((h + 7) / 8) * v.tiles_x;
    }
    return v;
}
static int elt_bytes(int dtype) { return dtype == SEC_F32 ? 4 : 2; }

// a 16-bit key no larger than the key of any bf16 logit whose sigmoid reaches `thr` (0 = no such bound): the logit of thr, lowered
// by more than bf16's and sigmoidf's rounding, mapped like f2key() >> 16, minus one key step (truncation of a negative value's
// bits rounds it UP)
static unsigned conservative_thr16(float thr, int dtype) {
    if (!(thr > 0.0f) || !(thr < 1.0f)) return 0u;
    float x = logf(thr / (1.0f - thr));
    x -= fabsf(x) / 64.0f + 0.01f;
    unsigned k16;
    if (dtype == SEC_F16) {                          // key16_of<__half>: the sign trick on the half's own bits
        if (!(fabsf(x) < 60000.0f)) return 0u;
        const unsigned u = __half_as_ushort(__float2half_rn(x));
        k16 = (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
    } else {
        unsigned u;
        __builtin_memcpy(&u, &x, 4);
        const unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        k16 = key >> 16;
    }
    return k16 > 2u ? k16 - 2u : 0u;                 // two key steps down: covers the rounding of the conversions above
}

static int predict_select_impl(const void *cls, const int64_t *h_cls_strides5, int batch, int anchors_per_loc, int h, int w,
                               int num_class, int k, float score_thr, unsigned *key_scratch, int *top_idx, float *top_score,
                               int *top_label, int *counts, int dtype, void *stream, const unsigned short *tile_live,
                               const void *cls_background) {
    if (!key_scratch) return SEC_E_WORKSPACE;
    if (!cls || !h_cls_strides5 || batch <= 0 || anchors_per_loc <= 0 || h <= 0 || w <= 0 || num_class <= 0 || k <= 0 ||
        k > kSelThreads || !top_idx || !top_score || !top_label || !counts)
        return SEC_E_INVALID;
    if ((tile_live != nullptr) != (cls_background != nullptr) || dtype < SEC_F32 || dtype > SEC_BF16) return SEC_E_INVALID;
    PredGeom g{batch, anchors_per_loc, h, w, num_class};
    View5 v = mkview_lazy(h_cls_strides5, cls, cls_background, tile_live, h, w, elt_bytes(dtype));
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)batch * anchors_per_loc * h * w;
#define SEC_SEL(T, K16)                                                                                                         \
    do {                                                                                                                        \
        hipLaunchKernelGGL(k_predict_keys<T>, dim3(div_up(total, kBlock)), dim3(kBlock), 0, st, (const T *)cls, v, g, key_scratch); \
            hipLaunchKernelGGL((k_predict_select<T, K16>), dim3(batch), dim3(kSelThreads), 0, st, (const T *)cls, v, g, k,        \
                               score_thr, key_scratch, top_idx, top_score, top_label, counts, thr_key32);                        \
    } while (0)
    const long long nfr = (long long)anchors_per_loc * h * w;
    // 32-bit key of a logit safely below the score threshold's (k_predict_select's shortcut; 0: none)
    unsigned thr_key32 = 0u;
    if (score_thr > 0.0f && score_thr < 1.0f) {
        float x = logf(score_thr / (1.0f - score_thr));
        x -= fabsf(x) * 1e-4f + 1e-5f;
        unsigned u;
        __builtin_memcpy(&u, &x, 4);
        thr_key32 = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    // 16-bit heads (bf16 / fp16): register-resident select on 16-bit keys
    static int use_thr = -1, chunked = -1;
    if (use_thr < 0) use_thr = 1;
    if (chunked < 0) chunked = 1;
    const bool h16 = dtype == SEC_BF16 || dtype == SEC_F16;
    const unsigned thr16 = (use_thr && h16) ? conservative_thr16(score_thr, dtype) : 0u;
    const long long reg_cap = (long long)kSelThreads * 72;           // keys one workgroup of k_predict_select_reg holds
    // chunked form: per-chunk top-K on the whole chip, then one workgroup per frame over the candidates.  Chunks of 8192 anchors
    // (KP 4) while their candidate lists fit the second stage, of 73 728 anchors (KP 36) for larger heads (up to 5.3 M anchors per
    // frame).  The candidate arrays live in key_scratch (4 bytes per anchor; they need 6 bytes per candidate slot).
    int kp = 0;
    if (h16 && chunked) {
        const long long c4 = (nfr + kSelChunk - 1) / kSelChunk, c36 = (nfr + reg_cap - 1) / reg_cap;
        if (c4 >= 2 && c4 * kSelThreads <= reg_cap) kp = 4;
        else if (nfr > reg_cap && c36 * kSelThreads <= reg_cap) kp = 36;
    }
    if (kp) {
        const long long per = (long long)kSelThreads * 2 * kp;
        const int chunks = (int)((nfr + per - 1) / per);
        const long long n2 = (long long)chunks * kSelThreads;
        if ((size_t)batch * n2 * 6 <= (size_t)total * 4) {
            unsigned short *ck = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(key_scratch) + (size_t)batch * n2 * 4);
            int *ci = reinterpret_cast<int *>(key_scratch);
            const int pairs = (int)((n2 / 2 + kSelThreads - 1) / kSelThreads);
#define SEC_SEL2(T, KP2) hipLaunchKernelGGL((k_predict_select_reg<T, KP2, true>), dim3(batch), dim3(kSelThreads), 0, st, (const T *)cls, v, g, k, \
                                            score_thr, ck, (int)n2, top_idx, top_score, top_label, counts, (int)n2, ci, thr16)
#define SEC_CHUNKED(T)                                                                                                                      \
    do {                                                                                                                                    \
        if (kp == 4)                                                                                                                        \
            hipLaunchKernelGGL((k_predict_select_chunk<T, 4>), dim3(chunks, batch), dim3(kSelThreads), 0, st, (const T *)cls, v, g, (int)nfr, k, \
                               chunks, thr16, ck, ci);                                                                                      \
        else                                                                                                                                \
            hipLaunchKernelGGL((k_predict_select_chunk<T, 36>), dim3(chunks, batch), dim3(kSelThreads), 0, st, (const T *)cls, v, g, (int)nfr, k, \
                               chunks, thr16, ck, ci);                                                                                      \
        if (pairs <= 5) SEC_SEL2(T, 5);                                                                                                     \
        else if (pairs <= 10) SEC_SEL2(T, 10);                                                                                              \
        else if (pairs <= 20) SEC_SEL2(T, 20);                                                                                              \
        else SEC_SEL2(T, 36);                                                                                                               \
    } while (0)
            if (dtype == SEC_BF16) SEC_CHUNKED(__hip_bfloat16);
            else SEC_CHUNKED(__half);
#undef SEC_CHUNKED
#undef SEC_SEL2
            return check_launch();
        }
    }
    if (h16 && nfr <= reg_cap) {                                     // one workgroup per frame over a compact 16-bit key array
        const int ns = (int)((nfr + 1) & ~1ll);
#define SEC_REG(T)                                                                                                                          \
    do {                                                                                                                                    \
        hipLaunchKernelGGL(k_predict_keys16<T>, dim3(div_up((long long)batch * ns, kBlock)), dim3(kBlock), 0, st, (const T *)cls, v, g, ns, \
                           reinterpret_cast<unsigned short *>(key_scratch));                                                                \
        hipLaunchKernelGGL((k_predict_select_reg<T, 36>), dim3(batch), dim3(kSelThreads), 0, st, (const T *)cls, v, g, k, score_thr,       \
                           reinterpret_cast<const unsigned short *>(key_scratch), ns, top_idx, top_score, top_label, counts, 0,            \
                           (const int *)nullptr, thr16);                                                                                    \
    } while (0)
        if (dtype == SEC_BF16) SEC_REG(__hip_bfloat16);
        else SEC_REG(__half);
#undef SEC_REG
        return check_launch();
    }
    if (dtype == SEC_F32) SEC_SEL(float, false);
    else if (dtype == SEC_BF16) SEC_SEL(__hip_bfloat16, true);
    else if (dtype == SEC_F16) SEC_SEL(__half, false);
    else return SEC_E_UNSUPPORTED;
#undef SEC_SEL
    return check_launch();
}

SEC_API int sec_predict_select(const void *cls, const int64_t *h_cls_strides5, int batch, int anchors_per_loc, int h, int w,
                               int num_class, int k, float score_thr, unsigned *key_scratch, int *top_idx, float *top_score,
                               int *top_label, int *counts, int dtype, void *stream) {
    return predict_select_impl(cls, h_cls_strides5, batch, anchors_per_loc, h, w, num_class, k, score_thr, key_scratch, top_idx, top_score,
                               top_label, counts, dtype, stream, nullptr, nullptr);
}
SEC_API int sec_predict_select_lazy(const void *cls, const int64_t *h_cls_strides5, int batch, int anchors_per_loc, int h, int w,
                                    int num_class, int k, float score_thr, unsigned *key_scratch, int *top_idx, float *top_score,
                                    int *top_label, int *counts, int dtype, const unsigned short *tile_live, const void *cls_background,
                                    void *stream) {
    if (!tile_live || !cls_background) return SEC_E_INVALID;
    return predict_select_impl(cls, h_cls_strides5, batch, anchors_per_loc, h, w, num_class, k, score_thr, key_scratch, top_idx, top_score,
                               top_label, counts, dtype, stream, tile_live, cls_background);
}

static int predict_decode_impl(const void *box, const int64_t *h_box_strides5, const void *dir, const int64_t *h_dir_strides5,
                               int num_dir_bins, int batch, int anchors_per_loc, int h, int w, int k, const float *anchors,
                               const int *top_idx, const float *top_score, int rotate, float *decoded, float *dets,
                               int *dir_label, int dtype, void *stream, const unsigned short *tile_live, const void *box_background,
                               const void *dir_background) {
    if (!box || !h_box_strides5 || batch <= 0 || k <= 0 || !anchors || !top_idx || !top_score || !decoded || !dets || !dir_label)
        return SEC_E_INVALID;
    if (dtype < SEC_F32 || dtype > SEC_BF16 || (tile_live && (!box_background || (dir && !dir_background)))) return SEC_E_INVALID;
    PredGeom g{batch, anchors_per_loc, h, w, 1};
    View5 vb = mkview_lazy(h_box_strides5, box, box_background, tile_live, h, w, elt_bytes(dtype));
    View5 vd = dir ? mkview_lazy(h_dir_strides5, dir, dir_background, tile_live, h, w, elt_bytes(dtype)) : View5{0, 0, 0, 0, 0, 0, nullptr, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(div_up((long long)batch * k, kBlock));
#define SEC_DEC(T) hipLaunchKernelGGL(k_predict_decode<T>, grid, dim3(kBlock), 0, st, (const T *)box, vb, (const T *)dir, vd, \
                                      num_dir_bins, g, k, anchors, top_idx, top_score, rotate, decoded, dets, dir_label)
    if (dtype == SEC_F32) SEC_DEC(float);
    else if (dtype == SEC_BF16) SEC_DEC(__hip_bfloat16);
    else if (dtype == SEC_F16) SEC_DEC(__half);
    else return SEC_E_UNSUPPORTED;
#undef SEC_DEC
    return check_launch();
}
SEC_API int sec_predict_decode(const void *box, const int64_t *h_box_strides5, const void *dir, const int64_t *h_dir_strides5,
                               int num_dir_bins, int batch, int anchors_per_loc, int h, int w, int k, const float *anchors,
                               const int *top_idx, const float *top_score, int rotate, float *decoded, float *dets,
                               int *dir_label, int dtype, void *stream) {
    return predict_decode_impl(box, h_box_strides5, dir, h_dir_strides5, num_dir_bins, batch, anchors_per_loc, h, w, k, anchors, top_idx,
                               top_score, rotate, decoded, dets, dir_label, dtype, stream, nullptr, nullptr, nullptr);
}
SEC_API int sec_predict_decode_lazy(const void *box, const int64_t *h_box_strides5, const void *dir, const int64_t *h_dir_strides5,
                                    int num_dir_bins, int batch, int anchors_per_loc, int h, int w, int k, const float *anchors,
                                    const int *top_idx, const float *top_score, int rotate, float *decoded, float *dets,
                                    int *dir_label, int dtype, const unsigned short *tile_live, const void *box_background,
                                    const void *dir_background, void *stream) {
    if (!tile_live || !box_background) return SEC_E_INVALID;
    return predict_decode_impl(box, h_box_strides5, dir, h_dir_strides5, num_dir_bins, batch, anchors_per_loc, h, w, k, anchors, top_idx,
                               top_score, rotate, decoded, dets, dir_label, dtype, stream, tile_live, box_background, dir_background);
}

SEC_API int sec_predict_finalize(const float *decoded, const float *top_score, const int *top_label, const int *dir_label,
                                 const int *keep, const int *num_keep, int batch, int k, int post_max, int use_direction,
                                 float dir_offset, float dir_limit_offset, int num_dir_bins, const float *range6,
                                 float *boxes, float *scores, int *labels, unsigned char *valid, void *stream) {
    if (!decoded || !top_score || !top_label || !keep || !num_keep || batch <= 0 || k <= 0 || post_max <= 0 || post_max > k ||
        !boxes || !scores || !labels || !valid || (use_direction && (!dir_label || num_dir_bins <= 0)))
        return SEC_E_INVALID;
    float period = use_direction ? 6.283185307179586f / (float)num_dir_bins : 0.0f;
    hipLaunchKernelGGL(k_predict_finalize, dim3(div_up((long long)batch * post_max, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       decoded, top_score, top_label, dir_label, keep, num_keep, batch, k, post_max, use_direction, dir_offset,
                       dir_limit_offset, period, range6, boxes, scores, labels, valid);
    return check_launch();
}
