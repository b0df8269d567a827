// Rotated-box IoU and greedy NMS on gfx950, replacing the numba.cuda kernels of
// second/core/non_max_suppression/nms_gpu.py (rotate_iou_kernel_eval :564-602, rotate_nms_kernel :404-437,
// nms_kernel :70-101, host nms_postprocess :109-126) and the CPU round trip of box_torch_ops.rotate_nms
// (second/pytorch/core/box_torch_ops.py:492-515 -> nms_cpu.py:17-28).
//
// Wave64 mapping: a 64-box column tile sits in LDS, the 64 lanes of a wave are the 64 columns, so the
// 64-bit suppression word of a row is ONE __ballot (the reference builds it bit by bit in a 64-iteration
// per-thread loop).  The greedy reduce runs on-device in a single wave (lane w holds removal word w), so
// kept indices never leave the GPU.  The fp32 polygon-clipping arithmetic follows nms_gpu.py:166-401
// operation for operation (compiled with -ffp-contract=off).
#include "common.hpp"

namespace sec {

__device__ __forceinline__ float tri_area(const float *a, const float *b, const float *c) {
    return ((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f;
}

__device__ __forceinline__ void box_corners(float *c, const float *b) {  // nms_gpu.py:353-376
    float ac = cosf(b[4]), as = sinf(b[4]);
    float cx = b[0], cy = b[1], xd = b[2], yd = b[3];
    float xs[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    float ys[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = ac * xs[i] + as * ys[i] + cx;
        c[2 * i + 1] = -as * xs[i] + ac * ys[i] + cy;
    }
}

__device__ __forceinline__ bool pt_in_quad(float x, float y, const float *c) {  // nms_gpu.py:308-325
    float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    float ap0 = x - c[0], ap1 = y - c[1];
    float abab = ab0 * ab0 + ab1 * ab1;
    float abap = ab0 * ap0 + ab1 * ap1;
    float adad = ad0 * ad0 + ad1 * ad1;
    float adap = ad0 * ap0 + ad1 * ap1;
    const float eps = -1e-6f;
    return abab - abap >= eps && abap >= eps && adad - adap >= eps && adap >= eps;
}

__device__ __forceinline__ bool seg_intersect(const float *p1, const float *p2, int i, int j, float *t) {  // :222-264
    float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    bool acd = DA1 * CA0 > CA1 * DA0;
    bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        bool abc = CA1 * BA0 > BA1 * CA0;
        bool abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            float DC0 = D0 - C0, DC1 = D1 - C1;
            float ABBA = A0 * B1 - B0 * A1;
            float CDDC = C0 * D1 - D0 * C1;
            float DH = BA1 * DC0 - BA0 * DC1;
            float Dx = ABBA * DC0 - BA0 * CDDC;
            float Dy = ABBA * DC1 - BA1 * CDDC;
            t[0] = Dx / DH;
            t[1] = Dy / DH;
            return true;
        }
    }
    return false;
}

// intersection area of two quads given by their corners (nms_gpu.py:329-350,172-219,379-393)
// The reference indexes its vertex list dynamically (append, insertion sort).  In registers that means scratch memory, in LDS
// (rounds 1-2) a ~100-clock round trip per access on a serial chain of a few hundred accesses: ~18 us per clip with one wave per
// SIMD, and a launch lasts as long as its slowest clip.  Here the list (at most 8 vertices are ever used -- the reference's
// int_pts holds 8; later ones are counted, not stored) lives in registers and every index is static: an append is eight selects
// on "n == slot", the insertion sort is unrolled with a `moving` predicate that replays the reference's while loop step by step
// (same comparisons in the same order, so ties and NaN keys fall exactly where the reference's sort leaves them), the centroid
// and the triangle fan run to 8 / 6 with "i < n" predicates.  No LDS, no data-dependent branches except the per-edge-pair
// "these two segments cross" block.
__device__ __forceinline__ void vl_append(float (&px)[8], float (&py)[8], int &n, bool hit, float vx, float vy) {
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
        const bool wr = hit && n == sl;
        px[sl] = wr ? vx : px[sl];
        py[sl] = wr ? vy : py[sl];
    }
    n += hit ? 1 : 0;
}

__device__ float quad_inter(const float (&c1)[8], const float (&c2)[8]) {
    float px[8], py[8], key[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) px[i] = py[i] = key[i] = 0.0f;
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vl_append(px, py, n, pt_in_quad(c1[2 * i], c1[2 * i + 1], c2), c1[2 * i], c1[2 * i + 1]);
        vl_append(px, py, n, pt_in_quad(c2[2 * i], c2[2 * i + 1], c1), c2[2 * i], c2[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t[2] = {0.0f, 0.0f};
            const bool hit = seg_intersect(c1, c2, i, j, t);
            vl_append(px, py, n, hit, t[0], t[1]);
        }
    if (n > 8) n = 8;
    if (n < 3) return 0.0f;
    // angular sort about the centroid (insertion sort on the reference's key)
    float cx = 0.0f, cy = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cx = i < n ? cx + px[i] : cx;
        cy = i < n ? cy + py[i] : cy;
    }
    cx /= (float)n;
    cy /= (float)n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float vx = px[i] - cx, vy = py[i] - cy;
        float d = sqrtf(vx * vx + vy * vy);
        vx = vx / d;
        vy = vy / d;
        if (vy < 0) vx = -2 - vx;
        key[i] = vx;                        // slots >= n are never compared (every step below is predicated on i < n)
    }
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        // reference: if (V[i-1] > V[i]) { temp = V[i]; j = i; while (j > 0 && V[j-1] > temp) { V[j] = V[j-1]; --j; } V[j] = temp; }
        const float temp = key[i], tx = px[i], ty = py[i];
        bool moving = i < n && key[i - 1] > temp;
#pragma unroll
        for (int j = i; j >= 1; --j) {
            const bool shift = moving && key[j - 1] > temp;
            key[j] = shift ? key[j - 1] : (moving ? temp : key[j]);
            px[j] = shift ? px[j - 1] : (moving ? tx : px[j]);
            py[j] = shift ? py[j - 1] : (moving ? ty : py[j]);
            moving = shift;
        }
        key[0] = moving ? temp : key[0];
        px[0] = moving ? tx : px[0];
        py[0] = moving ? ty : py[0];
    }
    float s = 0.0f;
    const float p0[2] = {px[0], py[0]};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float pa[2] = {px[i + 1], py[i + 1]}, pb[2] = {px[i + 2], py[i + 2]};
        const float a = fabsf(tri_area(p0, pa, pb));
        s = i < n - 2 ? s + a : s;
    }
    return s;
}

struct Standup { float x0, y0, x1, y1; };
__device__ __forceinline__ Standup standup_of(const float *c) {
    Standup s{c[0], c[1], c[0], c[1]};
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        s.x0 = fminf(s.x0, c[2 * i]); s.x1 = fmaxf(s.x1, c[2 * i]);
        s.y0 = fminf(s.y0, c[2 * i + 1]); s.y1 = fmaxf(s.y1, c[2 * i + 1]);
    }
    return s;
}
// far apart => the clipper finds no vertex => intersection exactly 0 (margin covers its 1e-6 tolerances)
__device__ __forceinline__ bool far_apart(const Standup &a, const Standup &b) {
    const float m = 1e-3f;
    return a.x0 > b.x1 + m || b.x0 > a.x1 + m || a.y0 > b.y1 + m || b.y0 > a.y1 + m;
}

// ---------------------------------------------------------------- IoU matrix (rotate_iou_gpu_eval)
__global__ __launch_bounds__(kBlock) void k_rotate_iou(const float *__restrict__ boxes, int N,
                                                      const float *__restrict__ qboxes, int K, int criterion,
                                                      float *__restrict__ iou) {
    __shared__ float qc[64][9];   // corners of the 64 query boxes of this column tile (+1 pad)
    __shared__ float qa[64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int kq = blockIdx.y * 64 + lane;
    if (w == 0 && kq < K) {
        float c[8];
        box_corners(c, qboxes + (size_t)kq * 5);
#pragma unroll
        for (int i = 0; i < 8; ++i) qc[lane][i] = c[i];
        qa[lane] = qboxes[(size_t)kq * 5 + 2] * qboxes[(size_t)kq * 5 + 3];
    }
    __syncthreads();
    float c1[8];
    float a1 = 0.0f;
    if (kq < K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c1[i] = qc[lane][i];
        a1 = qa[lane];
    }
    for (int rr = 0; rr < 16; ++rr) {
        int n = blockIdx.x * 64 + w * 16 + rr;
        if (n >= N) break;
        if (kq >= K) continue;
        float c2[8];
        box_corners(c2, boxes + (size_t)n * 5);
        float a2 = boxes[(size_t)n * 5 + 2] * boxes[(size_t)n * 5 + 3];
        float in = far_apart(standup_of(c1), standup_of(c2)) ? 0.0f : quad_inter(c1, c2);
        float v;
        if (criterion == -1) v = in / (a1 + a2 - in);
        else if (criterion == 0) v = in / a1;
        else if (criterion == 1) v = in / a2;
        else v = in;
        iou[(size_t)n * K + kq] = v;
    }
}

// ---------------------------------------------------------------- suppression bit matrix
// A tile is ROWS rows x 64 columns of one frame's pair matrix; tiles entirely below the diagonal are skipped.
// Two phases per tile:
//   A. every thread screens ROWS / 4 (row, col) pairs with the cheap standup test (corners / standup boxes of the
//      ROWS + 64 boxes of the tile are computed once into LDS); survivors are appended to an LDS queue with one
//      wave-level atomic per ballot;
//   B. the queue is processed densely -- one polygon clip per thread -- and hits are OR-ed into the tile's 64
//      suppression words (LDS atomics).
// The reference (and a lanes-are-columns mapping) runs the ~1000-instruction clipper for all 64 lanes whenever a
// single pair of the wave overlaps; with ~1 % of the pairs overlapping that wastes > 95 % of the lanes.
// ROWS = 16 (round 3; 64 before): the launch lasts as long as its busiest tile, and the candidates of a detector cluster
// -- with 64 x 64 tiles the tile of the top-scoring boxes queued > 1000 pairs (five clipper rounds of 256 threads, 49 us
// for the launch) while most workgroups idled; 16-row tiles spread the same pairs over four times as many workgroups.
#ifdef SEC_NMS_DEBUG
__device__ int g_nms_dbg[4];
__device__ float g_nms_vals[8 * 32];
#endif
template <int ROWS>
__global__ __launch_bounds__(kBlock) void k_nms_mask(const float *__restrict__ dets, const int *__restrict__ counts,
                                                    int max_n, int stride, float thresh, int kind, int semantics,
                                                    float eps, int words, unsigned long long *__restrict__ mask) {
    static_assert(ROWS % 4 == 0 && ROWS <= 64 && 64 % ROWS == 0, "a tile is ROWS rows (a multiple of the 4 waves) x 64 columns");
    // grid (workgroups per frame, batch): a workgroup walks the tiles of ITS frame's live rectangle (ceil(n / ROWS) x ceil(n / 64),
    // n read from the device) with a stride of gridDim.x.  A grid sized for max_n (1000 candidates: 8 000 workgroups at ROWS = 16)
    // spends its time launching workgroups that read counts[b] and leave -- ~6 rounds of them per CU before the few live ones.
    const int b = blockIdx.y;
    const bool screen = !(semantics & SEC_NMS_EXACT_CLIP);      // inscribed-circle lower bound before the clipper (see below)
    semantics &= 0xff;
    int n = counts[b];
    if (n > max_n) n = max_n;
    const int ncb = (n + 63) >> 6, nrb = (n + ROWS - 1) / ROWS;
    __shared__ float tile[2][64][10];              // [0] column boxes, [1] row boxes: 8 corner floats (or x1,y1,x2,y2), area
    __shared__ Standup su[2][64];
    __shared__ float ctr[2][64][5];                // centre, inscribed radius, offset of the outer inscribed circles (see phase A)
    __shared__ unsigned long long sup_words[ROWS];
    __shared__ unsigned short queue[ROWS * 64];
    __shared__ int qcount;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *base = dets + (size_t)b * max_n * stride;
    for (int ti = blockIdx.x; ti < nrb * ncb; ti += gridDim.x) {
    const int rb = ti / ncb, cb = ti - rb * ncb;
    if (cb * 64 + 63 <= rb * ROWS) continue;       // no column of the tile lies right of any of its rows
    if (tid == 0) qcount = 0;
    if (tid < ROWS) sup_words[tid] = 0ull;
    if (w < 2 && (w == 0 || lane < ROWS)) {
        int idx = w == 0 ? cb * 64 + lane : rb * ROWS + lane;
        if (idx < n) {
            const float *d = base + (size_t)idx * stride;
            if (kind == 0) {
                float c[8];
                box_corners(c, d);
#pragma unroll
                for (int i = 0; i < 8; ++i) tile[w][lane][i] = c[i];
                tile[w][lane][8] = d[2] * d[3];
                su[w][lane] = standup_of(c);
                ctr[w][lane][0] = d[0];
                ctr[w][lane][1] = d[1];
                ctr[w][lane][2] = 0.5f * fminf(d[2], d[3]);   // radius of the circle inscribed at the centre (any rotation)
                // the same circle slid along the long axis stays inside until it touches the short sides: +-(long - short) / 2
                // (taken from the corners, so no angle convention enters; shortened by 0.1 % against rounding)
                const bool ylong = d[3] >= d[2];
                const float lng = ylong ? d[3] : d[2], sht = ylong ? d[2] : d[3];
                const float f = (lng > 0.0f && sht > 0.0f) ? 0.4995f * (lng - sht) / lng : 0.0f;
                ctr[w][lane][3] = f * (ylong ? c[2] - c[0] : c[4] - c[2]);
                ctr[w][lane][4] = f * (ylong ? c[3] - c[1] : c[5] - c[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[w][lane][i] = d[i];
            }
        }
    }
    __syncthreads();
#ifdef SEC_NMS_DEBUG
    if (w < 2 && (w == 0 || lane < ROWS)) {     // right after the tile was written: does LDS hold what a second load + derivation gives?
        int idx = w == 0 ? cb * 64 + lane : rb * ROWS + lane;
        if (idx < n && kind == 0) {
            const float *d = base + (size_t)idx * stride;
            float c[8], c2[8];
            box_corners(c, d);
            box_corners(c2, d);
            bool bad = false, bad2 = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) { bad |= tile[w][lane][i] != c[i]; bad2 |= c[i] != c2[i]; }
            if (bad) atomicAdd(&g_nms_dbg[2], 1);
            if (bad2) {
                const int slot = atomicAdd(&g_nms_dbg[3], 1);
                if (slot < 8) {
                    float *o = g_nms_vals + slot * 32;
                    for (int i = 0; i < 8; ++i) { o[i] = c[i]; o[8 + i] = c2[i]; o[16 + i] = tile[w][lane][i]; }
                    for (int i = 0; i < 6; ++i) o[24 + i] = d[i];
                    o[30] = (float)idx; o[31] = (float)(w * 1000 + lane);
                }
            }
        }
    }
    __syncthreads();
#endif
    // ---- phase A: screen
#pragma unroll 1
    for (int it = 0; it < ROWS / 4; ++it) {
        const int rl = it * 4 + w, cl = lane;      // a wave = one row of the tile x 64 columns
        const int row = rb * ROWS + rl, col = cb * 64 + cl;
        bool cand = false;
        if (row < n && col < n && col > row) {
            if (kind == 0) {
                const Standup s1 = su[1][rl], s2 = su[0][cl];
                if (!far_apart(s1, s2)) {
                    cand = true;
                    if (semantics == 1) {          // CPU path: standup IoU (eps = 0) must be > 0 (nms_cpu.py:25, A.6)
                        float iw = fminf(s1.x1, s2.x1) - fmaxf(s1.x0, s2.x0);
                        float ih = fminf(s1.y1, s2.y1) - fmaxf(s1.y0, s2.y0);
                        cand = iw > 0.0f && ih > 0.0f;
                        if (cand) {
                            float ua = (s1.x1 - s1.x0) * (s1.y1 - s1.y0) + (s2.x1 - s2.x0) * (s2.y1 - s2.y0) - iw * ih;
                            cand = iw * ih / ua > 0.0f;
                        }
                    }
                    if (cand && thresh >= 0.0f && screen) {
                        // Certain suppression without clipping: the circles of radius min(w, l) / 2 about the two centres lie inside
                        // their boxes, so the lens they share is a LOWER bound of the polygon intersection, and IoU grows with the
                        // intersection.  If even that bound clears the threshold (with a margin far above fp32 rounding) the pair is
                        // decided; only undecided pairs are queued for the clipper.  The candidates of a trained detector cluster on
                        // the objects -- most overlapping pairs are decided here (car.fhd: iou_threshold 0.01, car.fhd.config:94).
                        // Three such circles per box (centre and both ends of the long axis; a car is ~2.4 x as long as wide, one
                        // circle covers a third of it): the closest of the nine pairs gives the bound.  Round 3: 3 600 -> 10 800 of the
                        // 14 700 standup survivors of a bench frame decided here.
                        const float rr = fminf(ctr[1][rl][2], ctr[0][cl][2]);
                        const float dx = ctr[1][rl][0] - ctr[0][cl][0], dy = ctr[1][rl][1] - ctr[0][cl][1];
                        const float ax = ctr[1][rl][3], ay = ctr[1][rl][4], bx = ctr[0][cl][3], by = ctr[0][cl][4];
                        float d2 = 3.0e38f;
#pragma unroll
                        for (int ka = -1; ka <= 1; ++ka)
#pragma unroll
                            for (int kb = -1; kb <= 1; ++kb) {
                                const float ex = dx + (float)ka * ax - (float)kb * bx, ey = dy + (float)ka * ay - (float)kb * by;
                                d2 = fminf(d2, ex * ex + ey * ey);
                            }
                        const float dd = sqrtf(d2);
                        if (dd < 1.9f * rr) {
                            const float lens = 2.0f * rr * rr * acosf(fminf(dd / (2.0f * rr), 1.0f)) - 0.5f * dd * sqrtf(fmaxf(4.0f * rr * rr - dd * dd, 0.0f));
                            const float lb = lens / (tile[1][rl][8] + tile[0][cl][8] - lens);
                            if (lb > 1.02f * thresh + 1e-4f) {
                                atomicOr(&sup_words[rl], 1ull << cl);
                                cand = false;
                            }
                        }
                    }
                } else if (semantics == 0 && 0.0f > thresh) {
                    atomicOr(&sup_words[rl], 1ull << cl);   // IoU is exactly 0 and the threshold is negative
                }
            } else {
                const float *c1 = tile[1][rl], *c2 = tile[0][cl];
                float e = semantics == 0 ? 1.0f : eps;
                float wv = fmaxf(fminf(c1[2], c2[2]) - fmaxf(c1[0], c2[0]) + e, 0.0f);
                float hv = fmaxf(fminf(c1[3], c2[3]) - fmaxf(c1[1], c2[1]) + e, 0.0f);
                float in = wv * hv;
                float sa = (c1[2] - c1[0] + e) * (c1[3] - c1[1] + e);
                float sb = (c2[2] - c2[0] + e) * (c2[3] - c2[1] + e);
                float v = in / (sa + sb - in);
                if (semantics == 0 ? v > thresh : v >= thresh) atomicOr(&sup_words[rl], 1ull << cl);
            }
        }
        unsigned long long m = __ballot(cand);
        if (m) {
            int qbase = 0;
            if (lane == 0) qbase = atomicAdd(&qcount, __popcll(m));
            qbase = __shfl(qbase, 0, 64);
            if (cand) queue[qbase + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((rl << 8) | cl);
        }
    }
    __syncthreads();
    // ---- phase B: dense polygon clipping of the survivors
    const int qn = qcount;
    for (int q = tid; q < qn; q += kBlock) {
        const int rl = queue[q] >> 8, cl = queue[q] & 255;
        float c1[8], c2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { c1[i] = tile[1][rl][i]; c2[i] = tile[0][cl][i]; }
        float in = quad_inter(c1, c2);
        float v = in / (tile[1][rl][8] + tile[0][cl][8] - in);
        if (semantics == 1 ? v >= thresh : v > thresh) atomicOr(&sup_words[rl], 1ull << cl);
    }
    __syncthreads();
    if (tid < ROWS && rb * ROWS + tid < n) mask[((size_t)b * max_n + rb * ROWS + tid) * words + cb] = sup_words[tid];
#ifdef SEC_NMS_DEBUG
    // debug build: is the tile still what the global loads delivered at the start?  (re-load, re-derive, compare)
    if (w < 2 && (w == 0 || lane < ROWS)) {
        int idx = w == 0 ? cb * 64 + lane : rb * ROWS + lane;
        if (idx < n && kind == 0) {
            const float *d = base + (size_t)idx * stride;
            float c[8];
            box_corners(c, d);
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) bad |= tile[w][lane][i] != c[i];
            if (bad) atomicAdd(&g_nms_dbg[w], 1);
        }
    }
#endif
    __syncthreads();                               // the next tile re-initialises the LDS state read above
    }
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d, 64);
    return v;
}

// Greedy reduce (host loop nms_postprocess, nms_gpu.py:109-126) on-device, one wave per batch item.
// Lane w owns removal word w.  Per 64-box block: (1) the intra-block chain is resolved on scalar state
// with the 64 diagonal words (one per lane, read with v_readlane), (2) each kept row is read once (its words for
// the later blocks, one per lane) and OR-ed into the owners' removal words.
__global__ __launch_bounds__(64) void k_nms_reduce(const unsigned long long *__restrict__ mask,
                                                  const int *__restrict__ counts, int max_n, int words, int post_max,
                                                  int *__restrict__ keep, int *__restrict__ num_keep) {
    int b = blockIdx.x, lane = threadIdx.x;
    int n = counts[b];
    if (n > max_n) n = max_n;
    int nwords = (n + 63) >> 6;
    unsigned long long remv = 0ull;
    const unsigned long long *mb = mask + (size_t)b * max_n * words;
    int *kp = keep + (size_t)b * max_n;
    int nk = 0;
    for (int c = 0; c < nwords; ++c) {
        int base = c * 64;
        int cnt = n - base < 64 ? n - base : 64;
        unsigned long long diag = lane < cnt ? mb[(size_t)(base + lane) * words + c] : 0ull;
        unsigned rlo = __builtin_amdgcn_readlane((unsigned)remv, c), rhi = __builtin_amdgcn_readlane((unsigned)(remv >> 32), c);
        unsigned long long alive = ~(((unsigned long long)rhi << 32) | rlo);
        if (cnt < 64) alive &= (1ull << cnt) - 1ull;
        unsigned long long kept = 0ull;
        int room = post_max > 0 ? post_max - nk : 0x7fffffff;
        while (alive && room > 0) {
            int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
            kept |= 1ull << i;
            --room;
            unsigned dlo = __builtin_amdgcn_readlane((unsigned)diag, i), dhi = __builtin_amdgcn_readlane((unsigned)(diag >> 32), i);
            alive &= ~((1ull << i) | (((unsigned long long)dhi << 32) | dlo));
        }
        bool mine = (kept >> lane) & 1ull;
        if (mine) kp[nk + __popcll(kept & ((1ull << lane) - 1ull))] = base + lane;
        nk += __popcll(kept);
        if (post_max > 0 && nk >= post_max) break;
        // lane w (> c) merges the kept rows' word w into its removal word: one coalesced 8-byte-per-lane row load per KEPT
        // box (a handful per block), four in flight, instead of a wave-wide OR-reduction per later word
        const bool owner = lane > c && lane < nwords;
        unsigned long long todo = kept;
        while (todo) {
            int i4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                i4[u] = todo ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo)) : -1;
                if (todo) todo &= todo - 1ull;
            }
            unsigned long long v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = (owner && i4[u] >= 0) ? mb[(size_t)(base + i4[u]) * words + lane] : 0ull;
            remv |= (v4[0] | v4[1]) | (v4[2] | v4[3]);
        }
    }
    if (lane == 0) num_keep[b] = nk;
}


}  // namespace sec

using namespace sec;

SEC_API int sec_rotate_iou_f32(const float *boxes, int n, const float *qboxes, int k, int criterion, float *iou,
                               void *stream) {
    if (n < 0 || k < 0 || (n > 0 && k > 0 && (!boxes || !qboxes || !iou))) return SEC_E_INVALID;
    if (n == 0 || k == 0) return SEC_OK;
    hipLaunchKernelGGL(k_rotate_iou, dim3(div_up(n, 64), div_up(k, 64)), dim3(kBlock), 0, (hipStream_t)stream, boxes, n,
                       qboxes, k, criterion, iou);
    return check_launch();
}

#ifdef SEC_NMS_DEBUG
// which unit misbehaves beside the RPN conv?  counters: [0] repeated global loads differ, [1] sinf / cosf of the same argument
// differ, [2] plain fp32 arithmetic differs, [3] an LDS word read back differs from what was written
template <int BIG, int VG>
__global__ __launch_bounds__(256) void k_unit_check(const float *__restrict__ data, int n, int iters, int *cnt) {
    __shared__ float lds[256 * 8];
    __shared__ float big[BIG ? 8192 : 1];               // BIG: the 40 KB LDS footprint of k_nms_mask
    if (BIG) big[threadIdx.x * 32 % 8192] = 1.0f;
    if (VG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");     // register footprint of k_nms_mask (80 VGPRs)
    if (VG == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    const int i = (blockIdx.x * 256 + threadIdx.x) % n;
    const float *p = data + (size_t)i * 6;
    float v0[6];
    for (int k = 0; k < 6; ++k) v0[k] = __builtin_nontemporal_load(p + k);
    const float s0 = sinf(v0[4]), c0 = cosf(v0[4]);
    const float a0 = s0 * v0[2] + c0 * v0[3] + v0[0] / (v0[3] + 1.5f);
    for (int k = 0; k < 8; ++k) lds[threadIdx.x * 8 + k] = v0[k % 6] + (float)k;
    int bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        float v[6];
        for (int k = 0; k < 6; ++k) {                    // plain cached loads (default policy, through the vector L1), not foldable
            const float *q = p + k;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v[k]) : "v"(q) : "memory");
        }
        for (int k = 0; k < 6; ++k) bad[0] += v[k] != v0[k];
        float arg = v0[4];
        asm volatile("" : "+v"(arg));                   // keep the compiler from folding the repeats
        const float s1 = sinf(arg), c1 = cosf(arg);
        bad[1] += (s1 != s0) + (c1 != c0);
        float x2 = v0[2], x3 = v0[3], x0 = v0[0];
        asm volatile("" : "+v"(x2), "+v"(x3), "+v"(x0));
        const float a1 = s0 * x2 + c0 * x3 + x0 / (x3 + 1.5f);
        bad[2] += a1 != a0;
        for (int k = 0; k < 8; ++k) bad[3] += lds[threadIdx.x * 8 + k] != v0[k % 6] + (float)k;
    }
    for (int k = 0; k < 4; ++k)
        if (bad[k]) atomicAdd(&cnt[k], bad[k]);
}
extern "C" __attribute__((visibility("default"))) int sec__debug_unit_check(const float *data, int n, int blocks, int iters, int *cnt, void *stream) {
    const char *e = getenv("SEC_UNIT_CHECK");
    const int mode = e ? atoi(e) : 0;
    if (mode == 1) hipLaunchKernelGGL((k_unit_check<1, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n, iters, cnt);
    else if (mode == 2) hipLaunchKernelGGL((k_unit_check<0, 80>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n, iters, cnt);
    else if (mode == 3) hipLaunchKernelGGL((k_unit_check<1, 80>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n, iters, cnt);
    else if (mode == 4) hipLaunchKernelGGL((k_unit_check<1, 88>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n, iters, cnt);
    else hipLaunchKernelGGL((k_unit_check<0, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, n, iters, cnt);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" __attribute__((visibility("default"))) int sec__debug_nms_vals(float *h256) {
    return hipMemcpyFromSymbol(h256, HIP_SYMBOL(g_nms_vals), 256 * sizeof(float)) == hipSuccess ? 0 : -4;
}
extern "C" __attribute__((visibility("default"))) int sec__debug_nms_counters(int *h4, int reset) {
    int z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h4, HIP_SYMBOL(g_nms_dbg), sizeof(z)) != hipSuccess) return -4;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_nms_dbg), z, sizeof(z)) != hipSuccess) return -4;
    return 0;
}
#endif
SEC_API size_t sec_nms_workspace_bytes(int batch, int max_n) {
    if (batch <= 0 || max_n <= 0) return 0;
    return align_up((size_t)batch * max_n * ((max_n + 63) / 64) * sizeof(unsigned long long));
}

SEC_API int sec_nms_sorted_f32(const float *dets, const int *counts, int batch, int max_n, int stride, float thresh,
                               int kind, int semantics, float eps, int post_max, int *keep, int *num_keep,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!dets || !counts || !keep || !num_keep || batch <= 0 || max_n <= 0 || max_n > 4096 ||
        stride < (kind == 0 ? 5 : 4) || kind < 0 || kind > 1 || (semantics & ~SEC_NMS_EXACT_CLIP) < 0 ||
        (semantics & ~SEC_NMS_EXACT_CLIP) > 1)
        return SEC_E_INVALID;
    if (!workspace || workspace_bytes < sec_nms_workspace_bytes(batch, max_n)) return SEC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int words = (max_n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)workspace;
    // a tile costs ~12 us start to finish (box corners, screen, one clipper round), so the launch is as long as the most tiles any
    // workgroup walks: enough workgroups that a few hundred candidates leave each at most one live tile (car.fhd batch 8, ~400
    // candidates per frame, nms_sorted: 64 per frame 44 us, 128: 35 us, 256: 28 us), capped so that 1000 candidates x a large batch
    // do not flood the chip: 2048 / batch, between 32 and 256
    int per_frame = 2048 / batch;
    per_frame = per_frame < 32 ? 32 : per_frame > 256 ? 256 : per_frame;
    const dim3 grid(per_frame, batch);
    hipLaunchKernelGGL(k_nms_mask<16>, grid, dim3(kBlock), 0, st, dets, counts, max_n, stride, thresh, kind, semantics, eps, words, mask);
    hipLaunchKernelGGL(k_nms_reduce, dim3(batch), dim3(64), 0, st, mask, counts, max_n, words, post_max, keep, num_keep);
    return check_launch();
}
