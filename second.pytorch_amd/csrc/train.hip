// Training-side kernels of the SECOND path on gfx950:
//
//   sec_assign_targets_f32  -- anchor <-> ground-truth matching + box encoding.  Replaces the worker-side numpy of
//                              second/core/target_ops.py:29-229 (create_target_np) with NearestIouSimilarity
//                              (second/core/region_similarity.py:73-93 -> box_np_ops.rbbox2d_to_near_bbox :286-298 +
//                              iou_jit(eps=0) :697-725) and GroundBox3dCoder.encode (box_np_ops.second_box_encode :36-81).
//   sec_second_loss_f32     -- VoxelNet.loss (second/pytorch/models/voxelnet.py:239-312): sigmoid focal classification loss
//                              (core/losses.py:236-296), smooth-L1 localisation loss on the sin-difference encoding
//                              (losses.py:135-185, voxelnet.py:704-754), direction softmax cross-entropy (:814-829,
//                              losses.py:358-392), NormByNumPositives weights (voxelnet.py:756-797) -- values AND the
//                              gradients w.r.t. the three head outputs in one pass; deterministic two-stage reduction.
//
// Both are HBM-bound elementwise / small-reduction work: one thread per (frame, anchor), the frame's ground-truth boxes
// staged in LDS.  fp32 throughout, operation order as in the numpy / torch originals (-ffp-contract=off).
#include "common.hpp"

namespace sec {

constexpr int kMaxGtLds = 256;          // ground-truth boxes of one frame processed per LDS chunk

__device__ __forceinline__ float limit_period_f(float v, float offset, float period) {
    return __fsub_rn(v, __fmul_rn(floorf(__fadd_rn(__fdiv_rn(v, period), offset)), period));
}

// rbbox2d_to_near_bbox of (x, y, w, l, r): the axis-aligned box of the nearer of the "standing" / "lying" orientation
__device__ __forceinline__ float4 near_bbox(float x, float y, float w, float l, float r) {
    const float kPi = 3.14159274101257324f;
    const float a = fabsf(limit_period_f(r, 0.5f, kPi));
    const bool swap = a > 0.785398185253143311f;      // np.pi / 4 in fp32
    const float dx = swap ? l : w, dy = swap ? w : l;
    return make_float4(__fsub_rn(x, __fdiv_rn(dx, 2.0f)), __fsub_rn(y, __fdiv_rn(dy, 2.0f)), __fadd_rn(x, __fdiv_rn(dx, 2.0f)),
                       __fadd_rn(y, __fdiv_rn(dy, 2.0f)));
}

// iou_jit(boxes = anchor, query = gt, eps = 0)
__device__ __forceinline__ float iou_eps0(const float4 a, const float4 q) {
    const float box_area = __fmul_rn(__fsub_rn(q.z, q.x), __fsub_rn(q.w, q.y));
    const float iw = __fsub_rn(fminf(a.z, q.z), fmaxf(a.x, q.x));
    if (!(iw > 0.0f)) return 0.0f;
    const float ih = __fsub_rn(fminf(a.w, q.w), fmaxf(a.y, q.y));
    if (!(ih > 0.0f)) return 0.0f;
    const float ua = __fsub_rn(__fadd_rn(__fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y)), box_area), __fmul_rn(iw, ih));
    return __fdiv_rn(__fmul_rn(iw, ih), ua);
}

// pass 1: per (frame, anchor) best ground truth (first index on ties, like np.argmax) and, per ground truth, the best overlap
// over all anchors (atomicMax on the float bits: overlaps are >= 0, so integer order == float order)
//
// assign_per_class (target_assigner.py:90-160): one launch pair per class handles the class's anchor range
// [a_begin, a_end) against the ground truth of that class only (`filter` = its 1-based id, 0 = every ground truth); boxes of
// other classes are skipped, and a frame without a box of the class labels the whole range background (len(gt_boxes) == 0).
struct AssignRange { int a_begin, a_end, filter; };

__global__ __launch_bounds__(kBlock) void k_assign_max(const float *__restrict__ anchors, int n_anchor,
                                                      const float *__restrict__ gt, const int *__restrict__ gt_classes,
                                                      const int *__restrict__ gt_offsets, AssignRange R,
                                                      float *__restrict__ a_max, int *__restrict__ a_arg,
                                                      int *__restrict__ gt_max_bits) {
    __shared__ float4 s_gt[kMaxGtLds];
    __shared__ unsigned char s_on[kMaxGtLds];
    const int b = blockIdx.y, a = R.a_begin + blockIdx.x * kBlock + threadIdx.x;
    const int g0 = gt_offsets[b], g1 = gt_offsets[b + 1];
    const int n_total = n_anchor;            // row pitch of the per-(frame, anchor) arrays
    n_anchor = min(n_anchor, R.a_end);       // below: a < n_anchor == this launch owns the anchor
    float4 abv = make_float4(0, 0, 0, 0);
    if (a < n_anchor) {
        const float *p = anchors + (size_t)a * 7;
        abv = near_bbox(p[0], p[1], p[3], p[4], p[6]);
    }
    float best = -1.0f;
    int arg = -1;
    for (int c0 = g0; c0 < g1; c0 += kMaxGtLds) {
        const int cn = min(kMaxGtLds, g1 - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < cn; i += kBlock) {
            const float *q = gt + (size_t)(c0 + i) * 7;
            s_gt[i] = near_bbox(q[0], q[1], q[3], q[4], q[6]);
            s_on[i] = (R.filter == 0 || (gt_classes ? gt_classes[c0 + i] : 1) == R.filter) ? 1 : 0;
        }
        __syncthreads();
        if (a < n_anchor) {
            for (int i = 0; i < cn; ++i) {
                if (!s_on[i]) continue;
                const float v = iou_eps0(abv, s_gt[i]);
                if (v > best) { best = v; arg = c0 + i - g0; }
                if (v > 0.0f) atomicMax(&gt_max_bits[c0 + i], __float_as_int(v));
            }
        }
    }
    if (a < n_anchor) {
        a_max[(size_t)b * n_total + a] = best;
        a_arg[(size_t)b * n_total + a] = arg;
    }
}

// pass 2: labels / box targets / importance of create_target_np (positive_fraction = None, no anchor pruning)
__global__ __launch_bounds__(kBlock) void k_assign_write(const float *__restrict__ anchors, int n_anchor,
                                                        const float *__restrict__ gt, const int *__restrict__ gt_classes,
                                                        const float *__restrict__ gt_importance,
                                                        const int *__restrict__ gt_offsets, AssignRange R,
                                                        const float *__restrict__ a_max,
                                                        const int *__restrict__ a_arg, const int *__restrict__ gt_max_bits,
                                                        float matched, float unmatched, int *__restrict__ labels,
                                                        float *__restrict__ targets, float *__restrict__ importance) {
    __shared__ float4 s_gt[kMaxGtLds];
    __shared__ float s_gmax[kMaxGtLds];
    __shared__ int s_any;
    const int b = blockIdx.y, a = R.a_begin + blockIdx.x * kBlock + threadIdx.x;
    const int g0 = gt_offsets[b], g1 = gt_offsets[b + 1];
    const size_t o = (size_t)b * n_anchor + a;
    n_anchor = min(n_anchor, R.a_end);
    if (threadIdx.x == 0) s_any = 0;
    float4 abv = make_float4(0, 0, 0, 0);
    const float *p = anchors + (size_t)(a < n_anchor ? a : 0) * 7;
    if (a < n_anchor) abv = near_bbox(p[0], p[1], p[3], p[4], p[6]);
    bool force = false;                       // "anchors_with_max_overlap": ties with some ground truth's best overlap
    for (int c0 = g0; c0 < g1; c0 += kMaxGtLds) {
        const int cn = min(kMaxGtLds, g1 - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < cn; i += kBlock) {
            const float *q = gt + (size_t)(c0 + i) * 7;
            s_gt[i] = near_bbox(q[0], q[1], q[3], q[4], q[6]);
            const float m = __int_as_float(gt_max_bits[c0 + i]);
            const bool on = R.filter == 0 || (gt_classes ? gt_classes[c0 + i] : 1) == R.filter;
            // a ground truth no anchor overlaps matches nothing (-1); one of another class can never be matched (-2: an IoU is >= 0)
            s_gmax[i] = !on ? -2.0f : (m == 0.0f ? -1.0f : m);
            if (on) s_any = 1;
        }
        __syncthreads();
        if (a < n_anchor)
            for (int i = 0; i < cn; ++i) force |= iou_eps0(abv, s_gt[i]) == s_gmax[i];
    }
    __syncthreads();
    if (a >= n_anchor) return;
    int label = -1;
    float imp = 1.0f;
    float t[7] = {0, 0, 0, 0, 0, 0, 0};
    if (s_any) {
        const float best = a_max[o];
        const int arg = a_arg[o];
        const bool pos = best >= matched;
        if (best < unmatched) label = 0;
        if (force || pos) label = gt_classes ? gt_classes[g0 + arg] : 1;
        if (pos && gt_importance) {
            // create_target_np reads gt_importance[index within the ground truth it was GIVEN]; assign_per_class hands it the
            // class's boxes but the frame's whole importance array (target_assigner.py:141): reproduce that indexing
            int local = arg;
            if (R.filter != 0 && gt_classes) {
                local = 0;
                for (int i = 0; i < arg; ++i) local += gt_classes[g0 + i] == R.filter ? 1 : 0;
            }
            imp = gt_importance[g0 + local];
        }
        if (label > 0) {   // second_box_encode(gt[arg], anchor)
            const float *q = gt + (size_t)(g0 + arg) * 7;
            const float diag = sqrtf(__fadd_rn(__fmul_rn(p[4], p[4]), __fmul_rn(p[3], p[3])));
            t[0] = __fdiv_rn(__fsub_rn(q[0], p[0]), diag);
            t[1] = __fdiv_rn(__fsub_rn(q[1], p[1]), diag);
            t[2] = __fdiv_rn(__fsub_rn(q[2], p[2]), p[5]);
            t[3] = logf(__fdiv_rn(q[3], p[3]));
            t[4] = logf(__fdiv_rn(q[4], p[4]));
            t[5] = logf(__fdiv_rn(q[5], p[5]));
            t[6] = __fsub_rn(q[6], p[6]);
        }
    } else {
        label = 0;
    }
    labels[o] = label;
    importance[o] = imp;
#pragma unroll
    for (int j = 0; j < 7; ++j) targets[o * 7 + j] = t[j];
}

// ------------------------------------------------------------------------------------------------ loss
struct LossParams {
    int batch, n_anchor, num_class, num_bins;
    float alpha, gamma, sigma, pos_w, neg_w, cls_w, loc_w, dir_w, dir_offset, sin_factor;
    float code_w[7];
};

constexpr int kCountChunks = 64;        // workgroups per frame of the positive count (deterministic: fixed chunks, fixed order)
__global__ __launch_bounds__(kBlock) void k_loss_count(const int *__restrict__ labels, int n_anchor, float *__restrict__ part,
                                                      const float *__restrict__ importance) {
    // per (frame, chunk): number of positives (weight normaliser) and sum of positive importance (direction weights)
    __shared__ float s[2][kBlock / 64];
    const int b = blockIdx.y, c = blockIdx.x;
    const int per = (n_anchor + kCountChunks - 1) / kCountChunks;
    const int lo = c * per, hi = min(n_anchor, lo + per);
    float np_ = 0.0f, wi = 0.0f;
    for (int a = lo + threadIdx.x; a < hi; a += kBlock) {
        const bool pos = labels[(size_t)b * n_anchor + a] > 0;
        np_ += pos ? 1.0f : 0.0f;
        wi += pos ? importance[(size_t)b * n_anchor + a] : 0.0f;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { np_ += __shfl_xor(np_, d, 64); wi += __shfl_xor(wi, d, 64); }
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = np_; s[1][threadIdx.x >> 6] = wi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0, a1 = 0;
        for (int i = 0; i < kBlock / 64; ++i) { a0 += s[0][i]; a1 += s[1][i]; }
        part[((size_t)b * kCountChunks + c) * 2] = a0;
        part[((size_t)b * kCountChunks + c) * 2 + 1] = a1;
    }
}

// per (frame, anchor): the three loss terms and their gradients; per-block partial sums [nblocks][6] =
// (cls, loc, dir, cls_pos, cls_neg, -).  cls_preds [B, N, num_class] etc. contiguous; gradients of the TOTAL loss
// loc_w * loc / B + cls_w * cls / B + dir_w * dir / B.
__global__ __launch_bounds__(kBlock) void k_loss_main(const float *__restrict__ cls, const float *__restrict__ box,
                                                     const float *__restrict__ dirp, const int *__restrict__ labels,
                                                     const float *__restrict__ reg, const float *__restrict__ anchors,
                                                     const float *__restrict__ importance, const float *__restrict__ cnt,
                                                     LossParams P, float *__restrict__ d_cls, float *__restrict__ d_box,
                                                     float *__restrict__ d_dir, float *__restrict__ partial) {
    const int b = blockIdx.y, a = blockIdx.x * kBlock + threadIdx.x;
    float s_cls = 0, s_loc = 0, s_dir = 0, s_pos = 0, s_neg = 0;
    __shared__ float s_norm[2];
    if (threadIdx.x < 2) {                // the frame's normalisers: fixed-order sum of the chunk counts, clamped to >= 1
        float acc = 0.0f;
        for (int c = 0; c < kCountChunks; ++c) acc += cnt[((size_t)b * kCountChunks + c) * 2 + threadIdx.x];
        s_norm[threadIdx.x] = fmaxf(acc, 1.0f);
    }
    __syncthreads();
    if (a < P.n_anchor) {
        const size_t o = (size_t)b * P.n_anchor + a;
        const int label = labels[o];
        const float imp = importance[o];
        const float inv_b = 1.0f / (float)P.batch;
        const float norm = s_norm[0];
        const bool pos = label > 0, neg = label == 0;
        // ---- classification (focal, background encoded as zeros): target one-hot over classes 1..num_class
        const float wcls = ((neg ? P.neg_w : 0.0f) + (pos ? P.pos_w : 0.0f)) / norm * imp;
        for (int c = 0; c < P.num_class; ++c) {
            const float x = cls[o * P.num_class + c];
            const float t = (label == c + 1) ? 1.0f : 0.0f;
            const float ce = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
            const float p = 1.0f / (1.0f + expf(-x));
            const float pt = t * p + (1.0f - t) * (1.0f - p);
            const float om = 1.0f - pt;
            const float mod = P.gamma == 2.0f ? om * om : (P.gamma == 0.0f ? 1.0f : powf(om, P.gamma));
            const float at = t * P.alpha + (1.0f - t) * (1.0f - P.alpha);
            const float l = mod * at * ce * wcls;
            s_cls += l;
            // _get_pos_neg_loss (voxelnet.py:20-34): one class -> by label; several -> columns 1.. vs column 0 of the loss
            if (P.num_class == 1) { s_pos += pos ? l : 0.0f; s_neg += neg ? l : 0.0f; }
            else { s_pos += c > 0 ? l : 0.0f; s_neg += c == 0 ? l : 0.0f; }
            // d/dx: at * w * [ dmod/dx * ce + mod * (p - t) ],  dmod/dx = -gamma * om^(gamma-1) * dpt/dx,  dpt/dx = (2t-1) p (1-p)
            const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
            const float dmod = P.gamma == 2.0f ? -2.0f * om * dpt : (P.gamma == 0.0f ? 0.0f : -P.gamma * powf(om, P.gamma - 1.0f) * dpt);
            d_cls[o * P.num_class + c] = at * wcls * (dmod * ce + mod * (p - t)) * P.cls_w * inv_b;
        }
        // ---- localisation (smooth L1 on the sin-difference encoding), positives only
        const float wreg = (pos ? 1.0f : 0.0f) / norm * imp;
        const float s2 = P.sigma * P.sigma;
        const float pr = box[o * 7 + 6] * P.sin_factor, tr = reg[o * 7 + 6] * P.sin_factor;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            float pv = box[o * 7 + j], tv = reg[o * 7 + j], dscale = 1.0f;
            if (j == 6) {
                pv = sinf(pr) * cosf(tr);
                tv = cosf(pr) * sinf(tr);
                dscale = (cosf(pr) * cosf(tr) + sinf(pr) * sinf(tr)) * P.sin_factor;     // both encodings depend on the prediction
            }
            const float diff = P.code_w[j] * (pv - tv);
            const float ad = fabsf(diff);
            const bool small = ad <= 1.0f / s2;
            const float l = small ? 0.5f * (ad * P.sigma) * (ad * P.sigma) : ad - 0.5f / s2;
            s_loc += l * wreg;
            const float dl = small ? s2 * diff : (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
            d_box[o * 7 + j] = dl * P.code_w[j] * dscale * wreg * P.loc_w * inv_b;
        }
        // ---- direction classifier (softmax cross-entropy on the heading bin of the ground truth)
        if (P.num_bins > 0) {
            const float kTwoPi = 6.28318548202514648f;
            const float rot_gt = reg[o * 7 + 6] + anchors[(size_t)a * 7 + 6];
            const float off = limit_period_f(rot_gt - P.dir_offset, 0.0f, kTwoPi);
            int bin = (int)floorf(off / (kTwoPi / (float)P.num_bins));
            bin = bin < 0 ? 0 : (bin > P.num_bins - 1 ? P.num_bins - 1 : bin);
            const float wdir = (pos ? imp : 0.0f) / s_norm[1];
            float mx = -3.0e38f;
            for (int c = 0; c < P.num_bins; ++c) mx = fmaxf(mx, dirp[o * P.num_bins + c]);
            float se = 0.0f;
            for (int c = 0; c < P.num_bins; ++c) se += expf(dirp[o * P.num_bins + c] - mx);
            const float lse = mx + logf(se);
            s_dir += (lse - dirp[o * P.num_bins + bin]) * wdir;
            for (int c = 0; c < P.num_bins; ++c) {
                const float sm = expf(dirp[o * P.num_bins + c] - lse);
                d_dir[o * P.num_bins + c] = (sm - (c == bin ? 1.0f : 0.0f)) * wdir * P.dir_w * inv_b;
            }
        }
    }
    // block reduction -> partial[(b * gridDim.x + blockIdx.x)][6]
    __shared__ float red[5][kBlock / 64];
    float v[5] = {s_cls, s_loc, s_dir, s_pos, s_neg};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v[i] += __shfl_xor(v[i], d, 64);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float acc = 0.0f;
        for (int w = 0; w < kBlock / 64; ++w) acc += red[threadIdx.x][w];
        partial[((size_t)b * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = acc;
    }
}

// out[0..5] = loss, cls_loss_reduced, loc_loss_reduced, dir_loss_reduced, cls_pos_loss, cls_neg_loss (fixed summation order)
__global__ __launch_bounds__(kBlock) void k_loss_final(const float *__restrict__ partial, int nparts, LossParams P,
                                                      float *__restrict__ out) {
    __shared__ float red[5][kBlock];
    float v[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nparts; i += kBlock)
#pragma unroll
        for (int j = 0; j < 5; ++j) v[j] += partial[(size_t)i * 6 + j];
#pragma unroll
    for (int j = 0; j < 5; ++j) red[j][threadIdx.x] = v[j];
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int j = 0; j < 5; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float inv_b = 1.0f / (float)P.batch;
        const float cls = red[0][0] * inv_b * P.cls_w, loc = red[1][0] * inv_b * P.loc_w, dir = red[2][0] * inv_b;
        out[0] = loc + cls + dir * P.dir_w;
        out[1] = cls;
        out[2] = loc;
        out[3] = dir;
        out[4] = red[3][0] * inv_b / P.pos_w;
        out[5] = red[4][0] * inv_b / P.neg_w;
    }
}


// ------------------------------------------------------------------------------------------------ loss on the stacked heads
// The training step's 1x1 heads are ONE 128 -> 64 convolution whose output y [B, H, W, 64] (channels last, 16 bit) stacks
// box [A * 7] | cls [A * NC] | dir [A * BINS] | zero padding (ops.Heads1x1Function; rpn.py:386-391).  The reference views each head
// as [B, A, H, W, code] (anchor n = (a * H + y) * W + x) and hands three fp32 copies to the loss; autograd then stitches the three
// gradients back into dY: ~35 small torch launches, 0.25 ms of a 3.5 ms step.  Here the loss reads y and writes dY:
//   k_heads_loss<.., false>  the six loss scalars (forward)
//   k_heads_loss<.., true>   dY = g * d loss / d y in y's own layout and dtype (rounded once, after the multiplication by the incoming
//                            gradient g -- the loss scale of fp16 training) + per-workgroup column sums for the bias gradient
// A thread owns a pixel: it loads the first 16-byte chunks of its row, runs the A anchors through the arithmetic of k_loss_main
// (same expressions, same order), and writes the row back -- all indices static, everything in registers.
template <typename HT> __device__ __forceinline__ float ht2f(HT v);
template <> __device__ __forceinline__ float ht2f(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float ht2f(__half v) { return __half2float(v); }
template <typename HT> __device__ __forceinline__ HT f2ht(float v);
template <> __device__ __forceinline__ __hip_bfloat16 f2ht(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ __half f2ht(float v) { return __float2half_rn(v); }

template <typename HT, int A, int NC, int BINS, bool GRAD>
__global__ __launch_bounds__(kBlock) void k_heads_loss(const HT *__restrict__ heads, int HW, int HC, const int *__restrict__ labels,
                                                      const float *__restrict__ reg, const float *__restrict__ anchors,
                                                      const float *__restrict__ importance, const float *__restrict__ cnt, LossParams P,
                                                      const float *__restrict__ g_loss, HT *__restrict__ d_heads,
                                                      float *__restrict__ partial, float *__restrict__ o_cls_pred,
                                                      float *__restrict__ o_cls_loss, float *__restrict__ o_loc_loss) {
    // o_*: forward only, optional -- the per-anchor tensors VoxelNet.loss also returns (voxelnet.py:299-309: cls_preds as fp32
    // [B, n_anchor, NC], cls_loss [B, n_anchor, NC], loc_loss [B, n_anchor, 7]) for the drop-in training forward
    constexpr int TOT = A * (7 + NC + BINS), NCH = (TOT + 7) / 8;       // 16-byte chunks of a row that hold head outputs
    constexpr int BOX0 = 0, CLS0 = A * 7, DIR0 = A * 7 + A * NC;
    const int b = blockIdx.y, pix = blockIdx.x * kBlock + threadIdx.x;
    __shared__ float s_norm[2];
    if (threadIdx.x < 2) {                // the frame's normalisers: fixed-order sum of the chunk counts, clamped to >= 1
        float acc = 0.0f;
        for (int c = 0; c < kCountChunks; ++c) acc += cnt[((size_t)b * kCountChunks + c) * 2 + threadIdx.x];
        s_norm[threadIdx.x] = fmaxf(acc, 1.0f);
    }
    __syncthreads();
    float s_cls = 0, s_loc = 0, s_dir = 0, s_pos = 0, s_neg = 0;
    float v[NCH * 8], d[NCH * 8];
#pragma unroll
    for (int i = 0; i < NCH * 8; ++i) { v[i] = 0.0f; d[i] = 0.0f; }
    const bool live = pix < HW;
    const float gs = GRAD ? (g_loss ? g_loss[0] : 1.0f) : 0.0f;
    if (live) {
        const uint4 *row = reinterpret_cast<const uint4 *>(heads + ((size_t)b * HW + pix) * HC);
        uint4 q[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) q[i] = row[i];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const HT *e = reinterpret_cast<const HT *>(&q[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i * 8 + j] = ht2f<HT>(e[j]);
        }
        const float inv_b = 1.0f / (float)P.batch;
        const float norm = s_norm[0];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const int n = a * HW + pix;                                  // the reference's anchor index
            const size_t o = (size_t)b * P.n_anchor + n;
            const int label = labels[o];
            const float imp = importance[o];
            const bool pos = label > 0, neg = label == 0;
            // ---- classification (focal, background encoded as zeros)
            const float wcls = ((neg ? P.neg_w : 0.0f) + (pos ? P.pos_w : 0.0f)) / norm * imp;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float x = v[CLS0 + a * NC + c];
                const float t = (label == c + 1) ? 1.0f : 0.0f;
                const float ce = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
                const float p = 1.0f / (1.0f + expf(-x));
                const float pt = t * p + (1.0f - t) * (1.0f - p);
                const float om = 1.0f - pt;
                const float mod = P.gamma == 2.0f ? om * om : (P.gamma == 0.0f ? 1.0f : powf(om, P.gamma));
                const float at = t * P.alpha + (1.0f - t) * (1.0f - P.alpha);
                const float l = mod * at * ce * wcls;
                s_cls += l;
                if (NC == 1) { s_pos += pos ? l : 0.0f; s_neg += neg ? l : 0.0f; }
                else { s_pos += c > 0 ? l : 0.0f; s_neg += c == 0 ? l : 0.0f; }
                if (!GRAD && o_cls_pred) o_cls_pred[o * NC + c] = x;
                if (!GRAD && o_cls_loss) o_cls_loss[o * NC + c] = l;
                if (GRAD) {
                    const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
                    const float dmod = P.gamma == 2.0f ? -2.0f * om * dpt : (P.gamma == 0.0f ? 0.0f : -P.gamma * powf(om, P.gamma - 1.0f) * dpt);
                    d[CLS0 + a * NC + c] = at * wcls * (dmod * ce + mod * (p - t)) * P.cls_w * inv_b;
                }
            }
            // ---- localisation (smooth L1 on the sin-difference encoding), positives only
            const float wreg = (pos ? 1.0f : 0.0f) / norm * imp;
            const float s2 = P.sigma * P.sigma;
            float tg[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) tg[j] = reg[o * 7 + j];
            const float pr = v[BOX0 + a * 7 + 6] * P.sin_factor, tr = tg[6] * P.sin_factor;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float pv = v[BOX0 + a * 7 + j], tv = tg[j], dscale = 1.0f;
                if (j == 6) {
                    pv = sinf(pr) * cosf(tr);
                    tv = cosf(pr) * sinf(tr);
                    dscale = (cosf(pr) * cosf(tr) + sinf(pr) * sinf(tr)) * P.sin_factor;
                }
                const float diff = P.code_w[j] * (pv - tv);
                const float ad = fabsf(diff);
                const bool small = ad <= 1.0f / s2;
                const float l = small ? 0.5f * (ad * P.sigma) * (ad * P.sigma) : ad - 0.5f / s2;
                s_loc += l * wreg;
                if (!GRAD && o_loc_loss) o_loc_loss[o * 7 + j] = l * wreg;
                if (GRAD) {
                    const float dl = small ? s2 * diff : (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
                    d[BOX0 + a * 7 + j] = dl * P.code_w[j] * dscale * wreg * P.loc_w * inv_b;
                }
            }
            // ---- direction classifier (softmax cross-entropy on the heading bin of the ground truth)
            if (BINS > 0) {
                const float kTwoPi = 6.28318548202514648f;
                const float rot_gt = tg[6] + anchors[(size_t)n * 7 + 6];
                const float off = limit_period_f(rot_gt - P.dir_offset, 0.0f, kTwoPi);
                int bin = (int)floorf(off / (kTwoPi / (float)BINS));
                bin = bin < 0 ? 0 : (bin > BINS - 1 ? BINS - 1 : bin);
                const float wdir = (pos ? imp : 0.0f) / s_norm[1];
                float mx = -3.0e38f;
#pragma unroll
                for (int c = 0; c < BINS; ++c) mx = fmaxf(mx, v[DIR0 + a * BINS + c]);
                float se = 0.0f;
#pragma unroll
                for (int c = 0; c < BINS; ++c) se += expf(v[DIR0 + a * BINS + c] - mx);
                const float lse = mx + logf(se);
                float at_bin = 0.0f;
#pragma unroll
                for (int c = 0; c < BINS; ++c) at_bin = c == bin ? v[DIR0 + a * BINS + c] : at_bin;
                s_dir += (lse - at_bin) * wdir;
                if (GRAD) {
#pragma unroll
                    for (int c = 0; c < BINS; ++c) {
                        const float sm = expf(v[DIR0 + a * BINS + c] - lse);
                        d[DIR0 + a * BINS + c] = (sm - (c == bin ? 1.0f : 0.0f)) * wdir * P.dir_w * inv_b;
                    }
                }
            }
        }
    }
    if (!GRAD) {
        // block reduction -> partial[(b * gridDim.x + blockIdx.x)][6]
        __shared__ float red[5][kBlock / 64];
        float r5[5] = {s_cls, s_loc, s_dir, s_pos, s_neg};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) r5[i] += __shfl_xor(r5[i], sft, 64);
            if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = r5[i];
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            float acc = 0.0f;
            for (int w = 0; w < kBlock / 64; ++w) acc += red[threadIdx.x][w];
            partial[((size_t)b * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = acc;
        }
    } else {
        // dY row: the gradient times the incoming scalar, rounded ONCE to the heads' dtype; the padding channels are zeros
        float dq[NCH * 8];
#pragma unroll
        for (int i = 0; i < NCH * 8; ++i) {
            const HT r = f2ht<HT>(d[i] * gs);
            dq[i] = ht2f<HT>(r);                          // the bias gradient sums what dY holds (autograd summed the rounded tensor)
            d[i] = dq[i];
        }
        if (live) {
            uint4 *orow = reinterpret_cast<uint4 *>(d_heads + ((size_t)b * HW + pix) * HC);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                uint4 q;
                HT *e = reinterpret_cast<HT *>(&q);
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = f2ht<HT>(d[i * 8 + j]);
                orow[i] = q;
            }
            for (int i = NCH; i < HC / 8; ++i) orow[i] = make_uint4(0, 0, 0, 0);
        }
        // column sums of this workgroup's rows -> partial[(b * gridDim.x + blockIdx.x)][NCH * 8] (fixed order)
        __shared__ float red[kBlock / 64][NCH * 8];
#pragma unroll
        for (int i = 0; i < NCH * 8; ++i) {
            float t = dq[i];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) t += __shfl_xor(t, sft, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = t;
        }
        __syncthreads();
        if (threadIdx.x < NCH * 8) {
            float acc = 0.0f;
            for (int w = 0; w < kBlock / 64; ++w) acc += red[w][threadIdx.x];
            partial[((size_t)b * gridDim.x + blockIdx.x) * (NCH * 8) + threadIdx.x] = acc;
        }
    }
}

// d_bias[c] = sum of the workgroups' column sums (a fixed order: run-to-run identical); channels behind the heads get 0.
// 32 columns x 8 slices of the partial list per workgroup, eight loads in flight per thread (one thread per channel walking 552
// partials one dependent load at a time took 34 us).
__global__ __launch_bounds__(kBlock) void k_heads_bias_final(const float *__restrict__ partial, int nparts, int cols, int HC, float *__restrict__ d_bias) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, part = threadIdx.x >> 5;          // cols <= 32 (the instantiated head shapes hold 20 or 24)
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (c < cols) {
        int i = part;
        for (; i + 56 < nparts; i += 64) {
            const float *p = partial + (size_t)i * cols + c;
            const float v0 = p[0], v1 = p[(size_t)8 * cols], v2 = p[(size_t)16 * cols], v3 = p[(size_t)24 * cols];
            const float v4 = p[(size_t)32 * cols], v5 = p[(size_t)40 * cols], v6 = p[(size_t)48 * cols], v7 = p[(size_t)56 * cols];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            a0 += v4; a1 += v5; a2 += v6; a3 += v7;
        }
        for (; i < nparts; i += 8) a0 += partial[(size_t)i * cols + c];
    }
    red[part][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < HC && threadIdx.x < 64) {
        float t = 0.0f;
        if (threadIdx.x < cols && threadIdx.x < 32)
            for (int w = 0; w < 8; ++w) t += red[w][threadIdx.x];
        d_bias[threadIdx.x] = t;
    }
}

// ---- clip_grad_norm_ + AdamW on ONE flat fp32 parameter buffer -----------------------------------------------------------------------
// second/pytorch/train.py:323-325: torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0); mixed_optimizer.step() (adam + fixed
// weight decay, car.fhd.config:180-188).  With torch.optim.AdamW over 69 parameter tensors a captured step spends ~0.6 ms here: the
// capturable implementation computes beta^step per PARAMETER (2 x 69 one-element pow launches) besides ~15 multi-tensor kernels.
// The device trainer keeps all master weights as views of one flat buffer whose gradient is the all-reduce bucket, so the update is
// two launches: (1) sum of squares of the flat gradient in fixed-order partials; the last workgroup to finish (ticket) adds them in
// index order, stores the norm and advances the step counter; (2) the element-wise update with the clip factor
// min(1, max_norm / (norm + 1e-6)) folded in.  Same formulas as torch: p *= 1 - lr * wd; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
constexpr int kAdamBlocks = 512;
// Dynamic loss scaling on the device (fp16 features; the reference trains mixed precision through apex amp, train.py:209-216,
// 318-322): scale4 (or NULL) = (loss scale, clean steps in a row, growth interval, skipped steps).  The gradients in `g` carry the
// scale; the norm is taken of g / scale.  A non-finite norm (every rank sees the same reduced bucket, so every rank decides alike)
// halves the scale and makes the update launch a no-op; `growth interval` clean steps in a row double it.  No host read.
__global__ __launch_bounds__(kBlock) void k_flat_sumsq(const float *__restrict__ g, long long n, float *__restrict__ part,
                                                      unsigned *__restrict__ ticket, float *__restrict__ state /* [norm, step, skip, scale used] */,
                                                      float *__restrict__ scale4) {
    __shared__ float red[kBlock];
    __shared__ bool last;
    float a = 0.0f;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) { const float v = g[i]; a += v * v; }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = red[0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double t = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock) t += (double)__hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ double redd[kBlock];
    redd[threadIdx.x] = t;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) redd[threadIdx.x] += redd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float norm = (float)sqrt(redd[0]);
        bool skip = false;
        state[3] = scale4 ? scale4[0] : 1.0f;          // the scale these gradients were produced with (the update divides by it)
        if (scale4) {
            norm /= scale4[0];
            skip = !(norm == norm) || norm > 3.0e38f;     // NaN or Inf somewhere in the reduced bucket
            if (skip) { scale4[0] = fmaxf(scale4[0] * 0.5f, 1.0f); scale4[1] = 0.0f; scale4[3] += 1.0f; }
            else if ((scale4[1] += 1.0f) >= scale4[2]) { scale4[0] = fminf(scale4[0] * 2.0f, 16777216.0f); scale4[1] = 0.0f; }
        }
        state[0] = norm;
        if (!skip) state[1] += 1.0f;
        state[2] = skip ? 1.0f : 0.0f;
        *ticket = 0u;                                   // ready for the next step (graph replays)
    }
}
// `hyper` != NULL: (lr, beta1, beta2, eps, weight decay, max gradient norm) are read from device memory -- a captured step follows a
// learning-rate schedule (the reference's one-cycle schedule changes lr every step, car.fhd.config:171-188) without re-capture.
__global__ __launch_bounds__(kBlock) void k_flat_adamw(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                      float *__restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                      float wd, float max_norm, const float *__restrict__ state,
                                                      const float *__restrict__ hyper) {
    if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; max_norm = hyper[5]; }
    if (state[2] != 0.0f) return;                        // overflow: the step is skipped on every rank
    const float unscale = 1.0f / state[3];
    const float norm = state[0], step = state[1];
    float clip = max_norm > 0.0f ? max_norm / (norm + 1e-6f) : 1.0f;
    if (clip > 1.0f) clip = 1.0f;
    const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
    const float step_size = lr / bc1, rs2 = sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float gi = g[i] * unscale * clip;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);                 // lerp, as torch
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / rs2 + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

}  // namespace sec

using namespace sec;

SEC_API size_t sec_assign_targets_workspace_bytes(int batch, int n_anchor, int n_gt) {
    if (batch < 0 || n_anchor < 0 || n_gt < 0) return 0;
    return align_up((size_t)batch * n_anchor * sizeof(float)) + align_up((size_t)batch * n_anchor * sizeof(int)) +
           align_up((size_t)(n_gt > 0 ? n_gt : 1) * sizeof(int)) + 256;
}

SEC_API int sec_assign_targets_f32(const float *anchors, int n_anchor, const float *gt_boxes, const int *gt_classes,
                                   const float *gt_importance, const int *gt_offsets, int n_gt, int batch,
                                   float matched_threshold, float unmatched_threshold, int *labels, float *bbox_targets,
                                   float *importance, void *workspace, size_t workspace_bytes, void *stream) {
    if (n_anchor <= 0 || batch <= 0 || n_gt < 0 || !anchors || !gt_offsets || !labels || !bbox_targets || !importance ||
        (n_gt > 0 && !gt_boxes))
        return SEC_E_INVALID;
    if (!workspace || workspace_bytes < sec_assign_targets_workspace_bytes(batch, n_anchor, n_gt)) return SEC_E_WORKSPACE;
    Arena ar(workspace, workspace_bytes);
    float *a_max = ar.take<float>((size_t)batch * n_anchor);
    int *a_arg = ar.take<int>((size_t)batch * n_anchor);
    int *gt_max = ar.take<int>(n_gt > 0 ? n_gt : 1);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = fill_words(gt_max, (size_t)(n_gt > 0 ? n_gt : 1) * sizeof(int), 0u, st))) return rc;     // (a kernel: this runs inside captured training steps)
    dim3 grid(div_up(n_anchor, kBlock), batch);
    const AssignRange all{0, n_anchor, 0};
    hipLaunchKernelGGL(k_assign_max, grid, dim3(kBlock), 0, st, anchors, n_anchor, gt_boxes, gt_classes, gt_offsets, all, a_max,
                       a_arg, gt_max);
    hipLaunchKernelGGL(k_assign_write, grid, dim3(kBlock), 0, st, anchors, n_anchor, gt_boxes, gt_classes, gt_importance,
                       gt_offsets, all, a_max, a_arg, gt_max, matched_threshold, unmatched_threshold, labels, bbox_targets,
                       importance);
    return check_launch();
}

SEC_API int sec_assign_targets_per_class_f32(const float *anchors, int n_anchor, const float *gt_boxes, const int *gt_classes,
                                             const float *gt_importance, const int *gt_offsets, int n_gt, int batch,
                                             int n_class, const int *h_class_anchor_begin, const int *h_class_ids,
                                             const float *h_matched, const float *h_unmatched, int *labels,
                                             float *bbox_targets, float *importance, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    if (n_anchor <= 0 || batch <= 0 || n_gt < 0 || n_class <= 0 || !anchors || !gt_offsets || !labels || !bbox_targets ||
        !importance || !h_class_anchor_begin || !h_class_ids || !h_matched || !h_unmatched || (n_gt > 0 && (!gt_boxes || !gt_classes)))
        return SEC_E_INVALID;
    if (h_class_anchor_begin[0] != 0 || h_class_anchor_begin[n_class] != n_anchor) return SEC_E_INVALID;
    for (int c = 0; c < n_class; ++c)
        if (h_class_anchor_begin[c + 1] < h_class_anchor_begin[c] || h_class_ids[c] < 0) return SEC_E_INVALID;
    if (!workspace || workspace_bytes < sec_assign_targets_workspace_bytes(batch, n_anchor, n_gt)) return SEC_E_WORKSPACE;
    Arena ar(workspace, workspace_bytes);
    float *a_max = ar.take<float>((size_t)batch * n_anchor);
    int *a_arg = ar.take<int>((size_t)batch * n_anchor);
    int *gt_max = ar.take<int>(n_gt > 0 ? n_gt : 1);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // one best-overlap word per ground truth.  assign_per_class: a box belongs to one class, so the ranges touch disjoint
    // words; assign_all (class id 0 = every ground truth): the best overlap is taken over ALL anchors (target_ops.py:108-112),
    // hence every range's first pass before any second pass.
    if ((rc = fill_words(gt_max, (size_t)(n_gt > 0 ? n_gt : 1) * sizeof(int), 0u, st))) return rc;     // (a kernel: this runs inside captured training steps)
    for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < n_class; ++c) {
            const AssignRange R{h_class_anchor_begin[c], h_class_anchor_begin[c + 1], h_class_ids[c]};
            if (R.a_end == R.a_begin) continue;
            dim3 grid(div_up(R.a_end - R.a_begin, kBlock), batch);
            if (pass == 0)
                hipLaunchKernelGGL(k_assign_max, grid, dim3(kBlock), 0, st, anchors, n_anchor, gt_boxes, gt_classes, gt_offsets, R,
                                   a_max, a_arg, gt_max);
            else
                hipLaunchKernelGGL(k_assign_write, grid, dim3(kBlock), 0, st, anchors, n_anchor, gt_boxes, gt_classes,
                                   gt_importance, gt_offsets, R, a_max, a_arg, gt_max, h_matched[c], h_unmatched[c], labels,
                                   bbox_targets, importance);
        }
    return check_launch();
}

SEC_API size_t sec_second_loss_workspace_bytes(int batch, int n_anchor) {
    if (batch <= 0 || n_anchor <= 0) return 0;
    return align_up((size_t)2 * batch * kCountChunks * sizeof(float)) + align_up((size_t)batch * div_up(n_anchor, kBlock) * 6 * sizeof(float)) + 256;
}

SEC_API int sec_second_loss_f32(const float *cls_preds, const float *box_preds, const float *dir_preds, const int *labels,
                                const float *reg_targets, const float *anchors, const float *importance, int batch,
                                int n_anchor, int num_class, int num_dir_bins, const float *h_params17, float *d_cls, float *d_box,
                                float *d_dir, float *out6, void *workspace, size_t workspace_bytes, void *stream) {
    if (batch <= 0 || n_anchor <= 0 || num_class <= 0 || num_dir_bins < 0 || !cls_preds || !box_preds || !labels || !reg_targets ||
        !anchors || !importance || !h_params17 || !d_cls || !d_box || !out6 || (num_dir_bins > 0 && (!dir_preds || !d_dir)))
        return SEC_E_INVALID;
    if (!workspace || workspace_bytes < sec_second_loss_workspace_bytes(batch, n_anchor)) return SEC_E_WORKSPACE;
    LossParams P;
    P.batch = batch; P.n_anchor = n_anchor; P.num_class = num_class; P.num_bins = num_dir_bins;
    // h_params17: alpha, gamma, sigma, pos_cls_weight, neg_cls_weight, cls_loss_weight, loc_loss_weight, dir_loss_weight,
    //             dir_offset, sin_error_factor, then (optional, else 1) nothing -- code weights follow in [10..16]
    P.alpha = h_params17[0]; P.gamma = h_params17[1]; P.sigma = h_params17[2]; P.pos_w = h_params17[3]; P.neg_w = h_params17[4];
    P.cls_w = h_params17[5]; P.loc_w = h_params17[6]; P.dir_w = h_params17[7]; P.dir_offset = h_params17[8];
    P.sin_factor = h_params17[9];
    for (int j = 0; j < 7; ++j) P.code_w[j] = h_params17[10 + j];
    Arena ar(workspace, workspace_bytes);
    float *cnt = ar.take<float>((size_t)2 * batch * kCountChunks);
    const int nb = div_up(n_anchor, kBlock);
    float *partial = ar.take<float>((size_t)batch * nb * 6);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_count, dim3(kCountChunks, batch), dim3(kBlock), 0, st, labels, n_anchor, cnt, importance);
    hipLaunchKernelGGL(k_loss_main, dim3(nb, batch), dim3(kBlock), 0, st, cls_preds, box_preds, dir_preds, labels, reg_targets,
                       anchors, importance, cnt, P, d_cls, d_box, d_dir, partial);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(kBlock), 0, st, partial, batch * nb, P, out6);
    return check_launch();
}

// ---- the loss on the stacked heads (k_heads_loss): forward values, backward dY + bias gradient
static bool heads_loss_shape_ok(int head_channels, int a, int nc, int bins) {
    return head_channels == 64 && a == 2 && nc == 1 && (bins == 2 || bins == 0);
}
SEC_API int sec_heads_loss_supported(int head_channels, int anchors_per_loc, int num_class, int num_dir_bins, int dtype) {
    return (dtype == SEC_BF16 || dtype == SEC_F16) && heads_loss_shape_ok(head_channels, anchors_per_loc, num_class, num_dir_bins) ? 1 : 0;
}
SEC_API size_t sec_heads_loss_workspace_bytes(int batch, int h, int w, int anchors_per_loc) {
    if (batch <= 0 || h <= 0 || w <= 0 || anchors_per_loc <= 0) return 0;
    const long long nb = div_up((long long)h * w, kBlock);
    return align_up((size_t)2 * batch * kCountChunks * sizeof(float)) + align_up((size_t)batch * nb * 64 * sizeof(float)) + 256;
}
static void fill_loss_params(LossParams &P, int batch, int n_anchor, int nc, int bins, const float *h) {
    P.batch = batch; P.n_anchor = n_anchor; P.num_class = nc; P.num_bins = bins;
    P.alpha = h[0]; P.gamma = h[1]; P.sigma = h[2]; P.pos_w = h[3]; P.neg_w = h[4];
    P.cls_w = h[5]; P.loc_w = h[6]; P.dir_w = h[7]; P.dir_offset = h[8];
    P.sin_factor = h[9];
    for (int j = 0; j < 7; ++j) P.code_w[j] = h[10 + j];
}
template <typename HT, bool GRAD>
static void launch_heads_loss(int bins, dim3 grid, hipStream_t st, const void *heads, int HW, int HC, const int *labels, const float *reg,
                              const float *anchors, const float *importance, const float *cnt, const LossParams &P, const float *g,
                              void *d_heads, float *partial, float *const *terms = nullptr) {
    float *t0 = terms ? terms[0] : nullptr, *t1 = terms ? terms[1] : nullptr, *t2 = terms ? terms[2] : nullptr;
    if (bins == 2)
        hipLaunchKernelGGL((k_heads_loss<HT, 2, 1, 2, GRAD>), grid, dim3(kBlock), 0, st, (const HT *)heads, HW, HC, labels, reg, anchors, importance,
                           cnt, P, g, (HT *)d_heads, partial, t0, t1, t2);
    else
        hipLaunchKernelGGL((k_heads_loss<HT, 2, 1, 0, GRAD>), grid, dim3(kBlock), 0, st, (const HT *)heads, HW, HC, labels, reg, anchors, importance,
                           cnt, P, g, (HT *)d_heads, partial, t0, t1, t2);
}
static int heads_loss_impl(bool grad, const void *heads, int dtype, int batch, int h, int w, int head_channels, int a, int nc, int bins,
                           const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                           const float *h_params17, const float *grad_loss, void *d_heads, float *d_bias, float *out6, void *workspace,
                           size_t workspace_bytes, bool counts_ready, void *stream, float *const *terms = nullptr) {
    if (!heads || batch <= 0 || h <= 0 || w <= 0 || !labels || !reg_targets || !anchors || !importance || !h_params17 ||
        (grad ? (!d_heads || !d_bias) : !out6))
        return SEC_E_INVALID;
    if (!sec_heads_loss_supported(head_channels, a, nc, bins, dtype)) return SEC_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < sec_heads_loss_workspace_bytes(batch, h, w, a)) return SEC_E_WORKSPACE;
    const int HW = h * w, n_anchor = a * HW;
    LossParams P;
    fill_loss_params(P, batch, n_anchor, nc, bins, h_params17);
    Arena ar(workspace, workspace_bytes);
    float *cnt = ar.take<float>((size_t)2 * batch * kCountChunks);
    const int nb = div_up(HW, kBlock);
    float *partial = ar.take<float>((size_t)batch * nb * 64);
    hipStream_t st = (hipStream_t)stream;
    if (!counts_ready) hipLaunchKernelGGL(k_loss_count, dim3(kCountChunks, batch), dim3(kBlock), 0, st, labels, n_anchor, cnt, importance);
    const dim3 grid(nb, batch);
    if (!grad) {
        if (dtype == SEC_BF16) launch_heads_loss<__hip_bfloat16, false>(bins, grid, st, heads, HW, head_channels, labels, reg_targets, anchors, importance, cnt, P, nullptr, nullptr, partial, terms);
        else launch_heads_loss<__half, false>(bins, grid, st, heads, HW, head_channels, labels, reg_targets, anchors, importance, cnt, P, nullptr, nullptr, partial, terms);
        hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(kBlock), 0, st, partial, batch * nb, P, out6);
    } else {
        if (dtype == SEC_BF16) launch_heads_loss<__hip_bfloat16, true>(bins, grid, st, heads, HW, head_channels, labels, reg_targets, anchors, importance, cnt, P, grad_loss, d_heads, partial);
        else launch_heads_loss<__half, true>(bins, grid, st, heads, HW, head_channels, labels, reg_targets, anchors, importance, cnt, P, grad_loss, d_heads, partial);
        const int cols = (a * (7 + nc + bins) + 7) / 8 * 8;
        hipLaunchKernelGGL(k_heads_bias_final, dim3(1), dim3(kBlock), 0, st, partial, batch * nb, cols, head_channels, d_bias);
    }
    return check_launch();
}
SEC_API int sec_heads_loss_fwd(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                               int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                               const float *h_params17, float *out6, void *workspace, size_t workspace_bytes, void *stream) {
    return heads_loss_impl(false, heads, dtype, batch, h, w, head_channels, anchors_per_loc, num_class, num_dir_bins, labels, reg_targets, anchors,
                           importance, h_params17, nullptr, nullptr, nullptr, out6, workspace, workspace_bytes, false, stream);
}
SEC_API int sec_heads_loss_fwd_terms(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                                     int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                                     const float *h_params17, float *out6, float *cls_preds_out, float *cls_loss_out, float *loc_loss_out,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    float *const terms[3] = {cls_preds_out, cls_loss_out, loc_loss_out};
    return heads_loss_impl(false, heads, dtype, batch, h, w, head_channels, anchors_per_loc, num_class, num_dir_bins, labels, reg_targets, anchors,
                           importance, h_params17, nullptr, nullptr, nullptr, out6, workspace, workspace_bytes, false, stream, terms);
}
SEC_API int sec_heads_loss_bwd(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                               int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                               const float *h_params17, const float *grad_loss, void *d_heads, float *d_bias, void *workspace,
                               size_t workspace_bytes, int counts_ready, void *stream) {
    return heads_loss_impl(true, heads, dtype, batch, h, w, head_channels, anchors_per_loc, num_class, num_dir_bins, labels, reg_targets, anchors,
                           importance, h_params17, grad_loss, d_heads, d_bias, nullptr, workspace, workspace_bytes, counts_ready != 0, stream);
}

SEC_API size_t sec_flat_adamw_workspace_bytes(void) { return align_up((size_t)kAdamBlocks * sizeof(float) + 256); }

static int flat_adamw_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float max_grad_norm, const float *hyper6, float *state4,
                           float *loss_scale4, void *workspace, size_t workspace_bytes, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state4 || n <= 0 || !workspace) return SEC_E_INVALID;
    if (workspace_bytes < sec_flat_adamw_workspace_bytes()) return SEC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    unsigned *ticket = (unsigned *)((char *)workspace + (size_t)kAdamBlocks * sizeof(float));      // zero before the first call
    int blocks = div_up(n, (long long)kBlock * 4);
    if (blocks > kAdamBlocks) blocks = kAdamBlocks;
    hipLaunchKernelGGL(k_flat_sumsq, dim3(blocks), dim3(kBlock), 0, st, grad, n, part, ticket, state4, loss_scale4);
    int ub = div_up(n, (long long)kBlock * 4);
    if (ub > 2048) ub = 2048;
    hipLaunchKernelGGL(k_flat_adamw, dim3(ub), dim3(kBlock), 0, st, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, max_grad_norm, (const float *)state4, hyper6);
    return check_launch();
}

SEC_API int sec_flat_adamw_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float max_grad_norm, float *state4, float *loss_scale4,
                               void *workspace, size_t workspace_bytes, void *stream) {
    return flat_adamw_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_grad_norm, nullptr, state4,
                           loss_scale4, workspace, workspace_bytes, stream);
}

SEC_API int sec_flat_adamw_dev_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const float *hyper6,
                                   float *state4, float *loss_scale4, void *workspace, size_t workspace_bytes, void *stream) {
    if (!hyper6) return SEC_E_INVALID;
    return flat_adamw_impl(param, grad, exp_avg, exp_avg_sq, n, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, hyper6, state4, loss_scale4, workspace,
                           workspace_bytes, stream);
}
