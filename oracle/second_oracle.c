/*
 * second_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Single-threaded plain-C restatement of the arithmetic on the SECOND hot path
 * (BASELINE.json north_star).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (second.pytorch_amd/) never does and fails loudly when
 * its HIP library is missing.
 *
 * PARITY STATUS
 *   - rotated IoU / bit-mask NMS / axis-aligned NMS / standup-IoU pre-filter:
 *     PINNED against golden vectors produced by executing the reference's own
 *     Python (second/core/non_max_suppression/nms_gpu.py, nms_cpu.py,
 *     second/core/box_np_ops.py) -- see tests/golden/make_golden.py.
 *   - voxel coordinates / voxel numbering / cap: PINNED against the in-repo
 *     restatement second/utils/simplevis.py:8-60 executed in pure Python.
 *   - rulebook + indice_conv: the arithmetic lives in traveller59/spconv v1.x
 *     (un-vendored, un-pinned dependency: README.md:104), absent from
 *     /root/reference.  "PARITY UNPINNED" at that boundary: the functions
 *     below restate spconv's published CPU algorithm (include/spconv/
 *     geometry.h getValidOutPos, indice.h getIndicePairsConv/SubM,
 *     spconv_ops.h indiceConv) from SURVEY.md Appendix A.3-A.5, and are
 *     anchored by an independent check: SubMConv3d / SparseConv3d == dense
 *     torch conv3d restricted to the active sites (tests/test_oracle_conv.py).
 *
 * All float math is IEEE fp32 (compile with -ffp-contract=off); integer
 * results are the bit-exact targets for the HIP kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* a1: VoxelGeneratorV2.__init__ grid size.                                   */
/* spconv/utils/__init__.py [recall]; same formula in-repo:                   */
/* second/utils/simplevis.py:26-29  grid = round((max-min)/voxel_size), fp32  */
/* np.round == round-half-to-even == rintf under the default rounding mode.   */
/* ------------------------------------------------------------------------ */
ORC_API void orc_grid_size(const float *range, const float *vsize, int *grid)
{
    for (int j = 0; j < 3; ++j) {
        float q = (range[3 + j] - range[j]) / vsize[j];
        grid[j] = (int)rintf(q);
    }
}

/* ------------------------------------------------------------------------ */
/* a2/a3: points_to_voxel (hard voxelisation).                               */
/* Follows SURVEY Appendix A.1 (spconv include/spconv/point2voxel.h          */
/* points_to_voxel_3d_np) and the in-repo copy second/utils/simplevis.py:     */
/* 31-50: per point, per axis j in x,y,z: c = floor((p[j]-min[j])/size[j]);   */
/* reject c<0 or c>=grid[j]; coor[2-j] = c (so coor = z,y,x); dense           */
/* coor->voxel-id lookup; new voxel id = running count; at the cap either     */
/* stop (`break`, simplevis.py:47-48) or skip the point (`continue`, spconv   */
/* C++ [recall]).  Up to max_points points per voxel are stored in arrival    */
/* order; voxels[] is zero initialised.                                       */
/*   cap_mode 0 = break (in-repo evidence, default), 1 = continue.            */
/* Returns voxel_num.  coors is [max_voxels,3] (z,y,x).                       */
/* ------------------------------------------------------------------------ */
ORC_API int orc_points_to_voxel(const float *points, int num_points, int num_features,
                                const float *vsize, const float *range,
                                int max_points, int max_voxels, int cap_mode,
                                float *voxels, int *coors, int *num_points_per_voxel)
{
    int grid[3];
    orc_grid_size(range, vsize, grid);
    const int64_t gx = grid[0], gy = grid[1], gz = grid[2];
    const int64_t vol = gx * gy * gz;
    int *lookup = (int *)malloc((size_t)vol * sizeof(int));
    if (!lookup) return -1;
    memset(lookup, 0xff, (size_t)vol * sizeof(int)); /* -1 */
    memset(voxels, 0, (size_t)max_voxels * max_points * num_features * sizeof(float));
    memset(num_points_per_voxel, 0, (size_t)max_voxels * sizeof(int));
    int voxel_num = 0;
    for (int i = 0; i < num_points; ++i) {
        const float *p = points + (size_t)i * num_features;
        int c3[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            float c = floorf((p[j] - range[j]) / vsize[j]);
            if (!(c >= 0.0f) || !(c < (float)grid[j])) { /* also drops NaN */
                failed = 1;
                break;
            }
            c3[2 - j] = (int)c;
        }
        if (failed) continue;
        const int64_t lin = ((int64_t)c3[0] * gy + c3[1]) * gx + c3[2];
        int vid = lookup[lin];
        if (vid == -1) {
            if (voxel_num >= max_voxels) {
                if (cap_mode == 0) break;
                continue;
            }
            vid = voxel_num++;
            lookup[lin] = vid;
            coors[vid * 3 + 0] = c3[0];
            coors[vid * 3 + 1] = c3[1];
            coors[vid * 3 + 2] = c3[2];
        }
        int n = num_points_per_voxel[vid];
        if (n < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + n) * num_features, p,
                   (size_t)num_features * sizeof(float));
            num_points_per_voxel[vid] = n + 1;
        }
    }
    free(lookup);
    return voxel_num;
}

/* a7: SimpleVoxel.forward (second/pytorch/models/voxel_encoder.py:220-225):  */
/* mean = sum over the max_points slots of the first nf features / num_points */
ORC_API void orc_simple_voxel_mean(const float *voxels, const int *num_points, int num_voxels,
                                   int max_points, int num_features, int nf, float *out)
{
    for (int v = 0; v < num_voxels; ++v)
        for (int f = 0; f < nf; ++f) {
            float s = 0.0f;
            for (int t = 0; t < max_points; ++t)
                s += voxels[((size_t)v * max_points + t) * num_features + f];
            out[(size_t)v * nf + f] = s / (float)num_points[v];
        }
}

/* ------------------------------------------------------------------------ */
/* Rulebook (a9, a10).  SURVEY Appendix A.4; spconv geometry.h/indice.h       */
/* [recall].  Index arithmetic is C int with truncating division, as there.   */
/* ------------------------------------------------------------------------ */

/* Enumerate the valid output positions reachable from one input position.    */
/* out: up to kvol entries of {z,y,x,offset}.  Order: last (x) dim fastest,   */
/* output coordinate DEscending from the upper bound.                         */
static int valid_out_pos(const int *in_pos, const int *ksize, const int *stride,
                         const int *pad, const int *dil, const int *out_shape, int *out)
{
    int lo[3], hi[3], cnt[3], ctr[3] = {0, 0, 0};
    int total = 1;
    for (int d = 0; d < 3; ++d) {
        lo[d] = (in_pos[d] - (ksize[d] - 1) * dil[d] - 1 + stride[d] + pad[d]) / stride[d];
        hi[d] = (in_pos[d] + pad[d]) / stride[d];
        cnt[d] = (hi[d] - lo[d]) / dil[d] + 1;
        total *= cnt[d];
    }
    int n = 0;
    for (int it = 0; it < total; ++it) {
        int valid = 1, m = 1, off = 0;
        for (int d = 2; d >= 0; --d) {
            int val = hi[d] - ctr[d] * dil[d];
            out[n * 4 + d] = val;
            if (val < 0 || val > out_shape[d] - 1) valid = 0;
            off += m * ((in_pos[d] - val * stride[d] + pad[d]) / dil[d]);
            m *= ksize[d];
        }
        out[n * 4 + 3] = off;
        if (valid) ++n;
        ctr[2] += 1;
        for (int c = 2; c > 0; --c)
            if (ctr[c] == cnt[c]) { ctr[c - 1] += 1; ctr[c] = 0; }
    }
    return n;
}

ORC_API void orc_conv_output_size(const int *in_shape, const int *ksize, const int *stride,
                                  const int *pad, const int *dil, int *out_shape)
{
    /* spconv/ops.py get_conv_output_size [recall], SURVEY A.3 */
    for (int d = 0; d < 3; ++d)
        out_shape[d] = (in_shape[d] + 2 * pad[d] - dil[d] * (ksize[d] - 1) - 1) / stride[d] + 1;
}

/* SubM rulebook.  indices [N,4] = (b,z,y,x).  pairs [K,2,N] (-1 padded),     */
/* pair_num [K].  Output sites == input sites.                                */
ORC_API int orc_rulebook_subm(const int *indices, int N, int batch_size, const int *shape,
                              const int *ksize, const int *dil, int *pairs, int *pair_num)
{
    const int K = ksize[0] * ksize[1] * ksize[2];
    const int64_t vol = (int64_t)shape[0] * shape[1] * shape[2];
    int *grid = (int *)malloc((size_t)(vol * batch_size) * sizeof(int));
    if (!grid) return -1;
    memset(grid, 0xff, (size_t)(vol * batch_size) * sizeof(int));
    memset(pairs, 0xff, (size_t)K * 2 * N * sizeof(int));
    memset(pair_num, 0, (size_t)K * sizeof(int));
    int stride[3] = {1, 1, 1}, pad[3];
    for (int d = 0; d < 3; ++d) pad[d] = (ksize[d] / 2) * dil[d]; /* forced for subm */
    for (int j = 0; j < N; ++j) {
        const int *c = indices + (size_t)j * 4;
        grid[((int64_t)c[1] * shape[1] + c[2]) * shape[2] + c[3] + vol * c[0]] = j;
    }
    int *cand = (int *)malloc((size_t)K * 4 * sizeof(int));
    for (int j = 0; j < N; ++j) {
        const int *c = indices + (size_t)j * 4;
        int n = valid_out_pos(c + 1, ksize, stride, pad, dil, shape, cand);
        for (int i = 0; i < n; ++i) {
            const int *p = cand + i * 4;
            int64_t lin = ((int64_t)p[0] * shape[1] + p[1]) * shape[2] + p[2] + vol * c[0];
            int o = grid[lin];
            if (o > -1) {
                int k = p[3];
                int cnum = pair_num[k]++;
                pairs[((size_t)k * 2 + 0) * N + cnum] = j;
                pairs[((size_t)k * 2 + 1) * N + cnum] = o;
            }
        }
    }
    free(cand);
    free(grid);
    return N;
}

/* Strided / regular sparse conv rulebook.  First-touch output numbering.     */
/* out_indices must hold N*K rows of 4.  Returns number of active outputs.    */
ORC_API int orc_rulebook_conv(const int *indices, int N, int batch_size, const int *in_shape,
                              const int *out_shape, const int *ksize, const int *stride,
                              const int *pad, const int *dil, int *out_indices, int *pairs,
                              int *pair_num)
{
    (void)in_shape;
    const int K = ksize[0] * ksize[1] * ksize[2];
    const int64_t vol = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
    int *grid = (int *)malloc((size_t)(vol * batch_size) * sizeof(int));
    if (!grid) return -1;
    memset(grid, 0xff, (size_t)(vol * batch_size) * sizeof(int));
    memset(pairs, 0xff, (size_t)K * 2 * N * sizeof(int));
    memset(pair_num, 0, (size_t)K * sizeof(int));
    int *cand = (int *)malloc((size_t)K * 4 * sizeof(int));
    int num_act = 0;
    for (int j = 0; j < N; ++j) {
        const int *c = indices + (size_t)j * 4;
        int n = valid_out_pos(c + 1, ksize, stride, pad, dil, out_shape, cand);
        for (int i = 0; i < n; ++i) {
            const int *p = cand + i * 4;
            int64_t lin = ((int64_t)p[0] * out_shape[1] + p[1]) * out_shape[2] + p[2] + vol * c[0];
            if (grid[lin] == -1) {
                out_indices[num_act * 4 + 0] = c[0];
                out_indices[num_act * 4 + 1] = p[0];
                out_indices[num_act * 4 + 2] = p[1];
                out_indices[num_act * 4 + 3] = p[2];
                grid[lin] = num_act++;
            }
            int k = p[3];
            int cnum = pair_num[k]++;
            pairs[((size_t)k * 2 + 0) * N + cnum] = j;
            pairs[((size_t)k * 2 + 1) * N + cnum] = grid[lin];
        }
    }
    free(cand);
    free(grid);
    return num_act;
}

/* ------------------------------------------------------------------------ */
/* a11: indice_conv forward (spconv_ops.h indiceConv [recall]):   */
/* out = 0; for each offset k, for each pair n: out[pairs[k,1,n]] +=          */
/* feat[pairs[k,0,n]] @ W[k].  W is [K,Cin,Cout] (= [kD,kH,kW,Cin,Cout]).     */
/* acc64 != 0 accumulates each output element in double (tolerance oracle);   */
/* acc64 == 0 keeps fp32 offset-major summation like the reference.           */
/* ------------------------------------------------------------------------ */
ORC_API void orc_indice_conv_fwd(const float *feat, int N_in, int Cin, const float *W, int K,
                                 int Cout, const int *pairs, const int *pair_num, int N_out,
                                 int acc64, float *out)
{
    if (acc64) {
        double *acc = (double *)calloc((size_t)N_out * Cout, sizeof(double));
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < pair_num[k]; ++n) {
                int i = pairs[((size_t)k * 2 + 0) * N_in + n];
                int o = pairs[((size_t)k * 2 + 1) * N_in + n];
                const float *f = feat + (size_t)i * Cin;
                const float *w = W + (size_t)k * Cin * Cout;
                double *a = acc + (size_t)o * Cout;
                for (int ci = 0; ci < Cin; ++ci) {
                    double fv = f[ci];
                    for (int co = 0; co < Cout; ++co) a[co] += fv * (double)w[(size_t)ci * Cout + co];
                }
            }
        for (size_t t = 0; t < (size_t)N_out * Cout; ++t) out[t] = (float)acc[t];
        free(acc);
        return;
    }
    memset(out, 0, (size_t)N_out * Cout * sizeof(float));
    float *tmp = (float *)malloc((size_t)Cout * sizeof(float));
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < pair_num[k]; ++n) {
            int i = pairs[((size_t)k * 2 + 0) * N_in + n];
            int o = pairs[((size_t)k * 2 + 1) * N_in + n];
            const float *f = feat + (size_t)i * Cin;
            const float *w = W + (size_t)k * Cin * Cout;
            for (int co = 0; co < Cout; ++co) tmp[co] = 0.0f;
            for (int ci = 0; ci < Cin; ++ci) {
                float fv = f[ci];
                for (int co = 0; co < Cout; ++co) tmp[co] += fv * w[(size_t)ci * Cout + co];
            }
            float *a = out + (size_t)o * Cout;
            for (int co = 0; co < Cout; ++co) a[co] += tmp[co];
        }
    free(tmp);
}

/* a12: indice_conv backward (spconv_ops.h indiceConvBackward [recall]):      */
/* dW[k] = gather(feat)^T . gather(dOut);  dFeat[i] += dOut[o] . W[k]^T       */
ORC_API void orc_indice_conv_bwd(const float *feat, int N_in, int Cin, const float *W, int K,
                                 int Cout, const int *pairs, const int *pair_num,
                                 const float *dout, float *dfeat, float *dW)
{
    double *af = (double *)calloc((size_t)N_in * Cin, sizeof(double));
    double *aw = (double *)calloc((size_t)K * Cin * Cout, sizeof(double));
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < pair_num[k]; ++n) {
            int i = pairs[((size_t)k * 2 + 0) * N_in + n];
            int o = pairs[((size_t)k * 2 + 1) * N_in + n];
            const float *f = feat + (size_t)i * Cin;
            const float *g = dout + (size_t)o * Cout;
            const float *w = W + (size_t)k * Cin * Cout;
            double *wk = aw + (size_t)k * Cin * Cout;
            double *fi = af + (size_t)i * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                double s = 0.0;
                for (int co = 0; co < Cout; ++co) {
                    s += (double)g[co] * (double)w[(size_t)ci * Cout + co];
                    wk[(size_t)ci * Cout + co] += (double)f[ci] * (double)g[co];
                }
                fi[ci] += s;
            }
        }
    for (size_t t = 0; t < (size_t)N_in * Cin; ++t) dfeat[t] = (float)af[t];
    for (size_t t = 0; t < (size_t)K * Cin * Cout; ++t) dW[t] = (float)aw[t];
    free(af);
    free(aw);
}

/* a8: SparseConvTensor.dense() -> [B,C,D,H,W] contiguous                      */
/* (second/pytorch/models/middle.py:206-210 expects N,C,D,H,W).               */
ORC_API void orc_sparse_to_dense(const float *feat, const int *indices, int N, int C,
                                 int batch_size, const int *shape, float *out)
{
    const size_t vol = (size_t)shape[0] * shape[1] * shape[2];
    memset(out, 0, (size_t)batch_size * C * vol * sizeof(float));
    for (int i = 0; i < N; ++i) {
        const int *c = indices + (size_t)i * 4;
        size_t sp = ((size_t)c[1] * shape[1] + c[2]) * shape[2] + c[3];
        for (int ch = 0; ch < C; ++ch)
            out[((size_t)c[0] * C + ch) * vol + sp] = feat[(size_t)i * C + ch];
    }
}

/* a21: PointPillarsScatter (second/pytorch/models/pointpillars.py:444-476):   */
/* canvas[b, :, y*nx + x] = feat[i, :]   (later rows overwrite earlier ones)  */
ORC_API void orc_pillar_scatter(const float *feat, const int *coords, int P, int C,
                                int batch_size, int ny, int nx, float *out)
{
    memset(out, 0, (size_t)batch_size * C * ny * nx * sizeof(float));
    for (int i = 0; i < P; ++i) {
        const int *c = coords + (size_t)i * 4;
        size_t sp = (size_t)c[2] * nx + c[3];
        for (int ch = 0; ch < C; ++ch)
            out[((size_t)c[0] * C + ch) * ny * nx + sp] = feat[(size_t)i * C + ch];
    }
}

/* ------------------------------------------------------------------------ */
/* Rotated IoU (a17/a18).  fp32 restatement of the numba.cuda device          */
/* functions in second/core/non_max_suppression/nms_gpu.py:166-401.           */
/* ------------------------------------------------------------------------ */
static float tri_area(const float *a, const float *b, const float *c)
{ /* nms_gpu.py:166-169 */
    return ((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f;
}

static float poly_area(const float *pts, int n)
{ /* nms_gpu.py:172-179: fan triangulation from vertex 0 */
    float s = 0.0f;
    for (int i = 0; i < n - 2; ++i) s += fabsf(tri_area(pts, pts + 2 * i + 2, pts + 2 * i + 4));
    return s;
}

static void sort_vertices(float *pts, int n)
{ /* nms_gpu.py:182-219: angular key, insertion sort */
    if (n <= 0) return;
    float cx = 0.0f, cy = 0.0f;
    for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
    cx /= (float)n;
    cy /= (float)n;
    float vs[16];
    for (int i = 0; i < n; ++i) {
        float vx = pts[2 * i] - cx, vy = pts[2 * i + 1] - cy;
        float d = sqrtf(vx * vx + vy * vy);
        vx = vx / d;
        vy = vy / d;
        if (vy < 0) vx = -2 - vx;
        vs[i] = vx;
    }
    for (int i = 1; i < n; ++i) {
        if (vs[i - 1] > vs[i]) {
            float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
            int j = i;
            while (j > 0 && vs[j - 1] > temp) {
                vs[j] = vs[j - 1];
                pts[2 * j] = pts[2 * j - 2];
                pts[2 * j + 1] = pts[2 * j - 1];
                --j;
            }
            vs[j] = temp;
            pts[2 * j] = tx;
            pts[2 * j + 1] = ty;
        }
    }
}

static int seg_intersect(const float *p1, const float *p2, int i, int j, float *t)
{ /* nms_gpu.py:222-264 */
    float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    float B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    float D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    int acd = DA1 * CA0 > CA1 * DA0;
    int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        int abc = CA1 * BA0 > BA1 * CA0;
        int abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            float DC0 = D0 - C0, DC1 = D1 - C1;
            float ABBA = A0 * B1 - B0 * A1;
            float CDDC = C0 * D1 - D0 * C1;
            float DH = BA1 * DC0 - BA0 * DC1;
            float Dx = ABBA * DC0 - BA0 * CDDC;
            float Dy = ABBA * DC1 - BA1 * CDDC;
            t[0] = Dx / DH;
            t[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}

static int pt_in_quad(float x, float y, const float *c)
{ /* nms_gpu.py:308-325 */
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

static void box_corners(float *c, const float *b)
{ /* nms_gpu.py:353-376: clockwise corners, rotated clockwise by angle */
    float ac = cosf(b[4]), as = sinf(b[4]);
    float cx = b[0], cy = b[1], xd = b[2], yd = b[3];
    float xs[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    float ys[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = ac * xs[i] + as * ys[i] + cx;
        c[2 * i + 1] = -as * xs[i] + ac * ys[i] + cy;
    }
}

static float rot_inter(const float *b1, const float *b2)
{ /* nms_gpu.py:329-350,379-393.  The reference's buffer is 16 floats (8      */
  /* points); a convex quad/quad intersection has at most 8 vertices but      */
  /* degenerate inputs may report more candidates -- we hold 24 and clamp     */
  /* the count used to 8 so the restatement stays memory-safe.                */
    float c1[8], c2[8], pts[48];
    box_corners(c1, b1);
    box_corners(c2, b2);
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        if (pt_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
        if (pt_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
    }
    float t[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersect(c1, c2, i, j, t)) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; ++n; }
    if (n > 8) n = 8;
    sort_vertices(pts, n);
    return poly_area(pts, n);
}

/* criterion: -1 IoU, 0 inter/area1, 1 inter/area2, 2 inter (nms_gpu.py:549-561) */
ORC_API float orc_rotate_iou_pair(const float *b1, const float *b2, int criterion)
{
    float a1 = b1[2] * b1[3], a2 = b2[2] * b2[3];
    float in = rot_inter(b1, b2);
    if (criterion == -1) return in / (a1 + a2 - in);
    if (criterion == 0) return in / a1;
    if (criterion == 1) return in / a2;
    return in;
}

/* rotate_iou_gpu_eval (nms_gpu.py:564-640): iou[n,k] = f(query k as rbox1, box n as rbox2) */
ORC_API void orc_rotate_iou(const float *boxes, int N, const float *qboxes, int K, int criterion,
                            float *iou)
{
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            iou[(size_t)n * K + k] = orc_rotate_iou_pair(qboxes + 5 * k, boxes + 5 * n, criterion);
}

/* Greedy rotated NMS on boxes already sorted by descending score.            */
/*   dets [N,stride] with (x,y,w,l,r,...) in the first 5 columns.             */
/*   semantics 0: numba.cuda spec -- rotate_nms_kernel + nms_postprocess      */
/*                (nms_gpu.py:404-437,109-126): suppress j>i if IoU > thr.    */
/*   semantics 1: CPU path used by predict -- rotate_nms_cc (nms_cpu.py:17-28)*/
/*                + spconv rotate_non_max_suppression_cpu (SURVEY A.6         */
/*                [recall]): skip pairs whose standup IoU <= 0, suppress if   */
/*                IoU >= thr.                                                  */
/* keep receives indices into the sorted order; returns the count.            */
static void standup_box(const float *b, float *s)
{ /* box_np_ops.py:405-425 (center_to_corner_box2d) + :278-283 (standup) */
    float rs = sinf(b[4]), rc = cosf(b[4]);
    const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f};
    const float ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    float minx = 0, miny = 0, maxx = 0, maxy = 0;
    for (int i = 0; i < 4; ++i) {
        float px = b[2] * nx[i], py = b[3] * ny[i];
        /* rotation_2d, box_np_ops.py:344-357: x' = x cos + y sin ; y' = -x sin + y cos */
        float x = px * rc + py * rs + b[0];
        float y = -px * rs + py * rc + b[1];
        if (i == 0) { minx = maxx = x; miny = maxy = y; }
        else {
            if (x < minx) minx = x;
            if (x > maxx) maxx = x;
            if (y < miny) miny = y;
            if (y > maxy) maxy = y;
        }
    }
    s[0] = minx; s[1] = miny; s[2] = maxx; s[3] = maxy;
}

static float aa_iou(const float *a, const float *b, float eps)
{ /* box_np_ops.py:696-725 iou_jit */
    float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + eps;
    if (iw > 0) {
        float ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + eps;
        if (ih > 0) {
            float ua = (a[2] - a[0] + eps) * (a[3] - a[1] + eps) +
                       (b[2] - b[0] + eps) * (b[3] - b[1] + eps) - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0f;
}

ORC_API void orc_standup_boxes(const float *dets, int N, int stride, float *out)
{
    for (int i = 0; i < N; ++i) standup_box(dets + (size_t)i * stride, out + 4 * i);
}

ORC_API void orc_standup_iou(const float *sb, int N, float eps, float *iou)
{
    for (int k = 0; k < N; ++k)
        for (int n = 0; n < N; ++n) iou[(size_t)n * N + k] = aa_iou(sb + 4 * n, sb + 4 * k, eps);
}

ORC_API int orc_rotate_nms_sorted(const float *dets, int N, int stride, float thresh,
                                  int semantics, int *keep)
{
    unsigned char *sup = (unsigned char *)calloc((size_t)N + 1, 1);
    float *sb = NULL;
    if (semantics == 1) {
        sb = (float *)malloc((size_t)N * 4 * sizeof(float) + 16);
        orc_standup_boxes(dets, N, stride, sb);
    }
    int nk = 0;
    for (int i = 0; i < N; ++i) {
        if (sup[i]) continue;
        keep[nk++] = i;
        const float *bi = dets + (size_t)i * stride;
        for (int j = i + 1; j < N; ++j) {
            if (sup[j]) continue;
            const float *bj = dets + (size_t)j * stride;
            if (semantics == 1) {
                if (!(aa_iou(sb + 4 * i, sb + 4 * j, 0.0f) > 0.0f)) continue;
                if (orc_rotate_iou_pair(bi, bj, -1) >= thresh) sup[j] = 1;
            } else {
                if (orc_rotate_iou_pair(bi, bj, -1) > thresh) sup[j] = 1;
            }
        }
    }
    free(sup);
    free(sb);
    return nk;
}

/* a19: axis-aligned NMS on sorted boxes (x1,y1,x2,y2,...).                   */
/*   semantics 0: nms_kernel + nms_postprocess (nms_gpu.py:21-32,70-126):     */
/*                "+1" pixel convention, suppress if IoU > thr.               */
/*   semantics 1: nms_jit (nms_cpu.py:30-60): eps-convention, IoU >= thr.     */
ORC_API int orc_nms_sorted(const float *dets, int N, int stride, float thresh, int semantics,
                           float eps, int *keep)
{
    unsigned char *sup = (unsigned char *)calloc((size_t)N + 1, 1);
    int nk = 0;
    for (int i = 0; i < N; ++i) {
        if (sup[i]) continue;
        keep[nk++] = i;
        const float *a = dets + (size_t)i * stride;
        for (int j = i + 1; j < N; ++j) {
            if (sup[j]) continue;
            const float *b = dets + (size_t)j * stride;
            if (semantics == 0) {
                float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.0f, 0.0f);
                float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.0f, 0.0f);
                float in = w * h;
                float sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
                float sbb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
                if (in / (sa + sbb - in) > thresh) sup[j] = 1;
            } else {
                float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + eps, 0.0f);
                float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + eps, 0.0f);
                float in = w * h;
                float sa = (a[2] - a[0] + eps) * (a[3] - a[1] + eps);
                float sbb = (b[2] - b[0] + eps) * (b[3] - b[1] + eps);
                if (in / (sa + sbb - in) >= thresh) sup[j] = 1;
            }
        }
    }
    free(sup);
    return nk;
}

/* a15: second_box_decode (second/pytorch/core/box_torch_ops.py:56-101),      */
/* plain variant (no angle vector, no smooth_dim); boxes/anchors are [N,7].   */
ORC_API void orc_box_decode(const float *enc, const float *anchors, int N, float *out)
{
    for (int i = 0; i < N; ++i) {
        const float *t = enc + 7 * i, *a = anchors + 7 * i;
        float *g = out + 7 * i;
        float diag = sqrtf(a[4] * a[4] + a[3] * a[3]);
        g[0] = t[0] * diag + a[0];
        g[1] = t[1] * diag + a[1];
        g[2] = t[2] * a[5] + a[2];
        g[3] = expf(t[3]) * a[3];
        g[4] = expf(t[4]) * a[4];
        g[5] = expf(t[5]) * a[5];
        g[6] = t[6] + a[6];
    }
}

/* ------------------------------------------------------------------------ */
/* a20: PillarFeatureNet.forward with ONE PFNLayer (last_layer=True) in eval  */
/* mode (second/pytorch/models/pointpillars.py:203-237 + PFNLayer :51-65):    */
/* decorate each point with (xyz - mean of the pillar's points) and           */
/* (xy - pillar centre), zero the padded slots, Linear(F+5 -> C, no bias),    */
/* BatchNorm1d folded to scale/shift, ReLU, max over ALL T slots (padded      */
/* slots contribute relu(shift), as in the reference).                        */
/* voxels [P,T,F], coords [P,4] (b,z,y,x), W [F+5, C] (= linear.weight^T).    */
/* ------------------------------------------------------------------------ */
ORC_API void orc_pfn_fwd(const float *voxels, const int *num_points, const int *coords, int P, int T,
                         int F, const float *W, const float *scale, const float *shift, int C,
                         float vx, float vy, float x_offset, float y_offset, float *out)
{
    const int IN = F + 5;
    float *feat = (float *)malloc((size_t)IN * sizeof(float));
    for (int p = 0; p < P; ++p) {
        const float *v = voxels + (size_t)p * T * F;
        int n = num_points[p];
        float mean[3];
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int t = 0; t < T; ++t) s += v[(size_t)t * F + j];
            mean[j] = s / (float)n;
        }
        float cx = (float)coords[p * 4 + 3] * vx + x_offset;
        float cy = (float)coords[p * 4 + 2] * vy + y_offset;
        float *o = out + (size_t)p * C;
        for (int c = 0; c < C; ++c) o[c] = -INFINITY;
        for (int t = 0; t < T; ++t) {
            const float *pt = v + (size_t)t * F;
            if (t < n) {
                for (int j = 0; j < F; ++j) feat[j] = pt[j];
                for (int j = 0; j < 3; ++j) feat[F + j] = pt[j] - mean[j];
                feat[F + 3] = pt[0] - cx;
                feat[F + 4] = pt[1] - cy;
            } else {
                for (int j = 0; j < IN; ++j) feat[j] = 0.0f;
            }
            for (int c = 0; c < C; ++c) {
                float y = 0.0f;
                for (int j = 0; j < IN; ++j) y += feat[j] * W[(size_t)j * C + c];
                y = y * scale[c] + shift[c];
                if (y < 0.0f) y = 0.0f;
                if (y > o[c]) o[c] = y;
            }
        }
    }
    free(feat);
}

/* ------------------------------------------------------------------------ */
/* a4: block filtering of points_to_voxel_3d_with_filtering (SURVEY A.2,      */
/* spconv point2voxel.h [recall, constants UNVERIFIED]; enabled by            */
/* second/configs/nuscenes/all.fhd.config:9-12).  Given the voxelisation      */
/* result, build per-(y,x)-block min/max of the z of the STORED points, then  */
/* keep voxel v iff the height span over the block_size x block_size window   */
/* around its block lies in (height_threshold, height_high_threshold).        */
/* Returns the number of kept voxels; keep[v] in {0,1}.                       */
/* ------------------------------------------------------------------------ */
ORC_API int orc_block_filter(const float *voxels, const int *coors /*[V,3] z,y,x*/, const int *num_points,
                             int V, int T, int F, int grid_x, int grid_y, int block_factor, int block_size,
                             float height_threshold, float height_high_threshold, unsigned char *keep)
{
    const int bx = (grid_x + block_factor - 1) / block_factor, by = (grid_y + block_factor - 1) / block_factor;
    float *mins = (float *)malloc((size_t)bx * by * sizeof(float));
    float *maxs = (float *)malloc((size_t)bx * by * sizeof(float));
    for (int i = 0; i < bx * by; ++i) { mins[i] = 99999999.0f; maxs[i] = -99999999.0f; }
    for (int v = 0; v < V; ++v) {
        int cy = coors[v * 3 + 1] / block_factor, cx = coors[v * 3 + 2] / block_factor;
        for (int t = 0; t < num_points[v] && t < T; ++t) {
            float z = voxels[((size_t)v * T + t) * F + 2];
            if (z < mins[cy * bx + cx]) mins[cy * bx + cx] = z;
            if (z > maxs[cy * bx + cx]) maxs[cy * bx + cx] = z;
        }
    }
    int kept = 0;
    for (int v = 0; v < V; ++v) {
        int cy = coors[v * 3 + 1] / block_factor, cx = coors[v * 3 + 2] / block_factor;
        int y0 = cy - block_size / 2, y1 = cy + block_size - block_size / 2;
        int x0 = cx - block_size / 2, x1 = cx + block_size - block_size / 2;
        if (y0 < 0) y0 = 0;
        if (x0 < 0) x0 = 0;
        if (y1 > by) y1 = by;
        if (x1 > bx) x1 = bx;
        float hmin = 99999999.0f, hmax = -99999999.0f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                if (mins[y * bx + x] < hmin) hmin = mins[y * bx + x];
                if (maxs[y * bx + x] > hmax) hmax = maxs[y * bx + x];
            }
        float span = hmax - hmin;
        keep[v] = (span > height_threshold && span < height_high_threshold) ? 1 : 0;
        kept += keep[v];
    }
    free(mins);
    free(maxs);
    return kept;
}
