/*
 * second_hip.h -- C ABI of libsecond_hip.so: the MI355X (gfx950) implementation of the SECOND hot path.
 *
 * This is the drop-in boundary of the project.  The reference reaches this
 * arithmetic through the `spconv` Python package (pybind11 / torch custom ops of traveller59/spconv
 * v1.x, absent from /root/reference) and through numba.cuda kernels; each entry point below names the
 * reference interface it replaces.  Host code binds it with ctypes (second.pytorch_amd/second_amd/
 * runtime.py); INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with h_ (small host arrays read at call time);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, no function
 *     synchronises the host, allocates or frees memory;
 *   - scratch memory is caller-provided: query `*_workspace_bytes`, pass a buffer at least that large;
 *   - return value: 0 = ok, negative = error (SEC_E_*); never throws;
 *   - dtype codes: SEC_F32 / SEC_F16 / SEC_BF16; index tensors are int32; coordinates are (b, z, y, x).
 */
#ifndef SECOND_HIP_H
#define SECOND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEC_OK 0
#define SEC_E_INVALID -1   /* bad argument */
#define SEC_E_WORKSPACE -2 /* workspace too small */
#define SEC_E_UNSUPPORTED -3
#define SEC_E_LAUNCH -4    /* hipGetLastError() != hipSuccess after a launch */

#define SEC_F32 0
#define SEC_F16 1
#define SEC_BF16 2

#define SEC_ABI_VERSION 9
int sec_abi_version(void);
/* Content checksum of `count` device tensors in one launch (+ one memset): sums [count][2] = (sum of the tensor's 32-bit words, sum of
 * word * (index + 1)), both mod 2^64.  h_ptrs / h_nbytes are HOST arrays (4-byte aligned pointers, byte counts that are multiples of
 * 4: SEC_E_UNSUPPORTED otherwise); the pointers become kernel arguments, so a captured launch keeps reading the same tensors.  What
 * compat.accelerate_model compares every call against the values at adoption: the reference's own optimizers update weights through
 * `.data` (torchplus/train/fastai_optim.py), which no version counter sees. */
int sec_tensors_checksum(const void *const *h_ptrs, const long long *h_nbytes, int count, unsigned long long *sums, void *stream);
/* last HIP error string seen by this library (thread-unsafe convenience for diagnostics) */
const char *sec_last_error(void);
/* measurement aid: the kernel instantiation (template arguments as rocprofv3 prints them, e.g.
 * "k_conv_rows_buf<__hip_bfloat16, 64, 64, 27, 3, 8, 3, 641>") that the last sec_indice_conv_fwd / sec_conv2d_nhwc call of the
 * calling thread dispatched to; "" if that call took a kernel that does not report itself.  bench.py keys the committed PMC
 * passes (profiles/rNN_traffic.json) on this string, so a counter file can never be attributed to a superseded kernel form. */
const char *sec_last_kernel_name(void);

/* ---------------------------------------------------------------------------------------------
 * points_to_voxel  -- replaces spconv.utils.VoxelGeneratorV2.generate / generate_multi_gpu
 *   (called at second/data/preprocess.py:301-316; built at second/builder/voxel_builder.py:23-32)
 *   and, batched, the coordinate padding of merge_second_batch (second/data/preprocess.py:44-50).
 *
 * `batch` clouds are concatenated in `points` [num_points, num_features]; cloud b owns rows
 * [point_offsets[b], point_offsets[b+1]).  Semantics per cloud = the sequential reference loop
 * (second/utils/simplevis.py:31-50): voxel numbering in first-occurrence order, the first
 * `max_points` points of a voxel kept in arrival order, at most `max_voxels` voxels
 * (cap_mode 0 = `break` at the cap, 1 = `continue`).  max_points <= 256 (the reference's configs keep 1 ... 100 points per voxel);
 * larger values: SEC_E_UNSUPPORTED before anything is enqueued, and sec_voxelize_workspace_bytes returns 0.
 * Outputs are compact: cloud b's voxels are rows [voxel_offsets[b], voxel_offsets[b+1]).
 *   voxels [batch*max_voxels, max_points, num_features] (zero padded), coors [.,4] = (b,z,y,x),
 *   num_points_per_voxel [.], voxel_offsets [batch+1];
 *   voxels may be NULL (then mean must be NULL too): the tensor is not written, the per-voxel point lists stay in the workspace
 *   for a consumer that walks them itself (sec_pfn_fwd_slots);
 *   mean (optional, may be NULL) [., mean_features] = SimpleVoxel.forward
 *   (second/pytorch/models/voxel_encoder.py:220-225) fused as an epilogue, stored as mean_dtype (SEC_F32, or the 16-bit
 *   dtype of the sparse stack that consumes it: the reference's `.to(dtype)` of example_convert_to_torch, train.py:36-38).
 * --------------------------------------------------------------------------------------------- */
size_t sec_voxelize_workspace_bytes(int num_points, int batch, int max_voxels, int max_points);
/* SimpleVoxel.forward alone (voxel_encoder.py:220-225), for callers that already HOLD the voxel tensor -- the example dict of
 * VoxelNet.forward (voxelnet.py:339-375: voxels [n, max_points, num_features] fp32, num_points [n]): mean [n, mean_features] =
 * sum over the point slots / num_points, stored as mean_dtype; same operation order as the fused epilogue of sec_voxelize_f32 (bit
 * identical).  num_dev (optional device int): rows at or past it are written as zeros.
 * sec_rows_differ_f32: flag[0] |= 1 if any of the `rows` rows of a [rows][n] differs (bit compare) from b [n]; the caller zeroes
 * `flag`.  Used to compare the anchors an example carries (voxelnet.py:358, [B, A * 7]) with the table a captured pipeline decodes
 * with, in one pass. */
int sec_simple_voxel_f32(const float *voxels, const int *num_points, int n, const int *num_dev, int max_points, int num_features,
                         int mean_features, void *mean, int mean_dtype, void *stream);
int sec_rows_differ_f32(const float *a, long long rows, const float *b, long long n, int *flag, void *stream);
int sec_voxelize_f32(const float *points, const int *point_offsets, int num_points, int num_features,
                     int batch, const float *h_range6, const float *h_voxel_size3, int max_points,
                     int max_voxels, int cap_mode, float *voxels, int *coors,
                     int *num_points_per_voxel, int *voxel_offsets, void *mean, int mean_features,
                     int mean_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rulebooks -- replace spconv.ops.get_indice_pairs (torch.ops.spconv.get_indice_pairs; CPU oracle
 * semantics of spconv include/spconv/indice.h).
 *
 * Besides spconv's pair lists (`pairs` [K,2,n_in] -1 padded, `pair_num` [K]; canonical order =
 * ascending input row within an offset) the builders emit the OUTPUT-MAJOR GATHER TABLE
 * `nbr_out` [n_out, K]: nbr_out[o][k] = input row feeding output o through kernel offset k, or -1,
 * which is what sec_indice_conv_fwd consumes, and the INPUT-MAJOR table `nbr_in` [n_in, K]
 * (nbr_in[j][k] = output row or -1) which the backward pass consumes.
 * --------------------------------------------------------------------------------------------- */
/* max_out_per_in sizes the output hash table (2 * n_in * max_out_per_in slots).  The exact bound is the number
 * of kernel offsets that can reach an output from one input: prod over dims of the largest residue class of
 * {k*dil mod stride} (= prod(ceil(k/s)) for dil 1; 8 for k3 s2); the same value must be passed as `out_per_in_hint` (0 = exact bound) to
 * both conv3d calls.  A smaller hint saves memset traffic; if the data then needs more slots the build
 * reports num_out[1] = INT_MAX (overflow) instead of hanging. */
size_t sec_rulebook_workspace_bytes(int n_in, int kvol, int max_out_per_in);

/* Static-capacity mode: every builder takes an optional device int `n_in_dev`; buffers are sized for
 * `n_in` rows, only the first *n_in_dev are live.  With the device-side counts of the voxeliser
 * (voxel_offsets[batch]) and of sec_rulebook_conv3d_build (num_out) a whole forward pass needs no host
 * synchronisation and can be captured in a hipGraph. */

/* SubMConv3d: outputs == inputs.  pairs/pair_num may be NULL (and must be in static-capacity mode). */
int sec_rulebook_subm3d(const int *indices, int n_in, const int *n_in_dev, int batch,
                        const int *h_shape3, const int *h_ksize3, const int *h_dilation3, int *nbr_out,
                        int *pairs, int *pair_num, void *workspace, size_t workspace_bytes,
                        void *stream);

/* SubMConv3d on the OUTPUT sites of a strided conv whose rulebook was just built: the strided build's hash table
 * (output cell -> output row) is still in `conv_workspace`, so this layer skips re-hashing its sites (the
 * subm1/subm2/subm3 groups of SpMiddleFHD each follow a SparseConv3d, middle.py:152-188).  `indices` must be the
 * out_indices of that build, the conv_* arguments the ones given to sec_rulebook_conv3d_build / _tables, and
 * the workspace must not have been reused in between.  Same nbr_out as sec_rulebook_subm3d. */
int sec_rulebook_subm3d_after_conv(const int *indices, int n_in, const int *n_in_dev, int batch,
                                   const int *h_shape3, const int *h_ksize3, const int *h_dilation3,
                                   int *nbr_out, const void *conv_workspace, size_t conv_workspace_bytes,
                                   int conv_n_in, const int *h_conv_ksize3, const int *h_conv_stride3,
                                   const int *h_conv_dilation3, int conv_out_per_in_hint, void *stream);

/* SubMConv3d on the voxels of a sec_voxelize_f32 call whose workspace is still intact (the first layer of SpMiddleFHD,
 * middle.py:146): the voxeliser's hash table (cell -> voxel row) is this layer's site lookup, no re-hash.  `indices` must be
 * that call's coors (all rows, unfiltered), vox_* its num_points / max_voxels / max_points arguments, h_vox_grid3_zyx its
 * grid in (z, y, x) order (<= h_shape3 per dim).  Same nbr_out as sec_rulebook_subm3d. */
int sec_rulebook_subm3d_after_voxelize(const int *indices, int n_in, const int *n_in_dev, int batch,
                                       const int *h_shape3, const int *h_ksize3, const int *h_dilation3,
                                       int *nbr_out, const void *vox_workspace, size_t vox_workspace_bytes,
                                       int vox_num_points, int vox_max_voxels, int vox_max_points,
                                       const int *h_vox_grid3_zyx, void *stream);

/* SparseConv3d, step 1: discover the active outputs in first-touch order (oracle numbering).
 *   A dim with both stride > 1 and dilation > 1 returns SEC_E_UNSUPPORTED (no SECOND config has one).
 *   out_indices [out_cap,4]; num_out = device int[2]: [0] live outputs clamped to out_cap (feed it to the
 *   next layer as n_in_dev / num_out_dev), [1] the raw count (raw > out_cap == capacity overflow, to be
 *   checked by the caller whenever it next synchronises). State is kept in `workspace`. */
int sec_rulebook_conv3d_build(const int *indices, int n_in, const int *n_in_dev, int batch,
                              const int *h_in_shape3,
                              const int *h_out_shape3, const int *h_ksize3, const int *h_stride3,
                              const int *h_padding3, const int *h_dilation3, int *out_indices,
                              int out_cap, int *num_out, int out_per_in_hint, int *prefill_nbr_out,
                              int prefill_nbr_out_rows, int *prefill_nbr_in, void *workspace,
                              size_t workspace_bytes, void *stream);
/* (build, continued) prefill_nbr_out / prefill_nbr_in: when the caller already owns the gather tables (static-capacity
 * pipelines know their row counts before the build) the build's numbering launch also writes their -1 fill;
 * pass prefilled = 1 to step 2 then.  NULL = step 2 fills them itself.
 * step 2: fill the tables. nbr_out has `nbr_out_rows` rows (>= number of outputs; rows are -1 filled).
 * nbr_in may be NULL when neither the backward pass nor the pair lists are wanted (inference). */
int sec_rulebook_conv3d_tables(int n_in, const int *h_ksize3, const int *h_stride3,
                               const int *h_dilation3, int out_per_in_hint, int *nbr_out,
                               int nbr_out_rows, int *nbr_in, int prefilled, int *pairs, int *pair_num,
                               void *workspace, size_t workspace_bytes, void *stream);
/* The same three steps with spconv's GPU output numbering (spconv src/spconv/indice.cu, the path the reference takes on a
 * GPU: outputs = unique(linear cell index) in ASCENDING order; SURVEY.md Appendix A.4 "numbering = sorted"), pair order inside
 * an offset = ascending input row (the reference's GPU path has atomic-arrival order there: compare as sets).  The gather
 * tables describe the same convolution as the first-touch form with the output rows permuted -- features after dense() are
 * identical.  No hash table: a bitmap over the batch * D * H * W output cells (batch * volume < 2^32, else
 * SEC_E_UNSUPPORTED), one single-pass scan, rank = prefix[word] + popcount.  num_out as in sec_rulebook_conv3d_build. */
size_t sec_rulebook_sorted_workspace_bytes(int n_in, int kvol, int batch, const int *h_out_shape3);
int sec_rulebook_conv3d_build_sorted(const int *indices, int n_in, const int *n_in_dev, int batch,
                                     const int *h_in_shape3, const int *h_out_shape3, const int *h_ksize3,
                                     const int *h_stride3, const int *h_padding3, const int *h_dilation3,
                                     int *out_indices, int out_cap, int *num_out, int *prefill_nbr_out,
                                     int prefill_nbr_out_rows, int *prefill_nbr_in,
                                     int *prefill_extra, long long prefill_extra_words,
                                     const void *in_sites_workspace, size_t in_sites_workspace_bytes,
                                     void *workspace, size_t workspace_bytes, void *stream);
/* (build_sorted) prefill_extra: prefill_extra_words more int32 words set to -1 by the same init launch -- the gather table of
 * the SubM layer that follows on this layer's outputs (pass prefilled = 1 to sec_rulebook_subm3d_after_conv_sorted then).
 * out_indices may be NULL when the tables call below is given the buffer instead (it writes the same rows:
 * one launch less).  in_sites_workspace (optional): the workspace of the sorted build that produced `indices` (this layer's
 * inputs are that layer's outputs, e.g. through SubM layers on the same sites; it must be unmodified) -- the output bitmap
 * is then derived from that bitmap with plain loads (3x3x3 stride-2 and (3,1,1) stride-(2,1,1) layers, padding <= 1)
 * instead of one atomic per candidate. */
int sec_rulebook_conv3d_tables_sorted(const int *indices, int n_in, const int *n_in_dev, int batch,
                                      const int *h_in_shape3, const int *h_out_shape3, const int *h_ksize3,
                                      const int *h_stride3, const int *h_padding3, const int *h_dilation3,
                                      int *nbr_out, int nbr_out_rows, int *nbr_in, int prefilled, int *out_indices,
                                      int out_cap, int *pairs, int *pair_num, void *workspace,
                                      size_t workspace_bytes, void *stream);
/* SubMConv3d on the outputs of sec_rulebook_conv3d_build_sorted (its bitmap + ranks are the site lookup) */
int sec_rulebook_subm3d_after_conv_sorted(const int *indices, int n_in, const int *n_in_dev, int batch,
                                          const int *h_shape3, const int *h_ksize3, const int *h_dilation3,
                                          int *nbr_out, int prefilled, const void *conv_workspace,
                                          size_t conv_workspace_bytes, void *stream);
void sec_conv_output_shape(const int *h_in_shape3, const int *h_ksize3, const int *h_stride3,
                           const int *h_padding3, const int *h_dilation3, int *h_out_shape3);

/* ---------------------------------------------------------------------------------------------
 * indice_conv -- replaces spconv.ops.indice_conv / indice_subm_conv (torch.ops.spconv.indice_conv_*;
 * spconv_ops.h indiceConv):  out[o,:] = sum_k feat[nbr_out[o][k],:] @ W[k]  (fp32 accumulate)
 * followed by the optional fused epilogue  y = relu?( y * scale[c] + shift[c] )  (folded BatchNorm1d +
 * ReLU of second/pytorch/models/middle.py:146-191; bias is a shift with scale = 1).
 *   weight: spconv layout [kD,kH,kW,Cin,Cout] == [K,Cin,Cout], dtype = feature dtype.
 *   packed_weight: result of sec_pack_conv_weight for the MFMA path, or NULL (generic path).
 *     fp32 features (the reference's default precision, train.py:232-235): the products run on the BF16 matrix pipe with split
 *     operands -- v = bf16(v) + bf16(v - bf16(v)), x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32, error <= 3 * 2^-18 per
 *     product -- for the channel plans of SpMiddleFHD (16->32, 32->32, 32->64, 64->32, 64->64).  With packed_weight = the
 *     PRE-SPLIT image (sec_packed_weight_x3_bytes(kvol, cin, cout) bytes: per offset k the sec_pack_conv_weight image of bf16(W[k])
 *     followed by that of bf16(W[k] - bf16(W[k])); 27- and 3-offset kernels) the launch is the software-pipelined inference kernel;
 *     with NULL the weights are split on the fly (training, data gradients).
 *   num_out_dev: optional device int overriding n_out (static-capacity, sync-free pipelines).
 * --------------------------------------------------------------------------------------------- */
size_t sec_packed_weight_bytes(int kvol, int cin, int cout, int dtype);
size_t sec_packed_weight_x3_bytes(int kvol, int cin, int cout);
int sec_pack_conv_weight(const void *weight, int kvol, int cin, int cout, int dtype, void *packed,
                         void *stream);
int sec_indice_conv_fwd(const void *features, int n_in, int cin, const void *weight,
                        const void *packed_weight, int kvol, int cout, const int *nbr_out, int n_out,
                        const int *num_out_dev, const float *scale, const float *shift, int relu,
                        void *out, int dtype, int out_dtype, void *stream);
/* Which kernel sec_indice_conv_fwd dispatches for a shape (no launch): 0 generic VALU, 1 register-tiled VALU (fp32),
 * 2 Cin=4 first layer, 3 one MFMA wave per tile, 4 split-K MFMA, 5 split-K MFMA per 32-column slice, 6 row-split MFMA
 * with LDS-staged operands, 7-10 its A/B forms, 11 the buffer-load row-split kernel (the SubMConv3d 64->64 kernel of the
 * roofline figure), 12 the Cin=4 first layer on MFMA.  The parity tests
 * use it to prove which kernel an oracle comparison exercised.  sec_indice_conv_set_variant forces one kernel family
 * (same numbers as the SEC_CONV_VARIANT environment variable; < 0 restores the automatic choice): process-wide,
 * not thread-safe, meant for A/B measurements and tests. */
int sec_indice_conv_fwd_plan(int cin, int cout, int kvol, int n_out, int dtype, int out_dtype, int has_packed);
int sec_indice_conv_set_variant(int variant);
/* Arithmetic of the fp32 sparse convolutions, forward and data gradient (the reference computes fp32 products and sums:
 * train.py:232-235 builds the network in torch.float32 unless enable_mixed_precision).  Process-wide, not thread-safe, consulted
 * when a launch is issued (a captured graph keeps what it was captured with).
 *   SEC_FP32_SPLIT16 (default): the split-operand form described above -- 16 significant bits per operand, fp32 accumulation.
 *     Results carry the label "bf16x3" wherever this library's callers report a dtype.
 *   SEC_FP32_EXACT: v_mfma_f32_32x32x2_f32 (16->32 ... 64->64) / VALU fma (4->16, 16->16): IEEE fp32 products, fp32 accumulation;
 *     packed_weight images of fp32 weights are ignored. */
#define SEC_FP32_SPLIT16 0
#define SEC_FP32_EXACT 1
int sec_set_fp32_mode(int mode);
int sec_get_fp32_mode(void);
/* backward (spconv_ops.h indiceConvBackward): dfeat[j,:] = sum_k dout[nbr_in[j][k],:] @ W[k]^T ;
 * dW[k] = sum_o feat[nbr_out[o][k],:]^T dout[o,:].  fp32 gradients for weights, feature dtype for dfeat.
 * For 16-bit dtypes dfeat runs on the MFMA forward kernels over re-packed transposed weights, which live in
 * `workspace` (sec_indice_conv_bwd_workspace_bytes; NULL / too small selects the slower generic kernel).
 * dweight is accumulated with atomics into a zeroed buffer: `dweight_zeroed` != 0 says the caller zeroed it already (the
 * forward's sec_pack_conv_weight_train does, in the launch it runs anyway); 0: the call zeroes it (one memset node). */
size_t sec_indice_conv_bwd_workspace_bytes(int kvol, int cin, int cout, int dtype);
int sec_indice_conv_bwd(const void *features, int n_in, int cin, const void *weight, int kvol, int cout,
                        const int *nbr_out, const int *nbr_in, int n_out, const void *dout,
                        void *dfeat, float *dweight, int dtype, void *workspace, size_t workspace_bytes,
                        const void *packed_dgrad, int dweight_zeroed, void *stream);
/* Mixed-precision training (fp32 master weights, 16-bit features; train.py:196-203 keeps fp32 copies the same way): the three
 * 16-bit images of one layer's weight [kvol][cin][cout] in ONE launch -- `weight16` (the plain rounding), `packed_fwd`
 * (sec_pack_conv_weight's image; NULL when sec_packed_weight_bytes(kvol, cin, cout) is 0) and `packed_dgrad` (the transposed image
 * sec_indice_conv_bwd otherwise builds per call, sec_packed_weight_bytes(kvol, cout, cin) bytes; `subm` != 0: offsets mirrored,
 * as for a rulebook without an input-major table; NULL to skip).  Pass `packed_dgrad` to sec_indice_conv_bwd.
 * `zero_dweight` (optional, [kvol][cin][cout] fp32): zeroed by the same launch -- the accumulator the layer's backward will use. */
int sec_pack_conv_weight_train(const float *weight, int kvol, int cin, int cout, int subm, int dtype, void *weight16,
                               void *packed_fwd, void *packed_dgrad, float *zero_dweight, void *stream);
/* The same for n layers in ONE launch (host arrays of per-layer arguments; entries of packed_fwd / packed_dgrad / zero_dweight may be
 * NULL as above): a training step packs every sparse layer's weights before its forward pass instead of launching once per layer. */
int sec_pack_conv_weight_train_multi(int n, const float *const *weights, const int *kvol, const int *cin, const int *cout,
                                     const int *subm, int dtype, void *const *weight16, void *const *packed_fwd,
                                     void *const *packed_dgrad, float *const *zero_dweight, void *stream);

/* SparseConvTensor.dense() (spconv/__init__.py; consumed at second/pytorch/models/middle.py:206-210).
 * Scatter rows into a zero-initialised dense tensor with arbitrary element strides so the same kernel
 * writes NCDHW-contiguous or the channels-last [B, C*D, H, W] layout the RPN consumes.
 * The function clears `out_elems` elements first. */
int sec_sparse_to_dense(const void *features, const int *indices, int n, int c, const int *num_dev,
                        void *out, size_t out_elems, int64_t stride_b, int64_t stride_c,
                        int64_t stride_z, int64_t stride_y, int64_t stride_x, int dtype, void *stream);

/* SparseConvTensor.dense() WITHOUT the dense image, for the first RPN convolution (middle.py:206-210 -> rpn.py:486-497):
 * site_map[b][z][y][x] = row + 1 of the active site (0 = none), d = 2 planes.  sec_conv2d_nhwc_gather then runs the 3x3 /
 * stride 1 / pad 1 convolution of the [B, 2 * 64, H, W] view straight from the feature rows [rows][64] (16-bit): input channel
 * z * 64 + c, i.e. the weights must be packed from `weight[:, perm]` with perm[z * 64 + c] = c * 2 + z (the reference's
 * channel order after `.view(N, C * D, H, W)` is c * D + z).  Tiles without any active site write act(bias).
 * `feature_rows` = rows the feature buffer holds (capacity; only rows named by the map are read). */
int sec_sparse_site_map(const int *indices, int n, const int *num_dev, int batch, int d, int h, int w,
                        int *site_map, void *stream);
/* The same map for the outputs of a sorted-numbering strided build (sec_rulebook_conv3d_build_sorted), read off that build's
 * bitmap in ONE launch (no zero fill, no scatter): rows are numbered by ascending cell index, so a cell's row is its rank.
 * conv_workspace = the build's workspace, (d, h, w) = its output grid; rows beyond min(*num_dev, rows_cap) read as absent. */
int sec_sparse_site_map_sorted(const void *conv_workspace, size_t conv_workspace_bytes, const int *num_dev, int rows_cap,
                               int batch, int d, int h, int w, int *site_map, void *stream);
/* The WHOLE rulebook stack of a sparse middle in one call (second/pytorch/models/middle.py:146-189: SubM and strided layers
 * alternate; every spconv.ops.get_indice_pairs call of the stack, spconv GPU numbering = ascending linear cell index).
 * Level 0 = the caller's rows `indices0` [n0, 4] (any order, e.g. the voxeliser's arrival order; live rows = *n0_dev when given);
 * level l = 1 .. levels = outputs of the l-th strided conv (3x3x3 stride 2, or (3,1,1) stride (2,1,1); padding 0 / 1 per dim;
 * level 1 must be 3x3x3 stride 2; anything else: SEC_E_UNSUPPORTED -> build layer by layer).  h_shapes [(levels + 1) * 3] = grids
 * (D, H, W) of levels 0 .. levels; h_ksize / h_stride / h_pad [levels * 3]; h_out_cap [levels] = rows reserved per level.
 * HOST arrays of DEVICE pointers, one per level: h_nbr_out[l-1] [cap_l, kvol_l] (input row of level l-1 per (output, offset),
 * -1 = none), h_out_indices[l-1] [cap_l, 4] (coordinates of the output rows), h_num_out[l-1] int[2] = (live outputs clamped to cap_l, raw count:
 * raw > cap_l = overflow, reported, never out of bounds); h_subm_nbr[l] for l = 0 .. levels: the 3x3x3 SubM table [cap_l, 27]
 * on level l's sites or NULL.  h_subm_nbr[0] needs the finished sec_voxelize_f32 call's workspace (its hash table is the site
 * lookup of level 0; arguments as sec_rulebook_subm3d_after_voxelize).  site_map (or NULL): [batch, D, H, W] of the LAST level,
 * row + 1 / 0 (sec_sparse_site_map_sorted).  Rows at or past num_out[0] of any table are unspecified.
 * Same tables, element for element, as the layer-by-layer sorted builds (sec_rulebook_conv3d_build_sorted + _tables_sorted +
 * sec_rulebook_subm3d_after_conv_sorted / _after_voxelize) -- in 4 + (levels - 1) launches instead of ~25. */
size_t sec_rulebook_chain_workspace_bytes(int batch, int levels, const int *h_shapes);
int sec_rulebook_chain_sorted(const int *indices0, int n0, const int *n0_dev, int batch, int levels, const int *h_shapes,
                              const int *h_ksize, const int *h_stride, const int *h_pad, const int *h_out_cap,
                              int *const *h_nbr_out, int *const *h_out_indices, int *const *h_num_out,
                              int *const *h_subm_nbr, const void *vox_workspace, size_t vox_workspace_bytes,
                              int vox_num_points, int vox_max_voxels, int vox_max_points, const int *h_vox_grid3_zyx,
                              int *site_map, void *workspace, size_t workspace_bytes, void *stream);
int sec_conv2d_nhwc_gather(const void *features, long long feature_rows, const int *site_map, int batch, int h,
                           int w, const void *packed_weight, const float *bias, int cout, int relu,
                           const unsigned short *tile_order, const int *live_counts, const void *background, void *y,
                           int dtype, void *stream);   /* tile_order .. background: see sec_rpn_tile_live; tile_order NULL = every tile tests its own map entries; background NULL with lists = lazy consumers (below) */

/* sec_conv2d_nhwc writing channels [y_channel_offset, y_channel_offset + cout) of a wider channels-last map [batch][ho][wo][y_channels]:
 * the deblocks of a multi-block RPN deposit their outputs straight into the concatenated feature map the heads read (`torch.cat(ups, dim=1)`,
 * rpn.py:386-391) -- no concat copy.  Shapes: those of the strided / patch conv kernel (3x3 s2 p1 from 64 or 128 channels; k == stride 4 from 64,
 * 2 from 128 channels; 1x1 from 256 or 384 channels; cout a multiple of 128, or 64 -> 64 for the 3x3), SEC_E_UNSUPPORTED otherwise; offsets and
 * widths multiples of 8 channels.  Same arithmetic as sec_conv2d_nhwc (bit-identical values). */
int sec_conv2d_nhwc_into(const void *x, int batch, int h, int w, int cin, const void *packed_weight, const float *bias, int cout,
                         int ksize, int stride, int pad, int relu, void *y, int y_channels, int y_channel_offset, int dtype, void *stream);

/* PointPillarsScatter + the first RPN conv without the canvas (second/pytorch/models/pointpillars.py:444-476 writes the pillar rows
 * into a zeroed [B, C, ny, nx] image that rpn.py:484-486 `ZeroPad2d(1) + Conv2d(3, stride 2)` then reads): the convolution
 * (+ bias + ReLU, as sec_conv2d_nhwc) of that image straight from the pillar feature rows `rows` [feature_rows][cin] (16-bit) and
 * site_map [batch][h][w] = row + 1 of the cell's pillar, 0 = none (sec_sparse_site_map with d = 1).
 * Shapes: cin 64, 3x3 / stride 2 / pad 1, cout 64 or a multiple of 128 (SEC_E_UNSUPPORTED otherwise).  Bit-identical to
 * sec_conv2d_nhwc on the scattered image; tiles whose input cells hold no pillar write act(bias). */
int sec_conv2d_nhwc_rows(const void *rows, long long feature_rows, const int *site_map, int batch, int h, int w, int cin,
                         const void *packed_weight, const float *bias, int cout, int ksize, int stride, int pad, int relu, void *y,
                         int dtype, void *stream);

/* The dense RPN (rpn.py:486-497: Conv2d 3x3 + BatchNorm2d + ReLU, repeated) on a BEV image that is empty except at the sparse
 * middle's sites (middle.py:206-210): a pixel of the j-th conv's output sees the image only within j + 1 steps, so a tile
 * farther than that from every site holds exactly what the same network computes for an EMPTY frame at that position -- for
 * any weights -- and need not be convolved.
 * sec_rpn_tile_live: from the site map of sec_sparse_site_map* ([batch][2][h][w]), for the j-th 3x3 / stride 1 / pad 1 conv
 * of the RPN (j = 0: the first one, sec_conv2d_nhwc_gather; j = 0 .. layers - 1 <= 7) and frame b: tile_order[j][b][.] = the
 * frame's 8 x 16 output tiles (row-major indices, ceil(h / 8) x ceil(w / 16) per frame), the LIVE tiles -- those holding a
 * pixel within j + 1 steps (Chebyshev) of a site -- first (ascending), the others from the end backwards;
 * live_counts[j][b] = the number of live tiles.  workspace: sec_rpn_tile_live_workspace_bytes (the BEV bitmap).
 * sec_conv2d_nhwc_tiles: the 128-channel 3x3 / s1 / p1 conv + bias + ReLU of sec_conv2d_nhwc; the live tiles of
 * tile_order / live_counts (one layer's [batch][tiles] / [batch] slices) are convolved, spread evenly over the XCDs, the others
 * are copied from `background` = this layer's output for an empty frame, [h][w][cout] in the feature dtype, which the caller
 * computes once per network and map size with the same entry points.  tile_order == NULL: every tile is convolved.
 * Outputs are bit-identical to sec_conv2d_nhwc when the lists are the ones sec_rpn_tile_live derives. */
/* sec_conv1x1_chain_nhwc_tiles: sec_conv1x1_chain_nhwc (the RPN's deblock + merged heads, rpn.py:275-285,386-391) on the live
 * tiles of the LAST 3x3 conv's lists (a 1x1 conv reaches no farther); the other tiles are copied from `background` = its own
 * output for an empty frame, [h][w][cout2]. */
size_t sec_rpn_tile_live_workspace_bytes(int batch, int h, int w);
int sec_rpn_tile_live(const int *site_map, int batch, int h, int w, int layers, unsigned short *tile_order, int *live_counts,
                      void *workspace, size_t workspace_bytes, void *stream);
int sec_conv2d_nhwc_tiles(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int cout,
                          int relu, const unsigned short *tile_order, const int *live_counts, const void *background, void *y,
                          int dtype, void *stream);
int sec_conv1x1_chain_nhwc_tiles(const void *x, int batch, int h, int w, const void *packed_w1, const float *bias1, int relu1,
                                 const void *packed_w2, const float *bias2, int cout2, const unsigned short *tile_order,
                                 const int *live_counts, const void *background, void *y, int dtype, void *stream);
/* LAZY background (round 4): the copies above move a background tile of EVERY layer through memory although its only readers are
 * the halos of the next conv's live tiles.  With the lazy forms a conv writes its live tiles only (background == NULL in
 * sec_conv2d_nhwc_gather / sec_conv2d_nhwc_tiles_lazy: the other tiles of y are left unwritten) and its consumer fetches every halo
 * pixel either from x or -- when the tile holding it was not live in the producing layer -- from `background_in` = the producing
 * layer's output for an empty frame ([h][w][128]), the very values a copy would have put there: bit-identical results.
 * sec_rpn_tile_live_masks: sec_rpn_tile_live + nbr_masks [layers][2][batch][tiles] uint16: for conv l >= 1 (slot 0 is unused) and a
 * tile, bit (dy + 1) * 3 + (dx + 1) = the neighbour tile (dy, dx) was live for conv l - 1 (or lies outside the image: zero padding
 * either way); [l][0][b][r] follows tile_order[l][b][r] for the live tiles, [l][1][b][t] is indexed by tile (read when a conv falls
 * back to the plain tile order above three quarters live tiles).
 * sec_conv2d_nhwc_tiles_lazy: sec_conv2d_nhwc_tiles reading x through nbr_masks (the [2][batch][tiles] slice of ITS layer) and
 * background_in; background (its own empty-frame output) may be NULL when every consumer of y is lazy too.
 * sec_conv1x1_chain_nhwc_tiles with relu1 | SEC_CHAIN_X_LIVE_ONLY is the lazy consumer of the LAST conv: it then uses the lists
 * whatever the live share is (a 1x1 conv reads no halo: the live tiles of the last conv's list are all it needs of x). */
#define SEC_CHAIN_X_LIVE_ONLY 2
int sec_rpn_tile_live_masks(const int *site_map, int batch, int h, int w, int layers, unsigned short *tile_order, int *live_counts,
                            unsigned short *nbr_masks, void *workspace, size_t workspace_bytes, void *stream);
int sec_conv2d_nhwc_tiles_lazy(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int cout,
                               int relu, const unsigned short *tile_order, const int *live_counts, const void *background,
                               const unsigned short *nbr_masks, const void *background_in, void *y, int dtype, void *stream);
/* The LAST 3x3 conv of a single-block RPN with its 1x1 tail in the epilogue (round 6): sec_conv2d_nhwc_tiles_lazy (128 -> 128, background
 * NULL) followed by sec_conv1x1_chain_nhwc_tiles (relu1 | SEC_CHAIN_X_LIVE_ONLY, background NULL, cout2 = 64) on the same lists, as ONE
 * launch -- the conv's 16-bit output tile goes from LDS straight into the two 1x1 GEMMs (deblock rpn.py:275-285, merged heads
 * rpn.py:386-391) and never reaches memory; y_heads [batch][h][w][64] holds the live tiles of the list (every tile when the conv falls
 * back to the plain tile order), bit-identical to the two launches.  cout2 != 64 or a dtype other than bf16 / f16: SEC_E_UNSUPPORTED. */
int sec_conv2d_nhwc_tiles_tail(const void *x, int batch, int h, int w, const void *packed_weight, const float *bias, int relu,
                               const unsigned short *tile_order, const int *live_counts, const unsigned short *nbr_masks,
                               const void *background_in, const void *packed_w1, const float *bias1, int relu1, const void *packed_w2,
                               const float *bias2, int cout2, void *y_heads, int dtype, void *stream);

/* Adjoint of sec_sparse_to_dense -- rows[i,:] = dense[indices[i]] -- i.e. the backward of
 * SparseConvTensor.dense() (upstream gets it from autograd through scatter_nd, spconv/__init__.py) and of
 * PointPillarsScatter (pointpillars.py:444-476) with stride_z = 0. */
int sec_dense_to_sparse(const void *dense, const int *indices, int n, int c, const int *num_dev, void *rows,
                        int64_t stride_b, int64_t stride_c, int64_t stride_z, int64_t stride_y,
                        int64_t stride_x, int dtype, void *stream);     /* num_dev: as in sec_sparse_to_dense (rows past it untouched) */

/* PointPillarsScatter.forward (second/pytorch/models/pointpillars.py:444-476): coords (b,z,y,x),
 * canvas [B, C, ny, nx] (strides given in elements), cleared first.  num_dev (optional device int): only the
 * first *num_dev pillars are live (static-capacity pipelines). */
int sec_pillar_scatter(const void *features, const int *coords, int p, int c, const int *num_dev,
                       void *out, size_t out_elems, int64_t stride_b, int64_t stride_c, int64_t stride_y,
                       int64_t stride_x, int dtype, void *stream);

/* PillarFeatureNet.forward with a single PFNLayer in eval mode (second/pytorch/models/pointpillars.py:
 * 203-237, 51-65; config nuscenes/all.pp.largea.config:10-15): point decoration (cluster / pillar-centre
 * offsets), Linear(F+5 -> C, no bias; weight_t = weight^T [F+5, C]), folded BatchNorm1d (scale, shift), ReLU,
 * max over the max_points slots -- one launch.  voxels [P, max_points, 4] fp32, coords [P,4] (b,z,y,x). */
int sec_pfn_fwd(const float *voxels, const int *num_points, const int *coords, int num_pillars,
                const int *num_dev, int max_points, int num_features, const float *weight_t,
                const float *scale, const float *shift, int channels, float vx, float vy,
                float x_offset, float y_offset, void *out, int out_dtype, void *stream);
/* The same on the voxeliser's point lists instead of a [P, T, 4] tensor: `points` is the flat point array sec_voxelize_f32 was
 * called on, (vox_workspace, vox_num_points, vox_batch, vox_max_voxels, max_points) identify that call's workspace, which still
 * holds every pillar's point indices (sec_voxelize_f32 may then be called with voxels = NULL: it skips writing the tensor and only
 * fills num_points_per_voxel).  Bit-identical to sec_pfn_fwd on the materialised pillars. */
int sec_pfn_fwd_slots(const float *points, const void *vox_workspace, size_t vox_workspace_bytes, int vox_num_points,
                      int vox_batch, int vox_max_voxels, int max_points, int num_features, const int *num_points_per_voxel,
                      const int *coords, int num_pillars, const int *num_dev, const float *weight_t, const float *scale,
                      const float *shift, int channels, float vx, float vy, float x_offset, float y_offset, void *out,
                      int out_dtype, void *stream);

/* PillarFeatureNet in TRAINING mode (PFNLayer.forward under train(), pointpillars.py:51-65, trained by train.py:316-322):
 * Linear(9, C, bias=False) on the decorated, masked points, BatchNorm1d with BATCH statistics over all P * T rows (padded slots
 * included, as the reference's [P, T, C] tensor has them), ReLU, max over the T slots -- and its backward -- without ever
 * materialising [P, T, C].  C <= 64, 4 point features, T <= 127.
 *   fwd : out [P, C] f32; argmax [P, C] int8 (slot of the maximum, -1 = a padded slot); stats f32[C * 11 + 9] = mean[C],
 *         invstd[C], M[C][9] = sum x_c * input_j, S1[9] = sum input_j (kept for the backward); running_mean / running_var (may be
 *         NULL) updated with `momentum` (unbiased variance), like torch.nn.BatchNorm1d.
 *   bwd : grad_out [P, C] f32 -> dweight_t [9][C] (the layout of weight_t), dgamma [C], dbeta [C]; no gradient for the points.
 * workspace: sec_pfn_train_workspace_bytes(num_pillars, channels) for either call. */
size_t sec_pfn_train_workspace_bytes(int num_pillars, int channels);
int sec_pfn_train_fwd(const float *voxels, const int *num_points, const int *coords, int num_pillars, int max_points,
                      int num_features, const float *weight_t, const float *gamma, const float *beta, float eps, float momentum,
                      float *running_mean, float *running_var, int channels, float vx, float vy, float x_offset, float y_offset,
                      float *out, signed char *argmax, float *stats, void *workspace, size_t workspace_bytes, void *stream);
int sec_pfn_train_bwd(const float *voxels, const int *num_points, const int *coords, int num_pillars, int max_points,
                      int num_features, const float *weight_t, const float *gamma, const float *stats, int channels, float vx,
                      float vy, float x_offset, float y_offset, const float *grad_out, const float *out, const signed char *argmax,
                      float *dweight_t, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream);

/* Block filtering of points_to_voxel_3d_with_filtering (spconv point2voxel.h; enabled by
 * second/configs/nuscenes/all.fhd.config:9-12): keep a voxel iff the z-span of the stored points inside the
 * block_size x block_size block window around it lies in (height_threshold, height_high_threshold); the
 * voxel arrays are compacted in order.  Input = output layout of sec_voxelize_f32 (coors (b,z,y,x)). */
size_t sec_block_filter_workspace_bytes(int rows, int batch, int grid_x, int grid_y, int block_factor);
int sec_voxel_block_filter_f32(const float *voxels, const int *coors, const int *num_points,
                               const int *voxel_offsets, int rows, int batch, int max_points,
                               int num_features, int grid_x, int grid_y, int block_factor, int block_size,
                               float height_threshold, float height_high_threshold, float *out_voxels,
                               int *out_coors, int *out_num_points, int *out_offsets, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Dense RPN helper (second/pytorch/models/rpn.py:486-497: Conv2d -> BatchNorm2d -> ReLU): in-place
 * y = relu?(x + bias[c]) on a channels-last [pixels, channels] activation (bias = folded BatchNorm). */
int sec_bias_act_nhwc(void *x, const float *bias, size_t pixels, int channels, int relu, int dtype,
                      void *stream);

/* Dense conv2d of the RPN (second/pytorch/models/rpn.py:468-497 blocks, :275-285 1x1 deconv, :386-391 heads):
 * channels-last [B,H,W,Cin] bf16/f16, implicit GEMM on MFMA with bias (folded BatchNorm2d) + ReLU fused.
 * weight [Cout,Cin,k,k] (torch layout) is re-packed once by sec_conv2d_pack_weight.  Cin, Cout multiples of 64.
 * Output [B,Ho,Wo,Cout] with Ho = (H + 2 pad - k) / stride + 1.
 * `relu` is a flag word: bit 0 = ReLU; bit 1 = the input is a scattered sparse tensor (the first RPN layer after
 * SparseConvTensor.dense(), middle.py:206-210): the 3x3/s1 kernel then tests each input tile and writes act(bias)
 * for all-zero ones without running the MFMA loop -- bit-identical results for finite weights. */
size_t sec_conv2d_packed_weight_bytes(int cout, int cin, int ksize, int dtype);
int sec_conv2d_pack_weight(const void *weight, int cout, int cin, int ksize, int dtype, void *packed,
                           void *stream);
int sec_conv2d_nhwc(const void *x, int batch, int h, int w, int cin, const void *packed_weight,
                    const float *bias, int cout, int ksize, int stride, int pad, int relu, void *y,
                    int dtype, void *stream);

/* The same 3x3 / stride 1 / pad 1, 128-input-channel convolution for fp32 networks -- the reference's default precision
 * (second/pytorch/train.py:232-235, 497-500: float_dtype = torch.float32 unless mixed precision is enabled; rpn.py:468-497) -- on
 * the bf16 matrix pipe: every fp32 operand travels as TWO bf16 planes, v = hi + lo with hi = bf16(v), lo = bf16(v - hi), and
 * the product is x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 (relative error <= 3 * 2^-18 per product: within the
 * 1e-4 feature tolerance; MIOpen's fp32 convolution of this layer runs at the fp32 MFMA rate, 4x slower).
 *   x_hi, x_lo, y_hi, y_lo : [B,H,W,128] / [B,H,W,Cout] bf16 planes (sec_split_f32_bf16x2 makes them, sec_merge_bf16x2_f32 adds
 *                            them back to fp32); bias + ReLU are applied in fp32 before the result is split;
 *   packed_weight_hi_lo    : sec_conv2d_pack_weight(bf16(W)) immediately followed by sec_conv2d_pack_weight(bf16(W - bf16(W)))
 *                            (each 9 * 128 * Cout * 2 bytes) + 16 zero bytes;
 *   relu                   : flag word as in sec_conv2d_nhwc (bit 1: all-zero input tiles write act(bias)).
 * n of the split / merge helpers = number of fp32 elements, a multiple of 4. */
int sec_split_f32_bf16x2(const float *x, long long n, void *hi, void *lo, void *stream);
int sec_merge_bf16x2_f32(const void *hi, const void *lo, long long n, float *y, void *stream);
int sec_conv2d_nhwc_x3(const void *x_hi, const void *x_lo, int batch, int h, int w, const void *packed_weight_hi_lo,
                       const float *bias, int cout, int relu, void *y_hi, void *y_lo, void *stream);
/* ... on the tiles a site of the sparse middle can reach only (the fp32 counterpart of sec_conv2d_nhwc_tiles / _tiles_lazy; same
 * lists, masks and semantics, every image as a (hi, lo) plane pair): tile_order / live_counts = one layer of sec_rpn_tile_live;
 * background_hi / _lo (both or neither) = this layer's output for an EMPTY frame, copied into the tiles that are not live (NULL: they
 * are left unwritten, for a lazy consumer); nbr_masks (may be NULL: x holds every tile) + background_in_hi / _lo = the PRODUCING
 * layer's empty-frame output, read for halo pixels of tiles the producer did not write.  Bit-identical to sec_conv2d_nhwc_x3 on
 * the full image. */
/* The fused 1x1 tail (sec_conv1x1_chain_nhwc: deblock + merged heads, rpn.py:275-285,386-391) for fp32 networks: x as (hi, lo) bf16
 * planes [pixels][128] (the output of the last sec_conv2d_nhwc_x3), both weight sets as sec_conv2d_pack_weight(bf16(W)) followed by
 * sec_conv2d_pack_weight(bf16(W - bf16(W))) (ksize 1) + 16 zero bytes, the heads as fp32 [pixels][cout2], cout2 in {64, 128}. */
int sec_conv1x1_chain_x3(const void *x_hi, const void *x_lo, long long pixels, const void *packed_w1_hi_lo, const float *bias1,
                         int relu1, const void *packed_w2_hi_lo, const float *bias2, int cout2, float *y, void *stream);
int sec_conv2d_nhwc_x3_tiles(const void *x_hi, const void *x_lo, int batch, int h, int w, const void *packed_weight_hi_lo,
                             const float *bias, int cout, int relu, const unsigned short *tile_order, const int *live_counts,
                             const void *background_hi, const void *background_lo, const unsigned short *nbr_masks,
                             const void *background_in_hi, const void *background_in_lo, void *y_hi, void *y_lo, void *stream);

/* Fused tail of the RPN at inference: y = W2 * act(W1 * x + bias1) + bias2 over `pixels` channels-last pixels with
 * 128 input and 128 intermediate channels -- the 1x1/stride-1 ConvTranspose2d deblock with folded BatchNorm + ReLU
 * (second/pytorch/models/rpn.py:275-285) followed by the merged conv_box / conv_cls / conv_dir_cls 1x1 heads
 * (rpn.py:386-391, 412-420).  cout2 in {64, 128}; weights from sec_conv2d_pack_weight (ksize 1); bias2 may be NULL.
 * The intermediate never reaches HBM. */
int sec_conv1x1_chain_nhwc(const void *x, long long pixels, const void *packed_w1, const float *bias1,
                           int relu1, const void *packed_w2, const float *bias2, int cout2, void *y,
                           int dtype, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training of the dense RPN with 16-bit activations over fp32 master weights -- replaces what autograd runs for
 * nn.Conv2d(128, 128, 3, padding=1, bias=False) + nn.BatchNorm2d(eps 1e-3, momentum 0.01) + nn.ReLU
 * (second/pytorch/models/rpn.py:486-497) inside loss.backward() (second/pytorch/train.py:316-322): MIOpen's igemm
 * backward-weights / backward-data kernels and ~8 torch kernels per BatchNorm + ReLU pair.
 *   forward conv        sec_conv2d_nhwc (bias NULL, relu 0);
 *   data gradient       sec_conv2d_nhwc on dY with the weights flipped and transposed: dX = conv(dY, W[ci][co][2-ky][2-kx]);
 *   weight gradient     sec_conv2d_wgrad_nhwc: dweight [cout][cin][k][k] fp32 (torch's layout) = sum over pixels of
 *                       x[p + tap] (x) dy[p]; 3x3 / stride 1 / pad 1 with cin = cout = 128, or 1x1 with cin = 128 and cout = 128
 *                       or 64 (dy then holds 64 channels: the stacked heads); at most 2^23 pixels; deterministic (partials in
 *                       the workspace, summed in a fixed order -- no float atomics);
 *   BatchNorm + ReLU    sec_bn_relu_fwd_nhwc: batch statistics over `pixels` rows of a channels-last [pixels][channels] tensor
 *                       (biased variance for the normalisation, running statistics updated with the unbiased one, as
 *                       torch.nn.BatchNorm2d does), z = act((y - mean) * invstd * gamma + beta); save_mean / save_invstd
 *                       [channels] are kept for the backward.  sec_bn_relu_bwd_nhwc: dy, dgamma, dbeta from dz and y (the ReLU
 *                       mask is recomputed from y).  channels % 8 == 0, <= 256, a divisor of 2048.
 * --------------------------------------------------------------------------------------------- */
/* fp32 master weight [cout][cin][k][k] -> the 16-bit packed images of BOTH launches of a training step in one pass: `packed_fwd`
 * (what sec_conv2d_pack_weight makes of the 16-bit rounding of the weight) and `packed_dgrad` (the same of the flipped, transposed
 * kernel W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx], the weights of the data-gradient convolution); each
 * sec_conv2d_packed_weight_bytes(cout, cin, ksize, dtype) bytes. */
int sec_conv2d_pack_weight_train(const float *weight, int cout, int cin, int ksize, int dtype, void *packed_fwd,
                                 void *packed_dgrad, void *stream);
/* n layers in ONE launch (host arrays of per-layer arguments). */
int sec_conv2d_pack_weight_train_multi(int n, const float *const *weights, const int *cout, const int *cin, const int *ksize,
                                       int dtype, void *const *packed_fwd, void *const *packed_dgrad, void *stream);
size_t sec_conv2d_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int ksize);
int sec_conv2d_wgrad_nhwc(const void *x, const void *dy, int batch, int h, int w, int cin, int cout, int ksize,
                          int stride, int pad, float *dweight, void *workspace, size_t workspace_bytes, int dtype,
                          void *stream);
size_t sec_bn_train_workspace_bytes(int channels);
/* pixels_dev (device int, or NULL): static-capacity rows -- `pixels` is the capacity, the first *pixels_dev rows are live;
 * statistics and both passes run over the live rows only, rows past them are neither read nor written. */
int sec_bn_relu_fwd_nhwc(const void *y, long long pixels, int channels, const float *gamma, const float *beta, float eps,
                         float momentum, float *running_mean, float *running_var, int relu, void *z, float *save_mean,
                         float *save_invstd, void *workspace, size_t workspace_bytes, int dtype, const int *pixels_dev,
                         void *stream);
int sec_bn_relu_bwd_nhwc(const void *dz, const void *y, long long pixels, int channels, const float *gamma,
                         const float *beta, const float *save_mean, const float *save_invstd, int relu, void *dy,
                         float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, int dtype,
                         const int *pixels_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rotated IoU / NMS -- replace the numba.cuda kernels of second/core/non_max_suppression/nms_gpu.py
 * (rotate_iou_kernel_eval :564-602, rotate_nms_kernel :404-437, nms_kernel :70-101, nms_postprocess
 * :109-126) and the CPU path rotate_nms_cc (nms_cpu.py:17-28 -> spconv rotate_non_max_suppression_cpu).
 * --------------------------------------------------------------------------------------------- */
/* iou[n,k] for boxes [N,5], qboxes [K,5] (x,y,w,l,r); criterion -1 IoU, 0 inter/area(qbox k), 1 inter/area(box n), 2 inter.
 * (nms_gpu.py:549-561 writes 0 -> inter/area1, 1 -> inter/area2 with rbox1 = the QUERY box: rotate_iou_kernel_eval passes
 *  block_qboxes first, nms_gpu.py:600-604; tests/golden/rotate_iou.npz pins all four criteria on asymmetric N != K inputs.) */
int sec_rotate_iou_f32(const float *boxes, int n, const float *qboxes, int k, int criterion,
                       float *iou, void *stream);
/* Batched greedy NMS on boxes ALREADY SORTED by descending score.
 *   dets [batch, max_n, stride] (first 5 columns x,y,w,l,r  -- or x1,y1,x2,y2 for the axis-aligned kind);
 *   counts [batch] device ints (boxes per item, <= max_n <= 4096).
 *   kind 0 rotated, 1 axis-aligned.  semantics 0 = numba.cuda spec (IoU > thr; axis-aligned uses the
 *   '+1' convention), 1 = CPU path (rotated: standup-IoU pre-filter then IoU >= thr; axis-aligned:
 *   eps convention, >= thr).
 *   keep [batch, max_n] receives kept positions (ascending), num_keep [batch] their count (capped at post_max, <=0 = no cap).
 *   semantics | SEC_NMS_EXACT_CLIP: every standup-overlapping pair goes through the reference's polygon clipper.  Without the
 *   flag a pair whose inscribed-circle LOWER bound of the IoU already clears the threshold is decided without clipping (same
 *   keep lists wherever the reference's clipper computes the true intersection; it can under-report it for vertex lists of
 *   more than 8 points / near-degenerate boxes, where the flag reproduces nms_gpu.py's own answer).
 *   workspace: sec_nms_workspace_bytes(batch, max_n). */
#define SEC_NMS_EXACT_CLIP 256
size_t sec_nms_workspace_bytes(int batch, int max_n);
int sec_nms_sorted_f32(const float *dets, const int *counts, int batch, int max_n, int stride,
                       float thresh, int kind, int semantics, float eps, int post_max, int *keep,
                       int *num_keep, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Post-processing of VoxelNet.predict (second/pytorch/models/voxelnet.py:377-645) without leaving the device.
 * RPN outputs are addressed in place through element strides of a [B, A, H, W, C] view (h_*_strides5 =
 * {sb, sa, sy, sx, sc}); anchor n = (a*H + y)*W + x as in target_assigner.generate_anchors.
 *   select  : best class per anchor (num_class > 1), top-k by score (k <= 1024) sorted descending;
 *             counts[b] = entries with sigmoid score >= score_thr (voxelnet.py:545-569, box_torch_ops.py:497-501).
 *             The reference masks by the threshold BEFORE its topk: the result is rows [0, counts[b]) of each frame; what
 *             the rows behind them hold is unspecified (the next-best anchors, or anchor 0 / score 0 when the bf16 path
 *             skipped its bisection because no more than k anchors reach the threshold)
 *   decode  : second_box_decode (box_torch_ops.py:56-101) of the selected anchors -> decoded [B,k,7];
 *             dets [B,k,6] = NMS rows (rotate: x,y,w,l,r,score ; else standup x1,y1,x2,y2,score,0); direction argmax
 *   finalize: gather of the NMS survivors (keep / num_keep of sec_nms_sorted_f32), direction fix
 *             (voxelnet.py:598-607), post_center_range mask (:611-621) -> boxes [B,post_max,7], scores, labels, valid
 * --------------------------------------------------------------------------------------------- */
int sec_predict_select(const void *cls, const int64_t *h_cls_strides5, int batch, int anchors_per_loc,
                       int h, int w, int num_class, int k, float score_thr,
                       unsigned *key_scratch /* batch * anchors_per_loc * h * w words */, int *top_idx,
                       float *top_score, int *top_label, int *counts, int dtype, void *stream);
int sec_predict_decode(const void *box, const int64_t *h_box_strides5, const void *dir,
                       const int64_t *h_dir_strides5, int num_dir_bins, int batch, int anchors_per_loc,
                       int h, int w, int k, const float *anchors, const int *top_idx,
                       const float *top_score, int rotate, float *decoded, float *dets, int *dir_label,
                       int dtype, void *stream);
/* LAZY heads: the producer of the head tensor (sec_conv1x1_chain_nhwc_tiles with background == NULL and SEC_CHAIN_X_LIVE_ONLY) wrote
 * only the 8 x 16 tiles its live list names -- the copy of the other ~60 % of the map (22 MB per batch of 8 for car.fhd) never
 * happens.  tile_live [batch][ceil(h/8) * ceil(w/16)] (row-major tiles): bit 4 set = the tile was written (the tile-indexed half of
 * layer `convs` of sec_rpn_tile_live_masks called with convs + 1 layers); elements of any other tile are read from *_background =
 * the same view (same sa / sy / sx / sc strides, ONE frame) of that producer's output for an empty frame -- exactly what the copy
 * would have written.  Results identical, bit for bit, to the eager entry points on the materialised tensor. */
int sec_predict_select_lazy(const void *cls, const int64_t *h_cls_strides5, int batch, int anchors_per_loc,
                            int h, int w, int num_class, int k, float score_thr, unsigned *key_scratch, int *top_idx,
                            float *top_score, int *top_label, int *counts, int dtype, const unsigned short *tile_live,
                            const void *cls_background, void *stream);
int sec_predict_decode_lazy(const void *box, const int64_t *h_box_strides5, const void *dir,
                            const int64_t *h_dir_strides5, int num_dir_bins, int batch, int anchors_per_loc,
                            int h, int w, int k, const float *anchors, const int *top_idx,
                            const float *top_score, int rotate, float *decoded, float *dets, int *dir_label,
                            int dtype, const unsigned short *tile_live, const void *box_background,
                            const void *dir_background, void *stream);
int sec_predict_finalize(const float *decoded, const float *top_score, const int *top_label,
                         const int *dir_label, const int *keep, const int *num_keep, int batch, int k,
                         int post_max, int use_direction, float dir_offset, float dir_limit_offset,
                         int num_dir_bins, const float *range6, float *boxes, float *scores, int *labels,
                         unsigned char *valid, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training side: what the reference computes in DataLoader workers (numpy) and in ~40 torch kernels.
 *
 * sec_assign_targets_f32 -- replaces TargetAssigner.assign / assign_per_class -> create_target_np
 *   (second/core/target_assigner.py:51-88, second/core/target_ops.py:29-229) for the configuration every shipped config
 *   uses: NearestIouSimilarity (second/core/region_similarity.py:73-93), GroundBox3dCoder.encode
 *   (second/core/box_np_ops.py:36-81), sample_positive_fraction = -1 (no sub-sampling), no anchor pruning.
 *   anchors [n_anchor,7]; the ground truth of frame b is gt_boxes[gt_offsets[b] .. gt_offsets[b+1]) (x,y,z,w,l,h,r),
 *   gt_classes (1-based, NULL = all 1), gt_importance (NULL = all 1).  Outputs per (frame, anchor): labels
 *   (class / 0 background / -1 don't care), bbox_targets [.,7] (zeros for non-positives), importance.
 * sec_second_loss_f32 -- replaces VoxelNet.loss (second/pytorch/models/voxelnet.py:239-312): SigmoidFocalClassificationLoss
 *   + WeightedSmoothL1LocalizationLoss on the sin-difference encoding + direction WeightedSoftmaxClassificationLoss
 *   (second/pytorch/core/losses.py:135-296,358-392), NormByNumPositives weighting (voxelnet.py:756-797).  One pass returns the
 *   six scalars out6 = (loss, cls_loss_reduced, loc_loss_reduced, dir_loss_reduced, cls_pos_loss, cls_neg_loss) AND the
 *   gradients of `loss` w.r.t. the head outputs (d_cls [B,N,num_class], d_box [B,N,7], d_dir [B,N,bins]).
 *   h_params17 (host): alpha, gamma, sigma, pos_cls_weight, neg_cls_weight, classification_weight, localization_weight,
 *   direction_loss_weight, direction_offset, sin_error_factor, code_weight[7].  Head tensors contiguous fp32.
 * --------------------------------------------------------------------------------------------- */
size_t sec_assign_targets_workspace_bytes(int batch, int n_anchor, int n_gt);
int sec_assign_targets_f32(const float *anchors, int n_anchor, const float *gt_boxes, const int *gt_classes,
                           const float *gt_importance, const int *gt_offsets, int n_gt, int batch,
                           float matched_threshold, float unmatched_threshold, int *labels, float *bbox_targets,
                           float *importance, void *workspace, size_t workspace_bytes, void *stream);
/* Anchor ranges with their own thresholds.  Range c = anchors [h_class_anchor_begin[c], h_class_anchor_begin[c+1]) (the
 * reference concatenates the anchor generators class-major, target_assigner.py:169-207), thresholds h_matched[c] /
 * h_unmatched[c].  h_class_ids[c] = k > 0: TargetAssigner.assign_per_class (target_assigner.py:90-160; all.fhd.config:295) --
 * the range is matched against the ground truth of class k only, a frame without such a box labels the range background, and
 * gt_importance is indexed the way the reference does (position within the class's boxes, applied to the frame's array).
 * h_class_ids[c] = 0: TargetAssigner.assign_all with per-anchor threshold arrays (target_assigner.py:53-88;
 * all.pp.largea.config:269) -- every ground truth, best overlap per ground truth taken over all ranges. */
int sec_assign_targets_per_class_f32(const float *anchors, int n_anchor, const float *gt_boxes, const int *gt_classes,
                                     const float *gt_importance, const int *gt_offsets, int n_gt, int batch, int n_class,
                                     const int *h_class_anchor_begin, const int *h_class_ids, const float *h_matched,
                                     const float *h_unmatched, int *labels, float *bbox_targets, float *importance,
                                     void *workspace, size_t workspace_bytes, void *stream);
size_t sec_second_loss_workspace_bytes(int batch, int n_anchor);
int sec_second_loss_f32(const float *cls_preds, const float *box_preds, const float *dir_preds, const int *labels,
                        const float *reg_targets, const float *anchors, const float *importance, int batch,
                        int n_anchor, int num_class, int num_dir_bins, const float *h_params17, float *d_cls, float *d_box,
                        float *d_dir, float *out6, void *workspace, size_t workspace_bytes, void *stream);
/* The same loss read straight from the STACKED heads of the training step: heads [batch, h, w, head_channels] channels last, 16 bit
 * = box [A*7] | cls [A*num_class] | dir [A*bins] | zero padding, the output of the three 1x1 head convolutions run as one
 * (second/pytorch/models/rpn.py:386-391; the reference views each as [B, A, H, W, code], anchor n = (a*H + y)*W + x, and hands three
 * fp32 copies to VoxelNet.loss, voxelnet.py:239-312, whose gradients autograd stitches back).  _fwd: out6 as sec_second_loss_f32.
 * _bwd: d_heads (the layout and dtype of `heads`, padding channels zero) = grad_loss[0] * d loss / d heads, rounded once after the
 * multiplication (grad_loss: device float, the gradient arriving at the loss -- the loss scale of fp16 training; NULL = 1), and
 * d_bias [head_channels] fp32 = its sum over pixels (fixed order).  Same arithmetic as sec_second_loss_f32, expression by expression.
 * counts_ready != 0: `workspace` is the one the matching _fwd call used, untouched since -- the frames' positive counts in it are reused.
 * sec_heads_loss_supported: 1 for the instantiated shapes (head_channels 64, A = 2, one class, 0 or 2 direction bins, bf16 / fp16) --
 * anything else returns SEC_E_UNSUPPORTED and the caller keeps the three-tensor path. */
int sec_heads_loss_supported(int head_channels, int anchors_per_loc, int num_class, int num_dir_bins, int dtype);
size_t sec_heads_loss_workspace_bytes(int batch, int h, int w, int anchors_per_loc);
int sec_heads_loss_fwd(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                       int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                       const float *h_params17, float *out6, void *workspace, size_t workspace_bytes, void *stream);
/* _fwd_terms: _fwd plus the per-anchor tensors VoxelNet.loss returns beside the scalars (voxelnet.py:299-309; read by train.py:312-313,
 * 326-329 for update_metrics and the per-code loss display): cls_preds_out [batch, A*h*w, num_class] = the cls logits as fp32 in the
 * reference's anchor order, cls_loss_out [batch, A*h*w, num_class], loc_loss_out [batch, A*h*w, 7] (weighted focal / smooth-L1 terms
 * before the batch mean); any of the three may be NULL. */
int sec_heads_loss_fwd_terms(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                             int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                             const float *h_params17, float *out6, float *cls_preds_out, float *cls_loss_out, float *loc_loss_out,
                             void *workspace, size_t workspace_bytes, void *stream);
int sec_heads_loss_bwd(const void *heads, int dtype, int batch, int h, int w, int head_channels, int anchors_per_loc, int num_class,
                       int num_dir_bins, const int *labels, const float *reg_targets, const float *anchors, const float *importance,
                       const float *h_params17, const float *grad_loss, void *d_heads, float *d_bias, void *workspace,
                       size_t workspace_bytes, int counts_ready, void *stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_grad_norm) + the AdamW step (second/pytorch/train.py:323-325; adam + fixed weight
 * decay, car.fhd.config:180-188) on ONE flat fp32 buffer of master weights whose flat gradient is the all-reduce bucket: two launches
 * (fixed-order sum of squares + element-wise update).  state4 (device float[4]) = (gradient norm of this step, step count, 1 if
 * this step was skipped, the loss scale its gradients carried): the call advances the count itself; zero it before the first step.  loss_scale4 (device float[4] or
 * NULL) = (loss scale, clean steps in a row, growth interval, skipped steps): dynamic loss scaling ON THE DEVICE for fp16 features
 * (train.py:209-216,318-322 does it through apex amp on the host) -- the gradients carry the scale, a non-finite norm halves the
 * scale and skips the update, `growth interval` clean steps double it; nothing is read back.  workspace:
 * sec_flat_adamw_workspace_bytes(), zeroed before the first call.  max_grad_norm <= 0: no clipping.  Same formulas and operation
 * order as torch.optim.AdamW / clip_grad_norm_. */
size_t sec_flat_adamw_workspace_bytes(void);
int sec_flat_adamw_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, float max_grad_norm, float *state4, float *loss_scale4,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The same step with its hyper-parameters in DEVICE memory: hyper6 = (lr, beta1, beta2, eps, weight_decay, max_grad_norm), read by
 * the kernels at run time -- a step captured in a hipGraph follows a learning-rate schedule (the reference's one-cycle schedule
 * changes lr every step: torchplus/train/learning_schedules_fastai.py via train.py:186-200) by updating six floats between replays. */
int sec_flat_adamw_dev_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, const float *hyper6,
                           float *state4, float *loss_scale4, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SECOND_HIP_H */
