/*
 * ancsh_hip.h -- C ABI of libancsh_hip.so, the MI355X (gfx950) implementation of the ANCSH
 * inference + pose-fit hot path of dragonlong/articulated-pose.
 *
 * Drop-in boundary: these are the entry points a binding of the reference's native launchers
 * would call.  Each declaration cites the reference interface it replaces (paths relative to the
 * reference checkout; ops/ = pointnet_plusplus/utils/tf_ops/).  Conventions shared by all:
 *   - plain C: pointers + sizes only, no torch / TensorFlow types;
 *   - every buffer (inputs, outputs, scratch) is a caller-owned DEVICE pointer, row-major,
 *     float32 / int32; the library allocates nothing and keeps no mutable global state
 *     except a per-thread error string;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are
 *     asynchronous, no hidden synchronisation;
 *   - return 0 on success, ANCSH_EINVAL (-1) on a bad argument (nothing launched),
 *     ANCSH_EHIP (-2) when HIP reports a launch error; ancsh_last_error() describes it.
 *     (The reference launchers return void and never check CUDA errors.)
 *
 * THROUGHPUT NOTE for a host that links this library directly.  The network kernels are throughput-bound, the pose fit is
 * latency-bound (stage B: one of a batch's 12 800 LM fits may run MINPACK's whole 4200-evaluation budget in a single lane, ~1.6 ms,
 * exactly as scipy does): a lone batch of 32 clouds takes ~3.5 ms, while >= 16 batches in flight on SEPARATE HIP streams sustain
 * ~1.57 ms per batch (bench.py keeps 20).  Each such stream needs its own hardware queue: export GPU_MAX_HW_QUEUES=32 BEFORE the
 * first HIP call -- with the runtime's default of 4 queues one batch's LM kernel holds back other batches' kernels queued behind
 * it (measured 3.5 ms per batch at 4 queues against 2.35 at 24 with identical kernels).  The Python package sets the variable on
 * import; a C / C++ host has to do it itself.  Create those streams ONCE and reuse them: a process that has used more than
 * ~20-24 distinct streams over its lifetime runs the same pipeline 5-25 % slower (every stream that has been used keeps a
 * hardware queue; profiles/r04_step_sensitivity.txt).  Nothing in this ABI creates streams or threads.
 */
#ifndef ANCSH_HIP_H
#define ANCSH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ANCSH_OK 0
#define ANCSH_EINVAL (-1)
#define ANCSH_EHIP (-2)

#define ANCSH_ACT_NONE 0
#define ANCSH_ACT_RELU 1
#define ANCSH_MAX_GROUPS 4 /* layers one grouped launch can hold (ancsh_*_grouped) */
#define ANCSH_ACT_RAW 2   /* y = the raw k-ordered accumulator: no bias, no BN (bias/scale/shift may be NULL) */

/* library / diagnostics */
int ancsh_abi_version(void);   /* 7 since round 6 (5: round 5, 4: round 4, 3: round 3).  Operator entry points are only ever added: a library of version v serves every caller
                                 * written for <= v.  The one removal, in 7: ancsh_hbm_copy -- bench.py's HBM-copy yardstick, never an operator -- left the library
                                 * (tools/microbench/membw.hip, its own .so) */
const char *ancsh_last_error(void);

/* ---- PointNet++ set-abstraction / feature-propagation operators -------------------------- */

/* Replaces farthestpointsamplingLauncher(b,n,m,inp,temp,out), ops/sampling/tf_sampling_g.cu:203
 * (op shell ops/sampling/tf_sampling.cpp:95-123).  inp (b,n,3) -> out (b,m) int32; seed index 0;
 * ties resolved as the reference's 512-thread block does (lowest k%512, then lowest k).
 * `temp` is the reference's scratch argument (32*n floats there).  For n <= 8192 it may be NULL: coordinates and
 * running minimum distances live in registers.  n > 8192 takes the large-cloud kernel, which keeps the running
 * minima in `temp` and then REQUIRES it non-NULL with room for b*n floats (EINVAL otherwise). */
int ancsh_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream);

/* Fused variant: also writes new_xyz (b,m,3) = gather_point(inp, out) (pointnet_util.py:47); `temp` as above. */
int ancsh_farthest_point_sample_gather(int b, int n, int m, const float *inp, float *temp, int *out_idx, float *out_xyz,
                                       void *stream);

/* Replaces probsampleLauncher(b,n,m,inp_p,inp_r,temp,out), ops/sampling/tf_sampling_g.cu:196 (op shell tf_sampling.cpp:66-92):
 * inp_p (b,n) non-negative weights, inp_r (b,m) uniform randoms in [0,1) -> out (b,m) int32 sampled indices by inverse CDF;
 * temp (b,n) float32 scratch receives the cumulative sums (same summation tree as the reference's block scan). */
int ancsh_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out, void *stream);

/* Replaces gatherpointLauncher(b,n,m,inp,idx,out), ops/sampling/tf_sampling_g.cu:206. */
int ancsh_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream);

/* Replaces queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt),
 * ops/grouping/tf_grouping_g.cu:125.  xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries ->
 * idx (b,m,nsample), pts_cnt (b,m).  A query with an empty ball gets idx = 0 and pts_cnt = 0
 * (the reference leaves those idx slots uninitialised). */
int ancsh_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                           int *idx, int *pts_cnt, void *stream);

/* Up to four independent ball queries in ONE launch (arrays of length nprob): e.g. both set-abstraction levels of a batch --
 * level 2 only needs the level-1 centroids (ops/grouping/tf_grouping_g.cu:125 twice, pointnet_util.py:48 for layer1 and
 * layer2).  Outputs identical to nprob ancsh_query_ball_point calls. */
int ancsh_query_ball_point_multi(int nprob, const int *b, const int *n, const int *m, const float *radius, const int *nsample,
                                 const float *const *xyz1, const float *const *xyz2, int *const *idx, int *const *pts_cnt,
                                 void *stream);

/* query_ball_point + group_point(xyz1, idx) in one launch -- the first two ops of sample_and_group (pointnet_util.py:47-49):
 * idx / pts_cnt as above, grouped_xyz (b, m, nsample, out_ld >= 3) receives xyz1[idx] in its first three columns, minus the
 * query point when center != 0 (:49 `grouped_xyz -= new_xyz`).  Same results as the two separate ops. */
int ancsh_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2, int center,
                               int *idx, int *pts_cnt, float *grouped_xyz, int out_ld, void *stream);
/* The same for nprob <= 4 independent problems in ONE launch (arrays of length nprob; e.g. both set-abstraction levels of a batch).
 * Outputs identical to nprob ancsh_query_ball_group_xyz calls. */
int ancsh_query_ball_group_xyz_multi(int nprob, const int *b, const int *n, const int *m, const float *radius, const int *nsample,
                                     const float *const *xyz1, const float *const *xyz2, const int *center, int *const *idx,
                                     int *const *pts_cnt, float *const *grouped_xyz, const int *out_ld, void *stream);

/* Replaces groupPointLauncher(b,n,c,m,nsample,points,idx,out), ops/grouping/tf_grouping_g.cu:133.
 * b <= 65535 clouds per call (the cloud is a grid dimension); the same limit holds for the _ex form. */
int ancsh_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                      void *stream);

/* Up to four independent group_point problems (arrays of length nprob) in ONE launch: the grouped xyz of several SA levels
 * (pointnet_util.py:49 for layer1 and layer2) AND feature gathers with 16-byte rows (c % 4 == 0, 16-byte aligned buffers:
 * pointnet_util.py:51); a problem of any other channel count is launched on its own.  Outputs identical to nprob ancsh_group_point
 * calls. */
int ancsh_group_point_multi(int nprob, const int *b, const int *n, const int *c, const int *m, const int *nsample,
                            const float *const *points, const int *const *idx, float *const *out, void *stream);

/* Replaces selectionSortLauncher(b,n,m,k,dist,outi,out), ops/grouping/tf_grouping_g.cu:129 (select_top_k, tf_grouping.py:22):
 * dist (b,m,n) -> outi (b,m,n) int32, out (b,m,n): per row the k smallest in ascending order in the first k columns (ties: lowest
 * index first) and the remaining entries in the order the reference's swaps leave them.  n <= 7680. */
int ancsh_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream);

/* knn_point(k, xyz1, xyz2), tf_grouping.py:48-74, in one launch: xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> val (b,m,k) squared
 * distances ascending, idx (b,m,k) int32 = the first k columns of select_top_k over the pairwise squared distances. */
int ancsh_knn_point(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val, int *idx, void *stream);

/* group_point writing into a wider row: out[b,j,s, out_off : out_off+c] with row stride out_ld
 * floats; if center != NULL (b,m,c) it is subtracted (grouped_xyz -= new_xyz,
 * pointnet_util.py:53) -- fuses group + translation-normalisation + concat (:57). */
int ancsh_group_point_ex(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                         const float *center, float *out, int out_ld, int out_off, void *stream);

/* Replaces threenn_cpu(b,n,m,xyz1,xyz2,dist,idx), ops/3d_interpolation/tf_interpolate.cpp:60
 * (a host function in the reference).  dist = squared distances; m < 3 leaves +inf / index 0. */
int ancsh_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, void *stream);

/* pointnet_util.py:219-222: weight = (1/max(dist,1e-10)) / sum_3(1/max(dist,1e-10)); rows = b*n. */
int ancsh_three_weights(int rows, const float *dist, float *weight, void *stream);

/* ancsh_three_nn + ancsh_three_weights in one launch (same values). */
int ancsh_three_nn_weights(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, float *weight,
                           void *stream);

/* Replaces threeinterpolate_cpu(b,m,c,n,points,idx,weight,out), tf_interpolate.cpp:107. */
int ancsh_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                            float *out, void *stream);

/* three_interpolate writing into out[b,j, out_off : out_off+c] with row stride out_ld (fuses the
 * concat of pointnet_util.py:226). */
int ancsh_three_interpolate_ex(int b, int m, int c, int n, const float *points, const int *idx,
                               const float *weight, float *out, int out_ld, int out_off, void *stream);

/* The input row of pointnet_fp_module in one launch (pointnet_util.py:218-229: three_interpolate, then
 * tf.concat([interpolated_points, points1])): out (b, n, out_ld) = [interpolated points2 (c2) | points1 (b, n, c1) | zeros].
 * Needs c2 % 4 == 0, out_ld % 4 == 0 >= c2 + c1, 16-byte aligned points2 / out; same arithmetic as ancsh_three_interpolate. */
int ancsh_fp_interpolate_concat(int b, int m, int c2, int n, const float *points2, const int *idx, const float *weight,
                                const float *points1, int c1, float *out, int out_ld, void *stream);

/* The same for several networks on the same clouds in one launch: points2 / out hold b clouds (network-major: the networks'
 * copies of cloud c are c, c + geo_batch, ...), the 3-NN arrays idx / weight hold geo_batch clouds and points1 holds
 * points1_batch (= b when every network has its own skip features, = geo_batch when the skip features are the input cloud
 * itself: fa_layer3, pointnet_plusplus/architectures.py:84); b % geo_batch == b % points1_batch == 0. */
int ancsh_fp_interpolate_concat_ex(int b, int m, int c2, int n, const float *points2, const int *idx, const float *weight,
                                   const float *points1, int c1, float *out, int out_ld, int geo_batch, int points1_batch,
                                   void *stream);

/* ---- shared per-point MLP (1x1 conv + bias + inference batch-norm + activation) ---------- */

/* Replaces tf_util.conv1d / conv2d with kernel 1x1 (+ batch_norm_for_conv*d + ReLU) as used by
 * pointnet_sa_module / pointnet_fp_module / the heads: pointnet_plusplus/utils/tf_util.py:52-185,
 * 512-531.  x (rows, cin) with row stride ldx; w (cin, cout) row-major (TF kernel [1,1,cin,cout]);
 * y = act( fma(acc + bias, scale, shift) ), acc = k-ordered f32 FMA chain (v_mfma_f32_32x32x2_f32);
 * scale/shift = folded inference BN (scale = gamma*rsqrt(var+1e-3), shift = beta - mean*scale;
 * pass ones/zeros for no BN).  y (rows, cout) with row stride ldy.
 * pool > 0: fuses tf.reduce_max over each run of `pool` consecutive rows (pointnet_util.py:134);
 * pool must be 64 or 128, rows % pool == 0, y is (rows/pool, cout). */
int ancsh_conv1x1(long rows, int cin, int cout, const float *x, int ldx, const float *w, const float *bias,
                  const float *scale, const float *shift, int act, float *y, int ldy, int pool, void *stream);

/* ancsh_conv1x1 whose accumulators do not start from zero: element (row, col) starts its k-ordered chain from
 * acc_init[row / init_rows][col] (acc_init: (ceil(rows/init_rows), cout) row-major, or NULL = plain ancsh_conv1x1).
 * With act = ANCSH_ACT_RAW on a first call over the leading input channels and acc_init on a second call over the
 * rest, the result is bit-identical to ONE call over the concatenated channels; pointnet_fp_module uses it when its
 * interpolation source is a single point per cloud (pointnet_util.py:218-229: the interpolated block is then the same
 * vector for every point, so its share of the dot product is computed once per cloud). */
int ancsh_conv1x1_ex(long rows, int cin, int cout, const float *x, int ldx, const float *w, const float *bias,
                     const float *scale, const float *shift, int act, float *y, int ldy, int pool, const float *acc_init,
                     int init_rows, void *stream);

/* ancsh_conv1x1_ex for wide layers (cout % 64 == 0) with the kernel in ancsh_sa_pack_weights' fragment order
 * (w_packed = ancsh_sa_pack_weights(cin, cout, w)).  Same results, bit for bit.  Two schedules: layers with 128 / 256 / 259 / 384
 * input channels, cout % 128 == 0 and no pooling take the small-layer schedule (csrc/conv_rowtile.hip: whole 32-row input tile in
 * LDS, any alignment of x); everything else the wave-independent kernel (csrc/conv_packed.hip: no barrier in the k loop), for
 * which x must be 16-byte aligned with ldx % 4 == 0. */
int ancsh_conv1x1_packed(long rows, int cin, int cout, const float *x, int ldx, const float *w_packed, const float *bias,
                         const float *scale, const float *shift, int act, float *y, int ldy, int pool,
                         const float *acc_init, int init_rows, void *stream);

/* GROUPED layer launches: the ANCSH and the NPCS network (main.py --nocs_type=ancsh / npcs) share their backbone's SHAPES
 * (pointnet_plusplus/architectures.py:62-86) and see the same clouds, so the same layer of both is evaluated in ONE launch on
 * stacked activations: group g (g < ngroups <= ANCSH_MAX_GROUPS) works on rows [g * rows, (g + 1) * rows) of x / y with its own
 * parameters w[g], bias[g], scale[g], shift[g] (host arrays of device pointers; bias / scale / shift may be NULL tables with
 * ANCSH_ACT_RAW).  `rows` is PER GROUP; acc_init holds ngroups * (rows / init_rows) rows (rows % init_rows == 0).  Every output
 * is the one the plain call computes, bit for bit.  The small layers these serve are latency-bound, so a second network's rows
 * ride along almost for free (4096 x 256 -> 256: 8.2 us alone, 12.7 us for two).
 * ancsh_conv1x1_packed_grouped: ancsh_conv1x1_packed per group.
 * ancsh_conv1x1_grouped: ancsh_conv1x1 per group (plain [cin][cout] kernels); one launch for the few-rows raw product
 * (rows <= 64 per group, ANCSH_ACT_RAW, no pooling), group by group otherwise. */
int ancsh_conv1x1_packed_grouped(int ngroups, long rows, int cin, int cout, const float *x, int ldx,
                                 const float *const *w_packed, const float *const *bias, const float *const *scale,
                                 const float *const *shift, int act, float *y, int ldy, int pool, const float *acc_init,
                                 int init_rows, void *stream);
int ancsh_conv1x1_grouped(int ngroups, long rows, int cin, int cout, const float *x, int ldx, const float *const *w,
                          const float *const *bias, const float *const *scale, const float *const *shift, int act, float *y,
                          int ldy, int pool, void *stream);

/* Whole body of pointnet_sa_module after sampling (pointnet_util.py:47-57 grouping + concat, :113-134 three shared-MLP
 * layers + max over nsample) in ONE launch; the grouped tensor and the per-layer activations stay in LDS.
 * xyz (b,n,3); new_xyz (b,m,3) and idx (b,m,64) from ancsh_farthest_point_sample_gather / ancsh_query_ball_point;
 * params = 12 device pointers {packed w, bias, scale, shift} for the 3 layers, where "packed w" is the layer's (cin_i, c_i)
 * kernel re-ordered ONCE by ancsh_sa_pack_weights; out (b*m, c3); nsample must be 64.  Results are bit-identical to the
 * unfused kernels (ancsh_group_point_ex + ancsh_conv1x1), which serve every other shape.
 *
 * ancsh_sa_module_fused: a level WITHOUT input features (cfeat must be 0, feats NULL): the ANCSH backbone's layer1, mlp 64,64,128
 * (pointnet_plusplus/architectures.py:62-65).
 *
 * ancsh_sa_module_fused_partial: a level WITH input features: layer2, mlp 128,128,256 (:66-70).  Its first layer's input row is
 * [x_j - c | f_j] (pointnet_util.py:55); the dot product is summed FEATURES FIRST, the three centred coordinates last (TensorFlow
 * does not define a summation order; every path of this library and the CPU oracle use this one).  The feature part does not
 * depend on the centroid, so the caller computes it once per SOURCE POINT --
 *     ancsh_conv1x1(b*n, c, c1, feats, c, w1 + 3*c1, NULL, NULL, NULL, ANCSH_ACT_RAW, partial, c1, 0, stream)
 * (kernel rows 3.. of the first layer) -- and passes `partial` (b,n,c1), 16-byte aligned; params[0] is then the packed kernel
 * rows 0..2 only (ancsh_sa_pack_weights(3, c1, w1, ...)).  The kernel gathers partial rows like feature rows and continues each
 * k-ordered chain with the coordinates: the layer costs n rows of matrix work instead of 64 m.
 *
 * SHAPE LIMITS (hard-coded kernel instantiations; anything else returns ANCSH_EINVAL before a launch and the caller --
 * pointnet_util.pointnet_sa_module -- takes the unfused ancsh_group_point_ex + ancsh_conv1x1 path with identical results):
 *   ancsh_sa_module_fused          cfeat == 0, (c1, c2, c3) == (64, 64, 128),   nsample == 64
 *   ancsh_sa_module_fused_partial  (c1, c2, c3) == (128, 128, 256),             nsample == 64
 * b * m <= 2^31 / 64 rows; any b, n, m otherwise. */
int ancsh_sa_module_fused(int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                          const float *feats, const float *new_xyz, const int *idx, const float *const *params, float *out,
                          void *stream);
int ancsh_sa_module_fused_partial(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                                  const float *partial, const float *new_xyz, const int *idx, const float *const *params,
                                  float *out, void *stream);

/* The same level of `ngroups` networks on the SAME clouds in one launch (see ancsh_conv1x1_packed_grouped): the geometry (xyz,
 * new_xyz, idx) of b clouds is shared, feats / partial / out hold ngroups * b clouds network-major, params holds 12 pointers per
 * network.  Outputs identical to ngroups separate calls. */
int ancsh_sa_module_fused_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                  const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                  const float *const *params, float *out, void *stream);
int ancsh_sa_module_fused_partial_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                          const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                          const float *const *params, float *out, void *stream);

/* EXPERIMENT, opt-in (never the default path; f32 is the arithmetic of record): the same fused level with every f32 product
 * emulated on the bf16 matrix pipe -- activations and weights split exactly into three bf16 terms, six
 * v_mfma_f32_32x32x16_bf16 products per f32 product (csrc/sa_bf16x3.hip).  Each product is reproduced to f32 precision; the ORDER
 * of the additions inside the instruction differs from the k-ordered chain of ancsh_sa_module_fused, so results agree to f32
 * summation noise, not bit for bit.  Shape: cfeat == 0 (feats ignored), mlp (64,64,128), nsample 64.  params = {packed, bias,
 * scale, shift} x 3, all 16-byte aligned, with `packed` from ancsh_sa_pack_weights_bf16x3 (ancsh_sa_packed_weight_bytes_bf16x3(k, n)
 * bytes; n % 32 == 0). */
long ancsh_sa_packed_weight_bytes_bf16x3(int k, int n);
int ancsh_sa_pack_weights_bf16x3(int k, int n, const float *w, void *packed, void *stream);
int ancsh_sa_module_fused_bf16x3(int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                                 const float *feats, const float *new_xyz, const int *idx, const float *const *params, float *out,
                                 void *stream);
/* ... and the counterpart of ancsh_sa_module_fused_partial (mlp 128,128,256): `partial` = the first layer's raw f32 partial sums per
 * source point (as there), params[0] = ancsh_sa_pack_weights_bf16x3(3, c1, kernel rows 0..2); the whole MLP of a neighbourhood
 * stays in one wave's registers (hidden layers computed transposed, fragments re-formed with v_permlane32_swap), no LDS. */
int ancsh_sa_module_fused_partial_bf16x3(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                                         const float *partial, const float *new_xyz, const int *idx, const float *const *params,
                                         float *out, void *stream);
/* ... and both for `ngroups` networks on the SAME clouds in one launch (the counterparts of ancsh_sa_module_fused_grouped /
 * ancsh_sa_module_fused_partial_grouped: shared geometry, partial / out network-major, 12 parameter pointers per network). */
int ancsh_sa_module_fused_bf16x3_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                         const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                         const float *const *params, float *out, void *stream);
int ancsh_sa_module_fused_partial_bf16x3_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                                 const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                                 const float *const *params, float *out, void *stream);
/* The same experiment with the F16x2 split scheme (csrc/bx3.h): x = hi + 2^-11 mid in two f16 terms, THREE v_mfma_f32_32x32x16_f16 products
 * into two accumulators instead of six bf16 products into one -- about 22 significant bits per operand (each product to ~7e-7 relative
 * instead of 6e-8), f16's range (|value| > 65504 -> inf -> the cloud's outputs NaN), half the matrix work.  Same arguments as the _bf16x3
 * entry points; weights packed by ancsh_sa_pack_weights_f16x2 (ancsh_sa_packed_weight_bytes_f16x2 bytes). */
long ancsh_sa_packed_weight_bytes_f16x2(int k, int n);
int ancsh_sa_pack_weights_f16x2(int k, int n, const float *w, void *packed, void *stream);
int ancsh_sa_module_fused_f16x2_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                        const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                        const float *const *params, float *out, void *stream);
int ancsh_sa_module_fused_partial_f16x2_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                                const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                                const float *const *params, float *out, void *stream);

/* Weight layout of ancsh_sa_module_fused: the MFMA B fragments of four consecutive k-steps as one 16-byte load per lane,
 *   packed[((slot*ceil(n/32) + j)*64 + lane)*4 + q] = w[2*(4*slot + q) + (lane >> 5)][j*32 + (lane & 31)]   (0 past row k-1 / column n-1).
 * ancsh_sa_packed_weight_floats(k, n) = number of floats `packed` must hold (-1 when k or n is not positive);
 * ancsh_sa_pack_weights re-orders a (k, n) row-major kernel on the device.  Done once per checkpoint, like the BN fold. */
long ancsh_sa_packed_weight_floats(int k, int n);
int ancsh_sa_pack_weights(int k, int n, const float *w, float *packed, void *stream);

/* A chain of per-point shared-MLP layers in ONE launch (the tail of the ANCSH graph: fa_layer3 convs, fc1, the
 * NOCS heads and the joint heads -- pointnet_plusplus/architectures.py:84-93, lib/architecture.py:105-129,195-206);
 * activations stay in two LDS tiles, only head logits are written.  x (rows, cin <= 131) with row stride ldx is
 * loaded into tile 0; op i computes act(BN(tile[src] . w + b)) with w (k, n), n == 128 or n <= 32, into tile dst
 * (0 or 1; dst == src runs the layer in place) or, when dst == -1, into the global matrix out (rows, n) with row stride out_ld.
 * ops: nops x 6 ints {k, n, act, src, dst, out_ld}; ptrs: nops x 5 device pointers {packed w, bias, scale, shift, out|NULL},
 * packed w = ancsh_sa_pack_weights(k, n, w) (any n: columns are zero-padded to a multiple of 32).
 * Each layer is the same arithmetic as ancsh_conv1x1 (bit-identical results). */
int ancsh_mlp_chain(long rows, int cin, const float *x, int ldx, int nops, const int *ops, const void *const *ptrs,
                    void *stream);

/* The same chain with ONE LDS tile per wave -- two waves per SIMD instead of one -- for ngroups <= 2 networks in one launch
 * (round 4).  Group g reads rows [g * rows, (g + 1) * rows) of x (row stride ldx) and runs ITS program.  Every 128-column layer
 * rewrites the wave's tile in place; a head block (n <= 32) writes (rows, n) to its global matrix `out` with row stride out_ld.
 * ops[g]: nops[g] x 5 ints {k, n, act, flags, out_ld}; ptrs[g]: nops[g] x 5 device pointers {packed w, bias, scale, shift, out | NULL}.
 * flags: 1 = the layer's 128-column output is also copied to this network's rows of `scratch`; 2 = the tile is reloaded from
 * them before the layer runs (the trunk `net` of lib/architecture.py:104 feeds both fc11_1 and fc3_0).  scratch: ngroups * rows * 128
 * floats, 16-byte aligned; NULL when no op carries a flag.  Bit-identical to ancsh_mlp_chain / ancsh_conv1x1. */
int ancsh_mlp_chain_grouped(int ngroups, long rows, int cin, const float *x, int ldx, const int *nops, const int *const *ops,
                            const void *const *const *ptrs, float *scratch, void *stream);

/* The MIDDLE of the backbone as chain launches (round 5, csrc/mid_chain.hip): layer3 (sample_and_group_all + 3 conv2d + reduce_max,
 * pointnet_util.py:66-91,113-134 via pointnet_plusplus/architectures.py:72-75), fa_layer1 and fa_layer2 (three_interpolate + concat +
 * 2 conv2d, pointnet_util.py:206-236 via architectures.py:78-82) of `ngroups` networks on the same b clouds.  A workgroup owns a 32-row
 * tile for a whole level, the layers' columns are split over its waves, activations never leave LDS; every output is bit-identical to
 * the layer-by-layer calls (ancsh_conv1x1_packed / ancsh_fp_interpolate_concat).  "packed w" = ancsh_sa_pack_weights of the layer's
 * (cin, cout) kernel.  Hard-coded instantiations -- the ANCSH backbone widths; any other shape returns ANCSH_EINVAL before a launch:
 *
 * ancsh_sa3_chain_grouped: xyz (b, npts, 3) shared by the networks, feats (ngroups * b, npts, cfeat = 256) network-major, npts % 32 == 0;
 *   params = per network 3 x {packed w, bias, scale, shift} for (3 + cfeat) -> c1 = 256 -> c2 = 512 -> c3 = 1024 (input row = [xyz | feats],
 *   pointnet_util.py:88); out (ngroups * b, npts / 32, c3): the maxima over each 32-row tile -- a cloud's output row is the element-wise
 *   maximum of its npts / 32 rows (taken by ancsh_fp_single_source_init; max is exact in any order).
 * ancsh_fp_single_source_init: the share of an FP module's first layer that comes from a ONE-point interpolation source
 *   (pointnet_util.py:218-224 with m == 1: weights exactly (1, 0, 0), so every point receives the source row): y (ngroups * b, cout) =
 *   the RAW k-ordered chain of max_q x[row][q][0:cin] over the plain kernel rows w[g][0:cin][0:cout]; x (ngroups * b, nparts, cin);
 *   cin % 64 == 0, cout % 128 == 0, x 16-byte aligned.
 * ancsh_fp1_chain_grouped: skip (ngroups * b * npts, cskip = 256), init (ngroups * b, c1) from the call above (row / npts selects it);
 *   params = per network 2 x {packed w, bias, scale, shift}: the first layer's kernel rows [cin_source:] (cskip -> c1 = 256) and the
 *   second layer (c1 -> c2 = 256); out (ngroups * b * npts, c2).
 * ancsh_fp2_chain_grouped: points2 (ngroups * b, m, c2 = 256) interpolation source, idx / weight (b, n, 3) of the shared geometry
 *   (ancsh_three_nn_weights), points1 (ngroups * b, n, c1 = 128) skip features, n % 32 == 0; the interpolation is
 *   p[i1] * w1 + p[i2] * w2 + p[i3] * w3 in that order (tf_interpolate.cpp:107-127); params = per network 2 x {packed w, bias, scale,
 *   shift} for (c2 + c1) -> n1 = 256 -> n2 = 128; out (ngroups * b * n, n2).
 * feats / skip / points2 / points1 and the packed kernels must be 16-byte aligned. */
int ancsh_sa3_chain_grouped(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                            const float *const *params, float *out, void *stream);
int ancsh_fp_single_source_init(int ngroups, int b, int cin, int cout, int nparts, const float *x, const float *const *w, float *y,
                                void *stream);
int ancsh_fp1_chain_grouped(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                            const float *const *params, float *out, void *stream);
int ancsh_fp2_chain_grouped(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                            const float *weight, const float *points1, const float *const *params, float *out, void *stream);

/* Diagnostic: the schedule the last ball-query launch of this process took (-1 wave per two queries = the default, 0..2 the opt-in
 * lane = query kernel selected with ANCSH_BQ_SCHEDULE=lanes<k>, -2 none yet).  A lanes<k> request that exceeds the kernel's LDS / word
 * limits falls back to the default schedule; results are identical either way (tests/test_ops_gpu.py). */
int ancsh_last_ball_query_schedule(void);

/* Round 5: ancsh_mlp_chain_grouped whose input rows are fa_layer3's [three_interpolate(points2) (c2 = 128) | xyz (3)]
 * (pointnet_util.py:218-229 via pointnet_plusplus/architectures.py:84-86) BUILT IN THE TILE LOAD: no (b * n, 132) concat buffer is written
 * or read and the interpolate + concat launch disappears.  points2 (ngroups * b, m, 128) network-major, 16-byte aligned; idx / weight
 * (b, n, 3) from ancsh_three_nn_weights and xyz (b, n, 3) are shared by the networks; n % 128 == 0; rows per network = b * n; nops / ops /
 * ptrs / scratch as ancsh_mlp_chain_grouped (the first op has k = 131).  Interpolation p[i1] * w1 + p[i2] * w2 + p[i3] * w3 in that order
 * (tf_interpolate.cpp:107-127): bit-identical to ancsh_fp_interpolate_concat_ex + ancsh_mlp_chain_grouped. */
int ancsh_mlp_chain_grouped_fp(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx, const float *weight,
                               const float *xyz, const int *nops, const int *const *ops, const void *const *const *ptrs, float *scratch,
                               void *stream);

/* EXPERIMENT, opt-in (like ancsh_sa_module_fused_bf16x3; f32 is the arithmetic of record): ancsh_mlp_chain_grouped_fp with every f32
 * product emulated by six bf16 MFMA products (csrc/tail_bf16x3.hip).  A wave owns 64 points and keeps their 128-channel activations in
 * registers as bf16x3 fragments (two register tiles: no save / restore of the trunk); n % 64 == 0.  ops[g]: nops[g] x 5 ints {k, n, act,
 * 0, out_ld}; ptrs[g]: nops[g] x 5 pointers {w packed by ancsh_sa_pack_weights_bf16x3, bias, scale, shift, out | NULL}, all 16-byte
 * aligned; a head block (out != NULL) is 128 -> n <= 32 with w packed as (128, 32) and bias / scale / shift holding 32 entries.  The op
 * list must have the shape  F H H H head+ [L head+] H H head+  (F = 131 -> 128 ReLU, H = 128 -> 128 ReLU, L = 128 -> 128 linear): the tail
 * of lib/architecture.py:98-139,195-208 with and without early_split_nocs. */
int ancsh_mlp_chain_grouped_fp_bf16x3(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx, const float *weight,
                                      const float *xyz, const int *nops, const int *const *ops, const void *const *const *ptrs,
                                      void *stream);
int ancsh_mlp_chain_grouped_fp_f16x2(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx, const float *weight,
                                     const float *xyz, const int *nops, const int *const *ops, const void *const *const *ptrs,
                                     void *stream);      /* the F16x2 scheme; weights packed by ancsh_sa_pack_weights_f16x2 */

/* ... and the MIDDLE of the backbone (ancsh_sa3_chain_grouped / ancsh_fp1_chain_grouped / ancsh_fp2_chain_grouped) on the 16-bit matrix pipe
 * (csrc/mid_bf16x3.hip): the activations of a 64-row tile live in LDS as the scheme's 16-bit planes, the workgroup's waves split every layer
 * by output channels.  Same arguments as the f32 forms with kernels packed by ancsh_sa_pack_weights_{bf16x3,f16x2} and 16-byte aligned bias /
 * scale / shift; npts (n) % 64 == 0, and layer3's out is (ngroups * b, npts / 64, 1024): the maxima of every 64-ROW tile (pass nparts =
 * npts / 64 to ancsh_fp_single_source_init).  ancsh_sa3_chain_grouped_bf16x3 returns an error: three bf16 planes of a 64 x 512 tile exceed the
 * 160 KB of LDS (layer3 then takes the f32 chain). */
int ancsh_sa3_chain_grouped_bf16x3(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                                   const float *const *params, float *out, void *stream);
int ancsh_sa3_chain_grouped_f16x2(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                                  const float *const *params, float *out, void *stream);
int ancsh_fp1_chain_grouped_bf16x3(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                                   const float *const *params, float *out, void *stream);
int ancsh_fp1_chain_grouped_f16x2(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                                  const float *const *params, float *out, void *stream);
int ancsh_fp2_chain_grouped_bf16x3(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                                   const float *weight, const float *points1, const float *const *params, float *out, void *stream);
int ancsh_fp2_chain_grouped_f16x2(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                                  const float *weight, const float *points1, const float *const *params, float *out, void *stream);

/* tf.reduce_max over nsample (pointnet_util.py:134): x (groups, nsample, c) -> y (groups, c). */
int ancsh_group_max(long groups, int nsample, int c, const float *x, float *y, void *stream);

/* ANCSH head activations + gocs composition, lib/architecture.py:124-159.
 *  logits (rows, ld) holds, per point, the raw head outputs laid out as
 *    [W(K) | nocs(3K) | scale(K) | trans(3K) | confi(1) | axis(3) | unitvec(3) | heatmap(1) | joint_cls(3)]
 *  (mixed_pred = 1, ANCSH) or [W(K) | nocs(3K) | confi(1) | axis(3) | unitvec(3) | heatmap(1) | joint_cls(3)]
 *  (mixed_pred = 0, NPCS).  Outputs (any may be NULL): W softmax (rows,K); nocs sigmoid (rows,3K);
 *  confi sigmoid (rows,1); heatmap sigmoid (rows,1); unitvec tanh (rows,3); axis tanh (rows,3);
 *  joint_cls softmax (rows,3); gocs = nocs*repeat(scale,3)+trans (rows,3K); scale sigmoid (rows,K);
 *  trans tanh (rows,3K). */
int ancsh_head_activations(long rows, int K, int mixed_pred, const float *logits, int ld, float *W, float *nocs,
                           float *confi, float *heatmap, float *unitvec, float *axis, float *joint_cls,
                           float *gocs, float *scale, float *trans, void *stream);

/* ---- pose fitting (host Python + numpy/scipy in the reference; device kernels here) ----------- */

/* Front end of the per-cloud loop body of solver_ransac_nonlinear (evaluation/parallel_ancsh_pose.py
 * :238-242, 259-260): labels = argmax(W, axis=1) (first maximum wins); the cloud's points are
 * partitioned by label in ascending point order, and the per-part (source, target) arrays
 *   source = nocs[idx, 3j:3j+3] (the point's own part-NOCS slot), target = P[idx, :3]
 * are written packed: rows [off[b*K+j], off[b*K+j+1]) of src/tgt (b*n rows in total, off has b*K+1
 * entries).  W (b,n,K), P (b,n,3), nocs (b,n,3K); labels (b,n) may be NULL; part_index (b,n) holds
 * the original point id of every packed row.  Optional by-products (NULL to skip): counts (b,K) points per part; rng0 / rng1
 * (b*(K-1), 2) = the [start,end) packed-row ranges of part 0 and of part j for every joint j = 1..K-1, i.e. the arguments
 * ancsh_ransac_joint takes (:274-283). */
int ancsh_pose_partition(int b, int n, int K, const float *W, const float *P, const float *nocs, int *labels,
                         int *part_index, int *off, float *src, float *tgt, int *counts, int *rng0, int *rng1, void *stream);

/* Non-finite inputs of the fit: a cloud whose P (b,n,3), nocs (b,n,3K), W (b,n,K) or joint_axis (b,n,3; may be NULL) holds a NaN or
 * +-Inf has no defined pose (in the reference np.linalg.svd raises LinAlgError on such a part, lib/d3_utils.py:214; np.argmax /
 * np.median over NaN rows are arbitrary): every value of its record (b, K, 26) float64 rows is set to NaN.  Call after the fit
 * kernels that write `record`; clouds without a non-finite value are not touched. */
int ancsh_pose_poison_records(int b, int n, int K, const float *P, const float *nocs, const float *W, const float *joint_axis,
                              double *record, void *stream);

/* jt_axis = np.median(joint_axis_per_point[joint_cls == j], 0), j = 1..K-1 (:295).
 * joint_axis (b,n,3), joint_cls (b,n) int32 -> out (b, K-1, 3) float32 (NaN for an empty selection). */
int ancsh_pose_joint_direction(int b, int n, int K, const float *joint_axis, const int *joint_cls, float *out,
                               void *stream);

/* Batched replacement of ransac(dataset, single_transformation_estimator, single_transformation_verifier,
 * inlier_th, niter) (:20-54, with transform_pts / rotate_pts / scale_pts of lib/d3_utils.py:206-246).
 * Problem p = rows [off[p], off[p+1]) of src/tgt (float32 (rows,3)).  draws (nprob, niter, 3) int32 =
 * the 3-point sample of every iteration (the reference draws np.random.randint(n, size=3) from the global
 * RNG, :38); NULL -> on-device counter-based generator keyed by `seed`.  max_n >= every problem size.
 * out_model (nprob,13) float64: R row-major (9), scale, translation (3) of the full-inlier refit;
 * out_inliers (rows) uint8 mask of the winning hypothesis; out_best (nprob,2): winning iteration (ties ->
 * earliest) and its inlier count.  scratch_scores: nprob*niter int32.  An empty problem yields NaNs and
 * out_best = (-1, 0) (the reference raises). */
int ancsh_ransac_single(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                        const int *draws, unsigned long long seed, int max_n, double *out_model,
                        unsigned char *out_inliers, int *out_best, int *scratch_scores, void *stream);

/* ancsh_ransac_single with the hypotheses scored from SCALAR registers.  Every lane of a wave tests the same point (lane =
 * hypothesis), so the points are wave-uniform operands: a padded copy of each part, four points per 96-byte record
 * {x[4] y[4] z[4]} source | {x[4] y[4] z[4]} target (scratch_quads, written by the call itself), is read with s_load_dwordx8/x16
 * and enters the packed-f32 residual arithmetic as the instruction's scalar source -- no LDS, no barrier, no staging pass per
 * workgroup.  scratch_quads: 32-byte aligned, ancsh_ransac_single_quads_floats(rows, nprob) floats where rows >= off[nprob]
 * (the row capacity of src/tgt; the kernels never write or read past that capacity even if off[] exceeds it).  Scores, winner,
 * inlier mask and model are those of ancsh_ransac_single bit for bit (evaluation/parallel_ancsh_pose.py:20-54, same replacement). */
long ancsh_ransac_single_quads_floats(long rows, int nprob);
int ancsh_ransac_single_ex(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                           const int *draws, unsigned long long seed, int max_n, double *out_model,
                           unsigned char *out_inliers, int *out_best, int *scratch_scores, float *scratch_quads, long rows,
                           void *stream);

/* Round 5: ancsh_ransac_single_ex (scratch_quads != NULL) / ancsh_ransac_single (NULL) with two optional extra outputs; scores,
 * winner, mask and out_model bit for bit as before.
 *   record (B, K, 26) float64, B = nprob / K: the per-part pose record the reference assembles in Python
 *     (evaluation/parallel_ancsh_pose.py:330-353), row (b, j) = [baseline R(9) s t(3) | nonlinear R(9) s t(3)]: this call fills columns
 *     0..12 of row p (and 13..25 when K == 1), ancsh_ransac_joint_rec columns 13..25 -- no assembly launches (torch.cat / clone) in
 *     the captured step.  NULL: not written.
 *   tie_stats (nprob, 2) int32 -- how implementation-sensitive the fit is:
 *     [0] points of the part whose residual norm under the WINNING hypothesis lies in [inlier_th - tie_window, inlier_th + tie_window)
 *         -- the verifier's `sqrt(sum(res**2)) < th` (:48-54) is a float32 threshold test, and an implementation whose 3-point model
 *         differs in the last bits (LAPACK's SVD there, Horn's quaternion here) may count such a point on the other side;
 *     [1] DEGENERATE CONTENDERS THAT WOULD CHANGE THE CONSENSUS SET: hypotheses whose score is within one inlier of the winning score,
 *         whose 3-point sample repeats an index (np.random.randint draws WITH replacement, :38) and which -- had they won -- would have
 *         handed the refit another inlier mask than the winner's (the winner itself counts when its own sample is degenerate).  Until
 *         round 5 every degenerate contender was counted (20-25 % of the fits at N = 1024: nearly all of them hypotheses with the
 *         winner's own mask, which cannot change the result whoever scores them).  The count is NEGATIVE when the winner's own sample is
 *         degenerate -- the one case in which the fit's consensus set is implementation-defined for certain (measured at 10000 / 200 on
 *         624 fits, profiles/r06_pose_tie_rate_K3.txt: 3 such fits, all 3 on another set than the reference arithmetic; a positive count
 *         fired on 22.9 % of the fits and held 1 of the other 4 flips: as a per-fit warning only the sign is sharp).  The centred points of such a sample are
 *         collinear, the 3 x 3 covariance has rank 1, and the rotation the reference takes from np.linalg.svd (lib/d3_utils.py:214) is
 *         LAPACK's completion of a null space that rounding noise selects -- implementation-defined in the reference itself.
 *     Measured at the reference's budgets on 1344 clouds, 8064 reported fits (profiles/r05_pose_tie_rate_full.txt): [0] was 0 in EVERY fit
 *     -- no threshold-tie flips at 10000 / 200 --; 32 fits (0.40 %) ended on another consensus set than the reference arithmetic, ALL
 *     with a repeated-index winner on one side; [1] > 0 in 22 of those 32 (and in 20-25 % / 5-10 % of all fits at N = 1024 / 2048) -- it
 *     sees the degenerate contenders of THIS implementation's arithmetic, while the other 10 had a degenerate winner only under LAPACK's
 *     choice of the free rotation about the sample's line, which no other implementation can score.
 *     NULL: not computed.  tie_window: absolute half-width on the norm, 0 <= tie_window < inlier_th. */
int ancsh_ransac_single_rec(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                            const int *draws, unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                            int *out_best, int *scratch_scores, float *scratch_quads, long rows, double *record, int K,
                            int *tie_stats, float tie_window, void *stream);

/* Batched replacement of ransac(dataset, joint_transformation_estimator, joint_transformation_verifier,
 * inlier_th, niter) (:20-33, 106-194; revolute objective :56-68; scipy least_squares(method='lm',
 * ftol=1e-4) = MINPACK lmdif restated on the 6x6 normal equations).  Problem p couples part 0
 * (rows [rng0[2p], rng0[2p+1])) and part j (rows [rng1[2p], rng1[2p+1])) of src/tgt through the joint
 * direction joint_dir[p] (3).  draws (nprob, niter, 6) int32 (3 samples of part 0, then 3 of part j) or
 * NULL.  out_model (nprob,26) float64: R0(9) s0 t0(3) R1(9) s1 t1(3) of the all-inlier refit;
 * out_inliers (nprob, 2, max_n) uint8; out_best (nprob) winning iteration; out_score (nprob) its score
 * ((n_inl0/3 + n_inl1/3)/2 as in :192).  scratch_scores nprob*niter float64, scratch_models
 * nprob*niter*26 float64, lm_stat (nprob,niter,2) int32 (MINPACK info, nfev) or NULL. */
int ancsh_ransac_joint(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                       const float *joint_dir, double inlier_th, int niter, const int *draws,
                       unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                       int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                       int *lm_stat, void *stream);

/* The same with an explicit schedule for the per-hypothesis LM fits.  Both schedules run the same MINPACK state machine; their
 * floating-point results agree to ~1e-7 (the eight-lane schedule recombines partial sums through a different instruction stream),
 * far inside the 1e-4 parity bar, and pick the same winning hypotheses on every test -- but NOT bit for bit.  The schedule is
 * therefore never chosen from the launch size: a cloud solved alone and the same cloud inside a batch of 32 give identical bytes
 * (tests/test_pose_gpu.py::test_stage_b_is_batch_size_invariant).
 *   ANCSH_LM_AUTO       = ANCSH_LM_THROUGHPUT (round 4; until round 3 launches of <= 2048 fits took the eight-lane schedule);
 *   ANCSH_LM_THROUGHPUT one lane per fit: least SIMD time; a small launch hands fewer hypotheses to each wave (scheduling only);
 *   ANCSH_LM_LATENCY    eight lanes per fit: the launch's long fits finish ~1.25x sooner, ~5 % less pipeline throughput.
 *                       Explicit only (AncshPipeline selects it for <= 2 batches in flight, a latency deployment). */
#define ANCSH_LM_AUTO 0
#define ANCSH_LM_THROUGHPUT 1
#define ANCSH_LM_LATENCY 2
int ancsh_ransac_joint_ex(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                          const float *joint_dir, double inlier_th, int niter, const int *draws,
                          unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                          int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                          int *lm_stat, int lm_schedule, void *stream);

/* ancsh_ransac_joint_ex with the same two optional outputs (see ancsh_ransac_single_rec): problem p = b * (K - 1) + q writes columns
 * 13..25 of record row (b, q + 1) and -- q == 0 -- of row (b, 0) (part 0 is reported from joint 1's fit, :327-329); tie_stats
 * (nprob, 2): [0] points of BOTH parts within tie_window of the threshold under the winning hypothesis' model (float64 test,
 * :186-194), [1] degenerate contenders: hypotheses within one inlier (1/6 of the joint score, :192) of the winning score with a
 * repeated index in either 3-point sample (:110-111). */
int ancsh_ransac_joint_rec(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                           const float *joint_dir, double inlier_th, int niter, const int *draws, unsigned long long seed,
                           int max_n, double *out_model, unsigned char *out_inliers, int *out_best, double *out_score,
                           double *scratch_scores, double *scratch_models, int *lm_stat, int lm_schedule, double *record, int K,
                           int *tie_stats, double tie_window, void *stream);

/* Batched estimateSimilarityUmeyama (lib/aligning.py:580-622; GT poses of evaluation/compute_gt_pose.py:87).
 * Problem p = rows [off[p], off[p+1]) of src/tgt.  out (nprob,32) float64: Scales(3) | Rotation(9, the
 * reference's TRANSPOSED matrix) | Translation(3) | OutTransform (4x4 row-major, 16) | pad(1). */
int ancsh_umeyama(int nprob, const int *off, const float *src, const float *tgt, double *out, void *stream);

/* Batched estimateSimilarityTransform (lib/aligning.py:17-32 with set_config :88-103, getRANSACInliers :485-507,
 * evaluateModel :540-547): 5-point Umeyama RANSAC, niter (reference: 100, max 128) iterations with the reference's
 * sequential bookkeeping (strictly better inlier ratio wins; stop once the best residual < PassThreshold/100; the
 * count_nonzero-of-indices quirk that never counts point 0), then Umeyama on the winner's inliers.
 * draws (nprob, niter, 5) int32 (the reference draws np.random.randint(n, size=5), :490) or NULL (device generator).
 * out (nprob,32) float64 laid out as ancsh_umeyama's, out[31] = BestInlierRatio; status (nprob): 0 ok, 1 = the reference
 * returns 4 x None (BestInlierRatio < 0.1; out row is NaN), -1 empty problem. */
int ancsh_estimate_similarity_transform(int nprob, const int *off, const float *src, const float *tgt, int niter,
                                        const int *draws, unsigned long long seed, double *out, int *status, void *stream);

/* iou_3d of lib/d3_utils.py:55-69 (per-part amodal box IoU of evaluation/compute_miou.py:212-225) for npairs box pairs in one
 * launch: bbox1, bbox2 (npairs, 8, 3) float64 corners in the reference's get_3d_bbox order; a nres^3 numpy.linspace grid over
 * the joint axis-aligned bounds; iou[p] = |inside both| / |inside either| (1.0 when the union is empty).  counts (npairs, 2)
 * int64 {intersection, union} is optional (NULL to skip). */
int ancsh_iou_3d(int npairs, int nres, const double *bbox1, const double *bbox2, double *iou, long *counts, void *stream);

/* Joint parameters from the per-point heads, evaluation/eval_joint_params.py:143-199, for b clouds in one launch (float32
 * inputs with the .h5 record's layouts, (b, n, .) row-major):
 *   gocs (b,n,gocs_channels) global NOCS, gocs_channels = 3*K (per part: a point reads its predicted part's triple) or 3;
 *   nocs (b,n,3K) part NOCS; mask (b,n,K) part scores (np.argmax = first maximum); heatmap (b,n); unitvec, joint_axis (b,n,3);
 *   joint_cls (b,n) int32 joint class per point (argmax of index_per_point / joint_cls_gt).
 * st (b,K,4) float64: scale_j = std(mean(y,1)) / std(mean(x,1)) and translation_j = mean(y - scale_j x, 0) over the points of
 *   predicted part j (x global, y part NOCS; :160-171), NaN for an empty part; nocs and st may BOTH be NULL to skip.
 * joint (b,K-1,6) float64: [joint point (3) | joint axis (3)] of joint j = 1..K-1 in global-NOCS space: per-channel median over
 *   the points of joint class j of  gocs + unitvec * (1 - heatmap) * 0.2  and of joint_axis (:176-187); axis_mean != 0 = the
 *   ground-truth variant (:189-199): the axis is the float32 mean instead of the median.  mask may be NULL when gocs_channels == 3.
 * Element arithmetic in float32 in numpy's order; medians are exact selections; NaN rows for a joint class without points. */
int ancsh_joint_params(int b, int n, int K, int gocs_channels, int axis_mean, const float *gocs, const float *nocs,
                       const float *mask, const float *heatmap, const float *unitvec, const float *joint_axis,
                       const int *joint_cls, double *st, double *joint, void *stream);

/* Amodal-box extents and boundaries of the predicted parts: the per-part block evaluation/compute_miou.py:196-208 and
 * evaluation/eval_pose_err.py:253-268 run one frame and one part at a time.  nocs (b,n,nocs_channels) float32 with 3K channels
 * (part j reads its own slot) or 3; mask (b,n,K) float32 (a point belongs to the part of its FIRST maximum, np.argmax); P rows of ldp
 * floats whose first three are the point; pose0 (b,12) float64 = part 0's fitted rotation row-major (9) and translation (3), rounded
 * to float32 inside like the reference's compose_rt.  Outputs per (cloud, part): scale_pred (b,K,3) float32 = 2 * max |nocs - 0.5|,
 * dynam (b,K) float64 = min x of the part's points in part 0's frame (the "dynamic boundary"), count (b,K) points of the part
 * (0 -> NaN outputs; the reference drops such a frame through its bare except).  K <= 8. */
int ancsh_part_extents(int b, int n, int K, int nocs_channels, const float *nocs, const float *mask, const float *P, int ldp,
                       const double *pose0, float *scale_pred, double *dynam, int *count, void *stream);

/* ---- input sampling in front of the network (lib/dataset.py:290-357) ------------------------ */

/* One launch for a ragged batch: cloud b owns raw rows [offsets[b], offsets[b+1]) of `rows` (nchan floats each:
 * x y z then per-point channels).  Sampled row i of cloud b = raw row perm[b*num_points+i] % n_raw (the reference's
 * tiling, :290-317, is this modulo), giving P (b,N,3) = xyz * norm_factor[b] (:346), chan_out (b,N,nchan-3) = the other
 * channels, mask_array (b,N,n_parts) = one-hot of int8(row[cls_col]) with numpy's negative-index rule (:357) and
 * joint_cls_mask (b,N) = row[jcls_col] > 0 (:353-355; jcls_col < 0: zeros). */
int ancsh_input_sample(int nclouds, int num_points, int nchan, const float *rows, const int *offsets, const int *perm,
                       const float *norm_factor, int cls_col, int jcls_col, int n_parts, float *P, float *chan_out,
                       float *mask_array, float *joint_cls_mask, void *stream);

/* ---- test-time losses of predict_and_save (lib/network.py:430-498, lib/loss.py:54-182) ------- */

/* One launch per batch; ptrs = 16 device pointers {W (b,n,K), nocs (b,n,3K), gocs (b,n,3K)|NULL, heatmap (b,n), unitvec (b,n,3),
 * joint_axis (b,n,3), index (b,n,3), cls_gt (b,n) int32 (-1 allowed), joint_cls_gt (b,n) int32, nocs_gt (b,n,3),
 * gocs_gt (b,n,3)|NULL, mask_array (b,n,K), heatmap_gt (b,n), unitvec_gt (b,n,3), orient_gt (b,n,3), joint_cls_mask (b,n)}.
 * out (b, 5+K+3) = [nocs_loss, gocs_loss, heatmap_loss, unitvec_loss, orient_loss | miou_loss[K] | index_loss[3]] per cloud,
 * i.e. loss_dict of compute_loss before collect_losses' batch means.  type_l: 0 = 'L2', 1 = 'L1' (cfg coord_regress_loss). */
int ancsh_test_losses(int b, int n, int K, int type_l, const void *const *ptrs, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANCSH_HIP_H */
