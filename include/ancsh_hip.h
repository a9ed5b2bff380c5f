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

/* library / diagnostics */
int ancsh_abi_version(void);
const char *ancsh_last_error(void);

/* ---- PointNet++ set-abstraction / feature-propagation operators -------------------------- */

/* Replaces farthestpointsamplingLauncher(b,n,m,inp,temp,out), ops/sampling/tf_sampling_g.cu:203
 * (op shell ops/sampling/tf_sampling.cpp:95-123).  inp (b,n,3) -> out (b,m) int32; seed index 0;
 * ties resolved as the reference's 512-thread block does (lowest k%512, then lowest k).
 * `temp` is the reference's 32*n-float scratch: accepted for signature parity, may be NULL
 * (running minimum distances live in registers); used only by the n > 8192 fallback
 * (which needs b*n floats). */
int ancsh_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream);

/* Fused variant: also writes new_xyz (b,m,3) = gather_point(inp, out) (pointnet_util.py:47). */
int ancsh_farthest_point_sample_gather(int b, int n, int m, const float *inp, int *out_idx, float *out_xyz,
                                       void *stream);

/* Replaces gatherpointLauncher(b,n,m,inp,idx,out), ops/sampling/tf_sampling_g.cu:206. */
int ancsh_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream);

/* Replaces queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt),
 * ops/grouping/tf_grouping_g.cu:125.  xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries ->
 * idx (b,m,nsample), pts_cnt (b,m).  A query with an empty ball gets idx = 0 and pts_cnt = 0
 * (the reference leaves those idx slots uninitialised). */
int ancsh_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                           int *idx, int *pts_cnt, void *stream);

/* Replaces groupPointLauncher(b,n,c,m,nsample,points,idx,out), ops/grouping/tf_grouping_g.cu:133. */
int ancsh_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                      void *stream);

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

/* Replaces threeinterpolate_cpu(b,m,c,n,points,idx,weight,out), tf_interpolate.cpp:107. */
int ancsh_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                            float *out, void *stream);

/* three_interpolate writing into out[b,j, out_off : out_off+c] with row stride out_ld (fuses the
 * concat of pointnet_util.py:226). */
int ancsh_three_interpolate_ex(int b, int m, int c, int n, const float *points, const int *idx,
                               const float *weight, float *out, int out_ld, int out_off, void *stream);

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

#ifdef __cplusplus
}
#endif
#endif /* ANCSH_HIP_H */
