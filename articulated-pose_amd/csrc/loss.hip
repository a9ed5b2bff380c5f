// loss.hip -- the test-time losses written to test_loss.txt, one launch per batch (gfx950).
//
// Reference: lib/network.py:430-498 (compute_loss, is_eval = False) over lib/loss.py:54-102 (compute_nocs_loss,
// MULTI_HEAD), :104-166 (compute_vect_loss with confidence = joint_cls_mask) and :169-182 (compute_miou_loss): ~60 TF
// element-wise / reduction ops per batch.  Here ONE workgroup per cloud streams its points once (every prediction and
// ground-truth row is read exactly once: HBM-bound, ~ (22 + 8K) floats per point) and reduces all 5 + K + 3 sums
// together: per-thread partial sums in registers, wave reduction by DPP, 4 waves combined through LDS.
//   out[b] = [ nocs, gocs, heatmap, unitvec, orient | miou_loss[0..K) | index_miou_loss[0..3) ]
//   nocs    = sum_i mean_n( mask[n,i] * d(nocs[n,3i:3i+3], nocs_gt[n]) )          d = L2 norm (type_l 0) or L1 norm (1)
//   heatmap = mean_n( |h[n] - h_gt[n]| * jmask[n] );  unitvec / orient = mean_n( d(v[n], v_gt[n]) * jmask[n] )
//   miou    = 1 - dot_k / (cnt_k + sumW_k - dot_k + 1e-10),  dot_k = sum_n [cls_gt[n] == k] W[n,k]   (one_hot(-1) = zero row)
// Sums are accumulated in float64 (TensorFlow reduces in float32 in an unspecified order; the difference is < 1e-6 relative).
#include "common.h"

namespace ancsh {

constexpr int LOSS_MAX_K = 8;
constexpr int LOSS_NACC = 5 + 3 * LOSS_MAX_K + 9;      // 5 scalars, (dot, sumW, cnt) x K, (dot, sumW, cnt) x 3

struct LossArgs {
    const float *W, *nocs, *gocs, *heatmap, *unitvec, *axis, *index;
    const int *cls_gt, *jcls_gt;
    const float *nocs_gt, *gocs_gt, *mask, *heatmap_gt, *unitvec_gt, *orient_gt, *jmask;
};

__device__ __forceinline__ float dist3(const float *a, const float *b, int type_l) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return type_l == 0 ? sqrtf(dx * dx + dy * dy + dz * dz) : fabsf(dx) + fabsf(dy) + fabsf(dz);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void test_losses_kernel(int n, int K, int type_l, LossArgs a, float *__restrict__ out) {
    __shared__ double red[4][LOSS_NACC];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t p0 = (size_t)b * n;
    double acc[LOSS_NACC];
#pragma unroll
    for (int i = 0; i < LOSS_NACC; ++i) acc[i] = 0.0;
    for (int i = tid; i < n; i += 256) {
        const size_t p = p0 + i;
        const float *ng = a.nocs_gt + p * 3;
        float s_nocs = 0.f, s_gocs = 0.f;
        const int lab = a.cls_gt[p];
#pragma unroll
        for (int k = 0; k < LOSS_MAX_K; ++k) {
            if (k < K) {
                const float m = a.mask[p * K + k];
                s_nocs += m * dist3(a.nocs + p * 3 * K + 3 * k, ng, type_l);
                if (a.gocs) s_gocs += m * dist3(a.gocs + p * 3 * K + 3 * k, a.gocs_gt + p * 3, type_l);
                const float w = a.W[p * K + k];
                acc[5 + 3 * k] += lab == k ? (double)w : 0.0;
                acc[5 + 3 * k + 1] += (double)w;
                acc[5 + 3 * k + 2] += lab == k ? 1.0 : 0.0;
            }
        }
        const float jm = a.jmask[p];
        acc[0] += (double)s_nocs;
        acc[1] += (double)s_gocs;
        acc[2] += (double)(fabsf(a.heatmap[p] - a.heatmap_gt[p]) * jm);
        acc[3] += (double)(dist3(a.unitvec + p * 3, a.unitvec_gt + p * 3, type_l) * jm);
        acc[4] += (double)(dist3(a.axis + p * 3, a.orient_gt + p * 3, type_l) * jm);
        const int jl = a.jcls_gt[p];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = a.index[p * 3 + k];
            acc[5 + 3 * LOSS_MAX_K + 3 * k] += jl == k ? (double)w : 0.0;
            acc[5 + 3 * LOSS_MAX_K + 3 * k + 1] += (double)w;
            acc[5 + 3 * LOSS_MAX_K + 3 * k + 2] += jl == k ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < LOSS_NACC; ++i) {
        const double s = wave_sum_f64(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    const int width = 5 + K + 3;
    if (tid < width) {
        auto total = [&](int i) { return red[0][i] + red[1][i] + red[2][i] + red[3][i]; };
        double v;
        if (tid < 5) {
            v = total(tid) / (double)n;                        // reduce_mean over the points
        } else {
            const int base = tid < 5 + K ? 5 + 3 * (tid - 5) : 5 + 3 * LOSS_MAX_K + 3 * (tid - 5 - K);
            const double dot = total(base), sw = total(base + 1), cnt = total(base + 2);
            v = 1.0 - dot / (cnt + sw - dot + 1e-10);          // DIVISION_EPS, lib/constants.py:1
        }
        out[(size_t)b * width + tid] = (float)v;
    }
}

}  // namespace ancsh

using namespace ancsh;

// ptrs: 16 device pointers in LossArgs order {W, nocs, gocs|NULL, heatmap, unitvec, joint_axis, index, cls_gt(i32), joint_cls_gt(i32),
// nocs_gt, gocs_gt|NULL, mask_array, heatmap_gt, unitvec_gt, orient_gt, joint_cls_mask}; out (b, 5 + K + 3) float32.
extern "C" int ancsh_test_losses(int b, int n, int K, int type_l, const void *const *ptrs, float *out, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0, "test_losses: bad shape b=%d n=%d", b, n);
    ANCSH_REQUIRE(K >= 1 && K <= LOSS_MAX_K, "test_losses: n_max_parts=%d must be in [1,%d]", K, LOSS_MAX_K);
    ANCSH_REQUIRE(type_l == 0 || type_l == 1, "test_losses: coord_regress_loss must be 0 (L2) or 1 (L1); Soft_L1 is not on the test path");
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(ptrs && out, "test_losses: null pointer");
    for (int i = 0; i < 16; ++i) ANCSH_REQUIRE(ptrs[i] || i == 2 || i == 10, "test_losses: null tensor pointer #%d", i);
    ANCSH_REQUIRE((ptrs[2] == nullptr) == (ptrs[10] == nullptr), "test_losses: gocs prediction and ground truth go together");
    LossArgs a;
    a.W = (const float *)ptrs[0]; a.nocs = (const float *)ptrs[1]; a.gocs = (const float *)ptrs[2]; a.heatmap = (const float *)ptrs[3];
    a.unitvec = (const float *)ptrs[4]; a.axis = (const float *)ptrs[5]; a.index = (const float *)ptrs[6];
    a.cls_gt = (const int *)ptrs[7]; a.jcls_gt = (const int *)ptrs[8];
    a.nocs_gt = (const float *)ptrs[9]; a.gocs_gt = (const float *)ptrs[10]; a.mask = (const float *)ptrs[11];
    a.heatmap_gt = (const float *)ptrs[12]; a.unitvec_gt = (const float *)ptrs[13]; a.orient_gt = (const float *)ptrs[14];
    a.jmask = (const float *)ptrs[15];
    hipLaunchKernelGGL(test_losses_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, K, type_l, a, out);
    return check_launch("test_losses");
}
