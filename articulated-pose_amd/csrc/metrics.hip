// metrics.hip -- evaluation-side kernels.
//
// iou_3d (lib/d3_utils.py:40-69, called per part by evaluation/compute_miou.py:212-225): two oriented boxes (8 corners each),
// a nres^3 grid over their joint axis-aligned bounds, an inside-box test per grid point and box, IoU = |both| / |either|
// (1 when the union is empty).  The reference builds the 125 000 grid points with itertools.product and tests them with
// numpy, ~25 ms per pair; here ONE WORKGROUP per pair strides over the grid in registers (no point is ever materialised),
// counts by ballot + popcount and reduces 4 waves through LDS.  float64 like the reference: grid coordinates are
// numpy.linspace's (start + i*step, the last one exactly stop), the projections up.u are sums of three products in
// (x, y, z) order and the bounds np.dot(u, u).
#include "common.h"

namespace ancsh {

struct BoxFrame {
    double o[3], u1[3], u2[3], u3[3], d1, d2, d3;
};

__device__ __forceinline__ void box_frame(const double *bb, BoxFrame &f) {      // bb: 8 x 3 corners, reference's order
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f.o[c] = bb[4 * 3 + c];
        f.u1[c] = bb[5 * 3 + c] - bb[4 * 3 + c];
        f.u2[c] = bb[7 * 3 + c] - bb[4 * 3 + c];
        f.u3[c] = bb[0 * 3 + c] - bb[4 * 3 + c];
    }
    f.d1 = f.u1[0] * f.u1[0] + f.u1[1] * f.u1[1] + f.u1[2] * f.u1[2];
    f.d2 = f.u2[0] * f.u2[0] + f.u2[1] * f.u2[1] + f.u2[2] * f.u2[2];
    f.d3 = f.u3[0] * f.u3[0] + f.u3[1] * f.u3[1] + f.u3[2] * f.u3[2];
}

__device__ __forceinline__ bool inside(const BoxFrame &f, double x, double y, double z) {
    const double ux = x - f.o[0], uy = y - f.o[1], uz = z - f.o[2];
    const double p1 = ux * f.u1[0] + uy * f.u1[1] + uz * f.u1[2];
    const double p2 = ux * f.u2[0] + uy * f.u2[1] + uz * f.u2[2];
    const double p3 = ux * f.u3[0] + uy * f.u3[1] + uz * f.u3[2];
    return (p1 > 0.0) & (p1 < f.d1) & (p2 > 0.0) & (p2 < f.d2) & (p3 > 0.0) & (p3 < f.d3);
}

__global__ __launch_bounds__(256) void iou_3d_kernel(int nres, const double *__restrict__ bbox1, const double *__restrict__ bbox2,
                                                     double *__restrict__ iou, long *__restrict__ counts) {
#pragma clang fp contract(off)
    __shared__ int red[2][4];
    const int pair = blockIdx.x;
    const double *b1 = bbox1 + (size_t)pair * 24, *b2 = bbox2 + (size_t)pair * 24;
    BoxFrame f1, f2;
    box_frame(b1, f1);
    box_frame(b2, f2);
    double lo[3], hi[3], step[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double mn = b1[c], mx = b1[c];
        for (int k = 0; k < 8; ++k) {
            mn = fmin(mn, fmin(b1[k * 3 + c], b2[k * 3 + c]));
            mx = fmax(mx, fmax(b1[k * 3 + c], b2[k * 3 + c]));
        }
        lo[c] = mn; hi[c] = mx;
        step[c] = (mx - mn) / (double)(nres - 1);          // numpy.linspace: step = delta / div
    }
    const long total = (long)nres * nres * nres;
    int both = 0, either = 0;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int iz = (int)(e % nres), iy = (int)((e / nres) % nres), ix = (int)(e / ((long)nres * nres));
        // linspace: y = arange(num) * step + start, then y[-1] = stop
        const double x = ix == nres - 1 ? hi[0] : (double)ix * step[0] + lo[0];
        const double y = iy == nres - 1 ? hi[1] : (double)iy * step[1] + lo[1];
        const double z = iz == nres - 1 ? hi[2] : (double)iz * step[2] + lo[2];
        const bool i1 = inside(f1, x, y, z), i2 = inside(f2, x, y, z);
        both += (i1 & i2) ? 1 : 0;
        either += (i1 | i2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { both += __shfl_xor(both, o, 64); either += __shfl_xor(either, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = both; red[1][wave] = either; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long I = (long)red[0][0] + red[0][1] + red[0][2] + red[0][3], U = (long)red[1][0] + red[1][1] + red[1][2] + red[1][3];
        iou[pair] = U == 0 ? 1.0 : (double)I / (double)U;
        if (counts) { counts[pair * 2] = I; counts[pair * 2 + 1] = U; }
    }
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_iou_3d(int npairs, int nres, const double *bbox1, const double *bbox2, double *iou, long *counts, void *stream) {
    ANCSH_REQUIRE(npairs >= 0 && nres >= 2 && nres <= 1024, "iou_3d: npairs=%d nres=%d (2..1024)", npairs, nres);
    if (npairs == 0) return ANCSH_OK;
    ANCSH_REQUIRE(bbox1 && bbox2 && iou, "iou_3d: null pointer");
    hipLaunchKernelGGL(iou_3d_kernel, dim3(npairs), dim3(256), 0, (hipStream_t)stream, nres, bbox1, bbox2, iou, counts);
    return check_launch("iou_3d");
}

namespace ancsh {

// ---- joint parameters from the per-point heads (evaluation/eval_joint_params.py:143-199) --------------------------------
// Per cloud, the reference (numpy, one sample at a time):
//   * per part j: x = global NOCS, y = part NOCS of the points labelled j (argmax of the mask);
//       scale_j = std(mean(y, axis=1)) / std(mean(x, axis=1)),  translation_j = mean(y - scale_j * x, axis=0)      (:160-171)
//   * per joint j >= 1: the points whose joint class is j vote   joint_pts = nocs_g + unitvec * (1 - heatmap) * 0.2   (:178-181);
//       joint point = per-channel MEDIAN of the votes, joint axis = per-channel median of joint_axis_per_point          (:183-184)
//       (ground-truth variant :192-199: the axis is the MEAN of the votes' orientations).
// One workgroup per (cloud, part) and per (cloud, joint).  float32 element arithmetic in numpy's order (the inputs are float32
// .h5 arrays); medians are exact selections (ordered compaction + bitonic sort per channel, as joint_direction_kernel in
// pose.hip); the reductions behind std / mean run in float64 (numpy: pairwise float32 -- equal to ~1e-7, tests bound 1e-6).
template <int K_MAX>
__device__ __forceinline__ int argmax_row(const float *m, int K) {
    int c = 0;
    float best = m[0];
    for (int k = 1; k < K; ++k) { const float v = m[k]; if (v > best) { best = v; c = k; } }    // np.argmax: first maximum
    return c;
}

__device__ __forceinline__ double block_sum_f64(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void joint_params_kernel(int n, int K, int G, int axis_mean, const float *__restrict__ gocs,
                                                           const float *__restrict__ nocs, const float *__restrict__ mask,
                                                           const float *__restrict__ heatmap, const float *__restrict__ unitvec,
                                                           const float *__restrict__ axis, const int *__restrict__ joint_cls,
                                                           double *__restrict__ st, double *__restrict__ joint) {
    extern __shared__ float jp_vals[];   // 6 * npow2 floats (joint blocks)
    __shared__ double red[4];
    __shared__ int wcnt[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t p0 = (size_t)b * n;
    if ((int)blockIdx.y < K) {           // ---- similarity global NOCS -> part NOCS of part j
        if (!nocs || !st) return;
        const int j = blockIdx.y;
        double sx = 0, sxx = 0, sy = 0, syy = 0, m = 0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = mask ? argmax_row<8>(mask + (p0 + i) * K, K) : 0;
            if (c != j) continue;
            const float *x = gocs + (p0 + i) * G + (G == 3 ? 0 : 3 * j), *y = nocs + (p0 + i) * 3 * K + 3 * j;
            const float xm = ((x[0] + x[1]) + x[2]) / 3.0f, ym = ((y[0] + y[1]) + y[2]) / 3.0f;    // np.mean(., axis=1) in float32
            sx += xm; sxx += (double)xm * xm; sy += ym; syy += (double)ym * ym; m += 1.0;
        }
        sx = block_sum_f64(sx, red); sxx = block_sum_f64(sxx, red); sy = block_sum_f64(sy, red); syy = block_sum_f64(syy, red);
        m = block_sum_f64(m, red);
        const double vx = sxx / m - (sx / m) * (sx / m), vy = syy / m - (sy / m) * (sy / m);
        const float scale = (float)sqrt(vy > 0 ? vy : 0.0) / (float)sqrt(vx > 0 ? vx : 0.0);      // float32 / float32 (np.std of float32)
        double t[3] = {0, 0, 0};
        for (int i = threadIdx.x; i < n; i += 256) {
            const int c = mask ? argmax_row<8>(mask + (p0 + i) * K, K) : 0;
            if (c != j) continue;
            const float *x = gocs + (p0 + i) * G + (G == 3 ? 0 : 3 * j), *y = nocs + (p0 + i) * 3 * K + 3 * j;
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) t[c3] += (double)(y[c3] - scale * x[c3]);
        }
        for (int c3 = 0; c3 < 3; ++c3) t[c3] = block_sum_f64(t[c3], red);
        if (threadIdx.x == 0) {
            double *o = st + ((size_t)b * K + j) * 4;
            o[0] = m > 0 ? (double)scale : NAN;
            for (int c3 = 0; c3 < 3; ++c3) o[1 + c3] = m > 0 ? t[c3] / m : NAN;
        }
        return;
    }
    // ---- joint j: ordered compaction of the votes (index order), then per-channel median (or mean of the axis)
    const int j = (int)blockIdx.y - K + 1;
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const bool f = i < n && joint_cls[p0 + i] == j;
        const unsigned long long mm = __ballot(f);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(mm);
        __syncthreads();
        int start = cnt;
        for (int w = 0; w < wave; ++w) start += wcnt[w];
        if (f) {
            const int pos = start + __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0));
            const int c = (mask && G != 3) ? argmax_row<8>(mask + (p0 + i) * K, K) : 0;
            const float *g = gocs + (p0 + i) * G + (G == 3 ? 0 : 3 * c);
            const float w1 = 1.0f - heatmap[p0 + i];
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
                jp_vals[c3 * npow2 + pos] = axis[(p0 + i) * 3 + c3];
                const float off = (unitvec[(p0 + i) * 3 + c3] * w1) * 0.2f;      // unitvec * (1 - heatmap) * thres_r, float32
                jp_vals[(3 + c3) * npow2 + pos] = g[c3] + off;
            }
        }
        cnt += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
    __syncthreads();
    double *o = joint + ((size_t)b * (K - 1) + (j - 1)) * 6;
    if (axis_mean && threadIdx.x < 3) {      // np.mean(orient_gt[idx], axis=0): float32 accumulation row by row, then / count
        const float *v = jp_vals + threadIdx.x * npow2;
        float s = 0.f;
        for (int e = 0; e < cnt; ++e) s = s + v[e];
        o[3 + threadIdx.x] = cnt > 0 ? (double)(s / (float)cnt) : NAN;
    }
    __syncthreads();
    int p2 = 1;
    while (p2 < cnt) p2 <<= 1;
    for (int e = cnt + threadIdx.x; e < p2; e += 256)
#pragma unroll
        for (int c = 0; c < 6; ++c) jp_vals[c * npow2 + e] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int e = threadIdx.x; e < p2; e += 256) {
                const int partner = e ^ s;
                if (partner > e) {
                    const bool up = (e & k) == 0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        float *v = jp_vals + c * npow2;
                        const float a = v[e], bb = v[partner];
                        if ((a > bb) == up) { v[e] = bb; v[partner] = a; }
                    }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x < 6) {
        const float *v = jp_vals + threadIdx.x * npow2;
        float med = NAN;
        if (cnt > 0) med = (cnt & 1) ? v[cnt / 2] : (v[cnt / 2 - 1] + v[cnt / 2]) * 0.5f;
        if (threadIdx.x >= 3) o[threadIdx.x - 3] = med;            // joint point
        else if (!axis_mean) o[3 + threadIdx.x] = med;            // joint axis
    }
}

// np.max / np.min PROPAGATE a NaN (compute_miou.py:196-208 takes np.max(abs(nocs - 0.5)) per part: a NaN prediction gives a NaN extent);
// fmaxf / fmin would drop it
__device__ __forceinline__ float np_maxf(float a, float b) { return a != a ? a : (b != b ? b : fmaxf(a, b)); }
__device__ __forceinline__ double np_min(double a, double b) { return a != a ? a : (b != b ? b : fmin(a, b)); }

// ---- amodal-box extents and boundaries of the predicted parts (evaluation/compute_miou.py:196-208, eval_pose_err.py:253-268) ----
// Per cloud and part j (points whose predicted mask row has its first maximum at j):
//   scale_pred_j = 2 * max |nocs_j - 0.5| per channel   (float32, numpy's ops: subtraction, abs, max; the doubling is exact)
//   dynam_j      = min over the part's points of the x coordinate of the point taken back through part 0's pose:
//                  ([P 1] . pinv(rt_0^T))[:, 0] with rt_0 = compose_rt(R0, t0) in FLOAT32 (:25-30) = sum_c (P_c - t0_c) * R0[c][0],
//                  float64 products of the float32-rounded pose (the reference's float32 pinv differs from this inverse by ~1e-7)
//   count_j      = points of the part (0 -> the reference's np.max raises and its bare except drops the frame)
// One workgroup per cloud, all its parts in one pass; HBM-bound: the mask, the point and the point's own NOCS slot are read once.
__global__ __launch_bounds__(256) void part_extents_kernel(int n, int K, int C, const float *__restrict__ nocs, const float *__restrict__ mask,
                                                           const float *__restrict__ P, int ldp, const double *__restrict__ pose0,
                                                           float *__restrict__ scale_pred, double *__restrict__ dynam,
                                                           int *__restrict__ count) {
    constexpr int KM = 8;
    __shared__ float smax[4][KM][3];
    __shared__ double smin[4][KM];
    __shared__ int scnt[4][KM];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t p0 = (size_t)b * n;
    const double *ps = pose0 + (size_t)b * 12;                  // R0 row-major (9) | t0 (3)
    const double r00 = (double)(float)ps[0], r10 = (double)(float)ps[3], r20 = (double)(float)ps[6];
    const double t0 = (double)(float)ps[9], t1 = (double)(float)ps[10], t2 = (double)(float)ps[11];
    const double m30 = (double)(float)(-(t0 * r00 + t1 * r10 + t2 * r20));      // the inverse's translation entry, float32 like the pinv's
    // one pass over the cloud: every part's running extents in registers (the part index only selects, it never addresses)
    float m[KM][3];
    double mn[KM];
    int cnt[KM];
#pragma unroll
    for (int j = 0; j < KM; ++j) { m[j][0] = m[j][1] = m[j][2] = -INFINITY; mn[j] = INFINITY; cnt[j] = 0; }
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = argmax_row<KM>(mask + (p0 + i) * K, K);
        const float *q = nocs + (p0 + i) * C + (C == 3 ? 0 : 3 * c);
        const float a0 = fabsf(q[0] - 0.5f), a1 = fabsf(q[1] - 0.5f), a2 = fabsf(q[2] - 0.5f);
        const float *x = P + (p0 + i) * ldp;
        // numpy: [x y z 1] . M[:, 0] accumulated left to right in float64
        const double v = (((double)x[0] * r00 + (double)x[1] * r10) + (double)x[2] * r20) + m30;
#pragma unroll
        for (int j = 0; j < KM; ++j) {
            const bool mine = c == j;
            m[j][0] = mine ? np_maxf(m[j][0], a0) : m[j][0];
            m[j][1] = mine ? np_maxf(m[j][1], a1) : m[j][1];
            m[j][2] = mine ? np_maxf(m[j][2], a2) : m[j][2];
            mn[j] = mine ? np_min(mn[j], v) : mn[j];
            cnt[j] += mine ? 1 : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        if (j >= K) break;                                      // K is uniform: the unused parts cost nothing past this point
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m[j][0] = np_maxf(m[j][0], __shfl_xor(m[j][0], o, 64)); m[j][1] = np_maxf(m[j][1], __shfl_xor(m[j][1], o, 64));
            m[j][2] = np_maxf(m[j][2], __shfl_xor(m[j][2], o, 64));
            mn[j] = np_min(mn[j], __shfl_xor(mn[j], o, 64));
            cnt[j] += __shfl_xor(cnt[j], o, 64);
        }
        if (lane == 0) { smax[wave][j][0] = m[j][0]; smax[wave][j][1] = m[j][1]; smax[wave][j][2] = m[j][2]; smin[wave][j] = mn[j]; scnt[wave][j] = cnt[j]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
        const int j = threadIdx.x;
        const size_t o = (size_t)b * K + j;
        const int c = scnt[0][j] + scnt[1][j] + scnt[2][j] + scnt[3][j];
        count[o] = c;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            scale_pred[o * 3 + k] = c > 0 ? 2.0f * np_maxf(np_maxf(smax[0][j][k], smax[1][j][k]), np_maxf(smax[2][j][k], smax[3][j][k])) : NAN;
        dynam[o] = c > 0 ? np_min(np_min(smin[0][j], smin[1][j]), np_min(smin[2][j], smin[3][j])) : NAN;
    }
}

}  // namespace ancsh

extern "C" int ancsh_part_extents(int b, int n, int K, int nocs_channels, const float *nocs, const float *mask, const float *P, int ldp,
                                  const double *pose0, float *scale_pred, double *dynam, int *count, void *stream) {
    using namespace ancsh;
    ANCSH_REQUIRE(b >= 0 && n > 0 && K >= 1 && K <= 8 && ldp >= 3, "part_extents: bad sizes b=%d n=%d K=%d ldp=%d", b, n, K, ldp);
    ANCSH_REQUIRE(nocs_channels == 3 || nocs_channels == 3 * K, "part_extents: nocs must have 3 or 3K = %d channels, got %d", 3 * K, nocs_channels);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(nocs && mask && P && pose0 && scale_pred && dynam && count, "part_extents: null pointer");
    hipLaunchKernelGGL(part_extents_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, K, nocs_channels, nocs, mask, P, ldp, pose0,
                       scale_pred, dynam, count);
    return check_launch("part_extents");
}

extern "C" int ancsh_joint_params(int b, int n, int K, int gocs_channels, int axis_mean, const float *gocs, const float *nocs,
                                  const float *mask, const float *heatmap, const float *unitvec, const float *joint_axis,
                                  const int *joint_cls, double *st, double *joint, void *stream) {
    using namespace ancsh;
    ANCSH_REQUIRE(b >= 0 && n > 0 && K >= 1 && K <= 8, "joint_params: bad sizes b=%d n=%d K=%d", b, n, K);
    ANCSH_REQUIRE(gocs_channels == 3 || gocs_channels == 3 * K, "joint_params: gocs must have 3 or 3K = %d channels, got %d", 3 * K, gocs_channels);
    ANCSH_REQUIRE(gocs_channels == 3 || mask, "joint_params: per-part global NOCS (3K channels) needs the part mask");
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(gocs && heatmap && unitvec && joint_axis && joint_cls && (joint || K == 1), "joint_params: null pointer");      // K == 1: no joint rows
    ANCSH_REQUIRE(!nocs == !st, "joint_params: nocs and st go together (both or neither)");
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    const size_t lds = (size_t)6 * npow2 * sizeof(float);
    ANCSH_REQUIRE(lds <= 144 * 1024, "joint_params: n %d too large for the LDS-resident medians", n);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)joint_params_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(joint_params_kernel, dim3(b, K + K - 1), dim3(256), lds, (hipStream_t)stream, n, K, gocs_channels, axis_mean ? 1 : 0,
                       gocs, nocs, mask, heatmap, unitvec, joint_axis, joint_cls, st, joint);
    return check_launch("joint_params");
}
