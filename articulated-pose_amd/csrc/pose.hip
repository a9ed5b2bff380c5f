// pose.hip -- batched pose fitting for gfx950: per-part RANSAC similarity fit (stage A), per-joint
// RANSAC with articulated Levenberg-Marquardt refinement (stage B), their full-inlier refits, part
// partitioning, joint-direction medians and Umeyama alignment.
//
// Reference (all host Python, one process per CPU core, ~10 s per cloud):
//   evaluation/parallel_ancsh_pose.py:20-33    ransac
//   evaluation/parallel_ancsh_pose.py:35-54    single_transformation_{estimator,verifier}
//   evaluation/parallel_ancsh_pose.py:56-68    objective_eval
//   evaluation/parallel_ancsh_pose.py:106-194  joint_transformation_{estimator,verifier}
//   evaluation/parallel_ancsh_pose.py:238-341  per-cloud orchestration
//   lib/d3_utils.py:150-163,206-246            Rodrigues, Kabsch, transform_pts, scale_pts
//   lib/aligning.py:580-622                    estimateSimilarityUmeyama
//
// CDNA4 mapping (this is ALU/latency-bound work on <= 24 KB of points per part, not HBM-bound):
//   * hypotheses are the parallel axis: one lane = one hypothesis.  The part's points sit in LDS; every
//     lane walks the same point at the same time, so the verifier's LDS reads are broadcasts
//     (conflict-free) and its inner loop is pure VALU;
//   * the winner is chosen by a deterministic arg-max over the per-hypothesis score array (strictly
//     greater wins => earliest iteration on ties, as the reference's `>` at :28);
//   * refits run one workgroup per problem: masks, means, 3x3 moment sums and the O(n^2) pairwise
//     scale sums are block reductions over LDS-resident inliers; the big LM shares MINPACK's control
//     flow across the workgroup (every thread holds the same 6x6 system);
//   * sample indices are an INPUT (`draws`), so a run is reproducible against numpy's randint stream;
//     with draws == NULL a counter-based device generator (splitmix64 of seed/problem/iteration) is used.
#include "common.h"
#include "pose_math.h"

namespace ancsh {
namespace pose {

constexpr int MODEL_A = 13;   // R(9) s t(3)
constexpr int MODEL_B = 26;   // R0(9) s0 t0(3) R1(9) s1 t1(3)

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ int device_draw(unsigned long long seed, int prob, int iter, int k, int n) {
    const unsigned long long h = splitmix64(seed ^ splitmix64(((unsigned long long)prob << 40) ^ ((unsigned long long)iter << 8) ^ (unsigned)k));
    return (int)(h % (unsigned long long)n);
}

// ---- block reductions (256 threads = 4 waves) -----------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sums K doubles across the workgroup; every thread receives every total.  red: LDS, >= K*nwaves doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *red, int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < K; ++i) red[wave * K + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += red[w * K + i];
        v[i] = s;
    }
}

// ---- similarity model from a point set held as arrays (thread-local, tiny n) --------------------------
// transform_pts on 3 sampled points: Kabsch rotation, pairwise scale, mean translation.
__device__ __forceinline__ void estimate_single3(const float s[3][3], const float t[3][3], float R[9], float &scale,
                                                 float tr[3]) {
    double sm[3], tm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sm[c] = ((double)s[0][c] + s[1][c] + s[2][c]) / 3.0;
        tm[c] = ((double)t[0][c] + t[1][c] + t[2][c]) / 3.0;
    }
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a * 3 + b] += (t[i][a] - tm[a]) * (s[i][b] - sm[b]);
    double q[4], Rd[9];
    horn_quat_fast(M, q);
    quat_to_mat(q, Rd);
    double ab = 0.0, aa = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            double ds = 0.0, dt = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double u = (double)s[i][c] - s[j][c], v = (double)t[i][c] - t[j][c];
                ds += u * u;
                dt += v * v;
            }
            ab += 2.0 * sqrt(ds) * sqrt(dt);   // ordered pairs (i,j) and (j,i); i == j contributes 0
            aa += 2.0 * ds;
        }
    const double sc = ab / (aa + 1e-6);
    scale = (float)sc;
#pragma unroll
    for (int a = 0; a < 9; ++a) R[a] = (float)Rd[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            acc += (double)t[i][a] - (double)scale * (Rd[a * 3 + 0] * s[i][0] + Rd[a * 3 + 1] * s[i][1] + Rd[a * 3 + 2] * s[i][2]);
        tr[a] = (float)(acc / 3.0);
    }
}

// single_transformation_verifier for one point, float32 like the reference's numpy expression
// target - scale*matmul(rotation, source) - translation ; sqrt(sum(res^2)) < th
__device__ __forceinline__ float residual_sq_f32(const float R[9], float sc, const float tr[3], float sx, float sy, float sz,
                                                 float tx, float ty, float tz) {
#pragma clang fp contract(off)   // products and differences individually rounded, as numpy evaluates them
    const float rx = __builtin_fmaf(R[2], sz, __builtin_fmaf(R[1], sy, R[0] * sx));
    const float ry = __builtin_fmaf(R[5], sz, __builtin_fmaf(R[4], sy, R[3] * sx));
    const float rz = __builtin_fmaf(R[8], sz, __builtin_fmaf(R[7], sy, R[6] * sx));
    const float ex = (tx - sc * rx) - tr[0], ey = (ty - sc * ry) - tr[1], ez = (tz - sc * rz) - tr[2];
    return (ex * ex + ey * ey) + ez * ez;
}
__device__ __forceinline__ bool inlier_f32(const float R[9], float sc, const float tr[3], float sx, float sy, float sz,
                                           float tx, float ty, float tz, float th_sq) {
    // reference: sqrt(sum) < th in float32.  sqrt is monotone and correctly rounded, so this equals
    // sum < T with T = min{x : sqrtf(x) >= th} (sq_threshold_f32, computed once on the host): no sqrt per point.
    return residual_sq_f32(R, sc, tr, sx, sy, sz, tx, ty, tz) < th_sq;
}

// Optional extra outputs of the two finish kernels (round 5; every pointer may be null):
//   record (B, K, 26) float64 -- the pose record of evaluation/parallel_ancsh_pose.py:330-353 per part, [baseline R(9) s t(3) |
//     nonlinear R(9) s t(3)]: stage A's problem p = b * K + j fills columns 0..12 of row p (and 13..25 when K == 1: the reference's
//     nonlinear entry of a one-part object is its baseline), stage B's problem p = b * (K - 1) + q fills columns 13..25 of row
//     b * K + q + 1 (and, q == 0, of row b * K: part 0 is reported from joint 1's fit, :327-329) -- no assembly launches afterwards;
//   tie (nprob, 2) int32 -- how implementation-sensitive the fit is: [0] points of the part(s) whose residual norm under the WINNING
//     hypothesis lies within +-window of the threshold (the verifiers' `sqrt(sum(res**2)) < th`, :48-54,186-194: an implementation
//     whose model differs in the last bits may count such a point on the other side), [1] DEGENERATE CONTENDERS: hypotheses -- the
//     winner included -- whose score is within one inlier of the winning score and whose 3-point sample repeats an index
//     (np.random.randint draws with replacement, :38,110-111); stage A (round 6) counts only those that WOULD CHANGE THE CONSENSUS SET
//     (the winner itself, or a contender whose inlier mask differs from the winner's).  Such a sample's centred points are collinear, its 3 x 3 covariance has
//     rank 1, and the rotation the reference takes from np.linalg.svd is LAPACK's completion of a null space that rounding noise
//     selects: implementation-defined in the reference itself (include/ancsh_hip.h has the measured figures).
constexpr int TIE_MAX_CAND = 16;       // degenerate contenders examined one by one in stage A's finish kernel
struct FitExtras {
    double *record;
    int K;
    int *tie;
    float lo_f, hi_f;        // stage A: squared-norm window [lo, hi)
    double lo_d, hi_d;       // stage B
    const int *draws;        // stage B: the sample streams (stage A's finish kernel already takes them)
    unsigned long long seed;
};

// the same predicate for two points at once on packed f32 (identical per-element roundings: every packed instruction
// is the IEEE operation applied to each half); returns how many of the two are inliers
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int inlier2_f32(const float R[9], float sc, const float tr[3], f32x2 sx, f32x2 sy, f32x2 sz,
                                           f32x2 tx, f32x2 ty, f32x2 tz, float th_sq) {
#pragma clang fp contract(off)
    const f32x2 rx = __builtin_elementwise_fma((f32x2)R[2], sz, __builtin_elementwise_fma((f32x2)R[1], sy, R[0] * sx));
    const f32x2 ry = __builtin_elementwise_fma((f32x2)R[5], sz, __builtin_elementwise_fma((f32x2)R[4], sy, R[3] * sx));
    const f32x2 rz = __builtin_elementwise_fma((f32x2)R[8], sz, __builtin_elementwise_fma((f32x2)R[7], sy, R[6] * sx));
    const f32x2 ex = (tx - sc * rx) - tr[0], ey = (ty - sc * ry) - tr[1], ez = (tz - sc * rz) - tr[2];
    const f32x2 s = (ex * ex + ey * ey) + ez * ez;
    return (s.x < th_sq ? 1 : 0) + (s.y < th_sq ? 1 : 0);
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pose.hip: the packed-f32 clamp counting (v_pk_mul_f32 ... clamp, DX10_CLAMP semantics) is written for gfx950 only"
#endif
// the same predicate, COUNTED on packed f32: cnt += (s < th ? 1.0f : 0.0f) per half, as  clamp((th - s) * 2^126)  -- one v_pk_fma_f32
// with the clamp modifier + v_pk_add_f32: two packed instructions per two points instead of two compares, two selects
// and an add with their VCC wait states.  Exact: th - s > 0 iff s < th (the difference of two distinct floats of this magnitude is
// never zero and far above the denormal range), any positive difference times 2^126 is >= 1, +inf (padding) and NaN clamp to 0
// (DX10_CLAMP), and the counts stay below 2^24.
__device__ __forceinline__ void inlier2_count_f32(const float R[9], float sc, const float tr[3], f32x2 sx, f32x2 sy, f32x2 sz,
                                                  f32x2 tx, f32x2 ty, f32x2 tz, float th_sq, f32x2 &cnt) {
#pragma clang fp contract(off)
    const f32x2 rx = __builtin_elementwise_fma((f32x2)R[2], sz, __builtin_elementwise_fma((f32x2)R[1], sy, R[0] * sx));
    const f32x2 ry = __builtin_elementwise_fma((f32x2)R[5], sz, __builtin_elementwise_fma((f32x2)R[4], sy, R[3] * sx));
    const f32x2 rz = __builtin_elementwise_fma((f32x2)R[8], sz, __builtin_elementwise_fma((f32x2)R[7], sy, R[6] * sx));
    const f32x2 ex = (tx - sc * rx) - tr[0], ey = (ty - sc * ry) - tr[1], ez = (tz - sc * rz) - tr[2];
    const f32x2 s = (ex * ex + ey * ey) + ez * ez;
    // one = clamp(fma(s, -2^126, th_sq * 2^126)): the fused product-sum is the exactly rounded (th_sq - s) * 2^126 (th_sq * 2^126 is exact
    // and finite for th_sq < 4, the launcher's guard), so its sign is that of th_sq - s without the separate subtraction
    const f32x2 nbig = {-0x1p126f, -0x1p126f};
    const f32x2 tbig = {th_sq * 0x1p126f, th_sq * 0x1p126f};
    f32x2 one;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(one) : "v"(s), "v"(nbig), "v"(tbig));
    cnt = cnt + one;
}

__device__ __forceinline__ void load_draw3(const int *draws, unsigned long long seed, int prob, int niter, int h, int k0,
                                           int stride, int n, int idx[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int v = draws ? draws[((size_t)prob * niter + h) * stride + k0 + k] : device_draw(seed, prob, h, k0 + k, n);
        idx[k] = v < 0 ? 0 : (v >= n ? n - 1 : v);
    }
}

// ================================ stage A ==========================================================
#ifndef A_CHUNK_N
#define A_CHUNK_N 2048
#endif
constexpr int A_CHUNK = A_CHUNK_N;   // points staged per LDS pass (24 B each)

__global__ __launch_bounds__(256) void ransac_single_score_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                                  const float *__restrict__ tgt, float th, int niter,
                                                                  const int *__restrict__ draws, unsigned long long seed,
                                                                  int *__restrict__ scores) {
    // structure-of-arrays tile so that ds_read_b128 hands each lane 4 consecutive points per coordinate and the residual
    // arithmetic runs on packed-f32 (v_pk_mul/fma/add_f32: two points per lane per instruction)
    __shared__ __attribute__((aligned(16))) float pl[6][A_CHUNK];
    const int prob = blockIdx.y, h = blockIdx.x * 256 + threadIdx.x;
    const int r0 = off[prob], n = off[prob + 1] - r0;
    const bool live = h < niter && n > 0;
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sc = 0.f, tr[3] = {0, 0, 0};
    if (live) {
        int id[3];
        load_draw3(draws, seed, prob, niter, h, 0, 3, n, id);
        float s3[3][3], t3[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s3[i][c] = src[(size_t)(r0 + id[i]) * 3 + c];
                t3[i][c] = tgt[(size_t)(r0 + id[i]) * 3 + c];
            }
        estimate_single3(s3, t3, R, sc, tr);
    }
    int cnt = 0;
    for (int base = 0; base < n; base += A_CHUNK) {
        const int m = (n - base) < A_CHUNK ? (n - base) : A_CHUNK;
        const int m4 = (m + 3) & ~3;
        __syncthreads();
        for (int e = threadIdx.x; e < m4 * 3; e += 256) {
            const int i = e / 3, c = e - i * 3;
            const bool in = i < m;
            // rows past the part's end are padded with a point that can never be an inlier (target at +inf)
            pl[c][i] = in ? src[(size_t)(r0 + base) * 3 + e] : 0.f;
            pl[3 + c][i] = in ? tgt[(size_t)(r0 + base) * 3 + e] : __builtin_inff();
        }
        __syncthreads();
        if (live) {
            // software pipelined: the six ds_read_b128 of the NEXT four points are issued before the arithmetic on the current four,
            // so the LDS latency hides under ~50 packed instructions instead of being waited out at the top of every trip
            const float4 *px = (const float4 *)pl[0], *py = (const float4 *)pl[1], *pz = (const float4 *)pl[2];
            const float4 *pa = (const float4 *)pl[3], *pb = (const float4 *)pl[4], *pc = (const float4 *)pl[5];
            const int nq = m4 >> 2;
            auto score4 = [&](const float4 &x, const float4 &y, const float4 &z, const float4 &a, const float4 &b, const float4 &c) {
                cnt += inlier2_f32(R, sc, tr, f32x2{x.x, x.y}, f32x2{y.x, y.y}, f32x2{z.x, z.y}, f32x2{a.x, a.y}, f32x2{b.x, b.y},
                                   f32x2{c.x, c.y}, th);
                cnt += inlier2_f32(R, sc, tr, f32x2{x.z, x.w}, f32x2{y.z, y.w}, f32x2{z.z, z.w}, f32x2{a.z, a.w}, f32x2{b.z, b.w},
                                   f32x2{c.z, c.w}, th);
            };
            // two register sets in ping-pong (no copies): set 1 is read while set 0 is scored and vice versa; reads past the last
            // quad re-read it (loads stay unconditional), an odd last quad is scored once
            float4 x0 = px[0], y0 = py[0], z0 = pz[0], a0 = pa[0], b0 = pb[0], c0 = pc[0];
            for (int q = 0; q < nq; q += 2) {
                const int q1 = q + 1 < nq ? q + 1 : q, q2 = q + 2 < nq ? q + 2 : q1;
                const float4 x1 = px[q1], y1 = py[q1], z1 = pz[q1], a1 = pa[q1], b1 = pb[q1], c1 = pc[q1];
                __builtin_amdgcn_sched_barrier(0);              // keep the reads above the arithmetic (the scheduler sinks them to their use)
                score4(x0, y0, z0, a0, b0, c0);
                __builtin_amdgcn_sched_barrier(0);
                x0 = px[q2]; y0 = py[q2]; z0 = pz[q2]; a0 = pa[q2]; b0 = pb[q2]; c0 = pc[q2];
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < nq) score4(x1, y1, z1, a1, b1, c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (h < niter) scores[(size_t)prob * niter + h] = cnt;
}

// ---- the same scores with the part's points in SCALAR registers ------------------------------------------------------------
// Every lane of a wave tests the SAME point (lane = hypothesis), so the points are wave-uniform operands: they are read with
// scalar loads (s_load_dwordx8 through the scalar cache) from a "quad" copy of the part -- four points per 96-byte record
// {x[4], y[4], z[4]} of the source, {x[4], y[4], z[4]} of the target, so that two neighbouring points form an aligned SGPR pair =
// one packed-f32 operand -- and enter v_pk_mul/fma/add_f32 as the instruction's single scalar source.  No LDS, no barrier, no
// staging pass per workgroup; the kernel is pure vector-ALU work.
// part p starts at quad 2 * (ceil(off[p] / 8) + p): parts never overlap and each owns an even number of quads
__device__ __forceinline__ int quad_start(int r0, int prob) { return 2 * ((r0 + 7) / 8 + prob); }

__global__ __launch_bounds__(256) void soa_quads_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                        const float *__restrict__ tgt, float *__restrict__ quads, int cap_quads) {
    const int prob = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int r0 = off[prob], n = off[prob + 1] - r0;
    if (i >= ((n + 7) & ~7)) return;           // whole PAIRS of quads: the scoring loop takes two quads per trip, unconditionally
    if (quad_start(r0, prob) + (i >> 2) >= cap_quads) return;
    float *q = quads + ((size_t)quad_start(r0, prob) + (i >> 2)) * 24 + (i & 3);
    const bool in = i < n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        q[c * 4] = in ? src[(size_t)(r0 + i) * 3 + c] : 0.f;
        q[12 + c * 4] = in ? tgt[(size_t)(r0 + i) * 3 + c] : __builtin_inff();      // padding: a point that is never an inlier
    }
}

#ifndef POSE_SCORE_WAVES
#define POSE_SCORE_WAVES 0          /* experiment: > 0 caps the scoring kernel's registers at 512 / POSE_SCORE_WAVES per lane so that its waves fit NEXT TO resident SA waves */
#endif
#if POSE_SCORE_WAVES > 0
#define POSE_SCORE_ATTR __attribute__((amdgpu_waves_per_eu(POSE_SCORE_WAVES)))
#else
#define POSE_SCORE_ATTR
#endif
__global__ __launch_bounds__(256) POSE_SCORE_ATTR void ransac_single_score_sreg_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                                       const float *__restrict__ tgt, const float *__restrict__ quads,
                                                                       int cap_quads, float th, int niter, const int *__restrict__ draws,
                                                                       unsigned long long seed, int *__restrict__ scores) {
    const int prob = blockIdx.y, h = blockIdx.x * 256 + threadIdx.x;
    const int r0 = off[prob], n = off[prob + 1] - r0;
    // lanes without a hypothesis (past niter) score the identity model and drop the result: no divergence inside the point loop
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sc = 0.f, tr[3] = {0, 0, 0};
    if (h < niter && n > 0) {
        int id[3];
        load_draw3(draws, seed, prob, niter, h, 0, 3, n, id);
        float s3[3][3], t3[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s3[i][c] = src[(size_t)(r0 + id[i]) * 3 + c];
                t3[i][c] = tgt[(size_t)(r0 + id[i]) * 3 + c];
            }
        estimate_single3(s3, t3, R, sc, tr);
    }
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    const int q0 = quad_start(r0, prob);
    const f32x8 *Q = reinterpret_cast<const f32x8 *>(quads + (size_t)q0 * 24);
    int nq = ((n + 7) >> 3) * 2;                       // even: the part's copy is padded to whole quad pairs
    nq = min(nq, (cap_quads - q0) & ~1);               // never past the caller's buffer (a capacity the caller got wrong)
    f32x2 cnt2 = {0.f, 0.f};                           // inlier counts of the even / odd points (exact small integers)
    auto score4 = [&](const f32x8 &v0, const f32x8 &v1, const f32x8 &v2) {      // x0..3 y0..3 | z0..3 a0..3 | b0..3 c0..3
        inlier2_count_f32(R, sc, tr, f32x2{v0[0], v0[1]}, f32x2{v0[4], v0[5]}, f32x2{v1[0], v1[1]}, f32x2{v1[4], v1[5]},
                          f32x2{v2[0], v2[1]}, f32x2{v2[4], v2[5]}, th, cnt2);
        inlier2_count_f32(R, sc, tr, f32x2{v0[2], v0[3]}, f32x2{v0[6], v0[7]}, f32x2{v1[2], v1[3]}, f32x2{v1[6], v1[7]},
                          f32x2{v2[2], v2[3]}, f32x2{v2[6], v2[7]}, th, cnt2);
    };
    // two SGPR sets in ping-pong.  Scalar loads return out of order, so the only wait there is is lgkmcnt(0) = "everything issued
    // so far": each trip is  issue(next) ; score(current, which arrived before that issue) ; wait  -- the wait sits AFTER the
    // arithmetic, where the load it covers has had a whole score4 (~220 clocks) to come back from the scalar cache.  The read
    // past the last pair re-reads it (loads stay unconditional).
    constexpr int LGKM0 = 0xC07F;                      // s_waitcnt lgkmcnt(0), vmcnt / expcnt untouched
    if (nq > 0) {
        f32x8 a0 = Q[0], a1 = Q[1], a2 = Q[2];
        __builtin_amdgcn_s_waitcnt(LGKM0);
        for (int q = 0; q < nq; q += 2) {
            const int q2 = q + 2 < nq ? q + 2 : q;
            const f32x8 b0 = Q[q * 3 + 3], b1 = Q[q * 3 + 4], b2 = Q[q * 3 + 5];
            __builtin_amdgcn_sched_barrier(0);
            score4(a0, a1, a2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(LGKM0);
            a0 = Q[q2 * 3]; a1 = Q[q2 * 3 + 1]; a2 = Q[q2 * 3 + 2];
            __builtin_amdgcn_sched_barrier(0);
            score4(b0, b1, b2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(LGKM0);
        }
    }
    if (h < niter) scores[(size_t)prob * niter + h] = n > 0 ? (int)(cnt2.x + cnt2.y) : 0;
}

// arg-max with earliest-iteration tie-break over a score array, whole workgroup; result broadcast.
template <class T>
__device__ __forceinline__ int block_argmax_first(const T *sc, int niter, T *best_out, void *lds) {
    T bv = (T)-1;
    int bi = 0x7fffffff;
    for (int h = threadIdx.x; h < niter; h += blockDim.x) {
        const T v = sc[h];
        if (v > bv) { bv = v; bi = h; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    T *rv = (T *)lds;
    int *ri = (int *)(rv + 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    bv = rv[0]; bi = ri[0];
    for (int w = 1; w < nw; ++w)
        if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
    __syncthreads();
    *best_out = bv;
    return bi;
}

// scale_pts' pair sums (lib/d3_utils.py:237-246) over the n LDS-resident points: A_ij = |s_i - s_j|, b_ij = |t_i - t_j| in
// FLOAT32 as numpy evaluates them on its float32 arrays ((dx^2 + dy^2) + dz^2, then sqrt; v_sqrt_f32 is within 1 ulp).
//   pr[0] += sum A*b, pr[1] += sum A*A, pr[2] += sum b*b   over all ordered pairs (i, j).
// Decomposition: wave w owns the column segment j in [w*seg, (w+1)*seg), lane l the rows l, l+64, ...: the j loop and its
// LDS addresses are WAVE-UNIFORM (scalar loop, broadcast reads), every thread does ~n*n/256 pairs.  A row segment (<= ~128
// terms of O(1)) is summed in float32 -- the reference sums all n^2 terms in float32 (sdot) -- and row segments in float64.
// Was: every thread one full row in float64 with an IEEE double sqrt per pair, 1.33 rows per thread on average but 2 for the
// slowest: 164 of the single-part refit kernel's 195 us.
__device__ __forceinline__ void pair_sums(const float (*cs)[3], const float (*ct)[3], int n, double (&pr)[3]) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int seg = (n + nw - 1) / nw;
    const int j0 = wave * seg, j1 = min(n, j0 + seg);
    constexpr int RB = 6;                              // rows a lane keeps in registers per pass: one LDS read of point j feeds RB pairs
    for (int ib = lane; ib < n; ib += 64 * RB) {
        float sx[RB], sy[RB], sz[RB], tx[RB], ty[RB], tz[RB], ab[RB], aa[RB], bb[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = min(ib + 64 * r, n - 1);     // rows past the end are computed on a clamped copy and dropped below
            sx[r] = cs[i][0]; sy[r] = cs[i][1]; sz[r] = cs[i][2];
            tx[r] = ct[i][0]; ty[r] = ct[i][1]; tz[r] = ct[i][2];
            ab[r] = aa[r] = bb[r] = 0.f;
        }
        for (int j = j0; j < j1; ++j) {
            const float cx = cs[j][0], cy = cs[j][1], cz = cs[j][2], dx = ct[j][0], dy = ct[j][1], dz = ct[j][2];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float ux = sx[r] - cx, uy = sy[r] - cy, uz = sz[r] - cz;
                const float vx = tx[r] - dx, vy = ty[r] - dy, vz = tz[r] - dz;
                const float A = __builtin_amdgcn_sqrtf((ux * ux + uy * uy) + uz * uz);
                const float b = __builtin_amdgcn_sqrtf((vx * vx + vy * vy) + vz * vz);
                ab[r] += A * b;
                aa[r] += A * A;
                bb[r] += b * b;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (ib + 64 * r < n) { pr[0] += (double)ab[r]; pr[1] += (double)aa[r]; pr[2] += (double)bb[r]; }
    }
}

// Full-inlier similarity refit of inliers compacted in LDS (cs/ct: n_in x 3 floats): transform_pts.
// Every thread returns the same model.  red: >= 16*4 doubles of LDS.
__device__ __forceinline__ void refit_similarity(const float (*cs)[3], const float (*ct)[3], int n_in, double *red,
                                                 double R[9], double &scale, double tr[3]) {
    double sums[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_in; i += 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { sums[c] += cs[i][c]; sums[3 + c] += ct[i][c]; }
    }
    block_sum<6>(sums, red, 4);
    double sm[3], tm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { sm[c] = sums[c] / n_in; tm[c] = sums[3 + c] / n_in; }
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_in; i += 256)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a * 3 + b] += (ct[i][a] - tm[a]) * (cs[i][b] - sm[b]);
    block_sum<9>(M, red, 4);
    double q[4];
    horn_quat(M, q);
    quat_to_mat(q, R);
    double pr[3] = {0, 0, 0};   // sum A*b, sum A*A (, sum b*b) over ordered pairs
    pair_sums(cs, ct, n_in, pr);
    block_sum<3>(pr, red, 4);
    scale = pr[0] / (pr[1] + 1e-6);
    const double sf = (double)(float)scale;   // the reference's scale is a float32 (float32 inputs)
#pragma unroll
    for (int a = 0; a < 3; ++a) tr[a] = tm[a] - sf * (R[a * 3 + 0] * sm[0] + R[a * 3 + 1] * sm[1] + R[a * 3 + 2] * sm[2]);
}

// ordered compaction of flagged points into LDS arrays; returns the count (uniform).
__device__ __forceinline__ int compact_flagged(bool flag, int i, int n, const float *src, const float *tgt, size_t row0,
                                               float (*cs)[3], float (*ct)[3], int base, int *wcnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    __syncthreads();
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int start = base;
    for (int w = 0; w < wave; ++w) start += wcnt[w];
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (flag) {
        const int pos = start + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cs[pos][c] = src[(row0 + i) * 3 + c];
            ct[pos][c] = tgt[(row0 + i) * 3 + c];
        }
    }
    return base + total;
}

__global__ __launch_bounds__(256) void ransac_single_finish_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                                   const float *__restrict__ tgt, float th, int niter,
                                                                   const int *__restrict__ draws, unsigned long long seed,
                                                                   const int *__restrict__ scores, int max_n,
                                                                   double *__restrict__ out_model,
                                                                   unsigned char *__restrict__ out_inliers,
                                                                   int *__restrict__ out_best, FitExtras E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = (double *)smem;                       // 64 doubles
    int *wcnt = (int *)(red + 64);                      // 4 ints for compact_flagged + 4 for the tie counts
    float(*cs)[3] = (float(*)[3])(wcnt + 8);
    float(*ct)[3] = cs + max_n;
    const int prob = blockIdx.x;
    const int r0 = off[prob], n = off[prob + 1] - r0;
    double *om = out_model + (size_t)prob * MODEL_A;
    if (n <= 0 || n > max_n) {   // empty part: the reference raises (randint(0)); report instead of dying
        if (threadIdx.x < MODEL_A) om[threadIdx.x] = NAN;
        if (E.record && threadIdx.x < (E.K == 1 ? 2 * MODEL_A : MODEL_A)) E.record[(size_t)prob * 26 + threadIdx.x] = NAN;
        if (E.tie && threadIdx.x < 2) E.tie[prob * 2 + threadIdx.x] = 0;
        if (threadIdx.x == 0) { out_best[prob * 2] = -1; out_best[prob * 2 + 1] = n <= 0 ? 0 : -2; }
        for (int i = threadIdx.x; i < n; i += 256) out_inliers[r0 + i] = 0;      // every row's flag is written by this kernel
        return;
    }
    int best_score;
    const int best = block_argmax_first<int>(scores + (size_t)prob * niter, niter, &best_score, red);
    // re-derive the winning hypothesis (same device function => same bits as when it was scored)
    int id[3];
    load_draw3(draws, seed, prob, niter, best, 0, 3, n, id);
    float s3[3][3], t3[3][3], R[9], sc, tr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s3[i][c] = src[(size_t)(r0 + id[i]) * 3 + c];
            t3[i][c] = tgt[(size_t)(r0 + id[i]) * 3 + c];
        }
    estimate_single3(s3, t3, R, sc, tr);
    int n_in = 0, n_border = 0, n_near = 0;             // the tie counts: per-wave partial sums (wave-uniform)
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        bool f = false, border = false;
        if (i < n) {
            const float *ps = src + (size_t)(r0 + i) * 3, *pt = tgt + (size_t)(r0 + i) * 3;
            const float rs = residual_sq_f32(R, sc, tr, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2]);
            f = rs < th;
            border = rs >= E.lo_f && rs < E.hi_f;
            out_inliers[r0 + i] = f ? 1 : 0;
        }
        n_border += __popcll(__ballot(border));
        n_in = compact_flagged(f, i, n, src, tgt, (size_t)r0, cs, ct, n_in, wcnt);
    }
    if (E.tie) {                                        // block-uniform
        // [1]: degenerate contenders (repeated-index sample, score within one inlier of the winner's) THAT WOULD CHANGE THE CONSENSUS SET:
        // the winner itself when its sample is degenerate, and every other such hypothesis whose inlier mask differs from the winner's
        // in at least one point.  (Round 5 counted every degenerate contender: 24.5 % of the fits at N = 1024, almost all of them
        // hypotheses with the winner's own mask, which cannot change the refit whoever scores them.)
        __shared__ int s_cand[TIE_MAX_CAND + 1];
        if (threadIdx.x == 0) s_cand[TIE_MAX_CAND] = 0;
        __syncthreads();
        const int *sp = scores + (size_t)prob * niter;
        for (int h = threadIdx.x; h < niter; h += 256) {
            if (sp[h] >= best_score - 1) {
                int d3[3];
                load_draw3(draws, seed, prob, niter, h, 0, 3, n, d3);
                if (d3[0] == d3[1] || d3[0] == d3[2] || d3[1] == d3[2]) {
                    const int slot = atomicAdd(&s_cand[TIE_MAX_CAND], 1);
                    if (slot < TIE_MAX_CAND) s_cand[slot] = h;
                }
            }
        }
        __syncthreads();
        const int found = s_cand[TIE_MAX_CAND], ncand = found < TIE_MAX_CAND ? found : TIE_MAX_CAND;
        n_near = found - ncand;                         // more contenders than slots: the rest counted as changing (conservative)
        bool winner_degenerate = false;
        for (int c = 0; c < ncand; ++c) {               // block-uniform trip count
            const int h = s_cand[c];
            if (h == best) { ++n_near; winner_degenerate = true; continue; }
            int d3[3];
            load_draw3(draws, seed, prob, niter, h, 0, 3, n, d3);
            float hs[3][3], ht[3][3], hR[9], hsc, htr[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    hs[i][cc] = src[(size_t)(r0 + d3[i]) * 3 + cc];
                    ht[i][cc] = tgt[(size_t)(r0 + d3[i]) * 3 + cc];
                }
            estimate_single3(hs, ht, hR, hsc, htr);
            int differs = 0;
            for (int i = threadIdx.x; i < n; i += 256) {          // the same thread wrote out_inliers[r0 + i] above
                const float *ps = src + (size_t)(r0 + i) * 3, *pt = tgt + (size_t)(r0 + i) * 3;
                const bool f = residual_sq_f32(hR, hsc, htr, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2]) < th;
                differs |= (int)(f != (out_inliers[r0 + i] != 0));
            }
            n_near += __syncthreads_or(differs) ? 1 : 0;
        }
        if ((threadIdx.x & 63) == 0) wcnt[4 + (threadIdx.x >> 6)] = n_border;
        __syncthreads();
        if (threadIdx.x == 0) {
            E.tie[prob * 2] = wcnt[4] + wcnt[5] + wcnt[6] + wcnt[7];
            // NEGATIVE when the winner's own sample is degenerate: that fit's consensus set is implementation-defined for certain
            // (r06_pose_tie_rate_K3.txt: 3 fits of 624, all 3 on another set than the reference arithmetic), where a positive count
            // only says that a degenerate hypothesis came close (22.9 % of the fits, 1 flip among them)
            E.tie[prob * 2 + 1] = winner_degenerate ? -n_near : n_near;               // block-uniform by construction
        }
    }
    __syncthreads();
    double Rd[9], scale, trd[3];
    if (n_in > 0) {
        refit_similarity(cs, ct, n_in, red, Rd, scale, trd);
    } else {
#pragma unroll
        for (int a = 0; a < 9; ++a) Rd[a] = NAN;
        scale = NAN; trd[0] = trd[1] = trd[2] = NAN;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 9; ++a) om[a] = Rd[a];
        om[9] = (double)(float)scale;
#pragma unroll
        for (int a = 0; a < 3; ++a) om[10 + a] = trd[a];
        out_best[prob * 2] = best;
        out_best[prob * 2 + 1] = best_score;
        if (E.record) {
            double *rec = E.record + (size_t)prob * 26;
#pragma unroll
            for (int a = 0; a < MODEL_A; ++a) rec[a] = om[a];
            if (E.K == 1) {
#pragma unroll
                for (int a = 0; a < MODEL_A; ++a) rec[MODEL_A + a] = om[a];
            }
        }
    }
}

// ================================ stage B ==========================================================
// Forward-difference normal equations of the articulated objective (objective_eval, isweight=False):
//   residuals  y0_i - Rod(x0_i, r0)   (depend on params 0..2)
//              y1_i - Rod(x1_i, r1)   (depend on params 3..5)
//              wj x [Rod(J, r0) - Rod(J, r1)]   (all six)
// One `PartFd` handles one rotation vector: base rod + 3 perturbed rods (MINPACK's h_j = eps|x_j|), the
// 3x3 diagonal block of A = J^T J, its 3 entries of g = J^T f, and d Rod(J)/d r for the joint rows.
// Only one part is live at a time, which keeps a whole LM solve inside the VGPR budget.
struct PartFd {
    Rod b, p[3];
    double rh[3];
    double blk[6];     // packed symmetric 3x3 : (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
    double gv[3];
    double ju[3];      // Rod(J, r)
    double jd[3][3];   // jd[c][q] = (Rod(J, r + h_q e_q)[c] - Rod(J, r)[c]) / h_q

    __device__ __forceinline__ void begin(const double r[3], const double J[3]) {
        b = rod_prepare(r[0], r[1], r[2]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double h = fd_step(r[q]);
            rh[q] = fast_rcp(h);
            p[q] = rod_prepare(r[0] + (q == 0 ? h : 0.0), r[1] + (q == 1 ? h : 0.0), r[2] + (q == 2 ? h : 0.0));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) blk[i] = 0.0;
        gv[0] = gv[1] = gv[2] = 0.0;
        rod_apply(b, J[0], J[1], J[2], ju[0], ju[1], ju[2]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double qx, qy, qz;
            rod_apply(p[q], J[0], J[1], J[2], qx, qy, qz);
            jd[0][q] = qx; jd[1][q] = qy; jd[2][q] = qz;     // finished in joint_finish (needs f of the joint row)
        }
    }
    // one point row-triple: f = y - Rod(x, r); a[c][q] = ((y - Rod_q(x)) - f) / h_q
    __device__ __forceinline__ void point(double px, double py, double pz, double yx, double yy, double yz) {
        double ox, oy, oz;
        rod_apply(b, px, py, pz, ox, oy, oz);
        const double f0 = yx - ox, f1 = yy - oy, f2 = yz - oz;
        double a[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double qx, qy, qz;
            rod_apply(p[q], px, py, pz, qx, qy, qz);
            a[0][q] = ((yx - qx) - f0) * rh[q];
            a[1][q] = ((yy - qy) - f1) * rh[q];
            a[2][q] = ((yz - qz) - f2) * rh[q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int r = 0; r <= q; ++r) blk[q * (q + 1) / 2 + r] += a[0][q] * a[0][r] + a[1][q] * a[1][r] + a[2][q] * a[2][r];
            gv[q] += a[0][q] * f0 + a[1][q] * f1 + a[2][q] * f2;
        }
    }
};

// combine the two parts and the joint rows into the packed 6x6 system
__device__ __forceinline__ void assemble_normal(const PartFd &p0, const double blk0[6], const double g0[3], const PartFd &p1,
                                                const double blk1[6], const double g1[3], double wj, double A[21], double g[6]) {
    // joint rows: f = u - w ; d f/d r0_q = (Rod_q(J; r0) - w - f)/h = (jd0 - u)/h ; d f/d r1_q = (u - Rod_q(J; r1) - f)/h = (w - jd1)/h
    double f[3], a[3][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = p0.ju[c] - p1.ju[c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[c][q] = (((p0.jd[c][q] - p1.ju[c]) - f[c])) * p0.rh[q];
            a[c][3 + q] = (((p0.ju[c] - p1.jd[c][q]) - f[c])) * p1.rh[q];
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double v = wj * (a[0][i] * a[0][j] + a[1][i] * a[1][j] + a[2][i] * a[2][j]);
            if (i < 3) v += blk0[i * (i + 1) / 2 + j];
            else if (j >= 3) v += blk1[(i - 3) * (i - 2) / 2 + (j - 3)];
            A[i * (i + 1) / 2 + j] = v;
        }
        g[i] = wj * (a[0][i] * f[0] + a[1][i] * f[1] + a[2][i] * f[2]) + (i < 3 ? g0[i] : g1[i - 3]);
    }
}

// Thread-local articulated problem on 3 + 3 sampled points.
struct HypProblem {
    double x0[3][3], y0[3][3], x1[3][3], y1[3][3], J[3], wj;

    __device__ __forceinline__ double cost(const double x[6]) const {
        const Rod r0 = rod_prepare(x[0], x[1], x[2]), r1 = rod_prepare(x[3], x[4], x[5]);
        double s = 0.0, ox, oy, oz;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rod_apply(r0, x0[i][0], x0[i][1], x0[i][2], ox, oy, oz);
            const double a = y0[i][0] - ox, b = y0[i][1] - oy, c = y0[i][2] - oz;
            s += a * a + b * b + c * c;
            rod_apply(r1, x1[i][0], x1[i][1], x1[i][2], ox, oy, oz);
            const double d = y1[i][0] - ox, e = y1[i][1] - oy, f = y1[i][2] - oz;
            s += d * d + e * e + f * f;
        }
        double ux, uy, uz, wx, wy, wz;
        rod_apply(r0, J[0], J[1], J[2], ux, uy, uz);
        rod_apply(r1, J[0], J[1], J[2], wx, wy, wz);
        s += wj * ((ux - wx) * (ux - wx) + (uy - wy) * (uy - wy) + (uz - wz) * (uz - wz));
        return s;
    }
    __device__ __forceinline__ void normal(const double x[6], double A[21], double g[6]) const {
        PartFd p0, p1;
        p0.begin(x, J);
#pragma unroll
        for (int i = 0; i < 3; ++i) p0.point(x0[i][0], x0[i][1], x0[i][2], y0[i][0], y0[i][1], y0[i][2]);
        p1.begin(x + 3, J);
#pragma unroll
        for (int i = 0; i < 3; ++i) p1.point(x1[i][0], x1[i][1], x1[i][2], y1[i][0], y1[i][1], y1[i][2]);
        assemble_normal(p0, p0.blk, p0.gv, p1, p1.blk, p1.gv, wj, A, g);
    }
};

// scale_pts both ways on 3 points: s = <A,b>/(<A,A>+1e-6), s_inv = <A,b>/(<b,b>+1e-6); float32 results
__device__ __forceinline__ void scales3(const float s[3][3], const float t[3][3], float &sc, float &sc_inv) {
    double ab = 0.0, aa = 0.0, bb = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            double ds = 0.0, dt = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double u = (double)s[i][c] - s[j][c], v = (double)t[i][c] - t[j][c];
                ds += u * u;
                dt += v * v;
            }
            ab += 2.0 * sqrt(ds) * sqrt(dt);
            aa += 2.0 * ds;
            bb += 2.0 * dt;
        }
    sc = (float)(ab / (aa + 1e-6));
    sc_inv = (float)(ab / (bb + 1e-6));
}

// centred source / pre-scaled centred target of 3 samples (float32 like the reference's arrays)
__device__ __forceinline__ void center_part3(const float s[3][3], const float t[3][3], float sc_inv, double xc[3][3],
                                             double yc[3][3]) {
    float sm[3], tm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sm[c] = ((s[0][c] + s[1][c]) + s[2][c]) / 3.0f;
        tm[c] = ((sc_inv * t[0][c] + sc_inv * t[1][c]) + sc_inv * t[2][c]) / 3.0f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xc[i][c] = (double)(s[i][c] - sm[c]);
            yc[i][c] = (double)(sc_inv * t[i][c] - tm[c]);
        }
    }
}

// ... and the Kabsch rotation vector between them (the LM start point)
__device__ __forceinline__ void prep_part3(const float s[3][3], const float t[3][3], float sc_inv, double xc[3][3],
                                           double yc[3][3], double rv[3]) {
    center_part3(s, t, sc_inv, xc, yc);
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // rotate_pts re-centres its (already centred) inputs; the means are ~1e-8 and do not move R
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a * 3 + b] += yc[i][a] * xc[i][b];
    double q[4];
    horn_quat_fast(M, q);
    quat_to_rotvec(q, rv);
}

// joint_transformation_verifier for one point (float64: the model rotation is float64 in the reference)
__device__ __forceinline__ double residual_sq_f64(const double R[9], double sc, const double tr[3], float sx, float sy, float sz,
                                                  float tx, float ty, float tz) {
    const double rx = R[0] * sx + R[1] * sy + R[2] * sz, ry = R[3] * sx + R[4] * sy + R[5] * sz,
                 rz = R[6] * sx + R[7] * sy + R[8] * sz;
    const double ex = ((double)tx - sc * rx) - tr[0], ey = ((double)ty - sc * ry) - tr[1], ez = ((double)tz - sc * rz) - tr[2];
    return ex * ex + ey * ey + ez * ez;
}
__device__ __forceinline__ bool inlier_f64(const double R[9], double sc, const double tr[3], float sx, float sy, float sz,
                                           float tx, float ty, float tz, double th) {
    return residual_sq_f64(R, sc, tr, sx, sy, sz, tx, ty, tz) < th;     // th = exact squared threshold (sq_threshold_f64)
}

// Stage B runs every hypothesis' articulated LM fit (joint_transformation_estimator, :106-184) as three kernels, because
// the fits' lengths are heavy-tailed (measured on the bench workload: mean 45 function evaluations, mean over waves of
// the slowest of 64 consecutive hypotheses 271, degenerate samples up to MINPACK's maxfev = 4200):
//   1. ransac_joint_init_kernel   (lock-step, thread per hypothesis): per-part scales + Kabsch start point -> models[0..9]
//   2. ransac_joint_lm_kernel     (one wave per HYP_CHUNK hypotheses): lanes run lm6_trip; a lane whose fit has terminated
//                                  takes the wave's next unstarted hypothesis, so SIMD time follows the SUM of the fit
//                                  lengths, not 64 x the slowest.  The order in which hypotheses run does not enter any
//                                  result (each is a function of its own draw only).
//   3. ransac_joint_model_kernel  (lock-step): rotations / translations of both parts from the fitted rotation vectors.
// Scratch layout per hypothesis (MODEL_B doubles) between the kernels: [0..5] rotation vectors, [6..7] 1/scale per part,
// [8..9] scale per part; kernel 3 overwrites the slot with the final model.
struct HypSamples {
    float s0[3][3], t0[3][3], s1[3][3], t1[3][3];
};
__device__ __forceinline__ void load_hyp_samples(const float *__restrict__ src, const float *__restrict__ tgt, const int *draws,
                                                 unsigned long long seed, int prob, int niter, int h, int a0, int n0, int a1,
                                                 int n1, HypSamples &q) {
    int i0[3], i1[3];
    load_draw3(draws, seed, prob, niter, h, 0, 6, n0, i0);
    load_draw3(draws, seed, prob, niter, h, 3, 6, n1, i1);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q.s0[i][c] = src[(size_t)(a0 + i0[i]) * 3 + c];
            q.t0[i][c] = tgt[(size_t)(a0 + i0[i]) * 3 + c];
            q.s1[i][c] = src[(size_t)(a1 + i1[i]) * 3 + c];
            q.t1[i][c] = tgt[(size_t)(a1 + i1[i]) * 3 + c];
        }
}

__global__ __launch_bounds__(64) void ransac_joint_init_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                               const float *__restrict__ src, const float *__restrict__ tgt,
                                                               int niter, const int *__restrict__ draws, unsigned long long seed,
                                                               double *__restrict__ scores, double *__restrict__ models) {
    const int prob = blockIdx.y, h = blockIdx.x * 64 + threadIdx.x;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    if (h >= niter) return;
    double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
    if (n0 <= 0 || n1 <= 0) {
        scores[(size_t)prob * niter + h] = -1.0;
        for (int i = 0; i < MODEL_B; ++i) mo[i] = NAN;
        return;
    }
    HypSamples q;
    load_hyp_samples(src, tgt, draws, seed, prob, niter, h, a0, n0, a1, n1, q);
    float sc0, sc0i, sc1, sc1i;
    scales3(q.s0, q.t0, sc0, sc0i);
    scales3(q.s1, q.t1, sc1, sc1i);
    double xc[3][3], yc[3][3], x[6];
    prep_part3(q.s0, q.t0, sc0i, xc, yc, x);
    prep_part3(q.s1, q.t1, sc1i, xc, yc, x + 3);
#pragma unroll
    for (int i = 0; i < 6; ++i) mo[i] = x[i];
    mo[6] = sc0i; mo[7] = sc1i; mo[8] = sc0; mo[9] = sc1;
}

constexpr int HYP_CHUNK = 256;    // hypotheses handed to one wave of a full launch
constexpr int HYP_CHUNK_SMALL = 64;   // ... of a launch too small to fill the chip with 256 per wave (scheduling only: same bits)
constexpr long HYP_SMALL_LAUNCH = 8192;   // fits per launch up to which the small chunk is used
constexpr int HYP_REFILL = 16;    // idle lanes that trigger a refill (a refill costs the whole wave ~1 trip of latency)

__global__ __launch_bounds__(64) void ransac_joint_lm_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                             const float *__restrict__ src, const float *__restrict__ tgt,
                                                             const float *__restrict__ joint_dir, int niter,
                                                             const int *__restrict__ draws, unsigned long long seed,
                                                             double *__restrict__ models, int *__restrict__ lm_stat, int chunk) {
    const int prob = blockIdx.y;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    if (n0 <= 0 || n1 <= 0) return;
    const int c1 = min(niter, (int)(blockIdx.x + 1) * chunk);
    int next = blockIdx.x * chunk;              // wave-uniform: first hypothesis of the chunk not yet handed out
    HypProblem P;
    P.J[0] = joint_dir[prob * 3]; P.J[1] = joint_dir[prob * 3 + 1]; P.J[2] = joint_dir[prob * 3 + 2];
    P.wj = 3.0;   // min(3,3) copies of the joint axis (:134)
    Lm6 S;
    bool active = false;
    int h = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (next < c1 && (n_idle >= HYP_REFILL || n_idle >= c1 - next || n_idle == 64)) {
            if (!active) {
                const int mine = next + __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
                if (mine < c1) {
                    h = mine;
                    const double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
                    HypSamples q;
                    load_hyp_samples(src, tgt, draws, seed, prob, niter, h, a0, n0, a1, n1, q);
                    center_part3(q.s0, q.t0, (float)mo[6], P.x0, P.y0);
                    center_part3(q.s1, q.t1, (float)mo[7], P.x1, P.y1);
#pragma unroll
                    for (int i = 0; i < 6; ++i) S.x[i] = mo[i];
                    lm6_begin(P, S);
                    active = true;
                }
            }
            next += n_idle;
        } else if (n_idle == 64) {
            break;
        }
        if (active && lm6_trip(P, S, 1e-4, 1e-8, 1e-8, 4200)) {
            double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
#pragma unroll
            for (int i = 0; i < 6; ++i) mo[i] = S.x[i];
#ifdef LM_COUNT
            if (lm_stat) { lm_stat[((size_t)prob * niter + h) * 2] = S.info | (S.trips << 4) | (S.nchol << 17); lm_stat[((size_t)prob * niter + h) * 2 + 1] = S.nfev; }
#else
            if (lm_stat) { lm_stat[((size_t)prob * niter + h) * 2] = S.info; lm_stat[((size_t)prob * niter + h) * 2 + 1] = S.nfev; }
#endif
            active = false;
        }
    }
}

// ---- lane-group-cooperative LM --------------------------------------------------------------------------------------------
// The kernel above keeps one fit in one lane: its duration is the LONGEST fit of the batch (1500-evaluation trajectories,
// ~1.6 ms), every one of its ~2000 instructions per evaluation issued for a single live lane.  Here EIGHT lanes share a fit.
// All eight hold the same Lm6 state and run the same MINPACK control flow (so no broadcast is ever needed and divergence only
// exists between groups); the two expensive callbacks are split across the group's lanes and recombined through a 0.9 KB
// group-private LDS record in a fixed order, so that every lane ends up with bit-identical A, g and cost:
//   normal(): role q8 = lane & 7 < 6 owns forward-difference column q8 (part q8 / 3, parameter q8 % 3): one perturbed
//             Rodrigues rotation instead of six, its 9 + 3 Jacobian entries; then every lane assembles the 6x6 system
//             from the six columns with the arithmetic of PartFd::point / assemble_normal above;
//   cost():   roles 0..5 own one (part, point) residual triple each, role 6 the joint-axis residual; the seven partial
//             sums are added in the serial order.
constexpr int COOP_G = 8;                 // lanes per fit
constexpr int COOP_XCH = 112;             // doubles per group record: col 54 | jd 18 | rh 6 | f 18 | ju 6 | cost 7
constexpr int COOP_HYP_PER_WAVE = 32;     // hypotheses handed to one wave (8 at a time, refilled)

__device__ __forceinline__ void group_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct HypProblemCoop {
    double x0[3][3], y0[3][3], x1[3][3], y1[3][3], J[3], wj;
    double *xch;      // this group's LDS record
    int role;         // lane & 7

    __device__ __forceinline__ double cost(const double x[6]) const {
        // role -> (part, point): 0..2 = part 0, 3..5 = part 1, 6 / 7 = joint axis
        const bool joint = role >= 6;
        const bool p1 = role >= 3 && role < 6;
        const int i = joint ? 0 : (p1 ? role - 3 : role);
        double partial;
        if (!joint) {
            const Rod r = p1 ? rod_prepare(x[3], x[4], x[5]) : rod_prepare(x[0], x[1], x[2]);
            const double px = p1 ? (i == 0 ? x1[0][0] : i == 1 ? x1[1][0] : x1[2][0]) : (i == 0 ? x0[0][0] : i == 1 ? x0[1][0] : x0[2][0]);
            const double py = p1 ? (i == 0 ? x1[0][1] : i == 1 ? x1[1][1] : x1[2][1]) : (i == 0 ? x0[0][1] : i == 1 ? x0[1][1] : x0[2][1]);
            const double pz = p1 ? (i == 0 ? x1[0][2] : i == 1 ? x1[1][2] : x1[2][2]) : (i == 0 ? x0[0][2] : i == 1 ? x0[1][2] : x0[2][2]);
            const double tx = p1 ? (i == 0 ? y1[0][0] : i == 1 ? y1[1][0] : y1[2][0]) : (i == 0 ? y0[0][0] : i == 1 ? y0[1][0] : y0[2][0]);
            const double ty = p1 ? (i == 0 ? y1[0][1] : i == 1 ? y1[1][1] : y1[2][1]) : (i == 0 ? y0[0][1] : i == 1 ? y0[1][1] : y0[2][1]);
            const double tz = p1 ? (i == 0 ? y1[0][2] : i == 1 ? y1[1][2] : y1[2][2]) : (i == 0 ? y0[0][2] : i == 1 ? y0[1][2] : y0[2][2]);
            double ox, oy, oz;
            rod_apply(r, px, py, pz, ox, oy, oz);
            const double a = tx - ox, b = ty - oy, c = tz - oz;
            partial = a * a + b * b + c * c;
        } else {
            const Rod r0 = rod_prepare(x[0], x[1], x[2]), r1 = rod_prepare(x[3], x[4], x[5]);
            double ux, uy, uz, wx, wy, wz;
            rod_apply(r0, J[0], J[1], J[2], ux, uy, uz);
            rod_apply(r1, J[0], J[1], J[2], wx, wy, wz);
            partial = (ux - wx) * (ux - wx) + (uy - wy) * (uy - wy) + (uz - wz) * (uz - wz);
        }
        double *cs = xch + 102;
        if (role < 7) cs[role] = partial;
        group_lds_fence();
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { s += cs[k]; s += cs[3 + k]; }     // the serial order: part 0 point k, part 1 point k
        s += wj * cs[6];
        group_lds_fence();                                             // the record is reused by the next callback
        return s;
    }

    __device__ __forceinline__ void normal(const double x[6], double A[21], double g[6]) const {
        double *col = xch, *jdx = xch + 54, *rhx = xch + 72, *fx = xch + 78, *jux = xch + 96;
        {   // ---- this lane's forward-difference column (roles 6, 7 shadow roles 0, 1 without writing) ----
            const int cr = role < 6 ? role : role - 6;
            const bool p1 = cr >= 3;
            const int q = p1 ? cr - 3 : cr;
            const double r0 = p1 ? x[3] : x[0], r1 = p1 ? x[4] : x[1], r2 = p1 ? x[5] : x[2];
            const Rod b = rod_prepare(r0, r1, r2);
            const double h = fd_step(q == 0 ? r0 : q == 1 ? r1 : r2);
            const double rh = fast_rcp(h);
            const Rod pr = rod_prepare(r0 + (q == 0 ? h : 0.0), r1 + (q == 1 ? h : 0.0), r2 + (q == 2 ? h : 0.0));
            const bool wr = role < 6;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double px = p1 ? x1[i][0] : x0[i][0], py = p1 ? x1[i][1] : x0[i][1], pz = p1 ? x1[i][2] : x0[i][2];
                const double yx = p1 ? y1[i][0] : y0[i][0], yy = p1 ? y1[i][1] : y0[i][1], yz = p1 ? y1[i][2] : y0[i][2];
                double ox, oy, oz, qx, qy, qz;
                rod_apply(b, px, py, pz, ox, oy, oz);
                const double f0 = yx - ox, f1 = yy - oy, f2 = yz - oz;
                rod_apply(pr, px, py, pz, qx, qy, qz);
                if (wr) {
                    col[cr * 9 + i * 3 + 0] = ((yx - qx) - f0) * rh;
                    col[cr * 9 + i * 3 + 1] = ((yy - qy) - f1) * rh;
                    col[cr * 9 + i * 3 + 2] = ((yz - qz) - f2) * rh;
                    if (q == 0) { fx[(p1 ? 9 : 0) + i * 3] = f0; fx[(p1 ? 9 : 0) + i * 3 + 1] = f1; fx[(p1 ? 9 : 0) + i * 3 + 2] = f2; }
                }
            }
            double ux, uy, uz, vx, vy, vz;
            rod_apply(b, J[0], J[1], J[2], ux, uy, uz);
            rod_apply(pr, J[0], J[1], J[2], vx, vy, vz);
            if (wr) {
                jdx[cr * 3] = vx; jdx[cr * 3 + 1] = vy; jdx[cr * 3 + 2] = vz;
                rhx[cr] = rh;
                if (q == 0) { jux[(p1 ? 3 : 0)] = ux; jux[(p1 ? 3 : 0) + 1] = uy; jux[(p1 ? 3 : 0) + 2] = uz; }
            }
        }
        group_lds_fence();
        // ---- every lane: the same assembly as PartFd::point (per part) + assemble_normal ----
        double blk[2][6], gv[2][3];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int e = 0; e < 6; ++e) blk[p][e] = 0.0;
            gv[p][0] = gv[p][1] = gv[p][2] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double a[3][3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int c = 0; c < 3; ++c) a[c][q] = col[(p * 3 + q) * 9 + i * 3 + c];
                const double f0 = fx[p * 9 + i * 3], f1 = fx[p * 9 + i * 3 + 1], f2 = fx[p * 9 + i * 3 + 2];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int r = 0; r <= q; ++r) blk[p][q * (q + 1) / 2 + r] += a[0][q] * a[0][r] + a[1][q] * a[1][r] + a[2][q] * a[2][r];
                    gv[p][q] += a[0][q] * f0 + a[1][q] * f1 + a[2][q] * f2;
                }
            }
        }
        double f[3], a[3][6];
#pragma unroll
        for (int c = 0; c < 3; ++c) f[c] = jux[c] - jux[3 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a[c][q] = ((jdx[q * 3 + c] - jux[3 + c]) - f[c]) * rhx[q];
                a[c][3 + q] = ((jux[c] - jdx[(3 + q) * 3 + c]) - f[c]) * rhx[3 + q];
            }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double v = wj * (a[0][i] * a[0][j] + a[1][i] * a[1][j] + a[2][i] * a[2][j]);
                if (i < 3) v += blk[0][i * (i + 1) / 2 + j];
                else if (j >= 3) v += blk[1][(i - 3) * (i - 2) / 2 + (j - 3)];
                A[i * (i + 1) / 2 + j] = v;
            }
            g[i] = wj * (a[0][i] * f[0] + a[1][i] * f[1] + a[2][i] * f[2]) + (i < 3 ? gv[0][i] : gv[1][i - 3]);
        }
        group_lds_fence();                                             // the record is reused by the next callback
    }
};

__global__ __launch_bounds__(64) void ransac_joint_lm_coop_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                                  const float *__restrict__ src, const float *__restrict__ tgt,
                                                                  const float *__restrict__ joint_dir, int niter,
                                                                  const int *__restrict__ draws, unsigned long long seed,
                                                                  double *__restrict__ models, int *__restrict__ lm_stat) {
    __shared__ double xch[64 / COOP_G][COOP_XCH];
    const int prob = blockIdx.y, lane = threadIdx.x;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    if (n0 <= 0 || n1 <= 0) return;
    const int c1 = min(niter, (int)(blockIdx.x + 1) * COOP_HYP_PER_WAVE);
    int next = blockIdx.x * COOP_HYP_PER_WAVE;          // wave-uniform: first hypothesis of the chunk not yet handed out
    const int grp = lane / COOP_G;
    const unsigned long long leaders = 0x0101010101010101ull;        // lane 0 of every group
    HypProblemCoop P;
    P.J[0] = joint_dir[prob * 3]; P.J[1] = joint_dir[prob * 3 + 1]; P.J[2] = joint_dir[prob * 3 + 2];
    P.wj = 3.0;   // min(3,3) copies of the joint axis (:134)
    P.xch = xch[grp];
    P.role = lane & (COOP_G - 1);
    Lm6 S;
    bool active = false;
    int h = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active) & leaders;     // one bit per idle group
        const int n_idle = __popcll(idle);
        if (next < c1 && n_idle > 0) {
            if (!active) {
                const int mine = next + __popcll(idle & ((1ull << (grp * COOP_G)) - 1ull));
                if (mine < c1) {
                    h = mine;
                    const double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
                    HypSamples q;
                    load_hyp_samples(src, tgt, draws, seed, prob, niter, h, a0, n0, a1, n1, q);
                    center_part3(q.s0, q.t0, (float)mo[6], P.x0, P.y0);
                    center_part3(q.s1, q.t1, (float)mo[7], P.x1, P.y1);
#pragma unroll
                    for (int i = 0; i < 6; ++i) S.x[i] = mo[i];
                    lm6_begin(P, S);
                    active = true;
                }
            }
            next += n_idle;
        } else if (n_idle == 64 / COOP_G) {
            break;
        }
        if (active && lm6_trip(P, S, 1e-4, 1e-8, 1e-8, 4200)) {
            if (P.role == 0) {
                double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
#pragma unroll
                for (int i = 0; i < 6; ++i) mo[i] = S.x[i];
                if (lm_stat) { lm_stat[((size_t)prob * niter + h) * 2] = S.info; lm_stat[((size_t)prob * niter + h) * 2 + 1] = S.nfev; }
            }
            active = false;
        }
    }
}

__global__ __launch_bounds__(64) void ransac_joint_model_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                                const float *__restrict__ src, const float *__restrict__ tgt,
                                                                int niter, const int *__restrict__ draws, unsigned long long seed,
                                                                double *__restrict__ models) {
    const int prob = blockIdx.y, h = blockIdx.x * 64 + threadIdx.x;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    if (h >= niter || n0 <= 0 || n1 <= 0) return;
    double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
    HypSamples q;
    load_hyp_samples(src, tgt, draws, seed, prob, niter, h, a0, n0, a1, n1, q);
    double x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = mo[i];
    const float sc0 = (float)mo[8], sc1 = (float)mo[9];
    double R0[9], R1[9], tr0[3], tr1[3];
    rotvec_to_mat(x, R0);
    rotvec_to_mat(x + 3, R1);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double u = 0.0, v = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            u += (double)q.t0[i][a] - (double)sc0 * (R0[a * 3] * q.s0[i][0] + R0[a * 3 + 1] * q.s0[i][1] + R0[a * 3 + 2] * q.s0[i][2]);
            v += (double)q.t1[i][a] - (double)sc1 * (R1[a * 3] * q.s1[i][0] + R1[a * 3 + 1] * q.s1[i][1] + R1[a * 3 + 2] * q.s1[i][2]);
        }
        tr0[a] = u / 3.0;
        tr1[a] = v / 3.0;
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) { mo[a] = R0[a]; mo[13 + a] = R1[a]; }
    mo[9] = sc0; mo[22] = sc1;
#pragma unroll
    for (int a = 0; a < 3; ++a) { mo[10 + a] = tr0[a]; mo[23 + a] = tr1[a]; }
}

// joint_transformation_verifier for every hypothesis: ONE WAVE per hypothesis, lanes stride over the
// points of both parts (coalesced 12-B rows from L2), inlier counts by ballot/popcount.
__global__ __launch_bounds__(256) void ransac_joint_verify_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                                 const float *__restrict__ src, const float *__restrict__ tgt,
                                                                 double th, int niter, const double *__restrict__ models,
                                                                 double *__restrict__ scores) {
    const int prob = blockIdx.y, lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= niter) return;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    if (n0 <= 0 || n1 <= 0) return;                       // score already -1
    const double *mo = models + ((size_t)prob * niter + h) * MODEL_B;
    double R0[9], R1[9], tr0[3], tr1[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) { R0[a] = mo[a]; R1[a] = mo[13 + a]; }
    const double sc0 = mo[9], sc1 = mo[22];
#pragma unroll
    for (int a = 0; a < 3; ++a) { tr0[a] = mo[10 + a]; tr1[a] = mo[23 + a]; }
    int c0 = 0, c1 = 0;
    for (int i = lane; i < n0; i += 64) {
        const float *ps = src + (size_t)(a0 + i) * 3, *pt = tgt + (size_t)(a0 + i) * 3;
        c0 += inlier_f64(R0, sc0, tr0, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2], th) ? 1 : 0;
    }
    for (int i = lane; i < n1; i += 64) {
        const float *ps = src + (size_t)(a1 + i) * 3, *pt = tgt + (size_t)(a1 + i) * 3;
        c1 += inlier_f64(R1, sc1, tr1, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2], th) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
    // (sum(inl0)/res0.shape[0] + sum(inl1)/res1.shape[0])/2 with res.shape[0] == 3 (sic, :192)
    if (lane == 0) scores[(size_t)prob * niter + h] = ((double)c0 / 3.0 + (double)c1 / 3.0) / 2.0;
}

// Workgroup-cooperative articulated problem over LDS-resident inliers (final refit :32 -> :106-184).
struct BlockProblem {
    const float (*x0)[3];
    const float (*y0)[3];
    const float (*x1)[3];
    const float (*y1)[3];
    int n0, n1;
    double J[3], wj;
    double *red;

    __device__ __forceinline__ double cost(const double x[6]) const {
        const Rod r0 = rod_prepare(x[0], x[1], x[2]), r1 = rod_prepare(x[3], x[4], x[5]);
        double s[1] = {0.0}, ox, oy, oz;
        for (int i = threadIdx.x; i < n0; i += 256) {
            rod_apply(r0, x0[i][0], x0[i][1], x0[i][2], ox, oy, oz);
            const double a = y0[i][0] - ox, b = y0[i][1] - oy, c = y0[i][2] - oz;
            s[0] += a * a + b * b + c * c;
        }
        for (int i = threadIdx.x; i < n1; i += 256) {
            rod_apply(r1, x1[i][0], x1[i][1], x1[i][2], ox, oy, oz);
            const double a = y1[i][0] - ox, b = y1[i][1] - oy, c = y1[i][2] - oz;
            s[0] += a * a + b * b + c * c;
        }
        block_sum<1>(s, red, 4);
        double ux, uy, uz, wx, wy, wz;
        rod_apply(r0, J[0], J[1], J[2], ux, uy, uz);
        rod_apply(r1, J[0], J[1], J[2], wx, wy, wz);
        return s[0] + wj * ((ux - wx) * (ux - wx) + (uy - wy) * (uy - wy) + (uz - wz) * (uz - wz));
    }
    __device__ __forceinline__ void normal(const double x[6], double A[21], double g[6]) const {
        PartFd p0, p1;
        p0.begin(x, J);
        for (int i = threadIdx.x; i < n0; i += 256) p0.point(x0[i][0], x0[i][1], x0[i][2], y0[i][0], y0[i][1], y0[i][2]);
        p1.begin(x + 3, J);
        for (int i = threadIdx.x; i < n1; i += 256) p1.point(x1[i][0], x1[i][1], x1[i][2], y1[i][0], y1[i][1], y1[i][2]);
        double v[18];
#pragma unroll
        for (int i = 0; i < 6; ++i) { v[i] = p0.blk[i]; v[9 + i] = p1.blk[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[6 + i] = p0.gv[i]; v[15 + i] = p1.gv[i]; }
        block_sum<18>(v, red, 4);
        assemble_normal(p0, v, v + 6, p1, v + 9, v + 15, wj, A, g);
    }
};

// pairwise scale sums of one part over LDS-resident inliers -> (s, s_inv) as float32
__device__ __forceinline__ void block_scales(const float (*cs)[3], const float (*ct)[3], int n, double *red, float &sc,
                                             float &sc_inv) {
    double pr[3] = {0, 0, 0};
    pair_sums(cs, ct, n, pr);
    block_sum<3>(pr, red, 4);
    sc = (float)(pr[0] / (pr[1] + 1e-6));
    sc_inv = (float)(pr[0] / (pr[2] + 1e-6));
}

// centre source, pre-scale + centre target IN PLACE (float32 like the reference's arrays); returns
// the uncentred means (for the translation) and the Kabsch rotation vector.
__device__ __forceinline__ void block_prep_part(float (*cs)[3], float (*ct)[3], int n, float sc_inv, double *red,
                                                double smean[3], double tmean[3], double rv[3]) {
    double sums[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) { sums[c] += cs[i][c]; sums[3 + c] += ct[i][c]; }
    block_sum<6>(sums, red, 4);
#pragma unroll
    for (int c = 0; c < 3; ++c) { smean[c] = sums[c] / n; tmean[c] = sums[3 + c] / n; }
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) {
        float xc[3], yc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xc[c] = cs[i][c] - (float)smean[c];
            yc[c] = sc_inv * ct[i][c] - (float)((double)sc_inv * tmean[c]);
            cs[i][c] = xc[c];
            ct[i][c] = yc[c];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a * 3 + b] += (double)yc[a] * xc[b];
    }
    block_sum<9>(M, red, 4);
    double q[4];
    horn_quat(M, q);
    quat_to_rotvec(q, rv);
}

__global__ __launch_bounds__(256) void ransac_joint_finish_kernel(const int *__restrict__ rng0, const int *__restrict__ rng1,
                                                                  const float *__restrict__ src, const float *__restrict__ tgt,
                                                                  const float *__restrict__ joint_dir, double th, int niter,
                                                                  const double *__restrict__ scores,
                                                                  const double *__restrict__ models, int max_n,
                                                                  double *__restrict__ out_model,
                                                                  unsigned char *__restrict__ out_inliers,
                                                                  int *__restrict__ out_best, double *__restrict__ out_score, FitExtras E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = (double *)smem;                      // 24*4 doubles
    int *wcnt = (int *)(red + 128);
    float(*c0s)[3] = (float(*)[3])(wcnt + 8);
    float(*c0t)[3] = c0s + max_n;
    float(*c1s)[3] = c0t + max_n;
    float(*c1t)[3] = c1s + max_n;
    const int prob = blockIdx.x;
    const int a0 = rng0[prob * 2], n0 = rng0[prob * 2 + 1] - a0;
    const int a1 = rng1[prob * 2], n1 = rng1[prob * 2 + 1] - a1;
    double *om = out_model + (size_t)prob * MODEL_B;
    unsigned char *oi0 = out_inliers + (size_t)prob * 2 * max_n, *oi1 = oi0 + max_n;
    // the record rows this joint fit reports (see FitExtras): part q + 1 always, part 0 from the cloud's first joint
    double *rec1 = nullptr, *rec0 = nullptr;
    if (E.record) {
        const int cloud = prob / (E.K - 1), q = prob - cloud * (E.K - 1);
        rec1 = E.record + ((size_t)cloud * E.K + q + 1) * 26 + MODEL_A;
        if (q == 0) rec0 = E.record + (size_t)cloud * E.K * 26 + MODEL_A;
    }
    if (n0 <= 0 || n1 <= 0 || n0 > max_n || n1 > max_n) {
        if (threadIdx.x < MODEL_B) om[threadIdx.x] = NAN;
        if (threadIdx.x < MODEL_A) { if (rec1) rec1[threadIdx.x] = NAN; if (rec0) rec0[threadIdx.x] = NAN; }
        if (E.tie && threadIdx.x < 2) E.tie[prob * 2 + threadIdx.x] = 0;
        if (threadIdx.x == 0) { out_best[prob] = -1; out_score[prob] = -1.0; }
        for (int i = threadIdx.x; i < 2 * max_n; i += 256) oi0[i] = 0;             // the whole (2, max_n) mask is written here:
        return;                                                                     // callers need not pre-clear it
    }
    for (int i = n0 + threadIdx.x; i < max_n; i += 256) oi0[i] = 0;
    for (int i = n1 + threadIdx.x; i < max_n; i += 256) oi1[i] = 0;
    double best_score;
    const int best = block_argmax_first<double>(scores + (size_t)prob * niter, niter, &best_score, red);
    const double *bm = models + ((size_t)prob * niter + best) * MODEL_B;
    double R0[9], R1[9], tr0[3], tr1[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) { R0[a] = bm[a]; R1[a] = bm[13 + a]; }
    const double hs0 = bm[9], hs1 = bm[22];
#pragma unroll
    for (int a = 0; a < 3; ++a) { tr0[a] = bm[10 + a]; tr1[a] = bm[23 + a]; }
    int m0 = 0, m1 = 0, n_border = 0, n_near = 0;
    for (int base = 0; base < n0; base += 256) {
        const int i = base + threadIdx.x;
        bool f = false, border = false;
        if (i < n0) {
            const float *ps = src + (size_t)(a0 + i) * 3, *pt = tgt + (size_t)(a0 + i) * 3;
            const double rs = residual_sq_f64(R0, hs0, tr0, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2]);
            f = rs < th;
            border = rs >= E.lo_d && rs < E.hi_d;
            oi0[i] = f ? 1 : 0;
        }
        n_border += __popcll(__ballot(border));
        m0 = compact_flagged(f, i, n0, src, tgt, (size_t)a0, c0s, c0t, m0, wcnt);
    }
    for (int base = 0; base < n1; base += 256) {
        const int i = base + threadIdx.x;
        bool f = false, border = false;
        if (i < n1) {
            const float *ps = src + (size_t)(a1 + i) * 3, *pt = tgt + (size_t)(a1 + i) * 3;
            const double rs = residual_sq_f64(R1, hs1, tr1, ps[0], ps[1], ps[2], pt[0], pt[1], pt[2]);
            f = rs < th;
            border = rs >= E.lo_d && rs < E.hi_d;
            oi1[i] = f ? 1 : 0;
        }
        n_border += __popcll(__ballot(border));
        m1 = compact_flagged(f, i, n1, src, tgt, (size_t)a1, c1s, c1t, m1, wcnt);
    }
    if (E.tie) {                                       // block-uniform; one inlier of either part moves the joint score by 1/6 (:192)
        const double *sp = scores + (size_t)prob * niter;
        const double near = best_score - (1.0 / 6.0 + 1e-9);
        for (int h = threadIdx.x; h < niter + 255 - (niter + 255) % 256; h += 256) {
            bool degenerate = false;
            if (h < niter && sp[h] >= near) {
                int i0[3], i1[3];
                load_draw3(E.draws, E.seed, prob, niter, h, 0, 6, n0, i0);
                load_draw3(E.draws, E.seed, prob, niter, h, 3, 6, n1, i1);
                degenerate = i0[0] == i0[1] || i0[0] == i0[2] || i0[1] == i0[2] || i1[0] == i1[1] || i1[0] == i1[2] || i1[1] == i1[2];
            }
            n_near += __popcll(__ballot(degenerate));
        }
        if ((threadIdx.x & 63) == 0) { wcnt[4 + (threadIdx.x >> 6)] = n_border; red[8 + (threadIdx.x >> 6)] = (double)n_near; }
        __syncthreads();
        if (threadIdx.x == 0) {
            E.tie[prob * 2] = wcnt[4] + wcnt[5] + wcnt[6] + wcnt[7];
            E.tie[prob * 2 + 1] = (int)(red[8] + red[9] + red[10] + red[11]);
        }
    }
    __syncthreads();
    if (m0 == 0 || m1 == 0) {   // the reference would produce NaNs (mean of an empty selection)
        if (threadIdx.x < MODEL_B) om[threadIdx.x] = NAN;
        if (threadIdx.x < MODEL_A) { if (rec1) rec1[threadIdx.x] = NAN; if (rec0) rec0[threadIdx.x] = NAN; }
        if (threadIdx.x == 0) { out_best[prob] = best; out_score[prob] = best_score; }
        return;
    }
    float sc0, sc0i, sc1, sc1i;
    block_scales(c0s, c0t, m0, red, sc0, sc0i);
    block_scales(c1s, c1t, m1, red, sc1, sc1i);
    double sm0[3], tm0[3], sm1[3], tm1[3], x[6];
    block_prep_part(c0s, c0t, m0, sc0i, red, sm0, tm0, x);
    block_prep_part(c1s, c1t, m1, sc1i, red, sm1, tm1, x + 3);
    __syncthreads();
    BlockProblem P;
    P.x0 = c0s; P.y0 = c0t; P.x1 = c1s; P.y1 = c1t;
    P.n0 = m0; P.n1 = m1;
    P.J[0] = joint_dir[prob * 3]; P.J[1] = joint_dir[prob * 3 + 1]; P.J[2] = joint_dir[prob * 3 + 2];
    P.wj = (double)(m0 < m1 ? m0 : m1);
    P.red = red;
    lmdif6(P, x, 1e-4, 1e-8, 1e-8, 4200, nullptr);
    rotvec_to_mat(x, R0);
    rotvec_to_mat(x + 3, R1);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 9; ++a) { om[a] = R0[a]; om[13 + a] = R1[a]; }
        om[9] = sc0; om[22] = sc1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // mean(target - scale*R*source) over the inliers
            om[10 + a] = tm0[a] - (double)sc0 * (R0[a * 3] * sm0[0] + R0[a * 3 + 1] * sm0[1] + R0[a * 3 + 2] * sm0[2]);
            om[23 + a] = tm1[a] - (double)sc1 * (R1[a * 3] * sm1[0] + R1[a * 3 + 1] * sm1[1] + R1[a * 3 + 2] * sm1[2]);
        }
        out_best[prob] = best;
        out_score[prob] = best_score;
#pragma unroll
        for (int a = 0; a < MODEL_A; ++a) {
            if (rec1) rec1[a] = om[MODEL_A + a];
            if (rec0) rec0[a] = om[a];
        }
    }
}

// ================================ glue kernels =====================================================
// labels = argmax(W, axis=1) (first maximum wins, np.argmax); ordered partition of the cloud's points
// by label; gather of the per-part source (own part-NOCS slot) / target (camera point) arrays
// (evaluation/parallel_ancsh_pose.py:238-242,259-260).
__global__ __launch_bounds__(256) void partition_kernel(int n, int K, const float *__restrict__ W, const float *__restrict__ P,
                                                        const float *__restrict__ nocs, int *__restrict__ labels,
                                                        int *__restrict__ part_index, int *__restrict__ off,
                                                        float *__restrict__ src, float *__restrict__ tgt,
                                                        int *__restrict__ counts, int *__restrict__ rng0, int *__restrict__ rng1) {
    __shared__ int wcnt[4];
    __shared__ int part_start[17];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *Wb = W + (size_t)b * n * K;
    int base = 0;
    for (int j = 0; j < K; ++j) {
        if (threadIdx.x == 0) { off[b * K + j] = b * n + base; part_start[j] = b * n + base; }
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int i = c0 + threadIdx.x;
            bool f = false;
            if (i < n) {
                int lab = 0;
                float bv = Wb[(size_t)i * K];
                for (int k = 1; k < K; ++k) {
                    const float v = Wb[(size_t)i * K + k];
                    if (v > bv) { bv = v; lab = k; }
                }
                f = lab == j;
                if (f && labels) labels[(size_t)b * n + i] = lab;
            }
            const unsigned long long m = __ballot(f);
            __syncthreads();
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int start = base;
            for (int w = 0; w < wave; ++w) start += wcnt[w];
            if (f) {
                const int pos = start + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                const size_t row = (size_t)b * n + pos;
                part_index[row] = i;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    src[row * 3 + c] = nocs[((size_t)b * n + i) * 3 * K + 3 * j + c];
                    tgt[row * 3 + c] = P[((size_t)b * n + i) * 3 + c];
                }
            }
            base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        }
    }
    if (threadIdx.x == 0 && b == gridDim.x - 1) off[gridDim.x * K] = gridDim.x * n;
    // optional by-products the joint stage consumes (:274-283): points per part and the [start, end) row ranges of part 0 /
    // part j for every joint j = 1..K-1 -- written here instead of by slicing / stacking `off` in separate launches
    if (threadIdx.x == 0) part_start[K] = (b + 1) * n;
    __syncthreads();
    if ((int)threadIdx.x < K && counts) counts[b * K + threadIdx.x] = part_start[threadIdx.x + 1] - part_start[threadIdx.x];
    if ((int)threadIdx.x >= 1 && (int)threadIdx.x < K && rng0 && rng1) {
        const int q = b * (K - 1) + (int)threadIdx.x - 1;
        rng0[q * 2] = part_start[0]; rng0[q * 2 + 1] = part_start[1];
        rng1[q * 2] = part_start[threadIdx.x]; rng1[q * 2 + 1] = part_start[threadIdx.x + 1];
    }
}

// A cloud with a non-finite value anywhere in the fit's inputs has no defined pose (the reference's own np.linalg.svd raises
// LinAlgError on it, np.argmax / np.median of NaN rows are arbitrary): its (K, 26) record rows become NaN, whatever the fit kernels
// made of it -- a poisoned cloud never yields a silently finite answer.  One workgroup per cloud; axis may be NULL.
__global__ __launch_bounds__(1024) void poison_records_kernel(int n, int K, const float *__restrict__ P, const float *__restrict__ nocs,
                                                              const float *__restrict__ W, const float *__restrict__ axis,
                                                              double *__restrict__ record) {
    const int b = blockIdx.x;
    bool bad = false;
    // 1024 threads, eight independent loads in flight per thread: the scan is a handful of memory latencies, not a per-element loop
    const auto scan = [&](const float *p, long cnt) {
        for (long i0 = threadIdx.x; i0 < cnt; i0 += 8 * 1024) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i0 + u * 1024 < cnt ? i0 + u * 1024 : i0];
#pragma unroll
            for (int u = 0; u < 8; ++u) bad = bad | !(fabsf(v[u]) <= 3.4028234664e38f);      // NaN and +-Inf fail the compare
        }
    };
    scan(P + (size_t)b * n * 3, (long)n * 3);
    scan(nocs + (size_t)b * n * 3 * K, (long)n * 3 * K);
    scan(W + (size_t)b * n * K, (long)n * K);
    if (axis) scan(axis + (size_t)b * n * 3, (long)n * 3);
    if (__syncthreads_or(bad))
        for (int i = threadIdx.x; i < K * 26; i += 1024) record[(size_t)b * K * 26 + i] = __builtin_nan("");
}

// jt_axis = np.median(joint_axis_per_point[joint_cls == j], 0)  (:295): one workgroup per (cloud, joint)
__global__ __launch_bounds__(256) void joint_direction_kernel(int n, int K, const float *__restrict__ axis,
                                                              const int *__restrict__ joint_cls, float *__restrict__ out) {
    extern __shared__ float vals[];   // 3 * npow2 floats
    __shared__ int wcnt[4];
    const int b = blockIdx.x, j = blockIdx.y + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const bool f = i < n && joint_cls[(size_t)b * n + i] == j;
        const unsigned long long m = __ballot(f);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int start = cnt;
        for (int w = 0; w < wave; ++w) start += wcnt[w];
        if (f) {
            const int pos = start + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
#pragma unroll
            for (int c = 0; c < 3; ++c) vals[c * npow2 + pos] = axis[((size_t)b * n + i) * 3 + c];
        }
        cnt += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
    int p2 = 1;
    while (p2 < cnt) p2 <<= 1;
    for (int e = cnt + threadIdx.x; e < p2; e += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) vals[c * npow2 + e] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int e = threadIdx.x; e < p2; e += 256) {
                const int partner = e ^ s;
                if (partner > e) {
                    const bool up = (e & k) == 0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float *v = vals + c * npow2;
                        const float a = v[e], bb = v[partner];
                        if ((a > bb) == up) { v[e] = bb; v[partner] = a; }
                    }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x < 3) {
        const float *v = vals + threadIdx.x * npow2;
        float med = NAN;
        if (cnt > 0) med = (cnt & 1) ? v[cnt / 2] : (v[cnt / 2 - 1] + v[cnt / 2]) * 0.5f;
        out[((size_t)b * (K - 1) + (j - 1)) * 3 + threadIdx.x] = med;
    }
}

// estimateSimilarityUmeyama (lib/aligning.py:580-622) for a batch of (source, target) point sets, float64.
// out (nprob, 29): Scales(3) | Rotation(9, the reference's TRANSPOSED matrix) | Translation(3) | OutTransform(4x4 row-major minus last row -> 12) ... see header
__global__ __launch_bounds__(256) void umeyama_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                      const float *__restrict__ tgt, double *__restrict__ out) {
    __shared__ double red[64];
    const int prob = blockIdx.x, r0 = off[prob], n = off[prob + 1] - r0;
    double *o = out + (size_t)prob * 32;
    if (n <= 0) {
        if (threadIdx.x < 32) o[threadIdx.x] = NAN;
        return;
    }
    double sums[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) { sums[c] += src[(size_t)(r0 + i) * 3 + c]; sums[3 + c] += tgt[(size_t)(r0 + i) * 3 + c]; }
    block_sum<6>(sums, red, 4);
    double sm[3], tm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { sm[c] = sums[c] / n; tm[c] = sums[3 + c] / n; }
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // Cov*n (9) + sum |s - sm|^2
    for (int i = threadIdx.x; i < n; i += 256) {
        double sc[3], tc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { sc[c] = src[(size_t)(r0 + i) * 3 + c] - sm[c]; tc[c] = tgt[(size_t)(r0 + i) * 3 + c] - tm[c]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a * 3 + b] += tc[a] * sc[b];
        acc[9] += sc[0] * sc[0] + sc[1] * sc[1] + sc[2] * sc[2];
    }
    block_sum<10>(acc, red, 4);
    if (threadIdx.x != 0) return;
    double M[9], q[4], R[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) M[a] = acc[a] / n;
    horn_quat(M, q);
    quat_to_mat(q, R);                       // R = U diag(1,1,d) Vh
    // sum of the (sign-corrected) singular values = tr(R^T Cov)
    double trace = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) trace += R[a * 3 + b] * M[a * 3 + b];
    const double varP = acc[9] / n;
    const double s = trace / varP;
    o[0] = o[1] = o[2] = s;
    // Rotation = (U Vh)^T ; Translation = tmean - smean . (s * Rotation) = tmean - s * R smean
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) o[3 + a * 3 + b] = R[b * 3 + a];
    double T[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        T[a] = tm[a] - s * (R[a * 3] * sm[0] + R[a * 3 + 1] * sm[1] + R[a * 3 + 2] * sm[2]);
        o[12 + a] = T[a];
    }
    // OutTransform = [[s*R, T],[0,0,0,1]] row-major 4x4
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) o[15 + a * 4 + b] = s * R[a * 3 + b];
        o[15 + a * 4 + 3] = T[a];
    }
    o[27] = 0; o[28] = 0; o[29] = 0; o[30] = 1; o[31] = 0;
}


// estimateSimilarityTransform (lib/aligning.py:17-32): 5-point Umeyama RANSAC (getRANSACInliers :485-507 with
// evaluateModel :540-547 and set_config :88-103), <= 100 iterations with the reference's SEQUENTIAL early stop,
// then Umeyama on the winner's inliers.  One workgroup per problem: thread h evaluates hypothesis h against all
// points (every hypothesis is independent), thread 0 then replays the reference's in-order bookkeeping
// (strictly-better inlier ratio wins; stop as soon as the best residual drops under StopThreshold) to find which
// hypothesis the sequential loop would have kept; the workgroup refits on its inliers.
// Quirk kept: nInliers = np.count_nonzero(InlierIdx) counts non-zero INDICES, so point 0 never counts (:544).
__device__ __forceinline__ void umeyama_small(const double (*s)[3], const double (*t)[3], int n, double T[12]) {
    double sm[3] = {0, 0, 0}, tm[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) { sm[c] += s[i][c]; tm[c] += t[i][c]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { sm[c] /= n; tm[c] /= n; }
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a * 3 + b] += (t[i][a] - tm[a]) * (s[i][b] - sm[b]);
            var += (s[i][a] - sm[a]) * (s[i][a] - sm[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) M[a] /= n;
    var /= n;
    double q[4], R[9];
    horn_quat(M, q);
    quat_to_mat(q, R);
    double tr = 0.0;
#pragma unroll
    for (int a = 0; a < 9; ++a) tr += R[a] * M[a];
    const double sc = tr / var;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) T[a * 4 + b] = sc * R[a * 3 + b];
        T[a * 4 + 3] = tm[a] - sc * (R[a * 3] * sm[0] + R[a * 3 + 1] * sm[1] + R[a * 3 + 2] * sm[2]);
    }
}

__global__ __launch_bounds__(128) void ransac_umeyama5_kernel(const int *__restrict__ off, const float *__restrict__ src,
                                                              const float *__restrict__ tgt, int niter,
                                                              const int *__restrict__ draws, unsigned long long seed,
                                                              double *__restrict__ out, int *__restrict__ status) {
    __shared__ double red[64];
    __shared__ double h_res[128], h_ratio[128];
    __shared__ double Tsel[12];
    __shared__ double thr[2];
    __shared__ int sel;
    const int prob = blockIdx.x, r0 = off[prob], n = off[prob + 1] - r0, h = threadIdx.x;
    double *o = out + (size_t)prob * 32;
    if (n <= 0 || niter > 128) {
        if (h < 32) o[h] = NAN;
        if (h == 0) status[prob] = -1;
        return;
    }
    // set_config: PassT = max(TargetNorm/SourceNorm, SourceNorm/TargetNorm) of the mean point norms; StopT = PassT/100
    double nm[2] = {0, 0};
    for (int i = h; i < n; i += 128) {
        const float *ps = src + (size_t)(r0 + i) * 3, *pt = tgt + (size_t)(r0 + i) * 3;
        nm[0] += sqrt((double)ps[0] * ps[0] + (double)ps[1] * ps[1] + (double)ps[2] * ps[2]);
        nm[1] += sqrt((double)pt[0] * pt[0] + (double)pt[1] * pt[1] + (double)pt[2] * pt[2]);
    }
    block_sum<2>(nm, red, 2);
    if (h == 0) {
        const double ts = (nm[1] / n) / (nm[0] / n), st = (nm[0] / n) / (nm[1] / n);
        thr[0] = st > ts ? st : ts;
        thr[1] = thr[0] / 100.0;
    }
    __syncthreads();
    const double passT = thr[0], stopT = thr[1];
    double T[12];
    if (h < niter) {
        double s5[5][3], t5[5][3];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int v = draws ? draws[((size_t)prob * niter + h) * 5 + k] : device_draw(seed, prob, h, k, n);
            v = v < 0 ? 0 : (v >= n ? n - 1 : v);
#pragma unroll
            for (int c = 0; c < 3; ++c) { s5[k][c] = src[(size_t)(r0 + v) * 3 + c]; t5[k][c] = tgt[(size_t)(r0 + v) * 3 + c]; }
        }
        umeyama_small(s5, t5, 5, T);
        double rs = 0.0;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const float *ps = src + (size_t)(r0 + i) * 3, *pt = tgt + (size_t)(r0 + i) * 3;
            double e2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double e = (double)pt[a] - (T[a * 4] * ps[0] + T[a * 4 + 1] * ps[1] + T[a * 4 + 2] * ps[2] + T[a * 4 + 3]);
                e2 += e * e;
            }
            rs += e2;                                   // Residual = norm(ResidualVec) = sqrt(sum of squared norms)
            if (sqrt(e2) < passT && i != 0) ++cnt;      // count_nonzero(InlierIdx): index 0 is never counted
        }
        h_res[h] = sqrt(rs);
        h_ratio[h] = (double)cnt / (double)n;
    }
    __syncthreads();
    if (h == 0) {
        double bestRes = 1e10, bestRatio = 0.0;
        int best = -1;                                   // -1: BestInlierIdx stays arange(n)
        for (int i = 0; i < niter; ++i) {
            if (h_ratio[i] > bestRatio) { bestRes = h_res[i]; bestRatio = h_ratio[i]; best = i; }
            if (bestRes < stopT) break;
        }
        sel = best;
        thr[0] = bestRatio;
    }
    __syncthreads();
    const int best = sel;
    const double bestRatio = thr[0];
    if (best >= 0 && h == best)
#pragma unroll
        for (int a = 0; a < 12; ++a) Tsel[a] = T[a];
    __syncthreads();
    if (bestRatio < 0.1) {                              // "[ WARN ] - Something is wrong. Small BestInlierRatio" -> 4 x None
        if (h < 32) o[h] = NAN;
        if (h == 0) status[prob] = 1;
        return;
    }
    // Umeyama over the winner's inliers (true inliers INCLUDING index 0: InlierIdx[0] holds it)
    double sums[7] = {0, 0, 0, 0, 0, 0, 0};
    auto is_in = [&](int i) {
        if (best < 0) return true;
        const float *ps = src + (size_t)(r0 + i) * 3, *pt = tgt + (size_t)(r0 + i) * 3;
        double e2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double e = (double)pt[a] - (Tsel[a * 4] * ps[0] + Tsel[a * 4 + 1] * ps[1] + Tsel[a * 4 + 2] * ps[2] + Tsel[a * 4 + 3]);
            e2 += e * e;
        }
        return sqrt(e2) < passT;
    };
    for (int i = h; i < n; i += 128)
        if (is_in(i)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { sums[c] += src[(size_t)(r0 + i) * 3 + c]; sums[3 + c] += tgt[(size_t)(r0 + i) * 3 + c]; }
            sums[6] += 1.0;
        }
    block_sum<7>(sums, red, 2);
    const double m = sums[6];
    double sm[3], tm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { sm[c] = sums[c] / m; tm[c] = sums[3 + c] / m; }
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = h; i < n; i += 128)
        if (is_in(i)) {
            double sc[3], tc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { sc[c] = src[(size_t)(r0 + i) * 3 + c] - sm[c]; tc[c] = tgt[(size_t)(r0 + i) * 3 + c] - tm[c]; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a * 3 + b] += tc[a] * sc[b];
            acc[9] += sc[0] * sc[0] + sc[1] * sc[1] + sc[2] * sc[2];
        }
    block_sum<10>(acc, red, 2);
    if (h != 0) return;
    double M[9], q[4], R[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) M[a] = acc[a] / m;
    horn_quat(M, q);
    quat_to_mat(q, R);
    double trace = 0.0;
#pragma unroll
    for (int a = 0; a < 9; ++a) trace += R[a] * M[a];
    const double s = trace / (acc[9] / m);
    o[0] = o[1] = o[2] = s;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) o[3 + a * 3 + b] = R[b * 3 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double Tv = tm[a] - s * (R[a * 3] * sm[0] + R[a * 3 + 1] * sm[1] + R[a * 3 + 2] * sm[2]);
        o[12 + a] = Tv;
#pragma unroll
        for (int b = 0; b < 3; ++b) o[15 + a * 4 + b] = s * R[a * 3 + b];
        o[15 + a * 4 + 3] = Tv;
    }
    o[27] = 0; o[28] = 0; o[29] = 0; o[30] = 1; o[31] = bestRatio;
    status[prob] = 0;
}

}  // namespace pose
}  // namespace ancsh

using namespace ancsh;
using namespace ancsh::pose;

// T = min{x >= 0 : sqrt(x) >= th} in the given precision (host libm sqrt is correctly rounded, like the
// reference's numpy sqrt), so that  sqrt(s) < th  <=>  s < T  exactly.
static float sq_threshold_f32(float th) {
    float t = th * th;
    while (sqrtf(t) >= th && t > 0.f) t = nextafterf(t, 0.f);
    while (sqrtf(t) < th) t = nextafterf(t, INFINITY);
    return t;
}
static double sq_threshold_f64(double th) {
    double t = th * th;
    while (sqrt(t) >= th && t > 0.0) t = nextafter(t, 0.0);
    while (sqrt(t) < th) t = nextafter(t, INFINITY);
    return t;
}

extern "C" int ancsh_pose_partition(int b, int n, int K, const float *W, const float *P, const float *nocs, int *labels,
                                    int *part_index, int *off, float *src, float *tgt, int *counts, int *rng0, int *rng1,
                                    void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && K >= 1 && K <= 16, "pose_partition: bad shape b=%d n=%d K=%d", b, n, K);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(W && P && nocs && part_index && off && src && tgt, "pose_partition: null pointer");
    ANCSH_REQUIRE((rng0 == nullptr) == (rng1 == nullptr), "pose_partition: rng0 and rng1 go together");
    hipLaunchKernelGGL(partition_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, n, K, W, P, nocs, labels, part_index, off,
                       src, tgt, counts, rng0, rng1);
    return check_launch("pose_partition");
}

extern "C" int ancsh_pose_poison_records(int b, int n, int K, const float *P, const float *nocs, const float *W, const float *joint_axis,
                                         double *record, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && K >= 1 && K <= 16, "pose_poison_records: bad shape b=%d n=%d K=%d", b, n, K);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(P && nocs && W && record, "pose_poison_records: null pointer");
    hipLaunchKernelGGL(poison_records_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, K, P, nocs, W, joint_axis, record);
    return check_launch("pose_poison_records");
}

extern "C" int ancsh_pose_joint_direction(int b, int n, int K, const float *joint_axis, const int *joint_cls, float *out,
                                          void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && K >= 1, "pose_joint_direction: bad shape");
    ANCSH_REQUIRE(n <= 8192, "pose_joint_direction: n %d > 8192", n);
    if (b == 0 || K == 1) return ANCSH_OK;
    ANCSH_REQUIRE(joint_axis && joint_cls && out, "pose_joint_direction: null pointer");
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    const size_t lds = (size_t)3 * npow2 * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)joint_direction_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(joint_direction_kernel, dim3(b, K - 1), dim3(256), lds, (hipStream_t)stream, n, K, joint_axis, joint_cls, out);
    return check_launch("pose_joint_direction");
}

static FitExtras no_extras() {
    FitExtras E;
    E.record = nullptr; E.K = 0; E.tie = nullptr;
    E.lo_f = E.hi_f = 0.f; E.lo_d = E.hi_d = 0.0;      // empty window: no point is "borderline"
    E.draws = nullptr; E.seed = 0;
    return E;
}

// min_K: 1 for the per-part fits (nprob = B * K), 2 for the joint fits (nprob = B * (K - 1)).  The tie window is an ABSOLUTE half-width
// on the residual norm: a point counts when th - w <= |res| < th + w, evaluated on the squared norm the verifiers already compute.
static int make_extras(const char *who, int nprob, double *record, int K, int min_K, int *tie, double th, double window, FitExtras &E) {
    E.record = record; E.K = K; E.tie = tie;
    if (record) {
        ANCSH_REQUIRE(K >= min_K && nprob % (K - (min_K - 1)) == 0, "%s: record needs K >= %d and nprob (%d) a multiple of %s (K = %d)", who, min_K, nprob,
                      min_K == 1 ? "K" : "K - 1", K);
    }
    if (tie) {
        ANCSH_REQUIRE(window >= 0.0 && window < th, "%s: tie_window %g must be in [0, inlier_th)", who, window);
        const double lo = (th - window) * (th - window), hi = (th + window) * (th + window);
        E.lo_f = (float)lo; E.hi_f = (float)hi; E.lo_d = lo; E.hi_d = hi;
    }
    return ANCSH_OK;
}

static long single_quads_needed(long rows, int nprob) { return 2 * ((rows + 7) / 8 + nprob) + 2; }     // quads (24 floats each)

extern "C" long ancsh_ransac_single_quads_floats(long rows, int nprob) {
    if (rows < 0 || nprob < 0) return -1;
    return 24 * single_quads_needed(rows, nprob);
}

static int ransac_single_impl(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                              const int *draws, unsigned long long seed, int max_n, double *out_model,
                              unsigned char *out_inliers, int *out_best, int *scratch_scores, float *scratch_quads, long rows,
                              FitExtras E, void *stream) {
    ANCSH_REQUIRE(nprob >= 0 && niter > 0 && max_n > 0, "ransac_single: bad sizes nprob=%d niter=%d max_n=%d", nprob, niter, max_n);
    ANCSH_REQUIRE(max_n <= 6144, "ransac_single: max_n %d > 6144 (refit keeps the inliers in LDS)", max_n);
    if (nprob == 0) return ANCSH_OK;
    ANCSH_REQUIRE(off && src && tgt && out_model && out_inliers && out_best && scratch_scores, "ransac_single: null pointer");
    hipStream_t st = (hipStream_t)stream;
    ANCSH_REQUIRE(inlier_th > 0.f, "ransac_single: inlier_th must be positive");
    inlier_th = sq_threshold_f32(inlier_th);     // the kernels compare squared residuals
    // the scalar-register kernel counts inliers as clamp((th_sq - s) * 2^126): exact while th_sq - s cannot be denormal, i.e. for any
    // threshold a fit would use; a squared threshold below 2^-100 or >= 4 (th_sq * 2^126 must stay finite) takes the compare-based kernel
    // (same scores by construction)
    if (scratch_quads && inlier_th >= 0x1p-100f && inlier_th < 4.0f) {
        ANCSH_REQUIRE(rows >= 0 && rows < (1L << 30), "ransac_single_ex: rows=%ld out of range", rows);
        ANCSH_REQUIRE((((uintptr_t)scratch_quads) & 31) == 0, "ransac_single_ex: scratch_quads must be 32-byte aligned");
        const int cap = (int)single_quads_needed(rows, nprob);
        hipLaunchKernelGGL(soa_quads_kernel, dim3((max_n + 7 + 255) / 256, nprob), dim3(256), 0, st, off, src, tgt, scratch_quads, cap);
        hipLaunchKernelGGL(ransac_single_score_sreg_kernel, dim3((niter + 255) / 256, nprob), dim3(256), 0, st, off, src, tgt,
                           (const float *)scratch_quads, cap, inlier_th, niter, draws, seed, scratch_scores);
    } else {
        hipLaunchKernelGGL(ransac_single_score_kernel, dim3((niter + 255) / 256, nprob), dim3(256), 0, st, off, src, tgt, inlier_th,
                           niter, draws, seed, scratch_scores);
    }
    const size_t lds = 64 * sizeof(double) + 8 * sizeof(int) + (size_t)2 * max_n * 3 * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)ransac_single_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ransac_single_finish_kernel, dim3(nprob), dim3(256), lds, st, off, src, tgt, inlier_th, niter, draws, seed,
                       scratch_scores, max_n, out_model, out_inliers, out_best, E);
    return check_launch("ransac_single");
}

extern "C" int ancsh_ransac_single(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                                   const int *draws, unsigned long long seed, int max_n, double *out_model,
                                   unsigned char *out_inliers, int *out_best, int *scratch_scores, void *stream) {
    return ransac_single_impl(nprob, off, src, tgt, inlier_th, niter, draws, seed, max_n, out_model, out_inliers, out_best,
                              scratch_scores, nullptr, 0, no_extras(), stream);
}

// The same call with the hypotheses scored from SCALAR registers: scratch_quads (32-byte aligned, ancsh_ransac_single_quads_floats(rows,
// nprob) floats, rows >= off[nprob]) receives a padded four-points-per-record copy of the parts; scores, winner, mask and model are
// those of ancsh_ransac_single bit for bit.
extern "C" int ancsh_ransac_single_ex(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                                      const int *draws, unsigned long long seed, int max_n, double *out_model,
                                      unsigned char *out_inliers, int *out_best, int *scratch_scores, float *scratch_quads,
                                      long rows, void *stream) {
    ANCSH_REQUIRE(scratch_quads, "ransac_single_ex: scratch_quads is NULL (ancsh_ransac_single is the call without it)");
    return ransac_single_impl(nprob, off, src, tgt, inlier_th, niter, draws, seed, max_n, out_model, out_inliers, out_best,
                              scratch_scores, scratch_quads, rows, no_extras(), stream);
}

// ancsh_ransac_single_ex (scratch_quads != NULL) / ancsh_ransac_single (NULL) that ALSO writes each fit straight into the pose record
// and / or reports how tie-sensitive it is (FitExtras above; include/ancsh_hip.h).  Everything else bit for bit as before.
extern "C" int ancsh_ransac_single_rec(int nprob, const int *off, const float *src, const float *tgt, float inlier_th, int niter,
                                       const int *draws, unsigned long long seed, int max_n, double *out_model,
                                       unsigned char *out_inliers, int *out_best, int *scratch_scores, float *scratch_quads,
                                       long rows, double *record, int K, int *tie_stats, float tie_window, void *stream) {
    FitExtras E = no_extras();
    if (int rc = make_extras("ransac_single_rec", nprob, record, K, 1, tie_stats, (double)inlier_th, (double)tie_window, E)) return rc;
    return ransac_single_impl(nprob, off, src, tgt, inlier_th, niter, draws, seed, max_n, out_model, out_inliers, out_best,
                              scratch_scores, scratch_quads, rows, E, stream);
}

static int ransac_joint_impl(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                             const float *joint_dir, double inlier_th, int niter, const int *draws,
                             unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                             int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                             int *lm_stat, int lm_schedule, FitExtras E, void *stream) {
    ANCSH_REQUIRE(lm_schedule >= ANCSH_LM_AUTO && lm_schedule <= ANCSH_LM_LATENCY, "ransac_joint: unknown lm_schedule %d", lm_schedule);
    ANCSH_REQUIRE(nprob >= 0 && niter > 0 && max_n > 0, "ransac_joint: bad sizes");
    ANCSH_REQUIRE(max_n <= 3072, "ransac_joint: max_n %d > 3072 (refit keeps both parts' inliers in LDS)", max_n);
    if (nprob == 0) return ANCSH_OK;
    ANCSH_REQUIRE(rng0 && rng1 && src && tgt && joint_dir && out_model && out_inliers && out_best && out_score &&
                      scratch_scores && scratch_models, "ransac_joint: null pointer");
    hipStream_t st = (hipStream_t)stream;
    ANCSH_REQUIRE(inlier_th > 0.0, "ransac_joint: inlier_th must be positive");
    inlier_th = sq_threshold_f64(inlier_th);      // the kernels compare squared residuals
    const dim3 per_hyp((niter + 63) / 64, nprob);
    hipLaunchKernelGGL(ransac_joint_init_kernel, per_hyp, dim3(64), 0, st, rng0, rng1, src, tgt, niter, draws, seed, scratch_scores,
                       scratch_models);
    // Two schedules of the same fits (identical MINPACK state machine, results equal to ~1e-7, not to the last bit: the eight-lane
    // callbacks are a different instruction stream and the f64 code is compiled with contraction on):
    //   * lane per fit: least SIMD time per fit -- the throughput schedule, and what ANCSH_LM_AUTO ALWAYS takes, so that a cloud's
    //     result does not depend on how many clouds share the launch.  Only the number of hypotheses handed to a wave follows the
    //     launch size (a small launch spreads over 4x more waves); the order in which hypotheses run enters no result;
    //   * eight lanes per fit (ANCSH_LM_LATENCY, explicit only): the long fits that set the launch's duration run ~1.25x faster
    //     (measured on 64 x 200 fits: 1.27 vs 1.6 ms; the tail is MINPACK's serial lmpar on rank-deficient samples, which no lane
    //     split shortens) at ~5 % lower pipeline throughput.
    if (lm_schedule == ANCSH_LM_LATENCY) {
        hipLaunchKernelGGL(ransac_joint_lm_coop_kernel, dim3((niter + COOP_HYP_PER_WAVE - 1) / COOP_HYP_PER_WAVE, nprob), dim3(64), 0, st,
                           rng0, rng1, src, tgt, joint_dir, niter, draws, seed, scratch_models, lm_stat);
    } else {
        const int chunk = (long)nprob * niter <= HYP_SMALL_LAUNCH ? HYP_CHUNK_SMALL : HYP_CHUNK;
        hipLaunchKernelGGL(ransac_joint_lm_kernel, dim3((niter + chunk - 1) / chunk, nprob), dim3(64), 0, st, rng0, rng1, src, tgt,
                           joint_dir, niter, draws, seed, scratch_models, lm_stat, chunk);
    }
    hipLaunchKernelGGL(ransac_joint_model_kernel, per_hyp, dim3(64), 0, st, rng0, rng1, src, tgt, niter, draws, seed, scratch_models);
    hipLaunchKernelGGL(ransac_joint_verify_kernel, dim3((niter + 3) / 4, nprob), dim3(256), 0, st, rng0, rng1, src, tgt, inlier_th,
                       niter, scratch_models, scratch_scores);
    const size_t lds = 128 * sizeof(double) + 8 * sizeof(int) + (size_t)4 * max_n * 3 * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)ransac_joint_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ransac_joint_finish_kernel, dim3(nprob), dim3(256), lds, st, rng0, rng1, src, tgt, joint_dir, inlier_th,
                       niter, scratch_scores, scratch_models, max_n, out_model, out_inliers, out_best, out_score, E);
    return check_launch("ransac_joint");
}

extern "C" int ancsh_ransac_joint(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                                  const float *joint_dir, double inlier_th, int niter, const int *draws,
                                  unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                                  int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                                  int *lm_stat, void *stream) {
    return ransac_joint_impl(nprob, rng0, rng1, src, tgt, joint_dir, inlier_th, niter, draws, seed, max_n, out_model, out_inliers,
                             out_best, out_score, scratch_scores, scratch_models, lm_stat, ANCSH_LM_AUTO, no_extras(), stream);
}

extern "C" int ancsh_ransac_joint_ex(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                                     const float *joint_dir, double inlier_th, int niter, const int *draws,
                                     unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                                     int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                                     int *lm_stat, int lm_schedule, void *stream) {
    return ransac_joint_impl(nprob, rng0, rng1, src, tgt, joint_dir, inlier_th, niter, draws, seed, max_n, out_model, out_inliers,
                             out_best, out_score, scratch_scores, scratch_models, lm_stat, lm_schedule, no_extras(), stream);
}

extern "C" int ancsh_ransac_joint_rec(int nprob, const int *rng0, const int *rng1, const float *src, const float *tgt,
                                      const float *joint_dir, double inlier_th, int niter, const int *draws,
                                      unsigned long long seed, int max_n, double *out_model, unsigned char *out_inliers,
                                      int *out_best, double *out_score, double *scratch_scores, double *scratch_models,
                                      int *lm_stat, int lm_schedule, double *record, int K, int *tie_stats, double tie_window,
                                      void *stream) {
    FitExtras E = no_extras();
    if (int rc = make_extras("ransac_joint_rec", nprob, record, K, 2, tie_stats, inlier_th, tie_window, E)) return rc;
    E.draws = draws; E.seed = seed;
    return ransac_joint_impl(nprob, rng0, rng1, src, tgt, joint_dir, inlier_th, niter, draws, seed, max_n, out_model, out_inliers,
                             out_best, out_score, scratch_scores, scratch_models, lm_stat, lm_schedule, E, stream);
}

extern "C" int ancsh_umeyama(int nprob, const int *off, const float *src, const float *tgt, double *out, void *stream) {
    ANCSH_REQUIRE(nprob >= 0, "umeyama: negative nprob");
    if (nprob == 0) return ANCSH_OK;
    ANCSH_REQUIRE(off && src && tgt && out, "umeyama: null pointer");
    hipLaunchKernelGGL(umeyama_kernel, dim3(nprob), dim3(256), 0, (hipStream_t)stream, off, src, tgt, out);
    return check_launch("umeyama");
}

extern "C" int ancsh_estimate_similarity_transform(int nprob, const int *off, const float *src, const float *tgt, int niter,
                                                   const int *draws, unsigned long long seed, double *out, int *status,
                                                   void *stream) {
    ANCSH_REQUIRE(nprob >= 0 && niter > 0 && niter <= 128, "estimate_similarity_transform: niter %d outside 1..128", niter);
    if (nprob == 0) return ANCSH_OK;
    ANCSH_REQUIRE(off && src && tgt && out && status, "estimate_similarity_transform: null pointer");
    hipLaunchKernelGGL(ransac_umeyama5_kernel, dim3(nprob), dim3(128), 0, (hipStream_t)stream, off, src, tgt, niter, draws, seed,
                       out, status);
    return check_launch("estimate_similarity_transform");
}
