// interpolate.hip -- three_nn / inverse-distance weights / three_interpolate for gfx950.
//
// Semantics: ops/3d_interpolation/tf_interpolate.cpp:60-127 -- in the reference these are
// single-threaded HOST loops behind a device->host->device round trip (the op registers
// DEVICE_CPU only, :187,222).  Here they are device kernels:
//   three_nn: one lane per query point; the known points of the cloud are staged in LDS as
//     float4 so the inner loop is one broadcast ds_read_b128 + 8 VALU per candidate; the
//     top-3 insertion uses the reference's strict '<' cascade (earliest index wins ties).
//     Arithmetic is the host compiler's: products and sums individually rounded (g++ -O2 on
//     x86-64 emits no FMA), hence this file is built with -ffp-contract=off and uses no fmaf.
//   three_interpolate: lanes run along the channel axis (coalesced 256 B per row segment).
#include "common.h"

namespace ancsh {

constexpr int NN_CHUNK = 2048;   // known points staged per LDS pass (32 KiB)
constexpr int NN_SEG = 8;        // lanes that share one query: each scans every NN_SEG-th known point

// The reference's insertion cascade with strict '<' (tf_interpolate.cpp:74-91) keeps the three smallest (distance, index)
// pairs in lexicographic order: an equal distance never displaces an earlier index.  That order is associative, so the
// candidates of a query can be split over NN_SEG lanes (each keeps its own top three with the same cascade, its candidates
// visited in ascending index) and the partial lists merged pairwise with the same (distance, index) comparison -- identical
// results, NN_SEG times the parallelism of the lane-per-query scan (which left half the SIMDs empty and ran 512-step loops).
struct Top3 {
    float d1, d2, d3;
    int i1, i2, i3;
};
__device__ __forceinline__ bool nn_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void nn_insert(Top3 &t, float d, int k) {          // k may be smaller than indices already held
    if (nn_less(d, k, t.d1, t.i1)) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k; }
    else if (nn_less(d, k, t.d2, t.i2)) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = d; t.i2 = k; }
    else if (nn_less(d, k, t.d3, t.i3)) { t.d3 = d; t.i3 = k; }
}

// weight != nullptr: also the interpolation weights of pointnet_util.py:219-222 (the arithmetic of three_weights_kernel below)
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2, float *__restrict__ dist,
                                                       int *__restrict__ idx, float *__restrict__ weight) {
    __shared__ float4 known[NN_CHUNK];
    const int b = blockIdx.y;
    const int seg = threadIdx.x & (NN_SEG - 1);
    const int j = blockIdx.x * (256 / NN_SEG) + (threadIdx.x / NN_SEG);
    const bool live = j < n;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    {
        const float *p = xyz1 + ((size_t)b * n + (live ? j : n - 1)) * 3;
        x1 = p[0]; y1 = p[1]; z1 = p[2];
    }
    // reference: double best = 1e40 (stores to float as +inf when never replaced); "no candidate" = (+inf, index 0), which
    // must lose against any real candidate at +inf distance only by index order -- real indices are >= 0, and a real
    // candidate with d = +inf and index 0 is the same pair, so the sentinel index is INT_MAX during the scan
    Top3 t = {INFINITY, INFINITY, INFINITY, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    const float *p2 = xyz2 + (size_t)b * m * 3;
    for (int base = 0; base < m; base += NN_CHUNK) {
        const int cnt = (m - base) < NN_CHUNK ? (m - base) : NN_CHUNK;
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) {
            const float *s = p2 + (size_t)(base + e) * 3;
            known[e] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
        for (int k = seg; k < cnt; k += NN_SEG) {
            const float4 q = known[k];
            const float dx = q.x - x1, dy = q.y - y1, dz = q.z - z1;
            const float d = dx * dx + dy * dy + dz * dz;   // ((dx*dx + dy*dy) + dz*dz), each op rounded
            // within a lane the indices ascend, so the reference's plain '<' cascade IS the lexicographic one here
            const int kk = base + k;
            if (d < t.d1) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = kk; }
            else if (d < t.d2) { t.d3 = t.d2; t.i3 = t.i2; t.d2 = d; t.i2 = kk; }
            else if (d < t.d3) { t.d3 = d; t.i3 = kk; }
        }
    }
    // merge the NN_SEG partial lists of a query (adjacent lanes) by butterfly exchange
#pragma unroll
    for (int o = 1; o < NN_SEG; o <<= 1) {
        const float e1 = __shfl_xor(t.d1, o, 64), e2 = __shfl_xor(t.d2, o, 64), e3 = __shfl_xor(t.d3, o, 64);
        const int f1 = __shfl_xor(t.i1, o, 64), f2 = __shfl_xor(t.i2, o, 64), f3 = __shfl_xor(t.i3, o, 64);
        nn_insert(t, e1, f1);
        nn_insert(t, e2, f2);
        nn_insert(t, e3, f3);
    }
    if (live && seg == 0) {
        float *od = dist + ((size_t)b * n + j) * 3;
        int *oi = idx + ((size_t)b * n + j) * 3;
        od[0] = t.d1; od[1] = t.d2; od[2] = t.d3;
        oi[0] = t.i1 == 0x7fffffff ? 0 : t.i1; oi[1] = t.i2 == 0x7fffffff ? 0 : t.i2; oi[2] = t.i3 == 0x7fffffff ? 0 : t.i3;
        if (weight) {
            const float d0 = fmaxf(t.d1, 1e-10f), d1 = fmaxf(t.d2, 1e-10f), d2 = fmaxf(t.d3, 1e-10f);
            const float r0 = __fdiv_rn(1.0f, d0), r1 = __fdiv_rn(1.0f, d1), r2 = __fdiv_rn(1.0f, d2);
            const float norm = (r0 + r1) + r2;
            float *ow = weight + ((size_t)b * n + j) * 3;
            ow[0] = __fdiv_rn(r0, norm); ow[1] = __fdiv_rn(r1, norm); ow[2] = __fdiv_rn(r2, norm);
        }
    }
}

__global__ void three_weights_kernel(int rows, const float *__restrict__ dist, float *__restrict__ weight) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float d0 = fmaxf(dist[r * 3 + 0], 1e-10f), d1 = fmaxf(dist[r * 3 + 1], 1e-10f), d2 = fmaxf(dist[r * 3 + 2], 1e-10f);
    float r0 = __fdiv_rn(1.0f, d0), r1 = __fdiv_rn(1.0f, d1), r2 = __fdiv_rn(1.0f, d2);
    float norm = (r0 + r1) + r2;
    weight[r * 3 + 0] = __fdiv_rn(r0, norm);
    weight[r * 3 + 1] = __fdiv_rn(r1, norm);
    weight[r * 3 + 2] = __fdiv_rn(r2, norm);
}

// out[b,j, off+l] = p[i1,l]*w1 + p[i2,l]*w2 + p[i3,l]*w3  (evaluation order of tf_interpolate.cpp:121)
__global__ __launch_bounds__(256) void three_interpolate_kernel(int m, int c, int n, const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out, int out_ld, int out_off,
                                                                long total) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / c;   // flat (b, j)
        const int l = (int)(e - row * c);
        const long bi = row / n;
        const float w1 = weight[row * 3], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const int a1 = idx[row * 3], a2 = idx[row * 3 + 1], a3 = idx[row * 3 + 2];
        const float *p = points + (size_t)bi * m * c;
        const float v = p[(size_t)a1 * c + l] * w1 + p[(size_t)a2 * c + l] * w2 + p[(size_t)a3 * c + l] * w3;
        out[(size_t)row * out_ld + out_off + l] = v;
    }
}

// c % 4 == 0 and 16-byte aligned rows: a lane produces four channels (one float4 load per neighbour, one float4 store), the
// row index and the three (idx, weight) pairs are decoded once per lane with 32-bit arithmetic.  Per-element arithmetic is
// the scalar kernel's ((p1*w1 + p2*w2) + p3*w3, each op rounded).
__global__ __launch_bounds__(256) void three_interpolate_vec4_kernel(int m, int c4, int n, const float *__restrict__ points,
                                                                     const int *__restrict__ idx, const float *__restrict__ weight,
                                                                     float *__restrict__ out, int out_ld, int out_off, unsigned total) {
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned row = e / (unsigned)c4;
        const int l4 = (int)(e - row * (unsigned)c4);
        const unsigned bi = row / (unsigned)n;
        const float w1 = weight[row * 3], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const int a1 = idx[row * 3], a2 = idx[row * 3 + 1], a3 = idx[row * 3 + 2];
        const float4 *p = reinterpret_cast<const float4 *>(points) + (size_t)bi * m * c4 + l4;
        const float4 x1 = p[(size_t)a1 * c4], x2 = p[(size_t)a2 * c4], x3 = p[(size_t)a3 * c4];
        float4 v;
        v.x = x1.x * w1 + x2.x * w2 + x3.x * w3;
        v.y = x1.y * w1 + x2.y * w2 + x3.y * w3;
        v.z = x1.z * w1 + x2.z * w2 + x3.z * w3;
        v.w = x1.w * w1 + x2.w * w2 + x3.w * w3;
        *reinterpret_cast<float4 *>(out + (size_t)row * out_ld + out_off + l4 * 4) = v;
    }
}

// The whole input row of an FP module in one launch (pointnet_util.py:218-229): out[row] = [three_interpolate(points2) (c2) |
// points1[row] (c1) | zeros up to out_ld].  One lane per 4 output floats; c2 % 4 == 0 so a float4 is either interpolated or tail.
// geo_rows / p1_rows: rows of the 3-NN arrays (idx, weight) / of points1 -- the row is taken MODULO them, so that several networks'
// features (points2, out: b * n rows, network-major) share one geometry and, where points1 is the input cloud itself, one points1.
__global__ __launch_bounds__(256) void fp_concat_kernel(int m, int c2, int n, const float *__restrict__ points2,
                                                        const int *__restrict__ idx, const float *__restrict__ weight,
                                                        const float *__restrict__ points1, int c1, float *__restrict__ out, int ld4,
                                                        unsigned total, unsigned geo_rows, unsigned p1_rows, unsigned wpc, bool p1_vec) {
    const int c24 = c2 / 4;
    // XCD-aware block -> cloud map (as in sa_fused.hip): workgroups go round-robin to the 8 XCDs, each with its own 4 MB L2, and a row
    // of points2 is gathered by ~3 * n / m output rows of ITS cloud.  With the identity map every XCD's L2 sees the points2 of all
    // clouds (FP2 of two networks: 8 MB); here XCD x takes the clouds x, x + 8, ... whole.  wpc = workgroups per cloud, 0 = identity
    // (set by the launcher only when the grid covers `total` exactly and the cloud count is a multiple of 8).
    unsigned blk = blockIdx.x;
    if (wpc) {
        const unsigned xcd = blk & 7u, j = blk >> 3;
        blk = (xcd + 8u * (j / wpc)) * wpc + j % wpc;
    }
    for (unsigned e = blk * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned row = e / (unsigned)ld4;
        const int q = (int)(e - row * (unsigned)ld4);
        float4 v;
        if (q < c24) {
            const unsigned bi = row / (unsigned)n;
            const unsigned grow = row % geo_rows;
            const float w1 = weight[grow * 3], w2 = weight[grow * 3 + 1], w3 = weight[grow * 3 + 2];
            const int a1 = idx[grow * 3], a2 = idx[grow * 3 + 1], a3 = idx[grow * 3 + 2];
            const float4 *p = reinterpret_cast<const float4 *>(points2) + (size_t)bi * m * c24 + q;
            const float4 x1 = p[(size_t)a1 * c24], x2 = p[(size_t)a2 * c24], x3 = p[(size_t)a3 * c24];
            v.x = x1.x * w1 + x2.x * w2 + x3.x * w3;
            v.y = x1.y * w1 + x2.y * w2 + x3.y * w3;
            v.z = x1.z * w1 + x2.z * w2 + x3.z * w3;
            v.w = x1.w * w1 + x2.w * w2 + x3.w * w3;
        } else {
            const int t = (q - c24) * 4;                    // first tail channel of this float4
            const float *s1 = points1 + (size_t)(row % p1_rows) * c1;
            if (p1_vec) {                                   // c1 % 4 == 0 and points1 16-byte aligned: whole float4 or nothing
                v = t < c1 ? *reinterpret_cast<const float4 *>(s1 + t) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                v.x = t < c1 ? s1[t] : 0.f;
                v.y = t + 1 < c1 ? s1[t + 1] : 0.f;
                v.z = t + 2 < c1 ? s1[t + 2] : 0.f;
                v.w = t + 3 < c1 ? s1[t + 3] : 0.f;
            }
        }
        reinterpret_cast<float4 *>(out)[(size_t)row * ld4 + q] = v;
    }
}

static int launch_interp(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                         float *out, int out_ld, int out_off, hipStream_t st) {
    ANCSH_REQUIRE(b >= 0 && m > 0 && c >= 0 && n >= 0, "ThreeInterpolate expects (b,m,c) points shape");
    ANCSH_REQUIRE(out_ld >= out_off + c && out_off >= 0, "three_interpolate: out_ld %d < out_off %d + c %d", out_ld, out_off, c);
    const long total = (long)b * n * c;
    if (total == 0) return ANCSH_OK;
    ANCSH_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
    if (c % 4 == 0 && out_ld % 4 == 0 && out_off % 4 == 0 && (((uintptr_t)points | (uintptr_t)out) % 16) == 0 && total / 4 < (1L << 31) &&
        (long)b * n < (1L << 30)) {
        const long t4 = total / 4;
        long blocks4 = (t4 + 255) / 256;
        if (blocks4 > 256L * 64) blocks4 = 256L * 64;
        hipLaunchKernelGGL(three_interpolate_vec4_kernel, dim3((unsigned)blocks4), dim3(256), 0, st, m, c / 4, n, points, idx, weight, out,
                           out_ld, out_off, (unsigned)t4);
        return check_launch("three_interpolate");
    }
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3((unsigned)blocks), dim3(256), 0, st, m, c, n, points, idx, weight, out,
                       out_ld, out_off, total);
    return check_launch("three_interpolate");
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                              void *stream) {
    ANCSH_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ThreeNN expects (b,n,3) xyz1 shape");
    if (b == 0 || n == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz1 && xyz2 && dist && idx, "three_nn: null pointer");
    dim3 grid((n + 256 / NN_SEG - 1) / (256 / NN_SEG), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx, (float *)nullptr);
    return check_launch("three_nn");
}

// three_nn + the interpolation weights of pointnet_util.py:219-222 in one launch (weight (b,n,3); same values as
// ancsh_three_nn followed by ancsh_three_weights)
extern "C" int ancsh_three_nn_weights(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                                      float *weight, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ThreeNN expects (b,n,3) xyz1 shape");
    if (b == 0 || n == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz1 && xyz2 && dist && idx && weight, "three_nn_weights: null pointer");
    dim3 grid((n + 256 / NN_SEG - 1) / (256 / NN_SEG), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx, weight);
    return check_launch("three_nn_weights");
}

extern "C" int ancsh_three_weights(int rows, const float *dist, float *weight, void *stream) {
    ANCSH_REQUIRE(rows >= 0, "three_weights: negative rows");
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE(dist && weight, "three_weights: null pointer");
    hipLaunchKernelGGL(three_weights_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, dist, weight);
    return check_launch("three_weights");
}

extern "C" int ancsh_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                                       const float *weight, float *out, void *stream) {
    return launch_interp(b, m, c, n, points, idx, weight, out, c, 0, (hipStream_t)stream);
}

static int fp_concat_launch(const char *who, int b, int m, int c2, int n, const float *points2, const int *idx, const float *weight,
                            const float *points1, int c1, float *out, int out_ld, int geo_batch, int p1_batch, void *stream) {
    ANCSH_REQUIRE(b >= 0 && m > 0 && c2 > 0 && n >= 0 && c1 >= 0, "%s: bad shape b=%d m=%d c2=%d n=%d c1=%d", who, b, m, c2, n, c1);
    ANCSH_REQUIRE(c2 % 4 == 0 && out_ld % 4 == 0 && out_ld >= c2 + c1, "%s: needs c2 %% 4 == 0, out_ld %% 4 == 0, out_ld >= c2 + c1 (c2=%d c1=%d out_ld=%d)", who, c2, c1, out_ld);
    ANCSH_REQUIRE(geo_batch > 0 && p1_batch > 0 && (b == 0 || (b % geo_batch == 0 && b % p1_batch == 0)),
                  "%s: b=%d must be a multiple of geo_batch=%d and points1_batch=%d", who, b, geo_batch, p1_batch);
    const long total = (long)b * n * (out_ld / 4);
    if (total == 0) return ANCSH_OK;
    ANCSH_REQUIRE(total < (1L << 31) && (long)b * n < (1L << 30), "%s: too many rows", who);
    ANCSH_REQUIRE(points2 && idx && weight && out && (c1 == 0 || points1), "%s: null pointer", who);
    ANCSH_REQUIRE((((uintptr_t)points2 | (uintptr_t)out) % 16) == 0, "%s: points2 / out must be 16-byte aligned", who);
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 64) blocks = 256L * 64;
    const long per_cloud = (long)n * (out_ld / 4);                      // float4 per cloud
    const unsigned wpc = (blocks * 256 == total && per_cloud % 256 == 0 && b % 8 == 0) ? (unsigned)(per_cloud / 256) : 0u;
    hipLaunchKernelGGL(fp_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m, c2, n, points2, idx, weight, points1,
                       c1, out, out_ld / 4, (unsigned)total, (unsigned)((long)geo_batch * n), (unsigned)((long)p1_batch * n), wpc,
                       c1 > 0 && c1 % 4 == 0 && ((uintptr_t)points1 % 16) == 0);
    return check_launch(who);
}

extern "C" int ancsh_fp_interpolate_concat(int b, int m, int c2, int n, const float *points2, const int *idx, const float *weight,
                                           const float *points1, int c1, float *out, int out_ld, void *stream) {
    return fp_concat_launch("fp_interpolate_concat", b, m, c2, n, points2, idx, weight, points1, c1, out, out_ld, b > 0 ? b : 1, b > 0 ? b : 1, stream);
}

// Several networks on the same clouds: points2 / out hold b clouds (network-major), idx / weight only geo_batch of them and points1
// points1_batch (cloud c reads row block c % geo_batch / c % points1_batch); b % geo_batch == b % points1_batch == 0.
extern "C" int ancsh_fp_interpolate_concat_ex(int b, int m, int c2, int n, const float *points2, const int *idx, const float *weight,
                                              const float *points1, int c1, float *out, int out_ld, int geo_batch, int points1_batch,
                                              void *stream) {
    return fp_concat_launch("fp_interpolate_concat_ex", b, m, c2, n, points2, idx, weight, points1, c1, out, out_ld, geo_batch, points1_batch, stream);
}

extern "C" int ancsh_three_interpolate_ex(int b, int m, int c, int n, const float *points, const int *idx,
                                          const float *weight, float *out, int out_ld, int out_off, void *stream) {
    return launch_interp(b, m, c, n, points, idx, weight, out, out_ld, out_off, (hipStream_t)stream);
}
