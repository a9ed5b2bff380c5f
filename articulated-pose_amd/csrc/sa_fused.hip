// sa_fused.hip -- one-launch PointNet++ set-abstraction body for gfx950:
//     group_point(xyz) - new_xyz | group_point(features)  ->  3 x (1x1 conv + bias + BN + ReLU)  ->  max over nsample
// (pointnet_plusplus/utils/pointnet_util.py:47-57 (grouping/concat) + :113-134 (MLP + reduce_max)).
//
// The reference materialises the (B, npoint, 64, 3+C) grouped tensor and three (B, npoint, 64, C_i) activations in
// HBM (4.2 MB + 12.6 MB per cloud for SA2) between five TF ops; here every WAVE owns 32*RT rows in a private LDS tile
// from the gather to the max -- SA1: RT = 2, a whole 64-sample neighbourhood (no workgroup synchronisation at all);
// SA2: RT = 1, half a neighbourhood (its 131-channel rows would not leave LDS for two workgroups per CU otherwise), the two
// halves meeting in one final pairwise max:
//   * gather: 16-B feature loads from the L2-resident level-1 features, centred xyz, into the wave's tile (ODD row
//     stride: conflict-free ds_read_b32 MFMA fragments, lane -> [row = lane&31][k = lane>>5]);
//   * each layer: v_mfma_f32_32x32x2_f32 over the wave's rows x ALL N output columns (RT * N/32 accumulators), the
//     weights going from L2 STRAIGHT INTO REGISTERS in the pre-packed fragment order of ancsh_sa_pack_weights (one 16-byte
//     load = the B fragments of four k-steps), both operands software pipelined through small register rings, the loads
//     issued in the shadow of the MFMAs (sched_barrier keeps them there), k loops fully unrolled (see the Makefile);
//   * epilogue in registers: bias, folded BN (one fmaf), ReLU, written back IN PLACE over the wave's own rows (all of a
//     layer's reads complete before its first write); the last layer instead takes the max over the wave's rows;
//   * RT = 1 only: the two waves of a neighbourhood combine their maxima through LDS (the only __syncthreads).
// No barrier inside the MLP means the 2..4 waves sharing a SIMD drift out of phase, so one wave's gather / epilogue
// (VALU, LDS, L2 latency) runs under another's MFMAs; the earlier workgroup-tiled version (4 waves x column slices, 3
// barriers) kept co-resident workgroups in lock-step and idled the matrix pipe ~35 % of the time.
// Arithmetic is the same k-ordered f32 fmaf chain as ancsh_conv1x1 / the CPU oracle => bit-identical outputs.
// HBM traffic per neighbourhood: 64 idx + the final C3 floats (the gathered features come from L2).
#include "common.h"
#include "wave_mlp.h"

namespace ancsh {

#ifndef SA1_RT
#define SA1_RT 2
#endif


// -DSA_STAMPS (diagnostic build only, scratch/sa_trace.py): every wave leaves its s_memtime phase stamps and hardware slot
// (HW_ID, XCC_ID) in the first ints of its output row instead of the features there, so the host can rebuild each SIMD's timeline.
#ifdef SA_STAMPS
#define SA_STAMP(i) do { st_[i] = (unsigned)(__builtin_readcyclecounter() - t0_); } while (0)
#else
#define SA_STAMP(i) do { } while (0)
#endif

#ifdef SA_STAMPS
__device__ __forceinline__ void sa_write_stamps(float *row, unsigned long long t0, const unsigned (&st)[8], bool writer) {
    const unsigned end = (unsigned)(__builtin_readcyclecounter() - t0);
    if (!writer) return;
    unsigned *o = reinterpret_cast<unsigned *>(row);
    o[0] = (unsigned)t0; o[1] = (unsigned)(t0 >> 32);
    o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    for (int i = 0; i < 7; ++i) o[4 + i] = st[i];
    o[11] = end;
}
#endif

// body shared by the two instantiations.  A wave owns 32*RT rows: RT = 2 -> a whole 64-sample neighbourhood (the max is
// wave-local, every weight fragment feeds two MFMAs); RT = 1 -> half a neighbourhood (the halves meet through LDS at the end).
// PARTIAL: `feats` holds, per source point, the first layer's RAW partial sums over the feature channels (C1 values, CF == C1:
// ancsh_conv1x1 with ANCSH_ACT_RAW on the kernel rows 3..3+c-1).  The first layer's dot product sums the feature channels first and
// the three centred coordinates last, so that partial sum is the same in every neighbourhood the point falls into: it is computed
// once per point (n rows) instead of once per neighbour (64 m rows), gathered like a feature row, and the layer here only CONTINUES
// the chain with the coordinates (two MFMA k-steps instead of 66 for SA2: a quarter of its matrix work).
// GROUPED launches (several networks on the same clouds): `groups` = ngroups * bgeo * m neighbourhoods; cloud c = g / m uses the
// GEOMETRY (xyz, new_xyz, idx) of cloud c % bgeo, the layer parameters GL.L[c / bgeo] and its own feature / output rows.
struct SaGroupLayers {
    SaLayer L[ANCSH_MAX_GROUPS][3];
};

template <int CF, int C1, int C2, int C3, int RT, bool PARTIAL = false>
__device__ __forceinline__ void sa_body(int n, int m, long groups, int bgeo, const float *__restrict__ xyz, const float *__restrict__ feats,
                                        const float *__restrict__ new_xyz, const int *__restrict__ idx, const SaGroupLayers &GL,
                                        float *__restrict__ out) {
    static_assert(!PARTIAL || CF == C1, "partial sums have the first layer's width");
    constexpr int CIN = PARTIAL ? 3 : 3 + CF;            // input channels the first layer still has to sum here
    constexpr int XOFF = PARTIAL ? CF : 0;               // tile column of the centred coordinates ...
    constexpr int FOFF = PARTIAL ? 0 : 3;                // ... and of the gathered feature (or partial-sum) row
    constexpr int WIN = PARTIAL ? CF + 4 : 3 + CF;
    constexpr int W0 = WIN > C1 ? WIN : C1, W1 = W0 > C2 ? W0 : C2;
    constexpr int LD = (W1 + 1) | 1;                     // odd, > widest layer input (column K of an odd K stays in-row)
    constexpr int ROWS = 32 * RT;                        // rows per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *T = smem + wave * (ROWS * LD);
    // XCD-aware workgroup -> neighbourhood map.  Workgroups go round-robin to the 8 XCDs (workgroup i runs on XCD i % 8), each with
    // its own 4 MB L2, and a source row is gathered by ~16 neighbourhoods of ITS cloud: with the identity map every XCD's L2 sees
    // the rows of all clouds (SA2: 32 x 256 KB of partial sums per network); here XCD x takes the clouds x, x + 8, ... whole, so its
    // L2 only ever holds b / 8 clouds' rows.
    constexpr int PER_WG = RT == 2 ? 4 : 2;              // neighbourhoods per workgroup
    long wg = blockIdx.x;
#ifndef SA_NO_XCD_MAP
    {
        const long wpc = m / PER_WG, clouds = groups / m;            // workgroups per cloud
        if ((clouds & 7) == 0 && wpc * PER_WG == m) {
            const long xcd = wg & 7, j = wg >> 3;
            wg = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
#endif
    const long g = RT == 2 ? wg * 4 + wave : wg * 2 + (wave >> 1);   // this wave's neighbourhood
    const int half = RT == 2 ? 0 : (wave & 1);           // RT = 1: rows half*32 .. +32 of it
    const bool live = g < groups;
    // which network (layer parameters) and which cloud's geometry; all wave-uniform
    const long cloud = (live ? g : groups - 1) / m;
    const int grp = (int)(cloud / bgeo);
    const long cg = cloud - (long)grp * bgeo;            // geometry cloud
    const long gg = cg * m + (g - cloud * m);            // neighbourhood index inside the geometry arrays
    const SaLayer L1 = GL.L[grp][0], L2 = GL.L[grp][1], L3 = GL.L[grp][2];
#ifdef SA_STAMPS
    const unsigned long long t0_ = __builtin_readcyclecounter();
    unsigned st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // layer 1's first weights are in flight during the gather
    float4 bw1[LayerCfg<CIN, C1>::DW + 1][C1 / 32];
    w_prologue<CIN, C1>(L1, bw1);
    // ---- gather: T[r][0:3] = xyz[idx] - new_xyz ; T[r][3:3+CF] = feats[idx] ; T[r][CIN] = 0 (odd-K pad column) --------
    if (live) {
        const long b = cloud;                            // feature rows: this network's copy of the cloud
        const int *gi = idx + gg * 64 + half * 32;
        if (lane < ROWS) {
            const int ii = gi[lane];
            const float *p = xyz + ((size_t)cg * n + ii) * 3;
            const float *c = new_xyz + (size_t)gg * 3;
            float *x = T + lane * LD + XOFF;
            x[0] = p[0] - c[0]; x[1] = p[1] - c[1]; x[2] = p[2] - c[2];
            if (CIN & 1) x[CIN] = 0.f;
        }
        if (CF > 0) {
            constexpr int V = CF / 4;                    // float4 per row
#pragma unroll 8
            for (int e = lane; e < ROWS * V; e += 64) {
                const int r = e / V, c4 = e % V;
                const int ii = gi[r];
                const float4 v = *reinterpret_cast<const float4 *>(feats + ((size_t)b * n + ii) * CF + c4 * 4);
                float *x = T + r * LD + FOFF + c4 * 4;
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            }
        }
    } else {
        for (int e = lane; e < ROWS * LD; e += 64) T[e] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    SA_STAMP(0);
    float none1[C1 / 32], none2[C2 / 32], pm[C3 / 32];
    {
        floatx16 acc[RT][C1 / 32];
        float ep1[3][C1 / 32];
        mfma_loop<CIN, C1, LD, RT, XOFF, PARTIAL ? 1 : 0>(T, L1, bw1, acc, ep1);
        SA_STAMP(1);
        float4 bw2[LayerCfg<C1, C2>::DW + 1][C2 / 32];
        w_prologue<C1, C2>(L2, bw2);                     // layer 2's first weights fly under layer 1's epilogue
        __builtin_amdgcn_sched_barrier(0);
        epilogue<C1, LD, false, RT>(T, acc, ep1, none1);
        SA_STAMP(2);
        floatx16 acc2[RT][C2 / 32];
        float ep2[3][C2 / 32];
        mfma_loop<C1, C2, LD, RT>(T, L2, bw2, acc2, ep2);
        SA_STAMP(3);
        float4 bw3[LayerCfg<C2, C3>::DW + 1][C3 / 32];
        w_prologue<C2, C3>(L3, bw3);
        __builtin_amdgcn_sched_barrier(0);
        epilogue<C2, LD, false, RT>(T, acc2, ep2, none2);
        SA_STAMP(4);
        floatx16 acc3[RT][C3 / 32];
        float ep3[3][C3 / 32];
        mfma_loop<C2, C3, LD, RT>(T, L3, bw3, acc3, ep3);
        SA_STAMP(5);
        epilogue<C3, LD, true, RT>(T, acc3, ep3, pm);
        SA_STAMP(6);
    }
    if (RT == 2) {                                       // the wave saw all 64 rows: done, no workgroup synchronisation at all
        if (lane < 32 && live) {
#pragma unroll
            for (int j = 0; j < C3 / 32; ++j) out[(size_t)g * C3 + j * 32 + lane] = pm[j];
        }
#ifdef SA_STAMPS
        sa_write_stamps(out + (size_t)g * C3, t0_, st_, live && lane == 0);
#endif
        return;
    }
    // ---- max over the neighbourhood's two row halves: odd waves hand their maxima to the even wave through LDS --------
    wave_lds_fence();
    if ((wave & 1) && lane < 32) {
#pragma unroll
        for (int j = 0; j < C3 / 32; ++j) T[j * 32 + lane] = pm[j];
    }
    __syncthreads();
    if (!(wave & 1) && lane < 32 && live) {
        const float *O = T + ROWS * LD;                  // the odd partner's tile
#pragma unroll
        for (int j = 0; j < C3 / 32; ++j) out[(size_t)g * C3 + j * 32 + lane] = nmax(pm[j], O[j * 32 + lane]);
    }
#ifdef SA_STAMPS
    __syncthreads();
    sa_write_stamps(out + (size_t)g * C3 + 16 * (wave & 1), t0_, st_, live && lane == 0);
#endif
}

// SA1 (3 -> 64 -> 64 -> 128): a wave owns a whole neighbourhood (RT = 2): 4..8 accumulators per layer, each weight fragment
// used twice, wave-local max; 66 KB of LDS per workgroup -> two workgroups (2 waves per SIMD) per CU
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sa1_fused_kernel(int n, int m, long groups, int bgeo, const float *__restrict__ xyz, const float *__restrict__ feats,
                      const float *__restrict__ new_xyz, const int *__restrict__ idx, SaGroupLayers GL, float *__restrict__ out) {
    sa_body<0, 64, 64, 128, SA1_RT>(n, m, groups, bgeo, xyz, feats, new_xyz, idx, GL, out);
}

// SA2 (3 + 128 -> 128 -> 128 -> 256) on per-point partial sums of its first layer: 68 KB of LDS, 128 accumulator + ~100 other
// registers: two workgroups per CU
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sa2_fused_kernel(int n, int m, long groups, int bgeo, const float *__restrict__ xyz, const float *__restrict__ partial,
                      const float *__restrict__ new_xyz, const int *__restrict__ idx, SaGroupLayers GL, float *__restrict__ out) {
    sa_body<128, 128, 128, 256, 1, true>(n, m, groups, bgeo, xyz, partial, new_xyz, idx, GL, out);
}

// packed[((slot*TN + j)*64 + lane)*4 + q] = W[2*(4*slot + q) + (lane>>5)][j*32 + (lane&31)], zero past row k-1
__global__ __launch_bounds__(256) void sa_pack_weights_kernel(int k, int n, const float *__restrict__ w, float *__restrict__ packed,
                                                              long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int tn = (n + 31) / 32;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long sj = e >> 8;
    const int j = (int)(sj % tn), slot = (int)(sj / tn);
    const int kk = 2 * (4 * slot + q) + (lane >> 5), col = j * 32 + (lane & 31);
    packed[e] = (kk < k && col < n) ? w[(size_t)kk * n + col] : 0.f;
}

static long sa_packed_floats(int k, int n) { return (long)(((k + 1) / 2 + 3) / 4) * ((n + 31) / 32) * 256; }

template <int CF, int C1, int C2, int C3, int RT, bool PARTIAL, class Kern>
static int launch_sa(Kern k, int ngroups, int b, int n, int m, const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                     const SaGroupLayers &GL, float *out, hipStream_t st) {
    constexpr int WIN = PARTIAL ? CF + 4 : 3 + CF;
    constexpr int W0 = WIN > C1 ? WIN : C1, W1 = W0 > C2 ? W0 : C2;
    constexpr int LD = (W1 + 1) | 1;
    const size_t lds = sizeof(float) * 4 * 32 * RT * LD;
    const long groups = (long)ngroups * b * m;
    const long per_wg = RT == 2 ? 4 : 2;                 // neighbourhoods per workgroup
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((groups + per_wg - 1) / per_wg)), dim3(256), lds, st, n, m, groups, b, xyz, feats, new_xyz, idx, GL, out);
    return check_launch("sa_module_fused");
}

}  // namespace ancsh

using namespace ancsh;

extern "C" long ancsh_sa_packed_weight_floats(int k, int n) {
    if (k <= 0 || n <= 0) return -1;
    return sa_packed_floats(k, n);
}

extern "C" int ancsh_sa_pack_weights(int k, int n, const float *w, float *packed, void *stream) {
    ANCSH_REQUIRE(k > 0 && n > 0, "sa_pack_weights: k=%d n=%d must be positive", k, n);
    ANCSH_REQUIRE(w && packed, "sa_pack_weights: null pointer");
    const long total = sa_packed_floats(k, n);
    hipLaunchKernelGGL(sa_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k, n, w, packed, total);
    return check_launch("sa_pack_weights");
}

static int sa_layers(const float *const *params, int ngroups, int c1, int c2, int c3, SaGroupLayers &GL, const char *who) {
    for (int g = 0; g < ANCSH_MAX_GROUPS; ++g) {
        const float *const *pp = params + 12 * (g < ngroups ? g : 0);
        for (int i = 0; i < 3; ++i) {
            SaLayer &L = GL.L[g][i];
            L.w = pp[4 * i]; L.bias = pp[4 * i + 1]; L.scale = pp[4 * i + 2]; L.shift = pp[4 * i + 3];
            L.ncol = i == 0 ? c1 : i == 1 ? c2 : c3; L.wstride = 0;
            ANCSH_REQUIRE(L.w && L.bias && L.scale && L.shift, "%s: null layer parameter", who);
        }
    }
    return ANCSH_OK;
}

static int sa_fused_impl(const char *who, int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                         const float *feats, const float *new_xyz, const int *idx, const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0, "%s: bad shape b=%d n=%d m=%d", who, b, n, m);
    ANCSH_REQUIRE(nsample == 64, "%s: nsample must be 64 (got %d)", who, nsample);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && new_xyz && idx && params && out && (cfeat == 0 || feats), "%s: null pointer", who);
    SaGroupLayers GL;
    if (int rc = sa_layers(params, ngroups, c1, c2, c3, GL, who)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (cfeat == 0 && c1 == 64 && c2 == 64 && c3 == 128)
        return launch_sa<0, 64, 64, 128, SA1_RT, false>(sa1_fused_kernel, ngroups, b, n, m, xyz, feats, new_xyz, idx, GL, out, st);
    set_error("%s: unsupported layer shape (cfeat=%d mlp=[%d,%d,%d]); levels with input features go through "
              "ancsh_sa_module_fused_partial, other shapes through the unfused path", who, cfeat, c1, c2, c3);
    return ANCSH_EINVAL;
}

static int sa_fused_partial_impl(const char *who, int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                                 const float *partial, const float *new_xyz, const int *idx, const float *const *params, float *out,
                                 void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0, "%s: bad shape b=%d n=%d m=%d", who, b, n, m);
    ANCSH_REQUIRE(nsample == 64, "%s: nsample must be 64 (got %d)", who, nsample);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && partial && new_xyz && idx && params && out, "%s: null pointer", who);
    ANCSH_REQUIRE((((uintptr_t)partial) & 15) == 0, "%s: partial must be 16-byte aligned", who);
    SaGroupLayers GL;
    if (int rc = sa_layers(params, ngroups, c1, c2, c3, GL, who)) return rc;
    if (c1 == 128 && c2 == 128 && c3 == 256)
        return launch_sa<128, 128, 128, 256, 1, true>(sa2_fused_kernel, ngroups, b, n, m, xyz, partial, new_xyz, idx, GL, out, (hipStream_t)stream);
    set_error("%s: unsupported layer shape (mlp=[%d,%d,%d]); use the unfused path", who, c1, c2, c3);
    return ANCSH_EINVAL;
}

// params: 12 device pointers = {packed w, bias, scale, shift} x 3 layers (see ancsh_conv1x1 for bias/scale/shift)
extern "C" int ancsh_sa_module_fused(int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                                     const float *feats, const float *new_xyz, const int *idx, const float *const *params,
                                     float *out, void *stream) {
    return sa_fused_impl("sa_module_fused", 1, b, n, m, nsample, cfeat, c1, c2, c3, xyz, feats, new_xyz, idx, params, out, stream);
}

// A level WITH input features: partial = the first layer's raw partial sums over the feature channels, one row of c1 values per
// source point (ancsh_conv1x1(b*n, c, c1, feats, ..., w + 3*c1, ..., ANCSH_ACT_RAW)); params[0] = the packed kernel rows 0..2
// (the coordinates), the rest as ancsh_sa_module_fused.
extern "C" int ancsh_sa_module_fused_partial(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                                             const float *partial, const float *new_xyz, const int *idx,
                                             const float *const *params, float *out, void *stream) {
    return sa_fused_partial_impl("sa_module_fused_partial", 1, b, n, m, nsample, c1, c2, c3, xyz, partial, new_xyz, idx, params, out, stream);
}

// The same level of `ngroups` networks on the same clouds in one launch: geometry (xyz, new_xyz, idx) of b clouds, feats / partial /
// out of ngroups * b clouds (network-major), params = 12 pointers per network.
extern "C" int ancsh_sa_module_fused_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                             const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                             const float *const *params, float *out, void *stream) {
    return sa_fused_impl("sa_module_fused_grouped", ngroups, b, n, m, nsample, cfeat, c1, c2, c3, xyz, feats, new_xyz, idx, params, out, stream);
}

extern "C" int ancsh_sa_module_fused_partial_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                                     const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                                     const float *const *params, float *out, void *stream) {
    return sa_fused_partial_impl("sa_module_fused_partial_grouped", ngroups, b, n, m, nsample, c1, c2, c3, xyz, partial, new_xyz, idx, params,
                                 out, stream);
}
