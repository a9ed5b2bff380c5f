// mid_chain.hip -- the MIDDLE of the PointNet++ backbone as three chain launches (gfx950): layer3 (group_all set abstraction),
// fa_layer1 and fa_layer2 (feature propagation) -- pointnet_plusplus/architectures.py:72-82, pointnet_util.py:66-91,113-134
// (sample_and_group_all + 3 x conv2d + reduce_max) and :206-236 (three_interpolate + concat + 2 x conv2d).
//
// Until round 4 these were nine layer launches per step (the 1x1-conv kernels of mlp.hip / conv_rowtile.hip / conv_packed.hip), a
// torch.cat and an interpolate + concat launch, every activation going through HBM / L2 between them; the six small ones are
// latency-bound (one 32 x 32 output tile per wave, 4096..16384 rows).  Here a WORKGROUP owns one 32-row tile for a whole level:
//   * the input rows are staged ONCE into LDS (layer3: [xyz | features] built in the load, no concat tensor; fa_layer2: the three
//     nearest level-2 rows gathered and blended in the load -- pointnet_util.py:218-224 -- next to the level-1 skip features);
//   * every layer is split by COLUMNS over the workgroup's waves (wave w: N / 32 / NW adjacent 32-column tiles, the k loop of
//     wave_mlp.h: v_mfma_f32_32x32x2_f32, packed weights straight from L2 into registers, activations from the shared tile);
//   * a layer's epilogue (bias + folded BN + ReLU) rewrites the tile IN PLACE between two workgroup barriers, so no activation
//     leaves the CU; only the level's output does (layer3: the maxima of the tile's 32 rows -- the consumer takes the maximum over a
//     cloud's row tiles, max is exact in any order).
// Same k-ordered f32 fmaf chain per output as ancsh_conv1x1 and the CPU oracle: bit-identical (tests/test_mlp_gpu.py).
#include "common.h"
#include "wave_mlp.h"

namespace ancsh {

struct MidLayer {
    const float *w, *bias, *scale, *shift;      // w: packed (ancsh_sa_pack_weights)
};
struct MidGroups {
    MidLayer L[ANCSH_MAX_GROUPS][3];
};

// this wave's slice of a layer with N output columns split over NW waves
template <int N, int NW>
__device__ __forceinline__ SaLayer wave_slice(const MidLayer &M, int wave) {
    constexpr int TNW = N / 32 / NW;
    SaLayer S;
    S.w = M.w + (size_t)wave * TNW * 256;       // tile j of slot s sits at float4 (s * N/32 + j) * 64
    S.bias = M.bias + wave * TNW * 32; S.scale = M.scale + wave * TNW * 32; S.shift = M.shift + wave * TNW * 32;
    S.ncol = TNW * 32;
    S.wstride = (N / 32) * 64;
    return S;
}

// ---- layer3: rows [xyz (3) | features (256)] -> 256 -> 512 -> 1024, max over the tile's 32 rows ------------------------------------
// 8 waves: 1 / 2 / 4 column tiles per wave.  LDS: Y = 32 x 261 (input), X = 32 x 513 (layer 1 writes its 256 columns, layer 2 reads
// them and -- after a barrier -- writes its 512 over them, layer 3 reads those): 99 KB, one workgroup per CU; npts / 32 workgroups
// per cloud.
constexpr int SA3_NW = 8, SA3_K1 = 259, SA3_N1 = 256, SA3_N2 = 512, SA3_N3 = 1024, SA3_LDY = 261, SA3_LDX = 513;

__global__ __launch_bounds__(64 * SA3_NW) void sa3_chain_kernel(int bgeo, int npts, long tiles, const float *__restrict__ xyz,
                                                                const float *__restrict__ feats, MidGroups GL,
                                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Y = smem, *X = smem + 32 * SA3_LDY;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = blockIdx.x;
    const int tpc = npts / 32;                                  // row tiles per cloud
    const long cloud = tile / tpc;
    const int rt = (int)(tile - cloud * tpc);
    const int grp = (int)(cloud / bgeo);
    const long cg = cloud - (long)grp * bgeo;                   // geometry cloud
    constexpr int TN1 = SA3_N1 / 32 / SA3_NW, TN2 = SA3_N2 / 32 / SA3_NW, TN3 = SA3_N3 / 32 / SA3_NW;
    const SaLayer L1 = wave_slice<SA3_N1, SA3_NW>(GL.L[grp][0], wave);
    const SaLayer L2 = wave_slice<SA3_N2, SA3_NW>(GL.L[grp][1], wave);
    const SaLayer L3 = wave_slice<SA3_N3, SA3_NW>(GL.L[grp][2], wave);
    float4 bw1[LayerCfg<SA3_K1, 32 * TN1>::DW + 1][TN1];
    w_prologue<SA3_K1, 32 * TN1, true>(L1, bw1);
    // ---- stage Y[r] = [xyz | features | 0]: 64 float4 of features per row, 512 threads = 8 rows per pass, 4 passes in flight ----------
    {
        const float *f = feats + ((size_t)cloud * npts + rt * 32) * 256;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(f + (size_t)(i * 512 + tid) * 4);
        if (tid < 96) {
            const int r = tid / 3, c = tid - r * 3;
            Y[r * SA3_LDY + c] = xyz[((size_t)cg * npts + rt * 32 + r) * 3 + c];
        } else if (tid < 128) {
            Y[(tid - 96) * SA3_LDY + SA3_K1] = 0.f;           // the pad column of the odd K
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 512 + tid, r = e >> 6, c4 = e & 63;
            float *d = Y + r * SA3_LDY + 3 + c4 * 4;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    }
    __syncthreads();
    float none1[TN1], none2[TN2], pm[TN3];
    {
        floatx16 acc[1][TN1];
        float ep[3][TN1];
        mfma_loop<SA3_K1, 32 * TN1, SA3_LDY, 1, 0, 0, true>(Y, L1, bw1, acc, ep);
        float4 bw2[LayerCfg<SA3_N1, 32 * TN2>::DW + 1][TN2];
        w_prologue<SA3_N1, 32 * TN2, true>(L2, bw2);
        __builtin_amdgcn_sched_barrier(0);
        epilogue<32 * TN1, SA3_LDX, false, 1>(X + wave * 32 * TN1, acc, ep, none1);
        __syncthreads();
        floatx16 acc2[1][TN2];
        float ep2[3][TN2];
        mfma_loop<SA3_N1, 32 * TN2, SA3_LDX, 1, 0, 0, true>(X, L2, bw2, acc2, ep2);
        float4 bw3[LayerCfg<SA3_N2, 32 * TN3>::DW + 1][TN3];
        w_prologue<SA3_N2, 32 * TN3, true>(L3, bw3);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                        // every wave has read the 256 columns layer 2 is about to overwrite
        epilogue<32 * TN2, SA3_LDX, false, 1>(X + wave * 32 * TN2, acc2, ep2, none2);
        __syncthreads();
        floatx16 acc3[1][TN3];
        float ep3[3][TN3];
        mfma_loop<SA3_N2, 32 * TN3, SA3_LDX, 1, 0, 0, true>(X, L3, bw3, acc3, ep3);
        epilogue<32 * TN3, SA3_LDX, true, 1>(X, acc3, ep3, pm);
    }
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < TN3; ++j) out[(size_t)tile * SA3_N3 + wave * 32 * TN3 + j * 32 + lane] = pm[j];
    }
}

// ---- the single-source share of fa_layer1's first layer ---------------------------------------------------------------------------
// fa_layer1 interpolates from ONE point per cloud (layer3's output): every level-2 point receives the same 1024 channels, so their
// share of the 1280 -> 256 product is one vector-matrix product per cloud (pointnet_util._fp_single_source).  x: per
// cloud `nparts` rows of cin values whose element-wise maximum is the cloud's row (layer3's per-tile maxima); y[cloud][col] = the RAW
// k-ordered fmaf chain over w[0:cin][col].  The chain is sequential in k, so the launch is bound by ONE wave's time: a wave owns one
// cloud x 128 columns (two chains per lane: a lone wave issues one VALU instruction per ~4.8 clocks, 2 x 1024 of them = 4 us); the
// cloud's row goes to LDS once (maxima taken in that load, every partial row in flight together); the kernel rows stream through
// a two-stage register ring (FI_U rows of 512 B per stage), the next stage's loads issued before the current stage's fmafs.
constexpr int FI_U = 32;

template <int NPARTS>
__global__ __launch_bounds__(64) void fp_init_kernel(int b, int cin, int cout, int nparts_rt, const float *__restrict__ x, ConvGroups G,
                                                     float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // cin
    const int lane = threadIdx.x;
    const int row = blockIdx.y, grp = row / b, col = blockIdx.x * 128 + lane * 2;
    const int nparts = NPARTS ? NPARTS : nparts_rt;
    const float *wc = G.wp[grp] + col;
    float2 wa[FI_U], wb[FI_U];
#pragma unroll
    for (int u = 0; u < FI_U; ++u) wa[u] = *reinterpret_cast<const float2 *>(wc + (size_t)u * cout);
    {
        const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)row * nparts * cin);
        const int c4n = cin / 4;
        for (int c0 = 0; c0 < c4n; c0 += 256) {                 // 4 float4 per lane per pass, all partial rows in flight
            float4 m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i * 64 + lane;
                m[i] = p[c < c4n ? c : c4n - 1];
            }
            if (NPARTS) {
                float4 v[NPARTS > 1 ? NPARTS - 1 : 1][4];
#pragma unroll
                for (int q = 1; q < NPARTS; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = c0 + i * 64 + lane;
                        v[q - 1][i] = p[(size_t)q * c4n + (c < c4n ? c : c4n - 1)];
                    }
#pragma unroll
                for (int q = 1; q < NPARTS; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        m[i].x = nmax(m[i].x, v[q - 1][i].x); m[i].y = nmax(m[i].y, v[q - 1][i].y);
                        m[i].z = nmax(m[i].z, v[q - 1][i].z); m[i].w = nmax(m[i].w, v[q - 1][i].w);
                    }
            } else {
                for (int q = 1; q < nparts; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = c0 + i * 64 + lane;
                        const float4 v = p[(size_t)q * c4n + (c < c4n ? c : c4n - 1)];
                        m[i].x = nmax(m[i].x, v.x); m[i].y = nmax(m[i].y, v.y); m[i].z = nmax(m[i].z, v.z); m[i].w = nmax(m[i].w, v.w);
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i * 64 + lane;
                if (c < c4n) reinterpret_cast<float4 *>(xs)[c] = m[i];
            }
        }
    }
    wave_lds_fence();
    float2 acc = make_float2(0.f, 0.f);
    const int last = cin - 1;
    for (int k0 = 0; k0 < cin; k0 += 2 * FI_U) {                // cin % (2 * FI_U) == 0 (launcher)
#pragma unroll
        for (int u = 0; u < FI_U; ++u) wb[u] = *reinterpret_cast<const float2 *>(wc + (size_t)(k0 + FI_U + u) * cout);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FI_U; u += 4) {
            const float4 xv = *reinterpret_cast<const float4 *>(xs + k0 + u);           // wave-uniform address: one broadcast read
            acc.x = __builtin_fmaf(xv.x, wa[u].x, acc.x);         acc.y = __builtin_fmaf(xv.x, wa[u].y, acc.y);
            acc.x = __builtin_fmaf(xv.y, wa[u + 1].x, acc.x);     acc.y = __builtin_fmaf(xv.y, wa[u + 1].y, acc.y);
            acc.x = __builtin_fmaf(xv.z, wa[u + 2].x, acc.x);     acc.y = __builtin_fmaf(xv.z, wa[u + 2].y, acc.y);
            acc.x = __builtin_fmaf(xv.w, wa[u + 3].x, acc.x);     acc.y = __builtin_fmaf(xv.w, wa[u + 3].y, acc.y);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FI_U; ++u) {                        // unconditional (row clamped): the last trip re-reads the last kernel row
            const int k = k0 + 2 * FI_U + u;
            wa[u] = *reinterpret_cast<const float2 *>(wc + (size_t)(k < last ? k : last) * cout);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FI_U; u += 4) {
            const float4 xv = *reinterpret_cast<const float4 *>(xs + k0 + FI_U + u);
            acc.x = __builtin_fmaf(xv.x, wb[u].x, acc.x);         acc.y = __builtin_fmaf(xv.x, wb[u].y, acc.y);
            acc.x = __builtin_fmaf(xv.y, wb[u + 1].x, acc.x);     acc.y = __builtin_fmaf(xv.y, wb[u + 1].y, acc.y);
            acc.x = __builtin_fmaf(xv.z, wb[u + 2].x, acc.x);     acc.y = __builtin_fmaf(xv.z, wb[u + 2].y, acc.y);
            acc.x = __builtin_fmaf(xv.w, wb[u + 3].x, acc.x);     acc.y = __builtin_fmaf(xv.w, wb[u + 3].y, acc.y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    *reinterpret_cast<float2 *>(y + (size_t)row * cout + col) = acc;
}

// ---- fa_layer1 after that: rows = level-2 points, skip features 256 -> 256 (chain continued from init[cloud]) -> 256 -----------------
// 8 waves x one column tile per layer; one 32 x 257 tile, rewritten in place.
constexpr int FP1_NW = 8, FP1_K = 256, FP1_N = 256, FP1_LD = 257;

__global__ __launch_bounds__(64 * FP1_NW) void fp1_chain_kernel(int npts, long tiles, const float *__restrict__ skip,
                                                                const float *__restrict__ init, int rows_per_group, MidGroups GL,
                                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 32;
    const long cloud = row0 / npts;                              // npts % 32 == 0: a tile never straddles clouds
    const int grp = (int)(row0 / rows_per_group);
    const SaLayer L1 = wave_slice<FP1_N, FP1_NW>(GL.L[grp][0], wave);
    const SaLayer L2 = wave_slice<FP1_N, FP1_NW>(GL.L[grp][1], wave);
    float4 bw1[LayerCfg<FP1_K, 32>::DW + 1][1];
    w_prologue<FP1_K, 32, true>(L1, bw1);
    {
        const float *f = skip + (size_t)row0 * FP1_K;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(f + (size_t)(i * 512 + tid) * 4);
        if (tid < 32) T[tid * FP1_LD + FP1_K] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 512 + tid, r = e >> 6, c4 = e & 63;
            float *d = T + r * FP1_LD + c4 * 4;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    }
    floatx16 acc[1][1];
    {
        const float a0 = init[(size_t)cloud * FP1_N + wave * 32 + l31];      // every row of the tile continues its cloud's chain
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = a0;
    }
    __syncthreads();
    float ep[3][1], none[1];
    mfma_loop<FP1_K, 32, FP1_LD, 1, 0, 2, true>(T, L1, bw1, acc, ep);
    float4 bw2[LayerCfg<FP1_N, 32>::DW + 1][1];
    w_prologue<FP1_N, 32, true>(L2, bw2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    epilogue<32, FP1_LD, false, 1>(T + wave * 32, acc, ep, none);
    __syncthreads();
    floatx16 acc2[1][1];
    float ep2[3][1];
    mfma_loop<FP1_N, 32, FP1_LD, 1, 0, 0, true>(T, L2, bw2, acc2, ep2);
    epilogue_global<32, 1, true>(out + wave * 32, FP1_N, 32, row0, tiles * 32, acc2, ep2);
}

// ---- fa_layer2: rows = level-1 points, [three_interpolate(level-2 features) (256) | level-1 features (128)] -> 256 -> 128 ------------
// 4 waves: 2 / 1 column tiles per wave; one 32 x 385 tile (49 KB: three workgroups per CU), rewritten in place.  The interpolation
// is the reference's expression p[i1] * w1 + p[i2] * w2 + p[i3] * w3 in that order, unfused (tf_interpolate.cpp:107-127).  XCD-aware
// tile -> cloud map as in fp_concat_kernel: a level-2 row is gathered by ~12 level-1 rows of its cloud.
constexpr int FP2_NW = 4, FP2_C2 = 256, FP2_C1 = 128, FP2_K1 = 384, FP2_N1 = 256, FP2_N2 = 128, FP2_LD = 385;

__global__ __launch_bounds__(64 * FP2_NW) void fp2_chain_kernel(int bgeo, int n, int m, long tiles, const float *__restrict__ points2,
                                                                const int *__restrict__ idx, const float *__restrict__ weight,
                                                                const float *__restrict__ points1, int rows_per_group, MidGroups GL,
                                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long tile = blockIdx.x;
    const int tpc = n / 32;
    {
        const long clouds = tiles / tpc;
        if ((clouds & 7) == 0) {
            const long xcd = tile & 7, j = tile >> 3;
            tile = (xcd + 8 * (j / tpc)) * tpc + j % tpc;
        }
    }
    const long row0 = tile * 32;
    const long cloud = tile / tpc;
    const long cg = cloud % bgeo;
    const int grp = (int)(row0 / rows_per_group);
    constexpr int TN1 = FP2_N1 / 32 / FP2_NW, TN2 = FP2_N2 / 32 / FP2_NW;
    const SaLayer L1 = wave_slice<FP2_N1, FP2_NW>(GL.L[grp][0], wave);
    const SaLayer L2 = wave_slice<FP2_N2, FP2_NW>(GL.L[grp][1], wave);
    float4 bw1[LayerCfg<FP2_K1, 32 * TN1>::DW + 1][TN1];
    w_prologue<FP2_K1, 32 * TN1, true>(L1, bw1);
    {
        // interpolated part: a wave takes the rows wave, wave + 4, ...: lane = one float4 of the 256 channels.  The 24 indices and
        // weights of its 8 rows are wave-uniform (scalar loads), then all 24 gathers are in flight together: two memory latencies per
        // workgroup instead of eight (the gather of a workgroup that runs alone on its CU is not hidden by anything)
        const float4 *p2 = reinterpret_cast<const float4 *>(points2) + (size_t)cloud * m * (FP2_C2 / 4) + lane;
        const long g0 = cg * n + (row0 - cloud * n);            // first row of the tile in the geometry arrays
        int ii[8][3];
        float w[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ii[j][q] = idx[(g0 + wave + 4 * j) * 3 + q];
                w[j][q] = weight[(g0 + wave + 4 * j) * 3 + q];
            }
        float4 a[4][3];                                         // two batches of four rows: 12 gathers in flight (register budget of 3 waves per SIMD)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[j][q] = p2[(size_t)ii[j][q] * (FP2_C2 / 4)];
        // skip part: 32 float4 per row, 256 threads = 8 rows per pass
        const float *f = points1 + (size_t)row0 * FP2_C1;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(f + (size_t)(i * 256 + tid) * 4);
        if (tid < 32) T[tid * FP2_LD + FP2_K1] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 nx[4][3];
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q) nx[j][q] = p2[(size_t)ii[4 + j][q] * (FP2_C2 / 4)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float *d = T + (wave + 4 * (4 * h + j)) * FP2_LD + lane * 4;
                const float w1 = w[4 * h + j][0], w2 = w[4 * h + j][1], w3 = w[4 * h + j][2];
                d[0] = a[j][0].x * w1 + a[j][1].x * w2 + a[j][2].x * w3;
                d[1] = a[j][0].y * w1 + a[j][1].y * w2 + a[j][2].y * w3;
                d[2] = a[j][0].z * w1 + a[j][1].z * w2 + a[j][2].z * w3;
                d[3] = a[j][0].w * w1 + a[j][1].w * w2 + a[j][2].w * w3;
            }
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a[j][q] = nx[j][q];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 256 + tid, r = e >> 5, c4 = e & 31;
            float *d = T + r * FP2_LD + FP2_C2 + c4 * 4;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    }
    __syncthreads();
    floatx16 acc[1][TN1];
    float ep[3][TN1], none[TN1];
    mfma_loop<FP2_K1, 32 * TN1, FP2_LD, 1, 0, 0, true>(T, L1, bw1, acc, ep);
    float4 bw2[LayerCfg<FP2_N1, 32 * TN2>::DW + 1][TN2];
    w_prologue<FP2_N1, 32 * TN2, true>(L2, bw2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    epilogue<32 * TN1, FP2_LD, false, 1>(T + wave * 32 * TN1, acc, ep, none);
    __syncthreads();
    floatx16 acc2[1][TN2];
    float ep2[3][TN2];
    mfma_loop<FP2_N1, 32 * TN2, FP2_LD, 1, 0, 0, true>(T, L2, bw2, acc2, ep2);
    epilogue_global<32 * TN2, 1, true>(out + wave * 32 * TN2, FP2_N2, 32 * TN2, row0, tiles * 32, acc2, ep2);
}

static int mid_layers(const float *const *params, int ngroups, int nlayers, MidGroups &GL, const char *who) {
    for (int g = 0; g < ANCSH_MAX_GROUPS; ++g) {
        const float *const *pp = params + 4 * nlayers * (g < ngroups ? g : 0);
        for (int i = 0; i < 3; ++i) {
            MidLayer &L = GL.L[g][i];
            const int s = i < nlayers ? i : 0;
            L.w = pp[4 * s]; L.bias = pp[4 * s + 1]; L.scale = pp[4 * s + 2]; L.shift = pp[4 * s + 3];
            ANCSH_REQUIRE(L.w && L.bias && L.scale && L.shift, "%s: null layer parameter", who);
            ANCSH_REQUIRE((((uintptr_t)L.w) & 15) == 0, "%s: packed kernels must be 16-byte aligned", who);
        }
    }
    return ANCSH_OK;
}

}  // namespace ancsh

using namespace ancsh;

// layer3 of `ngroups` networks on the same b clouds: xyz (b, npts, 3) shared, feats (ngroups * b, npts, 256) network-major;
// params = per network 3 x {packed w, bias, scale, shift} for 259 -> 256 -> 512 -> 1024; out (ngroups * b, npts / 32, 1024): the
// maxima over each 32-row tile (the level's output row of a cloud is the element-wise maximum of its npts / 32 rows).
extern "C" int ancsh_sa3_chain_grouped(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                                       const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "sa3_chain_grouped: ngroups=%d must be in [1,%d]", ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && npts > 0 && npts % 32 == 0, "sa3_chain_grouped: bad shape b=%d npts=%d (npts must be a multiple of 32)", b, npts);
    ANCSH_REQUIRE(cfeat == 256 && c1 == SA3_N1 && c2 == SA3_N2 && c3 == SA3_N3,
                  "sa3_chain_grouped: unsupported layer shape (cfeat=%d mlp=[%d,%d,%d]); use the layer-by-layer path", cfeat, c1, c2, c3);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && feats && params && out, "sa3_chain_grouped: null pointer");
    ANCSH_REQUIRE((((uintptr_t)feats) & 15) == 0, "sa3_chain_grouped: feats must be 16-byte aligned");
    MidGroups GL;
    if (int rc = mid_layers(params, ngroups, 3, GL, "sa3_chain_grouped")) return rc;
    const long tiles = (long)ngroups * b * (npts / 32);
    const size_t lds = sizeof(float) * 32 * (SA3_LDY + SA3_LDX);
    (void)hipFuncSetAttribute((const void *)sa3_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sa3_chain_kernel, dim3((unsigned)tiles), dim3(64 * SA3_NW), lds, (hipStream_t)stream, b, npts, tiles, xyz, feats, GL, out);
    return check_launch("sa3_chain_grouped");
}

// y (ngroups * b, cout) = RAW k-ordered chains of max_q x[(row * nparts + q)] (cin) over the plain kernels w[g] (cin, cout): the
// single-source share of an FP module's first layer (see fp_init_kernel).  b clouds per network.
extern "C" int ancsh_fp_single_source_init(int ngroups, int b, int cin, int cout, int nparts, const float *x, const float *const *w, float *y,
                                           void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "fp_single_source_init: ngroups=%d must be in [1,%d]", ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && cin > 0 && cin % (2 * FI_U) == 0 && cout > 0 && cout % 128 == 0 && nparts >= 1,
                  "fp_single_source_init: bad shape b=%d cin=%d (multiple of %d) cout=%d (multiple of 128) nparts=%d", b, cin, 2 * FI_U, cout, nparts);
    // the input row is staged in dynamic LDS: 48 KB is what a launch gets without raising MaxDynamicSharedMemorySize (the path's cin is 1024)
    ANCSH_REQUIRE((size_t)cin * sizeof(float) <= 48 * 1024 && (long)ngroups * b <= 65535, "fp_single_source_init: cin=%d (max 12288) / b=%d too large", cin, b);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(x && w && y, "fp_single_source_init: null pointer");
    ConvGroups G{};
    G.n = ngroups;
    for (int g = 0; g < CONV_MAX_GROUPS; ++g) {
        G.wp[g] = w[g < ngroups ? g : 0];
        ANCSH_REQUIRE(G.wp[g] && (((uintptr_t)G.wp[g]) & 7) == 0, "fp_single_source_init: null / unaligned kernel of group %d", g);
    }
    ANCSH_REQUIRE((((uintptr_t)y) & 7) == 0 && (((uintptr_t)x) & 15) == 0, "fp_single_source_init: x must be 16-byte, y 8-byte aligned");
    const dim3 grid(cout / 128, ngroups * b);
    const size_t lds = sizeof(float) * cin;
    hipStream_t st = (hipStream_t)stream;
    switch (nparts) {
    case 1: hipLaunchKernelGGL(fp_init_kernel<1>, grid, dim3(64), lds, st, b, cin, cout, nparts, x, G, y); break;
    case 4: hipLaunchKernelGGL(fp_init_kernel<4>, grid, dim3(64), lds, st, b, cin, cout, nparts, x, G, y); break;
    default: hipLaunchKernelGGL(fp_init_kernel<0>, grid, dim3(64), lds, st, b, cin, cout, nparts, x, G, y); break;
    }
    return check_launch("fp_single_source_init");
}

// fa_layer1 after ancsh_fp_single_source_init: skip (ngroups * b * npts, 256) level-2 features, init (ngroups * b, 256); params = per
// network 2 x {packed w, bias, scale, shift}: the first layer's kernel rows [1024:1280] (256 -> 256) and the second layer (256 -> 256);
// out (ngroups * b * npts, 256).
extern "C" int ancsh_fp1_chain_grouped(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                                       const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "fp1_chain_grouped: ngroups=%d must be in [1,%d]", ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && npts > 0 && npts % 32 == 0, "fp1_chain_grouped: bad shape b=%d npts=%d (npts must be a multiple of 32)", b, npts);
    ANCSH_REQUIRE(cskip == FP1_K && c1 == FP1_N && c2 == FP1_N, "fp1_chain_grouped: unsupported layer shape (%d -> %d -> %d); use the layer-by-layer path",
                  cskip, c1, c2);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(skip && init && params && out, "fp1_chain_grouped: null pointer");
    ANCSH_REQUIRE((((uintptr_t)skip) & 15) == 0, "fp1_chain_grouped: skip must be 16-byte aligned");
    MidGroups GL;
    if (int rc = mid_layers(params, ngroups, 2, GL, "fp1_chain_grouped")) return rc;
    const long tiles = (long)ngroups * b * (npts / 32);
    const size_t lds = sizeof(float) * 32 * FP1_LD;
    hipLaunchKernelGGL(fp1_chain_kernel, dim3((unsigned)tiles), dim3(64 * FP1_NW), lds, (hipStream_t)stream, npts, tiles, skip, init, b * npts, GL, out);
    return check_launch("fp1_chain_grouped");
}

// fa_layer2: points2 (ngroups * b, m, 256) level-2 features, idx / weight (b, n, 3) from ancsh_three_nn_weights (shared geometry),
// points1 (ngroups * b, n, 128) level-1 features; params = per network 2 x {packed w, bias, scale, shift} for 384 -> 256 -> 128;
// out (ngroups * b * n, 128).
extern "C" int ancsh_fp2_chain_grouped(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                                       const float *weight, const float *points1, const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "fp2_chain_grouped: ngroups=%d must be in [1,%d]", ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && m > 0 && n > 0 && n % 32 == 0, "fp2_chain_grouped: bad shape b=%d m=%d n=%d (n must be a multiple of 32)", b, m, n);
    ANCSH_REQUIRE(c2 == FP2_C2 && c1 == FP2_C1 && n1 == FP2_N1 && n2 == FP2_N2,
                  "fp2_chain_grouped: unsupported layer shape ([%d | %d] -> %d -> %d); use the layer-by-layer path", c2, c1, n1, n2);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(points2 && idx && weight && points1 && params && out, "fp2_chain_grouped: null pointer");
    ANCSH_REQUIRE((((uintptr_t)points2 | (uintptr_t)points1) & 15) == 0, "fp2_chain_grouped: points2 / points1 must be 16-byte aligned");
    MidGroups GL;
    if (int rc = mid_layers(params, ngroups, 2, GL, "fp2_chain_grouped")) return rc;
    const long tiles = (long)ngroups * b * (n / 32);
    const size_t lds = sizeof(float) * 32 * FP2_LD;
    (void)hipFuncSetAttribute((const void *)fp2_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fp2_chain_kernel, dim3((unsigned)tiles), dim3(64 * FP2_NW), lds, (hipStream_t)stream, b, n, m, tiles, points2, idx, weight, points1,
                       b * n, GL, out);
    return check_launch("fp2_chain_grouped");
}
