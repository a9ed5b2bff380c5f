// mid_bf16x3.hip -- EXPERIMENT (opt-in, never the default): the MIDDLE of the backbone -- layer3 (group-all set abstraction), fa_layer1,
// fa_layer2 (pointnet_plusplus/architectures.py:72-82; mid_chain.hip is the graded f32 form) -- on the 16-bit matrix pipe with the
// split-16 schemes of bx3.h (Bf16x3 / F16x2).
//
// These layers are too wide for the register tiles of sa_bf16x3.hip / tail_bf16x3.hip (256 .. 1024 channels), so the activations of a
// 64-row tile live in LDS as the scheme's 16-bit PLANES, row-major [plane][row][channel] with a padded row stride, and the four waves of
// a workgroup split every layer by output channels:
//   * a wave's k loop reads each activation fragment with ONE ds_read_b128 per plane and point block (a lane's 8 consecutive channels of
//     row l31 are the MFMA operand whichever way the tile is used) and feeds NPROD x (its tiles) x 2 MFMAs with it; its weight fragments
//     stream from L2 one k-block ahead (bx3.h's double buffer);
//   * hidden layers are computed TRANSPOSED (weights as the A operand): an accumulator then holds, per lane, one point and four runs of
//     four consecutive channels, so after BN / ReLU and the split each run is ONE 8-byte LDS store per plane -- the LDS does the
//     transposition the register kernels need v_permlane32_swap for;
//   * a layer's outputs overwrite the tile IN PLACE between two workgroup barriers (every wave has consumed the whole tile into its
//     accumulators before the first store), the layer that leaves the chip runs in the orientation its consumer wants: layer3's 1024-wide
//     last layer NORMAL (max over the tile's rows = max over accumulator registers; the consumer takes the maximum over a cloud's tiles,
//     exact in any order), the feature-propagation levels' last layers TRANSPOSED (a lane stores float4 runs of an output row).
// 64 rows per workgroup (two point blocks per weight fragment): layer3's three kernels are 2.9 MB in the F16x2 packing, and with 32 rows
// the launch would read 740 MB of weights from L2 for 8192 rows.
#include "bx3.h"

namespace ancsh {

constexpr int MS_P = 2;         // point blocks of 32 rows per workgroup
constexpr int MS_R = 32 * MS_P;

struct MsGroups {
    Bx3Layer L[ANCSH_MAX_GROUPS][3];
};

// halves per LDS row for K channels: K rounded up to the MFMA's 16-channel k-block + 8 (row stride = 16 bytes mod 128: the 16 lanes of a
// ds_read_b128 phase hit disjoint banks)
__host__ __device__ constexpr int ms_ld(int K) { return ((K + 15) / 16) * 16 + 8; }

template <class S>
struct MsTile {
    unsigned short *base;       // [S::NP][MS_R][ld]
    int ld;
    __device__ __forceinline__ unsigned short *at(int plane, int row, int ch) const { return base + ((size_t)plane * MS_R + row) * ld + ch; }
};

// stage f32 values v[0..7] (8 consecutive channels c0 .. c0 + 7 of one row; c0 % 8 == 0) as the scheme's planes
template <class S>
__device__ __forceinline__ void ms_store8(const MsTile<S> &T, int row, int c0, const float (&v)[8]) {
    u32 s[4][S::NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) S::split2(v[2 * j], v[2 * j + 1], s[j]);
#pragma unroll
    for (int pl = 0; pl < S::NP; ++pl) *reinterpret_cast<uint4 *>(T.at(pl, row, c0)) = make_uint4(s[0][pl], s[1][pl], s[2][pl], s[3][pl]);
}

// this wave's k loop over the tile: acc[j][p] (+)= W[tile j0 + j] * X[p] over KB k-blocks; TRANSPOSED: weights are the A operand.
// Wp: the layer's packed fragments + lane; TN: the layer's total 32-channel tiles (the packing's stride).
// D: how many k-blocks ahead the weight fragments are requested (a ring of D + 1 register slots).  A k-block is only NPROD x TW x 2 MFMAs
// (200 .. 800 clocks) against an L2 latency of ~600: with D = 1 layer3 ran at 65 us for 15 us of MFMA issue.
template <class S, int KB, int TW, bool TRANSPOSED, int D = 3>
__device__ __forceinline__ void ms_k_loop(const uint4 *Wp, int TN, int j0, const MsTile<S> &T, fx16 (&acc)[TW][MS_P][S::NACC]) {
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    uint4 w[D + 1][TW][S::NP];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < KB) {
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int pl = 0; pl < S::NP; ++pl) w[d][j][pl] = Wp[(size_t)((d * TN + j0 + j) * S::NP + pl) * 64];
        }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + D < KB) {
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int pl = 0; pl < S::NP; ++pl) w[(kb + D) % (D + 1)][j][pl] = Wp[(size_t)(((kb + D) * TN + j0 + j) * S::NP + pl) * 64];
        }
        uint4 x[MS_P][S::NP];
#pragma unroll
        for (int p = 0; p < MS_P; ++p)
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) x[p][pl] = *reinterpret_cast<const uint4 *>(T.at(pl, 32 * p + l31, 16 * kb + 8 * khalf));
        __builtin_amdgcn_sched_barrier(0);             // (without the fences the scheduler hoists every k-block's loads: 420 spilled registers)
#pragma unroll
        for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int p = 0; p < MS_P; ++p)
                    acc[j][p][S::PC[t]] = TRANSPOSED ? S::mfma(w[kb % (D + 1)][j][S::PW[t]], x[p][S::PA[t]], acc[j][p][S::PC[t]])
                                                     : S::mfma(x[p][S::PW[t]], w[kb % (D + 1)][j][S::PA[t]], acc[j][p][S::PC[t]]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// epilogue of a TRANSPOSED tile (channels 32 i ..): BN (+ ReLU) of the lane's 4 x 4 channels of point l31 of every point block, then either the
// split planes into the LDS tile (hidden layer) or float4 runs of a global f32 row
template <class S, bool RELU, bool TO_LDS>
__device__ __forceinline__ void ms_epilogue(const Bx3Layer &L, int i, const fx16 (&acc)[MS_P][S::NACC], const MsTile<S> &T, float *out, int out_ld,
                                            long row0) {
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const int c0 = 32 * i + 4 * khalf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4 *>(L.bias + c0 + 8 * q), s4 = *reinterpret_cast<const float4 *>(L.scale + c0 + 8 * q),
                     h4 = *reinterpret_cast<const float4 *>(L.shift + c0 + 8 * q);
        const f32x2v sc0 = {s4.x, s4.y}, sc1 = {s4.z, s4.w};
        const f32x2v shf0 = __builtin_elementwise_fma(f32x2v{b4.x, b4.y}, sc0, f32x2v{h4.x, h4.y}), shf1 = __builtin_elementwise_fma(f32x2v{b4.z, b4.w}, sc1, f32x2v{h4.z, h4.w});
        const f32x2v scl0 = sc0 * f32x2v{1.f / 2048.f, 1.f / 2048.f}, scl1 = sc1 * f32x2v{1.f / 2048.f, 1.f / 2048.f};
#pragma unroll
        for (int p = 0; p < MS_P; ++p) {
            f32x2v a01 = S::bn2(acc[p], 4 * q, sc0, scl0, shf0), a23 = S::bn2(acc[p], 4 * q + 2, sc1, scl1, shf1);
            if (RELU) { a01.x = nmax(a01.x, 0.f); a01.y = nmax(a01.y, 0.f); a23.x = nmax(a23.x, 0.f); a23.y = nmax(a23.y, 0.f); }
            if (TO_LDS) {
                u32 s01[S::NP], s23[S::NP];
                S::split2(a01.x, a01.y, s01);
                S::split2(a23.x, a23.y, s23);
#pragma unroll
                for (int pl = 0; pl < S::NP; ++pl) *reinterpret_cast<uint2 *>(T.at(pl, 32 * p + l31, c0 + 8 * q)) = make_uint2(s01[pl], s23[pl]);
            } else {
                *reinterpret_cast<float4 *>(out + (size_t)(row0 + 32 * p + l31) * out_ld + c0 + 8 * q) = make_float4(a01.x, a01.y, a23.x, a23.y);
            }
        }
    }
}

// a hidden layer K -> N over the whole workgroup, in place: every wave computes N / 32 / NW tiles for both point blocks, then -- between two
// barriers -- rewrites the tile.  init != nullptr: every row's accumulators start from init[0:N] (fa_layer1: the cloud's single-source share).
template <class S, int NW, int K, int N, bool RELU>
__device__ __forceinline__ void ms_hidden(const Bx3Layer &L, MsTile<S> &T, const float *init) {
    constexpr int KB = (K + 15) / 16, TN = N / 32, TW = TN / NW;
    static_assert(TN % NW == 0, "output tiles per wave");
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    fx16 acc[TW][MS_P][S::NACC];
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        bx3_zero<S, MS_P>(acc[j]);
        if (init) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(init + 32 * (wave * TW + j) + 4 * khalf + 8 * q);
#pragma unroll
                for (int p = 0; p < MS_P; ++p) { acc[j][p][0][4 * q] = v.x; acc[j][p][0][4 * q + 1] = v.y; acc[j][p][0][4 * q + 2] = v.z; acc[j][p][0][4 * q + 3] = v.w; }
            }
        }
    }
    ms_k_loop<S, KB, TW, true>(L.w + lane, TN, wave * TW, T, acc);
    __syncthreads();                                   // every wave has read the whole tile
    T.ld = ms_ld(N);                                   // the tile takes the output's row stride
#pragma unroll
    for (int j = 0; j < TW; ++j) ms_epilogue<S, RELU, true>(L, wave * TW + j, acc[j], T, nullptr, 0, 0);
    __syncthreads();                                   // (N % 16 == 0 for every layer here: the next layer reads no k-block padding)
}

// last layer of a feature-propagation level: K -> N, ReLU, f32 rows to global memory
template <class S, int NW, int K, int N>
__device__ __forceinline__ void ms_out_rows(const Bx3Layer &L, const MsTile<S> &T, float *out, long row0) {
    constexpr int KB = (K + 15) / 16, TN = N / 32, TW = TN / NW;
    static_assert(TN % NW == 0, "output tiles per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    fx16 acc[TW][MS_P][S::NACC];
#pragma unroll
    for (int j = 0; j < TW; ++j) bx3_zero<S, MS_P>(acc[j]);
    ms_k_loop<S, KB, TW, true>(L.w + lane, TN, wave * TW, T, acc);
#pragma unroll
    for (int j = 0; j < TW; ++j) ms_epilogue<S, true, false>(L, wave * TW + j, acc[j], T, out, N, row0);
}

// layer3's last layer: K -> N, ReLU, max over the tile's 64 rows -> out[0:N]; NORMAL orientation, the wave's N / 32 / NW tiles in groups of TG
template <class S, int NW, int K, int N, int TG>
__device__ __forceinline__ void ms_out_pooled(const Bx3Layer &L, const MsTile<S> &T, float *out) {
    constexpr int KB = (K + 15) / 16, TN = N / 32, TW = TN / NW;
    static_assert(TN % NW == 0 && TW % TG == 0, "output tiles per wave");
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int g = 0; g < TW; g += TG) {
        fx16 acc[TG][MS_P][S::NACC];
#pragma unroll
        for (int j = 0; j < TG; ++j) bx3_zero<S, MS_P>(acc[j]);
        ms_k_loop<S, KB, TG, false>(L.w + lane, TN, wave * TW + g, T, acc);
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int col = (wave * TW + g + j) * 32 + l31;
            const float sc = L.scale[col], shf = __builtin_fmaf(L.bias[col], sc, L.shift[col]);
            const f32x2v sc2 = {sc, sc}, scl2 = {sc * (1.f / 2048.f), sc * (1.f / 2048.f)}, shf2 = {shf, shf};
            float mx = 0.f;                            // the maximum starts at 0: the ReLU is implicit
#pragma unroll
            for (int p = 0; p < MS_P; ++p)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2v v = S::bn2(acc[j][p], r, sc2, scl2, shf2);
                    mx = nmax(nmax(mx, v.x), v.y);
                }
            mx = nmax(mx, __shfl_xor(mx, 32, 64));
            if (lane < 32) out[col] = mx;
        }
    }
}

// ---- layer3: rows [xyz (3) | features (256)] -> 256 -> 512 -> 1024, max over the tile's 64 rows ------------------------------------
// 8 waves (two per SIMD: 1 / 2 / 4 output tiles per wave and layer; with four waves the 512- and 1024-wide layers' accumulators spilled)
constexpr int SA3S_NW = 8;
template <class S>
__global__ __launch_bounds__(64 * SA3S_NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sa3_split16_kernel(int bgeo, int npts, long tiles, const float *__restrict__ xyz, const float *__restrict__ feats, MsGroups GL, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int tid = threadIdx.x;
    const long tile = blockIdx.x;
    const int tpc = npts / MS_R;                               // row tiles per cloud
    const long cloud = tile / tpc;
    const int rt = (int)(tile - cloud * tpc);
    const int grp = (int)(cloud / bgeo);
    const long cg = cloud - (long)grp * bgeo;                  // geometry cloud
    MsTile<S> T{smem16, ms_ld(259)};
    // stage [xyz | features | 0 ..]: channels 0..2 xyz, 3..258 features, 259..271 zero.  A thread owns 8 consecutive channels of a row.
    {
        constexpr int C8 = 272 / 8;                            // 34 groups of 8 channels per row
        for (int e = tid; e < MS_R * C8; e += 64 * SA3S_NW) {
            const int r = e / C8, c0 = (e - r * C8) * 8;
            const float *f = feats + ((size_t)cloud * npts + rt * MS_R + r) * 256;
            const float *x = xyz + ((size_t)cg * npts + rt * MS_R + r) * 3;
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                v[u] = c < 3 ? x[c] : (c < 259 ? f[c - 3] : 0.f);
            }
            ms_store8<S>(T, r, c0, v);
        }
    }
    __syncthreads();
    ms_hidden<S, SA3S_NW, 259, 256, true>(GL.L[grp][0], T, nullptr);
    ms_hidden<S, SA3S_NW, 256, 512, true>(GL.L[grp][1], T, nullptr);
    ms_out_pooled<S, SA3S_NW, 512, 1024, 2>(GL.L[grp][2], T, out + (size_t)tile * 1024);
}

// ---- fa_layer1: rows = level-2 points, skip features 256 -> 256 (chain continued from init[cloud]) -> 256 ----------------------------
constexpr int FPS_NW = 4;       // the feature-propagation levels: four waves, one per SIMD
template <class S>
__global__ __launch_bounds__(64 * FPS_NW) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fp1_split16_kernel(int npts, long tiles, const float *__restrict__ skip, const float *__restrict__ init, int rows_per_group, MsGroups GL,
                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * MS_R;
    const long cloud = row0 / npts;                            // npts % 64 == 0: a tile never straddles clouds
    const int grp = (int)(row0 / rows_per_group);
    MsTile<S> T{smem16, ms_ld(256)};
    for (int e = tid; e < MS_R * 32; e += 64 * FPS_NW) {        // 32 groups of 8 channels per row
        const int r = e >> 5, c0 = (e & 31) * 8;
        const float4 a = *reinterpret_cast<const float4 *>(skip + (size_t)(row0 + r) * 256 + c0), b = *reinterpret_cast<const float4 *>(skip + (size_t)(row0 + r) * 256 + c0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        ms_store8<S>(T, r, c0, v);
    }
    __syncthreads();
    ms_hidden<S, FPS_NW, 256, 256, true>(GL.L[grp][0], T, init + (size_t)cloud * 256);
    ms_out_rows<S, FPS_NW, 256, 256>(GL.L[grp][1], T, out, row0);
}

// ---- fa_layer2: rows = level-1 points, [three_interpolate(level-2 features) (256) | level-1 features (128)] -> 256 -> 128 ------------
template <class S>
__global__ __launch_bounds__(64 * FPS_NW) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fp2_split16_kernel(int bgeo, int n, int m, long tiles, const float *__restrict__ points2, const int *__restrict__ idx, const float *__restrict__ weight,
                        const float *__restrict__ points1, int rows_per_group, MsGroups GL, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    const int tid = threadIdx.x;
    long tile = blockIdx.x;
    const int tpc = n / MS_R;
    {
        const long clouds = tiles / tpc;                       // XCD-aware tile -> cloud map (mid_chain.hip): a level-2 row is gathered by ~12 level-1 rows of its cloud
        if ((clouds & 7) == 0) {
            const long xcd = tile & 7, j = tile >> 3;
            tile = (xcd + 8 * (j / tpc)) * tpc + j % tpc;
        }
    }
    const long row0 = tile * MS_R;
    const long cloud = tile / tpc;
    const long cg = cloud % bgeo;
    const int grp = (int)(row0 / rows_per_group);
    MsTile<S> T{smem16, ms_ld(384)};
    const long g0 = cg * n + (row0 - cloud * n);               // first row of the tile in the geometry arrays
    // interpolated part: p[i1] * w1 + p[i2] * w2 + p[i3] * w3 in that order, unfused (tf_interpolate.cpp:107-127); 32 groups of 8 channels per row
    for (int e = tid; e < MS_R * 32; e += 64 * FPS_NW) {
        const int r = e >> 5, c0 = (e & 31) * 8;
        const float *p2 = points2 + (size_t)cloud * m * 256 + c0;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 a[3][2];
        float w[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int ii = idx[(g0 + r) * 3 + q];
            w[q] = weight[(g0 + r) * 3 + q];
            a[q][0] = *reinterpret_cast<const float4 *>(p2 + (size_t)ii * 256);
            a[q][1] = *reinterpret_cast<const float4 *>(p2 + (size_t)ii * 256 + 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v[4 * h + 0] = a[0][h].x * w[0] + a[1][h].x * w[1] + a[2][h].x * w[2];
            v[4 * h + 1] = a[0][h].y * w[0] + a[1][h].y * w[1] + a[2][h].y * w[2];
            v[4 * h + 2] = a[0][h].z * w[0] + a[1][h].z * w[1] + a[2][h].z * w[2];
            v[4 * h + 3] = a[0][h].w * w[0] + a[1][h].w * w[1] + a[2][h].w * w[2];
        }
        ms_store8<S>(T, r, c0, v);
    }
    for (int e = tid; e < MS_R * 16; e += 64 * FPS_NW) {        // skip part: 16 groups of 8 channels per row
        const int r = e >> 4, c0 = (e & 15) * 8;
        const float4 a = *reinterpret_cast<const float4 *>(points1 + (size_t)(row0 + r) * 128 + c0), b = *reinterpret_cast<const float4 *>(points1 + (size_t)(row0 + r) * 128 + c0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        ms_store8<S>(T, r, 256 + c0, v);
    }
    __syncthreads();
    ms_hidden<S, FPS_NW, 384, 256, true>(GL.L[grp][0], T, nullptr);
    ms_out_rows<S, FPS_NW, 256, 128>(GL.L[grp][1], T, out, row0);
}

static int ms_layers(const float *const *params, int ngroups, int nlayers, MsGroups &GL, const char *who) {
    ANCSH_REQUIRE(params, "%s: null parameter table", who);
    for (int g = 0; g < ANCSH_MAX_GROUPS; ++g) {
        const float *const *pp = params + 4 * nlayers * (g < ngroups ? g : 0);
        for (int i = 0; i < 3; ++i) {
            Bx3Layer &L = GL.L[g][i];
            const int s = i < nlayers ? i : 0;
            L.w = reinterpret_cast<const uint4 *>(pp[4 * s]); L.bias = pp[4 * s + 1]; L.scale = pp[4 * s + 2]; L.shift = pp[4 * s + 3];
            ANCSH_REQUIRE(L.w && L.bias && L.scale && L.shift, "%s: null layer parameter", who);
            ANCSH_REQUIRE(((((uintptr_t)L.w) | (uintptr_t)L.bias | (uintptr_t)L.scale | (uintptr_t)L.shift) & 15) == 0, "%s: parameters must be 16-byte aligned", who);
        }
    }
    return ANCSH_OK;
}

template <class S>
static size_t ms_lds_bytes(int kmax) { return (size_t)S::NP * MS_R * ms_ld(kmax) * sizeof(unsigned short); }

template <class S>
static int sa3_split16(const char *who, int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                       const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && npts > 0 && npts % MS_R == 0, "%s: bad shape b=%d npts=%d (npts must be a multiple of %d)", who, b, npts, MS_R);
    ANCSH_REQUIRE(cfeat == 256 && c1 == 256 && c2 == 512 && c3 == 1024, "%s: unsupported layer shape (cfeat=%d mlp=[%d,%d,%d])", who, cfeat, c1, c2, c3);
    MsGroups GL;
    if (int rc = ms_layers(params, ngroups, 3, GL, who)) return rc;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && feats && out, "%s: null pointer", who);
    const long tiles = (long)ngroups * b * (npts / MS_R);
    const size_t lds = ms_lds_bytes<S>(512);
    ANCSH_REQUIRE(lds <= 160 * 1024, "%s: a 64-row tile of 512 channels needs %zu bytes of LDS in this scheme (160 KB per CU): layer3 takes the f32 chain", who, lds);
    (void)hipFuncSetAttribute((const void *)sa3_split16_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(sa3_split16_kernel<S>, dim3((unsigned)tiles), dim3(64 * SA3S_NW), lds, (hipStream_t)stream, b, npts, tiles, xyz, feats, GL, out);
    return check_launch(who);
}

template <class S>
static int fp1_split16(const char *who, int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                       const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && npts > 0 && npts % MS_R == 0, "%s: bad shape b=%d npts=%d (npts must be a multiple of %d)", who, b, npts, MS_R);
    ANCSH_REQUIRE(cskip == 256 && c1 == 256 && c2 == 256, "%s: unsupported layer shape (%d -> %d -> %d)", who, cskip, c1, c2);
    MsGroups GL;
    if (int rc = ms_layers(params, ngroups, 2, GL, who)) return rc;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(skip && init && out, "%s: null pointer", who);
    ANCSH_REQUIRE((((uintptr_t)skip | (uintptr_t)init | (uintptr_t)out) & 15) == 0, "%s: skip / init / out must be 16-byte aligned", who);
    const long tiles = (long)ngroups * b * (npts / MS_R);
    const size_t lds = ms_lds_bytes<S>(256);
    (void)hipFuncSetAttribute((const void *)fp1_split16_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fp1_split16_kernel<S>, dim3((unsigned)tiles), dim3(64 * FPS_NW), lds, (hipStream_t)stream, npts, tiles, skip, init, b * npts, GL, out);
    return check_launch(who);
}

template <class S>
static int fp2_split16(const char *who, int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                       const float *weight, const float *points1, const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && m > 0 && n > 0 && n % MS_R == 0, "%s: bad shape b=%d m=%d n=%d (n must be a multiple of %d)", who, b, m, n, MS_R);
    ANCSH_REQUIRE(c2 == 256 && c1 == 128 && n1 == 256 && n2 == 128, "%s: unsupported layer shape ([%d | %d] -> %d -> %d)", who, c2, c1, n1, n2);
    MsGroups GL;
    if (int rc = ms_layers(params, ngroups, 2, GL, who)) return rc;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(points2 && idx && weight && points1 && out, "%s: null pointer", who);
    ANCSH_REQUIRE((((uintptr_t)points2 | (uintptr_t)points1 | (uintptr_t)out) & 15) == 0, "%s: points2 / points1 / out must be 16-byte aligned", who);
    const long tiles = (long)ngroups * b * (n / MS_R);
    const size_t lds = ms_lds_bytes<S>(384);
    (void)hipFuncSetAttribute((const void *)fp2_split16_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fp2_split16_kernel<S>, dim3((unsigned)tiles), dim3(64 * FPS_NW), lds, (hipStream_t)stream, b, n, m, tiles, points2, idx, weight, points1,
                       b * n, GL, out);
    return check_launch(who);
}

}  // namespace ancsh

using namespace ancsh;

// The arguments of ancsh_sa3_chain_grouped / ancsh_fp1_chain_grouped / ancsh_fp2_chain_grouped (mid_chain.hip) with kernels packed by
// ancsh_sa_pack_weights_{bf16x3,f16x2} (the first layers' kernel rows as the f32 forms take them: layer3 all 259 rows in [xyz | features]
// order, fa_layer1 rows 1024.., fa_layer2 all 384) and 16-byte aligned bias / scale / shift.  Row tiles of 64: npts (n) % 64 == 0, and
// layer3's out is (ngroups * b, npts / 64, 1024): the maxima of every 64-row tile.
extern "C" int ancsh_sa3_chain_grouped_bf16x3(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                                              const float *const *params, float *out, void *stream) {
    return sa3_split16<Bf16x3>("sa3_chain_grouped_bf16x3", ngroups, b, npts, cfeat, c1, c2, c3, xyz, feats, params, out, stream);
}
extern "C" int ancsh_sa3_chain_grouped_f16x2(int ngroups, int b, int npts, int cfeat, int c1, int c2, int c3, const float *xyz, const float *feats,
                                             const float *const *params, float *out, void *stream) {
    return sa3_split16<F16x2>("sa3_chain_grouped_f16x2", ngroups, b, npts, cfeat, c1, c2, c3, xyz, feats, params, out, stream);
}
extern "C" int ancsh_fp1_chain_grouped_bf16x3(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                                              const float *const *params, float *out, void *stream) {
    return fp1_split16<Bf16x3>("fp1_chain_grouped_bf16x3", ngroups, b, npts, cskip, c1, c2, skip, init, params, out, stream);
}
extern "C" int ancsh_fp1_chain_grouped_f16x2(int ngroups, int b, int npts, int cskip, int c1, int c2, const float *skip, const float *init,
                                             const float *const *params, float *out, void *stream) {
    return fp1_split16<F16x2>("fp1_chain_grouped_f16x2", ngroups, b, npts, cskip, c1, c2, skip, init, params, out, stream);
}
extern "C" int ancsh_fp2_chain_grouped_bf16x3(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                                              const float *weight, const float *points1, const float *const *params, float *out, void *stream) {
    return fp2_split16<Bf16x3>("fp2_chain_grouped_bf16x3", ngroups, b, m, n, c2, c1, n1, n2, points2, idx, weight, points1, params, out, stream);
}
extern "C" int ancsh_fp2_chain_grouped_f16x2(int ngroups, int b, int m, int n, int c2, int c1, int n1, int n2, const float *points2, const int *idx,
                                             const float *weight, const float *points1, const float *const *params, float *out, void *stream) {
    return fp2_split16<F16x2>("fp2_chain_grouped_f16x2", ngroups, b, m, n, c2, c1, n1, n2, points2, idx, weight, points1, params, out, stream);
}
