// chain.hip -- a whole chain of per-point shared-MLP layers in ONE launch (gfx950).
//
// The tail of the ANCSH graph is ten 1x1 convolutions on the same N points: fa_layer3 (3 layers), fc1, the NOCS
// heads (fc2_*, fc11_1) and the joint heads (fc3_0, fc3_1, fc4_*) -- pointnet_plusplus/architectures.py:84-93,
// lib/architecture.py:105-129,195-206.  As separate launches each is a 20 us kernel that reads and writes a
// (B*N, 128) activation in HBM.  Here a WAVE owns 32 points and walks a small PROGRAM of layers with the activations in
// two private LDS tiles (odd row stride -> conflict-free MFMA fragment reads); weights stream from L2 in the packed
// fragment order; only the head logits leave the chip.  Per layer the arithmetic is the same k-ordered f32 fmaf chain
// (v_mfma_f32_32x32x2_f32) + bias + folded BN + ReLU as ancsh_conv1x1, so results are bit-identical.
// (Until round 2 a workgroup owned 64 rows, each wave a 32-column slice, two barriers per layer and half the waves idle
// through every head layer: 92 / 73 us for the 11- / 8-layer programs against 83 / 68 us now.)
#include "common.h"
#include "wave_mlp.h"

namespace ancsh {

constexpr int CH_MAX_OPS = 12;
constexpr int CH_LD = 133;      // LDS row stride (odd, >= 131 input channels + 1)

struct ChainOp {
    const float *w, *bias, *scale, *shift;
    float *out_g;               // non-null: write (rows, n) to global with row stride out_ld instead of an LDS tile
    int k, n, act, src, dst, out_ld;
};
struct ChainProg {
    int nops;
    ChainOp op[CH_MAX_OPS];
};

// Every WAVE owns 32 rows from the input load to the last head (the fused SA kernels' recipe, wave_mlp.h): both activation
// tiles are private to the wave, a layer is 32 rows x ALL its columns (4 accumulators for the 128-wide layers, 1 for a head
// block of <= 32 columns), written back by the wave itself -- no __syncthreads anywhere, no wave idles through a head layer,
// each weight fragment feeds 4 MFMAs, and the next layer's first weights are requested before the current layer's epilogue.
// 32768 rows = 1024 waves = one per SIMD (136 KB of LDS per workgroup of four).
constexpr int CW_PRE = 2;        // weight slots requested ahead across a layer boundary (= LayerCfg::DW of every layer here)

__device__ __forceinline__ void cw_prefetch(const ChainOp &L, float4 (&pre)[CW_PRE][4]) {
    // unconditional loads (a load under a branch drains vmcnt at the join): a head block's single tile is simply read four times
    const int tn = (L.n + 31) >> 5;
    const float4 *Wp = reinterpret_cast<const float4 *>(L.w) + (threadIdx.x & 63);
#pragma unroll
    for (int s2 = 0; s2 < CW_PRE; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[s2][j] = Wp[(size_t)(s2 * tn + (j < tn ? j : tn - 1)) * 64];
}

template <int K, int N, bool RELU>
__device__ __forceinline__ void cw_layer(const ChainOp &L, const ChainOp &NX, const float *Tsrc, float *Tdst, float4 (&pre)[CW_PRE][4],
                                         long row0, long rows) {
    constexpr int TN = N / 32;
    static_assert(LayerCfg<K, N>::DW == CW_PRE, "prefetch distance");
    SaLayer S;
    S.w = L.w; S.bias = L.bias; S.scale = L.scale; S.shift = L.shift; S.ncol = L.n; S.wstride = 0;
    float4 bw[CW_PRE + 1][TN];
#pragma unroll
    for (int s2 = 0; s2 < CW_PRE; ++s2)
#pragma unroll
        for (int j = 0; j < TN; ++j) bw[s2][j] = pre[s2][j];
    floatx16 acc[1][TN];
    float ep[3][TN], none[TN];
    mfma_loop<K, N, CH_LD, 1>(Tsrc, S, bw, acc, ep);
    cw_prefetch(NX, pre);                       // the next layer's first weights fly under this epilogue
    __builtin_amdgcn_sched_barrier(0);
    if (L.out_g) epilogue_global<N, 1, RELU>(L.out_g, L.out_ld, L.n, row0, rows, acc, ep);
    else epilogue<N, CH_LD, false, 1, RELU>(Tdst, acc, ep, none);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_chain_wave_kernel(long rows, int cin, const float *__restrict__ x, int ldx, ChainProg P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32 * CH_LD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *T = smem + wave * (2 * TILE);                       // this wave's two tiles
    const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (row0 >= rows) return;                                  // no barrier anywhere: a wave may simply leave
    float4 pre[CW_PRE][4];
    cw_prefetch(P.op[0], pre);
    // input rows -> tile 0 (columns >= cin zero: the odd-k tail of the first layer reads column cin)
    if ((ldx & 3) != 0 || (((uintptr_t)x) & 15) != 0) {        // rows not 16-B aligned: single floats
        for (int e = lane; e < 32 * CH_LD; e += 64) {
            const int r = e / CH_LD, c = e - r * CH_LD;
            T[e] = (c < cin && row0 + r < rows) ? x[(size_t)(row0 + r) * ldx + c] : 0.f;
        }
    } else {
        // 33 float4 per row cover columns 0..131 (column 132 is never read): a compile-time trip count, so all 17 loads of a lane
        // are in flight together; float4 slots past the row's last one re-read it (address clamped) and are masked to zero
        const int v4 = (cin + 3) / 4;
#pragma unroll
        for (int e0 = 0; e0 < 32 * 33; e0 += 64) {
            const int e = e0 + lane;
            const int r = e / 33, c4 = e - r * 33;
            const bool in = e < 32 * 33 && row0 + r < rows;
            const long row = row0 + r < rows ? row0 + r : rows - 1;
            const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)row * ldx + (c4 < v4 ? c4 : v4 - 1) * 4);
            if (e < 32 * 33) {
                float *d = T + r * CH_LD + c4 * 4;
                const int c = c4 * 4;
                d[0] = (in && c < cin) ? v.x : 0.f;
                d[1] = (in && c + 1 < cin) ? v.y : 0.f;
                d[2] = (in && c + 2 < cin) ? v.z : 0.f;
                d[3] = (in && c + 3 < cin) ? v.w : 0.f;
            }
        }
    }
    for (int i = 0; i < P.nops; ++i) {
        const ChainOp &L = P.op[i];
        const ChainOp &NX = P.op[i + 1 < P.nops ? i + 1 : i];
        const float *Ts = T + L.src * TILE;
        float *Td = T + (L.out_g ? 0 : L.dst) * TILE;
        const bool relu = L.act == ANCSH_ACT_RELU;
        if (L.k == 131) {
            if (relu) cw_layer<131, 128, true>(L, NX, Ts, Td, pre, row0, rows); else cw_layer<131, 128, false>(L, NX, Ts, Td, pre, row0, rows);
        } else if (L.n == 128) {
            if (relu) cw_layer<128, 128, true>(L, NX, Ts, Td, pre, row0, rows); else cw_layer<128, 128, false>(L, NX, Ts, Td, pre, row0, rows);
        } else {
            if (relu) cw_layer<128, 32, true>(L, NX, Ts, Td, pre, row0, rows); else cw_layer<128, 32, false>(L, NX, Ts, Td, pre, row0, rows);
        }
    }
}

// ---- one tile per wave, two waves per SIMD, several networks per launch (round 4) ----------------------------------------------
// The kernel above gives a wave TWO private tiles (the heads branch off a shared trunk) = 34 KB: four waves fill a CU's LDS, one
// per SIMD, and nothing runs under a wave's epilogues, gathers and weight waits (matrix pipe 0.61 busy, r03 SQ counters).  Here a
// wave owns ONE tile and every layer runs IN PLACE (the k loop has consumed the whole tile before the epilogue writes it -- the
// fused SA kernels' rule), so eight waves fit a CU: two per SIMD, one's epilogue under the other's MFMAs.  The only value with two
// 128-wide consumers is the trunk `net` (fc1's output feeds fc11_1 AND fc3_0): the op that produces it carries CH_SAVE -- its tile
// is also copied to a scratch row block in global memory (16 KB per wave, L2-resident) -- and the op that needs it back carries
// CH_RESTORE (2 x 16 float4 per lane: ~1 % of a wave's 2 x 10^5 cycles).  Several networks (their own programs, weights and row
// blocks of x / scratch) share one launch: blockIdx.y = network.  Same k-ordered f32 chains as everywhere: bit-identical.
constexpr int CH_SAVE = 1, CH_RESTORE = 2;
constexpr int CH1_MAX_GROUPS = 2;            // programs per launch (kernel arguments: 2 x 0.8 KB)
struct Chain1Op {
    const float *w, *bias, *scale, *shift;
    float *out_g;               // non-null: write (rows, n) to global with row stride out_ld; else the layer rewrites the tile in place
    int k, n, act, flags, out_ld;
};
struct Chain1Prog {
    int nops;
    Chain1Op op[CH_MAX_OPS];
};
struct Chain1Groups {
    long x_stride, scratch_stride;          // floats between consecutive groups' first rows of x / scratch
    Chain1Prog prog[CH1_MAX_GROUPS];
};
// Round 5: the chain's input rows BUILT IN THE LOAD instead of read from a concat buffer -- fa_layer3's input
// [three_interpolate(level-1 features) (128) | xyz (3)] (pointnet_util.py:218-229): points2 (groups * b, m, 128) network-major,
// idx / weight (b, n, 3) and xyz (b, n, 3) shared by the networks.  points2 == nullptr: the plain loader (x).
struct ChainFpLoad {
    const float *points2, *weight, *xyz;
    const int *idx;
    int n, m, b;                            // points per cloud (n % 128 == 0), interpolation sources per cloud, clouds per network
};

// rows [row0, row0 + 32) of a row-major global matrix (row stride ld floats, 16-byte aligned rows) -> the tile's first 4 * V4 columns.
// BATCH float4 loads of a lane are in flight together (the input load, with nothing else live, takes all 17; the restore inside the
// layer loop keeps to 4: the accumulator-sized register budget of two waves per SIMD has no room for 64 more).
template <int V4, int BATCH>
__device__ __forceinline__ void tile_load_f4(float *T, const float *__restrict__ g, int ld, int cols, long row0, long rows) {
    const int lane = threadIdx.x & 63;
    const int v4 = (cols + 3) / 4;
    constexpr int NE = (32 * V4 + 63) / 64;
#pragma unroll 1
    for (int b0 = 0; b0 < NE; b0 += BATCH) {
        float4 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 64 + lane;
            const int r = e / V4, c4 = e - r * V4;
            const long row = row0 + r < rows ? row0 + r : rows - 1;
            v[u] = *reinterpret_cast<const float4 *>(g + (size_t)(r < 32 ? row : row0) * ld + (c4 < v4 ? c4 : v4 - 1) * 4);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 64 + lane;
            const int r = e / V4, c4 = e - r * V4;
            const bool in = row0 + r < rows;
            if (b0 + u < NE && e < 32 * V4) {
                float *d = T + r * CH_LD + c4 * 4;
                const int c = c4 * 4;
                d[0] = (in && c < cols) ? v[u].x : 0.f;
                d[1] = (in && c + 1 < cols) ? v[u].y : 0.f;
                d[2] = (in && c + 2 < cols) ? v[u].z : 0.f;
                d[3] = (in && c + 3 < cols) ? v[u].w : 0.f;
            }
        }
    }
}

// rows [row0, row0 + 32) of fa_layer3's input: T[r][0:128] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3 (that order, unfused:
// tf_interpolate.cpp:107-127), T[r][128:131] = xyz, T[r][131] = 0.  A lane owns float4 column c4 = lane & 31 of the rows
// 2 * it + (lane >> 5): sixteen rows per lane, four at a time (12 gathers in flight; the rows are L2-resident under the XCD-aware map).
__device__ __forceinline__ void tile_load_fp3(float *T, const ChainFpLoad &F, int grp, long row0) {
    const int lane = threadIdx.x & 63, c4 = lane & 31, half = lane >> 5;
    const long cloud = row0 / F.n;                             // cloud inside this network (a tile never straddles clouds)
    const long g0 = cloud * F.n + (row0 - cloud * F.n);        // first row in the geometry arrays
    const float4 *p2 = reinterpret_cast<const float4 *>(F.points2) + ((size_t)grp * F.b + cloud) * F.m * 32 + c4;
#pragma unroll 1
    for (int it0 = 0; it0 < 16; it0 += 4) {
        float4 a[4][3];
        float w[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = g0 + 2 * (it0 + u) + half;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                w[u][q] = F.weight[r * 3 + q];
                a[u][q] = p2[(size_t)F.idx[r * 3 + q] * 32];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float *d = T + (2 * (it0 + u) + half) * CH_LD + c4 * 4;
            d[0] = a[u][0].x * w[u][0] + a[u][1].x * w[u][1] + a[u][2].x * w[u][2];
            d[1] = a[u][0].y * w[u][0] + a[u][1].y * w[u][1] + a[u][2].y * w[u][2];
            d[2] = a[u][0].z * w[u][0] + a[u][1].z * w[u][1] + a[u][2].z * w[u][2];
            d[3] = a[u][0].w * w[u][0] + a[u][1].w * w[u][1] + a[u][2].w * w[u][2];
        }
    }
    if (lane < 32) {
        const float *x = F.xyz + (g0 + lane) * 3;
        float *d = T + lane * CH_LD + 128;
        d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; d[3] = 0.f;
    }
}

__device__ __forceinline__ void tile_store_128(const float *T, float *__restrict__ g, long row0, long rows) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int b0 = 0; b0 < 16; b0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (b0 + u) * 64 + lane, r = e >> 5, c4 = e & 31;
            const float *s = T + r * CH_LD + c4 * 4;
            const float4 v = make_float4(s[0], s[1], s[2], s[3]);
            if (row0 + r < rows) *reinterpret_cast<float4 *>(g + (size_t)(row0 + r) * 128 + c4 * 4) = v;
        }
    }
}

// (a separate function on purpose: written inline in c1_layer the same loads cost hipcc 627 spilled registers at two waves per SIMD)
__device__ __forceinline__ void c1_prefetch(const Chain1Op &L, float4 (&pre)[CW_PRE][4]) {
    const int tn = (L.n + 31) >> 5;
    const float4 *Wp = reinterpret_cast<const float4 *>(L.w) + (threadIdx.x & 63);
#pragma unroll
    for (int s2 = 0; s2 < CW_PRE; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[s2][j] = Wp[(size_t)(s2 * tn + (j < tn ? j : tn - 1)) * 64];
}

template <int K, int N, bool RELU>
__device__ __forceinline__ void c1_layer(const Chain1Op &L, const Chain1Op &NX, float *T, float4 (&pre)[CW_PRE][4], long row0, long rows) {
    constexpr int TN = N / 32;
    static_assert(LayerCfg<K, N>::DW == CW_PRE, "prefetch distance");
    SaLayer S;
    S.w = L.w; S.bias = L.bias; S.scale = L.scale; S.shift = L.shift; S.ncol = L.n; S.wstride = 0;
    float4 bw[CW_PRE + 1][TN];
#pragma unroll
    for (int s2 = 0; s2 < CW_PRE; ++s2)
#pragma unroll
        for (int j = 0; j < TN; ++j) bw[s2][j] = pre[s2][j];
    floatx16 acc[1][TN];
    float ep[3][TN], none[TN];
    mfma_loop<K, N, CH_LD, 1>(T, S, bw, acc, ep);
    c1_prefetch(NX, pre);                       // the next layer's first weights fly under this epilogue
    __builtin_amdgcn_sched_barrier(0);
    if (L.out_g) epilogue_global<N, 1, RELU>(L.out_g, L.out_ld, L.n, row0, rows, acc, ep);
    else epilogue<N, CH_LD, false, 1, RELU>(T, acc, ep, none);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_chain1_kernel(long rows, int cin, const float *__restrict__ x, int ldx, float *__restrict__ scratch, Chain1Groups G, ChainFpLoad F) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32 * CH_LD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Chain1Prog &P = G.prog[blockIdx.y];
    if (x) x += (size_t)blockIdx.y * G.x_stride;
    if (scratch) scratch += (size_t)blockIdx.y * G.scratch_stride;
    float *T = smem + wave * TILE;                             // this wave's tile
    long blk = blockIdx.x;
    if (F.points2) {
        // XCD-aware workgroup -> cloud map (as fp_concat_kernel): workgroups go round-robin to the 8 XCDs, a level-1 row is gathered by
        // ~6 points of ITS cloud: XCD x takes the clouds x, x + 8, ... whole, so its L2 holds b / 8 clouds' rows per network
        const long wpc = F.n / 128;                            // workgroups per cloud
        if ((F.b & 7) == 0) {
            const long xcd = blk & 7, j = blk >> 3;
            blk = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
    const long row0 = (blk * 4 + wave) * 32;
    if (row0 >= rows) return;                                  // no barrier anywhere: a wave may simply leave
    float4 pre[CW_PRE][4];
    c1_prefetch(P.op[0], pre);
    // input rows -> the tile (columns >= cin zero: the odd-k tail of the first layer reads column cin)
    if (F.points2) {
        tile_load_fp3(T, F, blockIdx.y, row0);
    } else if ((ldx & 3) != 0 || (((uintptr_t)x) & 15) != 0) {        // rows not 16-B aligned: single floats
        for (int e = lane; e < 32 * CH_LD; e += 64) {
            const int r = e / CH_LD, c = e - r * CH_LD;
            T[e] = (c < cin && row0 + r < rows) ? x[(size_t)(row0 + r) * ldx + c] : 0.f;
        }
    } else {
        tile_load_f4<33, 17>(T, x, ldx, cin, row0, rows);      // 33 float4 per row cover columns 0..131 (column 132 is never read)
    }
    for (int i = 0; i < P.nops; ++i) {
        const Chain1Op &L = P.op[i];
        const Chain1Op &NX = P.op[i + 1 < P.nops ? i + 1 : i];
        if (L.flags & CH_RESTORE) {                            // block-uniform
            wave_lds_fence();                                  // the previous layer's reads of the tile have completed
            tile_load_f4<32, 4>(T, scratch, 128, 128, row0, rows);
        }
        const bool relu = L.act == ANCSH_ACT_RELU;
        if (L.k == 131) {
            if (relu) c1_layer<131, 128, true>(L, NX, T, pre, row0, rows); else c1_layer<131, 128, false>(L, NX, T, pre, row0, rows);
        } else if (L.n == 128) {
            if (relu) c1_layer<128, 128, true>(L, NX, T, pre, row0, rows); else c1_layer<128, 128, false>(L, NX, T, pre, row0, rows);
        } else {
            if (relu) c1_layer<128, 32, true>(L, NX, T, pre, row0, rows); else c1_layer<128, 32, false>(L, NX, T, pre, row0, rows);
        }
        if (L.flags & CH_SAVE) {
            wave_lds_fence();                                  // the epilogue's tile is complete
            tile_store_128(T, scratch, row0, rows);
        }
    }
}

}  // namespace ancsh

using namespace ancsh;

// ops: nops x 6 ints {k, n, act, src, dst (-1 = global), out_ld}; ptrs: nops x 5 device pointers {packed w, bias, scale, shift, out}
extern "C" int ancsh_mlp_chain(long rows, int cin, const float *x, int ldx, int nops, const int *ops,
                               const void *const *ptrs, void *stream) {
    ANCSH_REQUIRE(rows >= 0 && cin > 0 && cin <= 131 && ldx >= cin, "mlp_chain: bad input shape rows=%ld cin=%d ldx=%d", rows, cin, ldx);
    ANCSH_REQUIRE(nops > 0 && nops <= CH_MAX_OPS, "mlp_chain: nops %d outside 1..%d", nops, CH_MAX_OPS);
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE(x && ops && ptrs, "mlp_chain: null pointer");
    ChainProg P;
    P.nops = nops;
    for (int i = 0; i < nops; ++i) {
        ChainOp &o = P.op[i];
        o.k = ops[6 * i]; o.n = ops[6 * i + 1]; o.act = ops[6 * i + 2]; o.src = ops[6 * i + 3]; o.dst = ops[6 * i + 4]; o.out_ld = ops[6 * i + 5];
        o.w = (const float *)ptrs[5 * i]; o.bias = (const float *)ptrs[5 * i + 1]; o.scale = (const float *)ptrs[5 * i + 2];
        o.shift = (const float *)ptrs[5 * i + 3]; o.out_g = (float *)ptrs[5 * i + 4];
        ANCSH_REQUIRE(((o.k == 131 && o.n == 128) || o.k == 128) && (o.n == 128 || (o.n >= 1 && o.n <= 32)), "mlp_chain: op %d has unsupported shape %d -> %d", i, o.k, o.n);
        ANCSH_REQUIRE(o.src >= 0 && o.src < 2 && (o.dst == -1 || (o.dst >= 0 && o.dst < 2)), "mlp_chain: op %d bad tiles", i);
        ANCSH_REQUIRE((o.dst == -1) == (o.out_g != nullptr), "mlp_chain: op %d: global output iff dst == -1", i);
        ANCSH_REQUIRE(o.dst != -1 || o.out_ld >= o.n, "mlp_chain: op %d out_ld < n", i);
        ANCSH_REQUIRE(o.w && o.bias && o.scale && o.shift, "mlp_chain: op %d null parameter", i);
        ANCSH_REQUIRE(o.act == ANCSH_ACT_NONE || o.act == ANCSH_ACT_RELU, "mlp_chain: op %d bad activation", i);
    }
    const size_t lds = sizeof(float) * 4 * 2 * 32 * CH_LD;
    (void)hipFuncSetAttribute((const void *)mlp_chain_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mlp_chain_wave_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), lds, (hipStream_t)stream, rows, cin, x, ldx, P);
    return check_launch("mlp_chain");
}

// The chain with ONE tile per wave (two waves per SIMD) for ngroups <= 2 networks in one launch: group g reads rows
// [g * rows, (g + 1) * rows) of x and runs ITS program.  ops[g]: nops[g] x 5 ints {k, n, act, flags, out_ld}; ptrs[g]: nops[g] x 5 device
// pointers {packed w, bias, scale, shift, out | NULL}.  A layer with out == NULL rewrites the tile in place; flags: 1 = also copy the
// layer's 128-column output to this network's scratch rows, 2 = reload the tile from them before the layer.  scratch: ngroups * rows * 128
// floats (16-byte aligned), NULL when no op carries a flag.
static int chain_grouped_impl(int ngroups, long rows, int cin, const float *x, int ldx, const int *nops, const int *const *ops,
                              const void *const *const *ptrs, float *scratch, const ChainFpLoad &F, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= CH1_MAX_GROUPS, "mlp_chain_grouped: ngroups %d outside 1..%d", ngroups, CH1_MAX_GROUPS);
    ANCSH_REQUIRE(rows >= 0 && cin > 0 && cin <= 131 && (F.points2 || ldx >= cin), "mlp_chain_grouped: bad input shape rows=%ld cin=%d ldx=%d", rows, cin, ldx);
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE((x || F.points2) && nops && ops && ptrs, "mlp_chain_grouped: null pointer");
    ANCSH_REQUIRE((((uintptr_t)scratch) & 15) == 0, "mlp_chain_grouped: scratch must be 16-byte aligned");
    Chain1Groups G;
    G.x_stride = rows * (long)ldx;
    G.scratch_stride = rows * 128L;
    for (int g = 0; g < ngroups; ++g) {
        ANCSH_REQUIRE(nops[g] > 0 && nops[g] <= CH_MAX_OPS && ops[g] && ptrs[g], "mlp_chain_grouped: group %d: nops %d outside 1..%d", g, nops[g], CH_MAX_OPS);
        Chain1Prog &P = G.prog[g];
        P.nops = nops[g];
        bool saved = false;
        for (int i = 0; i < nops[g]; ++i) {
            Chain1Op &o = P.op[i];
            const int *q = ops[g] + 5 * i;
            o.k = q[0]; o.n = q[1]; o.act = q[2]; o.flags = q[3]; o.out_ld = q[4];
            o.w = (const float *)ptrs[g][5 * i]; o.bias = (const float *)ptrs[g][5 * i + 1]; o.scale = (const float *)ptrs[g][5 * i + 2];
            o.shift = (const float *)ptrs[g][5 * i + 3]; o.out_g = (float *)ptrs[g][5 * i + 4];
            ANCSH_REQUIRE(((o.k == 131 && o.n == 128) || o.k == 128) && (o.n == 128 || (o.n >= 1 && o.n <= 32)),
                          "mlp_chain_grouped: group %d op %d has unsupported shape %d -> %d", g, i, o.k, o.n);
            ANCSH_REQUIRE(o.n == 128 || o.out_g, "mlp_chain_grouped: group %d op %d: a head block (n <= 32) writes to global memory", g, i);
            ANCSH_REQUIRE(!o.out_g || o.out_ld >= o.n, "mlp_chain_grouped: group %d op %d out_ld < n", g, i);
            ANCSH_REQUIRE(o.w && o.bias && o.scale && o.shift, "mlp_chain_grouped: group %d op %d null parameter", g, i);
            ANCSH_REQUIRE(o.act == ANCSH_ACT_NONE || o.act == ANCSH_ACT_RELU, "mlp_chain_grouped: group %d op %d bad activation", g, i);
            ANCSH_REQUIRE((o.flags & ~(CH_SAVE | CH_RESTORE)) == 0 && (!(o.flags & CH_SAVE) || (!o.out_g && o.n == 128)),
                          "mlp_chain_grouped: group %d op %d bad flags %d", g, i, o.flags);
            ANCSH_REQUIRE(!(o.flags & CH_RESTORE) || saved, "mlp_chain_grouped: group %d op %d restores before any save", g, i);
            ANCSH_REQUIRE(!(o.flags & (CH_SAVE | CH_RESTORE)) || scratch, "mlp_chain_grouped: group %d op %d needs the scratch buffer", g, i);
            saved = saved || (o.flags & CH_SAVE);
        }
    }
    for (int g = ngroups; g < CH1_MAX_GROUPS; ++g) G.prog[g].nops = 0;
    const size_t lds = sizeof(float) * 4 * 32 * CH_LD;
    (void)hipFuncSetAttribute((const void *)mlp_chain1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mlp_chain1_kernel, dim3((unsigned)((rows + 127) / 128), ngroups), dim3(256), lds, (hipStream_t)stream, rows, cin, x, ldx,
                       scratch, G, F);
    return check_launch("mlp_chain_grouped");
}

extern "C" int ancsh_mlp_chain_grouped(int ngroups, long rows, int cin, const float *x, int ldx, const int *nops, const int *const *ops,
                                       const void *const *const *ptrs, float *scratch, void *stream) {
    ChainFpLoad F{};
    return chain_grouped_impl(ngroups, rows, cin, x, ldx, nops, ops, ptrs, scratch, F, stream);
}

// ancsh_mlp_chain_grouped whose input rows are fa_layer3's [three_interpolate(points2) (128) | xyz (3)] (pointnet_util.py:218-229,
// pointnet_plusplus/architectures.py:84-86) BUILT IN THE TILE LOAD: no (b * n, 132) concat buffer is written or read (34.6 MB each
// way per step at 2 x 32 x 1024 points), and the interpolate + concat launch disappears.  points2 (ngroups * b, m, 128) network-major;
// idx / weight (b, n, 3) from ancsh_three_nn_weights and xyz (b, n, 3) are shared by the networks; n % 128 == 0 (a workgroup's four
// tiles stay inside one cloud).  Same bits as ancsh_fp_interpolate_concat_ex + ancsh_mlp_chain_grouped.
extern "C" int ancsh_mlp_chain_grouped_fp(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx, const float *weight,
                                          const float *xyz, const int *nops, const int *const *ops, const void *const *const *ptrs, float *scratch,
                                          void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0 && n % 128 == 0, "mlp_chain_grouped_fp: bad shape b=%d n=%d (a multiple of 128) m=%d", b, n, m);
    ANCSH_REQUIRE(c2 == 128, "mlp_chain_grouped_fp: the interpolated part must have 128 channels (got %d); use ancsh_fp_interpolate_concat + ancsh_mlp_chain_grouped", c2);
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "mlp_chain_grouped_fp: ngroups=%d must be in [1,%d]", ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(nops && ops && ptrs, "mlp_chain_grouped_fp: null pointer (program table)");      // checked for an empty batch too
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(points2 && idx && weight && xyz, "mlp_chain_grouped_fp: null pointer");
    ANCSH_REQUIRE((((uintptr_t)points2) & 15) == 0, "mlp_chain_grouped_fp: points2 must be 16-byte aligned");
    ChainFpLoad F;
    F.points2 = points2; F.idx = idx; F.weight = weight; F.xyz = xyz; F.n = n; F.m = m; F.b = b;
    return chain_grouped_impl(ngroups, (long)b * n, 131, nullptr, 0, nops, ops, ptrs, scratch, F, stream);
}
