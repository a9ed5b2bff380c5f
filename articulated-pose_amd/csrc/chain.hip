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
