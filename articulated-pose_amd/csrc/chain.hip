// chain.hip -- a whole chain of per-point shared-MLP layers in ONE launch (gfx950).
//
// The tail of the ANCSH graph is ten 1x1 convolutions on the same N points: fa_layer3 (3 layers), fc1, the NOCS
// heads (fc2_*, fc11_1) and the joint heads (fc3_0, fc3_1, fc4_*) -- pointnet_plusplus/architectures.py:84-93,
// lib/architecture.py:105-129,195-206.  As separate launches each is a 20 us kernel that reads and writes a
// (B*N, 128) activation in HBM.  Here a workgroup owns ROWS points and walks a small PROGRAM of layers with the
// activations in two LDS tiles (odd row stride -> conflict-free MFMA fragment reads); weights stream from L2 in
// k-chunks; only the head logits leave the chip.  Per layer the arithmetic is the same k-ordered f32 fmaf chain
// (v_mfma_f32_32x32x2_f32) + bias + folded BN + ReLU as ancsh_conv1x1, so results are bit-identical.
#include "common.h"

namespace ancsh {

typedef float floatx16 __attribute__((ext_vector_type(16)));

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

// One layer with COMPILE-TIME k (128 or 131) so that the whole k loop unrolls into straight-line code: the
// accumulators then stay in registers from the first MFMA to the epilogue and the compiler can place counted vmcnt waits
// (a run-time trip count made it copy accumulators in and out of the MFMA block and drain every prefetch).
// WIDE (n == 128): wave w owns columns 32w..32w+31 of all ROWS rows (ROWS/32 accumulators).
// Narrow heads (n <= 32): waves 0..ROWS/32-1 own one 32x32 tile each; the other waves idle through the layer.
// Weights never touch LDS and arrive PRE-PACKED (ancsh_sa_pack_weights: one 16-byte load per lane = the B fragments of four
// k-steps of one 32-column tile, zero beyond row k-1 and column n-1): a wave streams its own tile two slots ahead of the
// MFMAs, activations one k-step pair ahead; sched_barrier pins that order (left alone the scheduler sinks every load to its
// use and each MFMA pair then waits out an L2 round trip: 55 load -> vmcnt(0) -> MFMA sequences in the previous version).
__device__ __forceinline__ float ch_f4(const float4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

template <int ROWS, int K, bool WIDE>
__device__ __forceinline__ void chain_layer(const ChainOp &L, float *smem, int offA, int offO, long row0, long rows) {
    // tiles are addressed as smem + integer offset (never through a selected pointer) so that every access
    // stays an LDS (ds_*) instruction; a pointer array indexed at run time degrades to FLAT loads
    constexpr int RT = WIDE ? ROWS / 32 : 1;
    constexpr int NK = (K + 1) / 2;            // MFMA k-steps
    constexpr int NS = (NK + 3) / 4;           // packed weight slots
    constexpr int DW = 2, DA = 2;              // prefetch distances: weight slots / activation k-steps
    const float *A = smem + offA;
    float *O = smem + offO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const bool active = WIDE || wave < ROWS / 32;
    const int rt0 = WIDE ? 0 : wave;
    const int tile = WIDE ? wave : 0;
    const int col = tile * 32 + l31;
    const int N = L.n;
    const bool cok = col < N;
    const int tn_all = (N + 31) / 32;
    floatx16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const float4 *Wp = reinterpret_cast<const float4 *>(L.w) + (size_t)tile * 64 + lane;
    auto wload = [&](float4 &b, int slot) {       // slot: compile-time after unrolling
        if (slot < NS) b = Wp[(size_t)slot * tn_all * 64];
    };
    const float *Af = A + (size_t)(rt0 * 32 + l31) * CH_LD + khalf;
    float4 bw[DW + 1];
    float aw[DA + 1][RT];
    float bs = 0.f, sc = 0.f, sh = 0.f;
    if (active) {
#pragma unroll
        for (int s2 = 0; s2 < DW; ++s2) wload(bw[s2], s2);
        bs = L.bias[cok ? col : 0]; sc = L.scale[cok ? col : 0]; sh = L.shift[cok ? col : 0];     // land under the k loop
    }
    __syncthreads();                 // the source tile (input load or previous layer's epilogue) is complete
    if (active) {
#pragma unroll
        for (int s2 = 0; s2 < DA; ++s2)
#pragma unroll
            for (int i = 0; i < RT; ++i) aw[s2][i] = Af[(size_t)i * 32 * CH_LD + 2 * s2];
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) {
            const int slot = s2 >> 2, q = s2 & 3;
#pragma unroll
            for (int i = 0; i < RT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s2 % (DA + 1)][i], ch_f4(bw[slot % (DW + 1)], q), acc[i], 0, 0, 0);
            if (q == 0) wload(bw[(slot + DW) % (DW + 1)], slot + DW);
            if (s2 + DA < NK) {
#pragma unroll
                for (int i = 0; i < RT; ++i) aw[(s2 + DA) % (DA + 1)][i] = Af[(size_t)i * 32 * CH_LD + 2 * (s2 + DA)];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!L.out_g && offO == offA) __syncthreads();      // in-place layer: every wave has finished reading the tile
    if (!active) return;
    const bool relu = L.act == ANCSH_ACT_RELU;
    float *og = L.out_g;
    const int old = L.out_ld;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (rt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            float v = __builtin_fmaf(acc[i][r] + bs, sc, sh);
            v = relu ? fmaxf(v, 0.f) : v;
            if (og) {
                if (cok && row0 + row < rows) og[(size_t)(row0 + row) * old + col] = v;
            } else {
                O[(size_t)row * CH_LD + col] = v;
            }
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(256) void mlp_chain_kernel(long rows, int cin, const float *__restrict__ x, int ldx,
                                                        ChainProg P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE0 = 0, TILE = ROWS * CH_LD;             // smem = [tile 0 | tile 1]
    float *buf0 = smem + TILE0;
    const long row0 = (long)blockIdx.x * ROWS;
    // input tile -> tile 0 (columns >= cin zero: the odd-k tail of the first layer reads column cin)
    if ((ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int v4 = (cin + 3) / 4;                      // float4 per row (row stride ldx is 16-B aligned)
        for (int e = threadIdx.x; e < ROWS * v4; e += 256) {
            const int r = e / v4, c4 = e - r * v4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < rows) v = *reinterpret_cast<const float4 *>(x + (size_t)(row0 + r) * ldx + c4 * 4);
            float *d = buf0 + (size_t)r * CH_LD + c4 * 4;
            const int c = c4 * 4;
            d[0] = c < cin ? v.x : 0.f;
            if (c + 1 < CH_LD) d[1] = c + 1 < cin ? v.y : 0.f;
            if (c + 2 < CH_LD) d[2] = c + 2 < cin ? v.z : 0.f;
            if (c + 3 < CH_LD) d[3] = c + 3 < cin ? v.w : 0.f;
        }
        for (int e = threadIdx.x; e < ROWS * (CH_LD - v4 * 4); e += 256) {     // remaining columns of each row
            const int w = CH_LD - v4 * 4, r = e / w, c = v4 * 4 + e - r * w;
            buf0[(size_t)r * CH_LD + c] = 0.f;
        }
    } else {
        for (int e = threadIdx.x; e < ROWS * CH_LD; e += 256) {
            const int r = e / CH_LD, c = e - r * CH_LD;
            float v = 0.f;
            if (c < cin && row0 + r < rows) v = x[(size_t)(row0 + r) * ldx + c];
            buf0[e] = v;
        }
    }
    for (int i = 0; i < P.nops; ++i) {
        const ChainOp &L = P.op[i];
        const int oa = TILE0 + L.src * TILE, oo = TILE0 + (L.out_g ? 0 : L.dst) * TILE;
        if (L.k == 131) chain_layer<ROWS, 131, true>(L, smem, oa, oo, row0, rows);
        else if (L.n == 128) chain_layer<ROWS, 128, true>(L, smem, oa, oo, row0, rows);
        else chain_layer<ROWS, 128, false>(L, smem, oa, oo, row0, rows);
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
#ifndef CH_ROWS
#define CH_ROWS 64
#endif
    constexpr int ROWS = CH_ROWS;
    const size_t lds = sizeof(float) * (2 * ROWS * CH_LD + 16);
    auto k = mlp_chain_kernel<ROWS>;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((rows + ROWS - 1) / ROWS)), dim3(256), lds, (hipStream_t)stream, rows, cin, x, ldx, P);
    return check_launch("mlp_chain");
}
