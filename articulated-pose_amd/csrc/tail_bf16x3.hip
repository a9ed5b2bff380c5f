// tail_bf16x3.hip -- EXPERIMENT (opt-in, never the default): the per-point tail of the network (fa_layer3's three convs, fc1 and every
// head: pointnet_plusplus/architectures.py:84-93, lib/architecture.py:105-129,195-206; chain.hip is the graded f32 form) with every f32
// product emulated by six bf16 MFMA products (bx3.h; arithmetic and error statement in sa_bf16x3.hip).
//
// The fused SA levels' recipe (sa_bf16x3.hip): a WAVE owns 64 points (two blocks of 32) from the input load to the last head, and the
// 128-channel activations of both blocks stay in its REGISTERS as bf16x3 MFMA fragments (two register tiles X / Y of 192 VGPRs each; a
// hidden layer reads one and writes the other, computed transposed so that its accumulators are the next layer's fragments after
// bias / BN / ReLU, the split and one v_permlane32_swap per register pair).  No LDS activation tile, no barrier, and -- two register tiles
// instead of chain.hip's one LDS tile -- no save / restore of the trunk: fc11_1 reads X and writes Y (X = fc1's output stays), the NOCS
// head reads Y, fc3_0 reads X again.  Every weight fragment feeds both point blocks (the weight stream from L2 bounds this kernel
// otherwise); a head block (<= 32 outputs) is one transposed tile whose accumulators go straight to the logits.
//
// The input rows [three_interpolate(level-1 features) (128) | xyz (3)] (pointnet_util.py:218-229) are built in the load like
// chain.hip's tile_load_fp3 (coalesced 512-byte row gathers, p[i1] * w1 + p[i2] * w2 + p[i3] * w3 in that order) into a wave-private
// f32 LDS tile, from which the first layer -- the only one that does not start from registers -- streams its fragments k-block by
// k-block with all four output tiles' accumulators live.
#include "bx3.h"

namespace ancsh {

constexpr int TB_MAX_OPS = 12;
constexpr int TB_LD = 132;                  // f32 LDS input tile: row stride in floats (16-byte aligned rows: 128 interpolated + xyz + 0)
constexpr int TB_MAX_GROUPS = 2;

struct TailBxOp {
    Bx3Layer L;
    float *out_g;                           // head block: logits + column offset (row stride out_ld); nullptr for a hidden layer
    int k, n, act, out_ld;
};
// A program has a FIXED shape, so that the kernel is straight-line code and the compiler sees when a register tile dies (a runtime
// interpreter over src / dst tile numbers kept both tiles live everywhere: 732 spilled registers):
//   op 0                 131 -> 128 ReLU          input -> X
//   ops 1, 2, 3          128 -> 128               X -> Y -> X -> Y            (Y = the trunk `net`, fc1's output)
//   nh1 head blocks      128 -> n <= 32           from Y
//   split == 1:  one LINEAR hidden op Y -> X (fc11_1; the trunk stays in Y), then nh2 head blocks from X
//   two hidden ops       Y -> X -> Y              (fc3_0, fc3_1)
//   nh3 head blocks      from Y
struct TailBxProg {
    int nops, split, nh1, nh2, nh3;
    TailBxOp op[TB_MAX_OPS];
};
struct TailBxGroups {
    TailBxProg prog[TB_MAX_GROUPS];
};
struct TailBxLoad {
    const float *points2, *weight, *xyz;
    const int *idx;
    int n, m, b;                            // points per cloud (n % 64 == 0), interpolation sources per cloud, clouds per network
};

// 64 rows of fa_layer3's input -> the wave's f32 tile T[64][TB_LD]: a lane owns float4 column c4 = lane & 31 of the rows 2 * it + (lane >> 5)
__device__ __forceinline__ void tb_load_input(float *T, const TailBxLoad &F, int grp, long row0) {
    const int lane = threadIdx.x & 63, c4 = lane & 31, half = lane >> 5;
    const long cloud = row0 / F.n;                             // cloud inside this network (64 rows never straddle clouds)
    const long g0 = row0;                                      // first row in the geometry arrays (b clouds of n rows, shared by the networks)
    const float4 *p2 = reinterpret_cast<const float4 *>(F.points2) + ((size_t)grp * F.b + cloud) * F.m * 32 + c4;
#pragma unroll 1
    for (int it0 = 0; it0 < 32; it0 += 4) {
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
            float4 v;
            v.x = a[u][0].x * w[u][0] + a[u][1].x * w[u][1] + a[u][2].x * w[u][2];
            v.y = a[u][0].y * w[u][0] + a[u][1].y * w[u][1] + a[u][2].y * w[u][2];
            v.z = a[u][0].z * w[u][0] + a[u][1].z * w[u][1] + a[u][2].z * w[u][2];
            v.w = a[u][0].w * w[u][0] + a[u][1].w * w[u][1] + a[u][2].w * w[u][2];
            *reinterpret_cast<float4 *>(T + (2 * (it0 + u) + half) * TB_LD + c4 * 4) = v;
        }
    }
    {
        const float *x = F.xyz + (g0 + lane) * 3;
        *reinterpret_cast<float4 *>(T + lane * TB_LD + 128) = make_float4(x[0], x[1], x[2], 0.f);
    }
}

// first layer: K = 131 (9 k-blocks: 8 of interpolated channels from the LDS tile, 1 holding xyz), N = 128, both point blocks; k-block
// outer so that the input fragments are formed once and dropped: acc[P][4] (128 VGPRs) live, Y written at the end.
template <class S, int P, bool RELU>
__device__ __forceinline__ void tb_first(const Bx3Layer &L, const float *T, BxFrag (&Y)[P][8][S::NP]) {
    constexpr int TM = 4, KB = 9;
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const uint4 *Wp = L.w + lane;
    fx16 acc[TM][P][S::NACC];
#pragma unroll
    for (int i = 0; i < TM; ++i) bx3_zero<S, P>(acc[i]);
    uint4 w[2][TM][S::NP];
    EpiRaw raw;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl) w[0][i][pl] = Wp[(size_t)(i * S::NP + pl) * 64];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < S::NP; ++pl) w[(kb + 1) & 1][i][pl] = Wp[(size_t)(((kb + 1) * TM + i) * S::NP + pl) * 64];
        }
        BxFrag X[P][S::NP];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float *src = T + (32 * p + l31) * TB_LD + 16 * kb + 8 * khalf;
            float4 v0, v1;
            if (kb < 8) {
                v0 = *reinterpret_cast<const float4 *>(src);
                v1 = *reinterpret_cast<const float4 *>(src + 4);
            } else {                          // channels 128..143: xyz in the lower lanes' elements 0..2, zeros elsewhere
                const float4 x = *reinterpret_cast<const float4 *>(T + (32 * p + l31) * TB_LD + 128);
                v0 = khalf ? make_float4(0.f, 0.f, 0.f, 0.f) : x;
                v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            u32 s0[S::NP], s1[S::NP], s2[S::NP], s3[S::NP];
            S::split2(v0.x, v0.y, s0);
            S::split2(v0.z, v0.w, s1);
            S::split2(v1.x, v1.y, s2);
            S::split2(v1.z, v1.w, s3);
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) X[p][pl] = BxFrag{{s0[pl], s1[pl], s2[pl], s3[pl]}};
        }
        if (kb == KB - 1) raw = bx3_epi_load(L, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[i][p][S::PC[t]] = S::mfma(w[kb & 1][i][S::PW[t]], bx_u4<S>(X[p][S::PA[t]]), acc[i][p][S::PC[t]]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) bx3_tile_epilogue<S, P, RELU, 8, TM>(L, i, raw, acc[i], Y);
}

// a head block: n <= 32 outputs (weights / bias / scale / shift padded to 32 columns by the host) in the NORMAL orientation (activations
// as the A operand, weights as B -- the same fragment registers either way): register r of lane (khalf, l31) holds output column l31 of
// point (r & 3) + 8 (r >> 2) + 4 khalf, so that a store instruction writes 32 consecutive columns of a logits row (the transposed
// orientation scattered single floats over 64 rows per instruction: 8192 partial-line writes per wave)
template <class S, int P>
__device__ __forceinline__ void tb_head(const TailBxOp &O, const BxFrag (&X)[P][8][S::NP], long row0) {
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const uint4 *Wp = O.L.w + lane;
    fx16 acc[P][S::NACC];
    bx3_zero<S, P>(acc);
    uint4 w[8][S::NP];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl) w[kb][pl] = Wp[(size_t)(kb * S::NP + pl) * 64];           // all loads in flight: one L2 latency per head
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int t = 0; t < S::NPROD; ++t)              // (activation plane PW[t], weight plane PA[t]): the same products, roles swapped
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc[p][S::PC[t]] = S::mfma(bx_u4<S>(X[p][kb][S::PW[t]]), w[kb][S::PA[t]], acc[p][S::PC[t]]);
    }
    const bool relu = O.act == ANCSH_ACT_RELU;
    const float sc = O.L.scale[l31], shf = __builtin_fmaf(O.L.bias[l31], sc, O.L.shift[l31]);      // bias folded into the shift
    const f32x2v sc2 = {sc, sc}, scl2 = {sc * (1.f / 2048.f), sc * (1.f / 2048.f)}, shf2 = {shf, shf};
    if (l31 < O.n) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2v v = S::bn2(acc[p], r, sc2, scl2, shf2);
                float *o = O.out_g + (size_t)(row0 + 32 * p + (r & 3) + 8 * (r >> 2) + 4 * khalf) * O.out_ld + l31;      // r even: rows row, row + 1
                o[0] = relu ? nmax(v.x, 0.f) : v.x;
                o[O.out_ld] = relu ? nmax(v.y, 0.f) : v.y;
            }
    }
}

// KEEP: the input tile is read again later (fc11_1 leaves the trunk in place): output-tile-outer order, which never holds more than X + Y
template <class S, int P, bool RELU, bool KEEP = false>
__device__ __forceinline__ void tb_hidden(const TailBxOp &O, const BxFrag (&X)[P][8][S::NP], BxFrag (&Y)[P][8][S::NP]) {
    if (KEEP) {
        const float *const none[P] = {nullptr, nullptr};
        bx3_hidden<S, 8, 128, P, RELU>(O.L, X, Y, none);
    } else {
        bx3_hidden_kouter<S, 8, 128, P, RELU>(O.L, X, Y);
    }
}

template <class S>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tail_bx3_kernel(long rows, TailBxGroups G, TailBxLoad F) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const TailBxProg &Pg = G.prog[blockIdx.y];
    float *T = smem + wave * (64 * TB_LD);
    long blk = blockIdx.x;
    {
        // XCD-aware workgroup -> cloud map (chain.hip): XCD x takes the clouds x, x + 8, ... whole, so its L2 holds their level-1 rows
        const long wpc = F.n / 256;                            // workgroups per cloud (4 waves x 64 rows)
        if (wpc > 0 && (F.b & 7) == 0 && F.n % 256 == 0) {
            const long xcd = blk & 7, j = blk >> 3;
            blk = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
    const long row0 = (blk * 4 + wave) * 64;
    if (row0 >= rows) return;                                  // no barrier anywhere: a wave may simply leave
#if !defined(BX3_KO_INPUT)     /* (timing experiment only: no input gather) */
    tb_load_input(T, F, blockIdx.y, row0);
#endif
    bx3_fence();
    BxFrag X[P][8][S::NP], Y[P][8][S::NP];
    tb_first<S, P, true>(Pg.op[0].L, T, X);
    tb_hidden<S, P, true>(Pg.op[1], X, Y);
    tb_hidden<S, P, true>(Pg.op[2], Y, X);
    tb_hidden<S, P, true>(Pg.op[3], X, Y);                 // Y = the trunk
    int i = 4;
    for (int h = 0; h < Pg.nh1; ++h, ++i) tb_head<S, P>(Pg.op[i], Y, row0);
    if (Pg.split) {                                        // block-uniform
        tb_hidden<S, P, false, true>(Pg.op[i], Y, X);      // fc11_1 (no activation: lib/architecture.py:111); the trunk stays in Y
        ++i;
        for (int h = 0; h < Pg.nh2; ++h, ++i) tb_head<S, P>(Pg.op[i], X, row0);
    }
    tb_hidden<S, P, true>(Pg.op[i], Y, X);
    tb_hidden<S, P, true>(Pg.op[i + 1], X, Y);
    i += 2;
    for (int h = 0; h < Pg.nh3; ++h, ++i) tb_head<S, P>(Pg.op[i], Y, row0);
}

}  // namespace ancsh

using namespace ancsh;

// The tail programs of ngroups <= 2 networks on the same b clouds of n points (n % 64 == 0), input rows built in the load from
// points2 (ngroups * b, m, 128) network-major, idx / weight (b, n, 3) (ancsh_three_nn_weights) and xyz (b, n, 3).
// ops[g]: nops[g] x 5 ints {k, n, act, 0, out_ld} (ancsh_mlp_chain_grouped's layout; the flags entry is ignored: two register tiles need no
// save / restore); ptrs[g]: nops[g] x 5 device pointers {bf16x3-packed w, bias, scale, shift, out | NULL}, all 16-byte aligned.
// A head block (out != NULL) is 128 -> n <= 32 with w packed as (128, 32) and bias / scale / shift holding 32 entries (padding: any finite
// value).  The op list must have the shape  F H H H head+ [L head+] H H head+  (F = 131 -> 128 ReLU, H = 128 -> 128 ReLU, L = 128 -> 128 without
// activation): lib/architecture.py's
// tail with and without early_split_nocs.
template <class S>
static int tail_split16(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx, const float *weight, const float *xyz,
                        const int *nops, const int *const *ops, const void *const *const *ptrs, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= TB_MAX_GROUPS, "mlp_chain_grouped_fp_bf16x3: ngroups %d outside 1..%d", ngroups, TB_MAX_GROUPS);
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0 && n % 64 == 0, "mlp_chain_grouped_fp_bf16x3: bad shape b=%d n=%d (a multiple of 64) m=%d", b, n, m);
    ANCSH_REQUIRE(c2 == 128, "mlp_chain_grouped_fp_bf16x3: the interpolated part must have 128 channels (got %d)", c2);
    ANCSH_REQUIRE(nops && ops && ptrs, "mlp_chain_grouped_fp_bf16x3: null pointer (program table)");
    TailBxGroups G;
    for (int g = 0; g < ngroups; ++g) {
        ANCSH_REQUIRE(nops[g] > 0 && nops[g] <= TB_MAX_OPS && ops[g] && ptrs[g], "mlp_chain_grouped_fp_bf16x3: group %d: nops %d outside 1..%d", g, nops[g], TB_MAX_OPS);
        TailBxProg &P = G.prog[g];
        P.nops = nops[g];
        char shape[TB_MAX_OPS + 1];
        for (int i = 0; i < nops[g]; ++i) {
            TailBxOp &o = P.op[i];
            const int *q = ops[g] + 5 * i;
            o.k = q[0]; o.n = q[1]; o.act = q[2]; o.out_ld = q[4];
            o.L.w = (const uint4 *)ptrs[g][5 * i]; o.L.bias = (const float *)ptrs[g][5 * i + 1]; o.L.scale = (const float *)ptrs[g][5 * i + 2];
            o.L.shift = (const float *)ptrs[g][5 * i + 3]; o.out_g = (float *)ptrs[g][5 * i + 4];
            ANCSH_REQUIRE(o.L.w && o.L.bias && o.L.scale && o.L.shift, "mlp_chain_grouped_fp_bf16x3: group %d op %d null parameter", g, i);
            ANCSH_REQUIRE(((((uintptr_t)o.L.w) | (uintptr_t)o.L.bias | (uintptr_t)o.L.scale | (uintptr_t)o.L.shift) & 15) == 0,
                          "mlp_chain_grouped_fp_bf16x3: group %d op %d: parameters must be 16-byte aligned", g, i);
            ANCSH_REQUIRE(o.act == ANCSH_ACT_NONE || o.act == ANCSH_ACT_RELU, "mlp_chain_grouped_fp_bf16x3: group %d op %d bad activation", g, i);
            if (i == 0) {
                ANCSH_REQUIRE(o.k == 131 && o.n == 128 && !o.out_g && o.act == ANCSH_ACT_RELU,
                              "mlp_chain_grouped_fp_bf16x3: group %d: op 0 must be the 131 -> 128 ReLU layer", g);
                shape[i] = 'F';
            } else if (o.out_g) {
                ANCSH_REQUIRE(o.k == 128 && o.n >= 1 && o.n <= 32 && o.out_ld >= o.n,
                              "mlp_chain_grouped_fp_bf16x3: group %d op %d: a head block is 128 -> n <= 32 (got %d -> %d, ld %d)", g, i, o.k, o.n, o.out_ld);
                shape[i] = 'h';
            } else {
                ANCSH_REQUIRE(o.k == 128 && o.n == 128, "mlp_chain_grouped_fp_bf16x3: group %d op %d: a hidden layer is 128 -> 128 (got %d -> %d)", g, i, o.k, o.n);
                shape[i] = o.act == ANCSH_ACT_RELU ? 'H' : 'L';       // L: linear (no activation)
            }
        }
        shape[nops[g]] = 0;
        // F H H H h+ [H h+] H H h+
        int i = 0;
        const auto run = [&](char c) { int k = 0; while (shape[i] == c) { ++i; ++k; } return k; };
        bool ok = run('F') == 1 && run('H') == 3;
        P.nh1 = run('h');
        ok = ok && P.nh1 >= 1;
        P.split = 0; P.nh2 = 0;
        if (ok && run('L') == 1) {
            P.split = 1;
            P.nh2 = run('h');
            ok = P.nh2 >= 1;
        }
        ok = ok && run('H') == 2;
        P.nh3 = run('h');
        ok = ok && P.nh3 >= 1 && shape[i] == 0;
        ANCSH_REQUIRE(ok, "mlp_chain_grouped_fp_bf16x3: group %d: op list %s is not F H H H h+ [L h+] H H h+ (F first layer, H hidden ReLU, L hidden linear, h head block)", g, shape);
    }
    for (int g = ngroups; g < TB_MAX_GROUPS; ++g) G.prog[g].nops = 0;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(points2 && idx && weight && xyz, "mlp_chain_grouped_fp_bf16x3: null pointer");
    ANCSH_REQUIRE((((uintptr_t)points2) & 15) == 0, "mlp_chain_grouped_fp_bf16x3: points2 must be 16-byte aligned");
    TailBxLoad F;
    F.points2 = points2; F.idx = idx; F.weight = weight; F.xyz = xyz; F.n = n; F.m = m; F.b = b;
    const long rows = (long)b * n;
    const size_t lds = sizeof(float) * 4 * 64 * TB_LD;
    (void)hipFuncSetAttribute((const void *)tail_bx3_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tail_bx3_kernel<S>, dim3((unsigned)((rows + 255) / 256), ngroups), dim3(256), lds, (hipStream_t)stream, rows, G, F);
    return check_launch("mlp_chain_grouped_fp_bf16x3");
}

extern "C" int ancsh_mlp_chain_grouped_fp_bf16x3(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx,
                                                 const float *weight, const float *xyz, const int *nops, const int *const *ops,
                                                 const void *const *const *ptrs, void *stream) {
    return tail_split16<Bf16x3>(ngroups, b, n, m, c2, points2, idx, weight, xyz, nops, ops, ptrs, stream);
}
// ... the same with the F16x2 scheme (bx3.h): weights packed by ancsh_sa_pack_weights_f16x2
extern "C" int ancsh_mlp_chain_grouped_fp_f16x2(int ngroups, int b, int n, int m, int c2, const float *points2, const int *idx,
                                                const float *weight, const float *xyz, const int *nops, const int *const *ops,
                                                const void *const *const *ptrs, void *stream) {
    return tail_split16<F16x2>(ngroups, b, n, m, c2, points2, idx, weight, xyz, nops, ops, ptrs, stream);
}
