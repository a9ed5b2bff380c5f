// sa_fused.hip -- one-launch PointNet++ set-abstraction body for gfx950:
//     group_point(xyz) - new_xyz | group_point(features)  ->  3 x (1x1 conv + bias + BN + ReLU)  ->  max over nsample
// (pointnet_plusplus/utils/pointnet_util.py:47-57 (grouping/concat) + :113-134 (MLP + reduce_max)).
//
// The reference materialises the (B, npoint, 64, 3+C) grouped tensor and three (B, npoint, 64, C_i) activations in
// HBM (4.2 MB + 12.6 MB per cloud for SA2) between five TF ops; here a workgroup keeps its 128 rows (= two
// 64-sample neighbourhoods) in LDS from the gather to the max:
//   * gather: 16-B feature loads from the L2-resident (512 x 128) level-1 features, centred xyz, into an LDS tile
//     with ODD row stride (conflict-free ds_read_b32 MFMA fragments: lane -> [row = lane&31][k = lane>>5]);
//   * each layer: v_mfma_f32_32x32x2_f32 over the LDS activation tile x weights that go from L2 STRAIGHT INTO
//     REGISTERS (an MFMA B fragment is one weight per lane; each wave prefetches the next 16 k-rows of its own
//     column slice while the current ones feed the matrix pipe) -- no weight staging, no barrier inside a layer;
//     a wave owns 64 rows x 32..128 columns, i.e. 2..8 independent accumulators;
//   * epilogue in registers: bias, folded BN (one fmaf), ReLU, write the next layer's LDS tile; the last layer
//     instead takes the max over its 64 rows (2 row tiles x 16 regs x 2 lane halves) and stores (npoint, C3).
// Arithmetic is the same k-ordered f32 fmaf chain as ancsh_conv1x1 / the CPU oracle => bit-identical outputs.
// HBM traffic per neighbourhood: 64 idx + the final C3 floats (the gathered features come from L2).
#include "common.h"

namespace ancsh {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct SaLayer {
    const float *w, *bias, *scale, *shift;
};


// One MLP layer over the workgroup's LDS tile.  A: [SA_ROWS][LDA] (row-major, odd LDA), W global [K][N].
// POOL = false: out_lds[SA_ROWS][LDO] = relu(bn(A.W + b));  POOL = true: out_g[group][N] = max over each wave's 64 rows.
// The B (weight) fragments never touch LDS: lane (k = lane>>5, col = lane&31) of an MFMA needs exactly W[k][col], so
// each wave loads its own 32-column slices straight from L2 into registers, one SA_KC-row chunk ahead of the MFMAs
// that consume them (two register buffers, ping-pong).  No weight staging, no barriers inside a layer.
template <int SA_ROWS, int SA_KC, int K, int N, int LDA, int LDO, bool POOL>
__device__ __forceinline__ void sa_layer(const float *__restrict__ A, const SaLayer L, float *__restrict__ out_lds,
                                         float *__restrict__ out_g, long group0) {
    constexpr int NWR = SA_ROWS / 64;          // wave rows: each wave owns 64 rows (one neighbourhood)
    constexpr int NWC = 4 / NWR;               // wave columns
    constexpr int WCOLS = N / NWC;             // columns per wave
    static_assert(WCOLS % 32 == 0, "a wave needs at least one 32-column MFMA tile");
    constexpr int TN = WCOLS / 32;             // column tiles per wave
    constexpr int NCH = (K + SA_KC - 1) / SA_KC;
    constexpr int KS = SA_KC / 2;              // MFMA k-steps per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave / NWC, ch = wave % NWC;
    const int khalf = lane >> 5, l31 = lane & 31;

    floatx16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float *Wl = L.w + (size_t)khalf * N + ch * WCOLS + l31;
    auto wload = [&](float (&b)[KS][TN], int c) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = c * SA_KC + 2 * s + khalf;
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s][j] = (c < NCH && k < K) ? Wl[(size_t)(c * SA_KC + 2 * s) * N + j * 32] : 0.f;
        }
    };
    const float *Af = A + (size_t)(rh * 64 + l31) * LDA + khalf;
    auto compute = [&](const float (&b)[KS][TN], int c) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (c * SA_KC + 2 * s < K) {             // compile-time after unrolling when c is a constant, else uniform
                const float a0 = Af[c * SA_KC + 2 * s], a1 = Af[c * SA_KC + 2 * s + 32 * LDA];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[s][j], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[s][j], acc[1][j], 0, 0, 0);
                }
            }
        }
    };
    float b0[KS][TN], b1[KS][TN];
    wload(b0, 0);
    __syncthreads();                 // the A tile (gather or previous layer's epilogue) is complete
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
        wload(b1, c + 1);
        compute(b0, c);
        if (c + 1 < NCH) {
            wload(b0, c + 2);
            compute(b1, c + 1);
        }
    }
    // ---- epilogue ----------------------------------------------------------------------------------
    if (!POOL && out_lds == nullptr) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = ch * WCOLS + j * 32 + l31;
        const float bs = L.bias[col], sc = L.scale[col], sh = L.shift[col];
        float pmax = 0.f;    // post-ReLU values are >= 0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaxf(__builtin_fmaf(acc[i][j][r] + bs, sc, sh), 0.f);
                if (POOL) {
                    pmax = fmaxf(pmax, v);
                } else {
                    const int row = rh * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    out_lds[(size_t)row * LDO + col] = v;
                }
            }
        if (POOL) {
            pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
            if (khalf == 0) out_g[(size_t)(group0 + rh) * N + col] = pmax;
        }
    }
}

template <int SA_ROWS, int SA_KC, int CF, int C1, int C2, int C3>
__global__ __launch_bounds__(256) void sa_fused_kernel(int n, int m, long groups, const float *__restrict__ xyz,
                                                       const float *__restrict__ feats, const float *__restrict__ new_xyz,
                                                       const int *__restrict__ idx, SaLayer L1, SaLayer L2, SaLayer L3,
                                                       float *__restrict__ out) {
    constexpr int CIN = 3 + CF;
    constexpr int LDX = (CIN & 1) ? CIN : CIN + 1;
    constexpr int LD1 = C1 + 1, LD2 = C2 + 1;
    constexpr int XSZ = SA_ROWS * (LDX > LD2 ? LDX : LD2) + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufX = smem;                                   // gathered input, later layer-2 output
    float *buf1 = bufX + XSZ;                             // layer-1 output

    const int tid = threadIdx.x;
    const long group0 = (long)blockIdx.x * (SA_ROWS / 64);  // one or two 64-sample neighbourhoods per workgroup
    // ---- gather: X[r][0:3] = xyz[idx] - new_xyz ; X[r][3:3+CF] = feats[idx] ---------------------------
    if (tid < SA_ROWS) {
        const long g = group0 + (tid >> 6);
        if (g < groups) {
            const long b = g / m;
            const int ii = idx[g * 64 + (tid & 63)];
            const float *p = xyz + ((size_t)b * n + ii) * 3;
            const float *c = new_xyz + (size_t)g * 3;
            float *x = bufX + (size_t)tid * LDX;
            x[0] = p[0] - c[0]; x[1] = p[1] - c[1]; x[2] = p[2] - c[2];
        }
    }
    if (CF > 0) {
        constexpr int V = CF / 4;                          // float4 per row
        for (int e = tid; e < SA_ROWS * V; e += 256) {
            const int r = e / V, c4 = e % V;
            const long g = group0 + (r >> 6);
            if (g < groups) {
                const long b = g / m;
                const int ii = idx[g * 64 + (r & 63)];
                const float4 v = *reinterpret_cast<const float4 *>(feats + ((size_t)b * n + ii) * CF + c4 * 4);
                float *x = bufX + (size_t)r * LDX + 3 + c4 * 4;
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            }
        }
    }
    if (tid == 0) bufX[SA_ROWS * LDX] = 0.f;               // the k = CIN read of the last row (odd CIN) lands here
    // (sa_layer starts with a barrier)
    sa_layer<SA_ROWS, SA_KC, CIN, C1, LDX, LD1, false>(bufX, L1, buf1, nullptr, 0);
    if (tid == 0) buf1[SA_ROWS * LD1] = 0.f;
    sa_layer<SA_ROWS, SA_KC, C1, C2, LD1, LD2, false>(buf1, L2, bufX, nullptr, 0);
    sa_layer<SA_ROWS, SA_KC, C2, C3, LD2, 1, true>(bufX, L3, nullptr, out, group0);
}

template <int SA_ROWS, int SA_KC, int CF, int C1, int C2, int C3>
static int launch_sa(int b, int n, int m, const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                     const SaLayer &L1, const SaLayer &L2, const SaLayer &L3, float *out, hipStream_t st) {
    constexpr int CIN = 3 + CF;
    constexpr int LDX = (CIN & 1) ? CIN : CIN + 1;
    constexpr int LD1 = C1 + 1, LD2 = C2 + 1;
    constexpr int XSZ = SA_ROWS * (LDX > LD2 ? LDX : LD2) + 4;
    const size_t lds = sizeof(float) * (XSZ + SA_ROWS * LD1 + 4);
    const long groups = (long)b * m;
    auto k = sa_fused_kernel<SA_ROWS, SA_KC, CF, C1, C2, C3>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    constexpr int GP = SA_ROWS / 64;
    hipLaunchKernelGGL(k, dim3((unsigned)((groups + GP - 1) / GP)), dim3(256), lds, st, n, m, groups, xyz, feats, new_xyz, idx, L1, L2, L3, out);
    return check_launch("sa_module_fused");
}

}  // namespace ancsh

using namespace ancsh;

// params: 12 device pointers = {w, bias, scale, shift} x 3 layers (see ancsh_conv1x1 for their meaning)
extern "C" int ancsh_sa_module_fused(int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                                     const float *feats, const float *new_xyz, const int *idx, const float *const *params,
                                     float *out, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0, "sa_module_fused: bad shape b=%d n=%d m=%d", b, n, m);
    ANCSH_REQUIRE(nsample == 64, "sa_module_fused: nsample must be 64 (got %d)", nsample);
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(((long)b * m) % 2 == 0, "sa_module_fused: b*m = %ld must be even (two neighbourhoods per workgroup)", (long)b * m);
    ANCSH_REQUIRE(xyz && new_xyz && idx && params && out && (cfeat == 0 || feats), "sa_module_fused: null pointer");
    SaLayer L[3];
    for (int i = 0; i < 3; ++i) {
        L[i].w = params[4 * i]; L[i].bias = params[4 * i + 1]; L[i].scale = params[4 * i + 2]; L[i].shift = params[4 * i + 3];
        ANCSH_REQUIRE(L[i].w && L[i].bias && L[i].scale && L[i].shift, "sa_module_fused: null layer parameter");
    }
    hipStream_t st = (hipStream_t)stream;
    if (cfeat == 0 && c1 == 64 && c2 == 64 && c3 == 128)
        return launch_sa<128, 16, 0, 64, 64, 128>(b, n, m, xyz, feats, new_xyz, idx, L[0], L[1], L[2], out, st);
    if (cfeat == 128 && c1 == 128 && c2 == 128 && c3 == 256)
        return launch_sa<64, 16, 128, 128, 128, 256>(b, n, m, xyz, feats, new_xyz, idx, L[0], L[1], L[2], out, st);
    set_error("sa_module_fused: unsupported layer shape (cfeat=%d mlp=[%d,%d,%d]); use the unfused path", cfeat, c1, c2, c3);
    return ANCSH_EINVAL;
}
