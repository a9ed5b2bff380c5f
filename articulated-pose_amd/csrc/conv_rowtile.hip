// conv_rowtile.hip -- shared per-point MLP layer for the SMALL layers of the backbone (4096..16384 rows, 128..384 input channels:
// SA3's first two layers, FP1, FP2, the per-point partial sums of SA2's first layer), reached through ancsh_conv1x1_packed.
//
// Same operator and the same arithmetic as ancsh_conv1x1 (tf_util.conv1d / conv2d 1x1 + bias + inference BN + ReLU;
// pointnet_plusplus/utils/tf_util.py:52-185, pointnet_util.py:118-134,228-234): one k-ordered f32 fmaf chain per output on
// v_mfma_f32_32x32x2_f32, identical bits.  These launches are LATENCY-bound: a 4096 x 256 -> 256 layer is 1024 output tiles of
// 32 x 32, i.e. exactly one wave per SIMD running K/2 dependent MFMAs (3.4 us), and the workgroup-tiled kernel (mlp.hip) spends
// 13-17 us on it because every 16-k chunk of A and B goes global -> registers -> LDS behind two barriers.  Here
//   * a workgroup owns ONE 32-row tile and four adjacent 32-column tiles (one per wave);
//   * the whole 32 x K input tile is brought into LDS once, every load of a lane in flight together, one barrier;
//   * each wave then runs the k loop of wave_mlp.h on the shared tile: its weights in the packed fragment order straight from L2
//     two slots ahead, activations eight k-steps ahead, k fully unrolled (K is a template parameter), epilogue from registers.
#include "common.h"
#include "wave_mlp.h"

namespace ancsh {

#ifndef ROWTILE_MAX_ROWS
#define ROWTILE_MAX_ROWS (1L << 30)
#endif

template <int K>
__global__ __launch_bounds__(256) void conv_rowtile_kernel(long rows, int cout, const float *__restrict__ x, int ldx,
                                                           const float *__restrict__ wp, const float *__restrict__ bias,
                                                           const float *__restrict__ scale, const float *__restrict__ shift, int act,
                                                           float *__restrict__ y, int ldy, const float *__restrict__ acc_init,
                                                           int init_rows, ConvGroups G) {
    CONV_SELECT_GROUP(G, x, y, acc_init, wp, bias, scale, shift)
    constexpr int LD = (K + 1) | 1;                            // odd, column K readable (zero) when K is odd
    extern __shared__ __attribute__((aligned(16))) float T[];  // 32 x LD
    const int tid = threadIdx.x, lane = tid & 63, khalf = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 32;
    const int tile = blockIdx.y * 4 + wave;                    // this wave's 32-column tile
    const bool raw = act == ANCSH_ACT_RAW;
    SaLayer L;
    L.w = wp + (size_t)tile * 256;
    L.wstride = (cout / 32) * 64;
    L.ncol = 32;
    // raw accumulators: no epilogue constants exist; the (unused) loads of the k loop are pointed at valid memory
    L.bias = raw ? wp : bias + tile * 32; L.scale = raw ? wp : scale + tile * 32; L.shift = raw ? wp : shift + tile * 32;
    float4 bw[LayerCfg<K, 32>::DW + 1][1];
    w_prologue<K, 32, true>(L, bw);                            // the first weights fly while the input tile is staged
    floatx16 acc[1][1];
    if (acc_init) {                                            // the chain continues acc_init[row / init_rows][col]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            row = row < rows ? row : rows - 1;
            acc[0][0][r] = acc_init[(size_t)((unsigned)row / (unsigned)init_rows) * cout + tile * 32 + l31];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    }
    // ---- the 32 x K input tile: unconditional loads (row clamped), values past the last row zeroed -----------------------------
    if ((ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        constexpr int V4 = (K + 3) / 4;
        constexpr int NV = (32 * V4 + 255) / 256;
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, r = e / V4, c4 = e - r * V4;
            const long row = (e < 32 * V4 && row0 + r < rows) ? row0 + r : (row0 < rows ? row0 : rows - 1);
            v[i] = *reinterpret_cast<const float4 *>(x + (size_t)row * ldx + (e < 32 * V4 ? c4 : 0) * 4);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, r = e / V4, c = (e - r * V4) * 4;
            if (e < 32 * V4) {
                const bool in = row0 + r < rows;
                float *d = T + r * LD + c;
                d[0] = (in && c < K) ? v[i].x : 0.f;
                if (c + 1 < LD) d[1] = (in && c + 1 < K) ? v[i].y : 0.f;
                if (c + 2 < LD) d[2] = (in && c + 2 < K) ? v[i].z : 0.f;
                if (c + 3 < LD) d[3] = (in && c + 3 < K) ? v[i].w : 0.f;
            }
        }
    } else {
        constexpr int KP = K + (K & 1);                        // odd K: column K is the zero pad of the last k-step
        constexpr int NS = (32 * KP + 255) / 256;
        float v[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = tid + 256 * i, r = e / KP, c = e - r * KP;
            const bool ok = e < 32 * KP && c < K && row0 + r < rows;
            const long row = ok ? row0 + r : (row0 < rows ? row0 : rows - 1);
            v[i] = x[(size_t)row * ldx + (ok ? c : 0)];
            v[i] = ok ? v[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = tid + 256 * i, r = e / KP, c = e - r * KP;
            if (e < 32 * KP) T[r * LD + c] = v[i];
        }
    }
    __syncthreads();
    float ep[3][1];
    mfma_loop<K, 32, LD, 1, 0, 2, true>(T, L, bw, acc, ep);
    if (raw) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (row < rows) y[(size_t)row * ldy + tile * 32 + l31] = acc[0][0][r];
        }
    } else if (act == ANCSH_ACT_RELU) {
        epilogue_global<32, 1, true>(y + tile * 32, ldy, 32, row0, rows, acc, ep);
    } else {
        epilogue_global<32, 1, false>(y + tile * 32, ldy, 32, row0, rows, acc, ep);
    }
}

template <int K>
static void rowtile_go(long rows, int cout, const float *x, int ldx, const float *wp, const float *bias, const float *scale,
                       const float *shift, int act, float *y, int ldy, const float *acc_init, int init_rows, const ConvGroups &G,
                       hipStream_t st) {
    constexpr int LD = (K + 1) | 1;
    const size_t lds = sizeof(float) * 32 * LD;
    auto k = conv_rowtile_kernel<K>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((rows + 31) / 32), cout / 128, G.n), dim3(256), lds, st, rows, cout, x, ldx, wp, bias, scale, shift,
                       act, y, ldy, acc_init, init_rows, G);
}

// true when the shape is one this kernel serves (the caller has validated the arguments): launched; false: not launched
bool conv_rowtile_launch(long rows, int cin, int cout, const float *x, int ldx, const float *wp, const float *bias,
                         const float *scale, const float *shift, int act, float *y, int ldy, const float *acc_init, int init_rows,
                         const ConvGroups &G, hipStream_t st) {
    if (cout % 128 != 0 || rows > ROWTILE_MAX_ROWS) return false;
    switch (cin) {
    case 128: rowtile_go<128>(rows, cout, x, ldx, wp, bias, scale, shift, act, y, ldy, acc_init, init_rows, G, st); return true;
    case 256: rowtile_go<256>(rows, cout, x, ldx, wp, bias, scale, shift, act, y, ldy, acc_init, init_rows, G, st); return true;
    case 259: rowtile_go<259>(rows, cout, x, ldx, wp, bias, scale, shift, act, y, ldy, acc_init, init_rows, G, st); return true;
    case 384: rowtile_go<384>(rows, cout, x, ldx, wp, bias, scale, shift, act, y, ldy, acc_init, init_rows, G, st); return true;
    default: return false;
    }
}

}  // namespace ancsh
