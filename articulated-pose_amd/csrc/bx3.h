// bx3.h -- building blocks of the split-bf16 ("bf16x3") kernels: an f32 value is split EXACTLY into three bf16 terms and an f32 product is
// emulated by six v_mfma_f32_32x32x16_bf16 products (see sa_bf16x3.hip for the arithmetic and its error statement).  Shared by the fused
// set-abstraction levels (sa_bf16x3.hip) and the per-point tail chain (tail_bf16x3.hip).  OPT-IN experiment code: f32 is the graded path.
#pragma once
#include "common.h"

namespace ancsh {

typedef float fx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

struct Bx3Layer {
    const uint4 *w;             // packed fragments: [(kb * TN + j) * 3 + plane][64 lanes] x 16 bytes
    const float *bias, *scale, *shift;
};

__device__ __forceinline__ unsigned short bx3_bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bx3_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void bx3_split(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
#pragma clang fp contract(off)
    h = bx3_bf(x);
    const float r1 = x - bx3_f(h);
    m = bx3_bf(r1);
    l = bx3_bf(r1 - bx3_f(m));
}

__device__ __forceinline__ void bx3_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- register-resident variant ---------------------------------------------------------------------------------------------
// The 8 bf16 a lane holds of an activation fragment -- point l31, channels 16 kb + 8 (lane >> 5) + 0..7 -- are the SAME registers
// whether the tile is used as the A operand (points as rows) or as the B operand (points as columns).  A hidden layer is therefore
// computed TRANSPOSED, D^T = W^T X^T (weights as A, activations as B): its accumulator then holds, per lane, point l31 and the
// output channels 32 i + 4 (lane >> 5) + 8 q + 0..3 (q = 0..3) -- four runs of four CONSECUTIVE channels, i.e. after bias / BN /
// ReLU, the split and v_cvt_pk_bf16_f32, halves of the next layer's fragments; one v_permlane32_swap per register pair exchanges
// the runs the two lane halves owe each other.  The activations of the whole MLP never leave the registers: no LDS tile, no
// scattered 2-byte stores, no barrier -- a wave owns a whole 64-sample neighbourhood (two point blocks, so every weight fragment
// feeds two MFMA sets: the weight stream from L2 is what bounds this kernel otherwise) and the last layer runs in the normal
// orientation so that the max over the points is a max over accumulator registers.
typedef unsigned int u32;
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
struct BxFrag {
    u32 r[4];
};
__device__ __forceinline__ bfx8 bx_as(const BxFrag &f) { return __builtin_bit_cast(bfx8, f); }

// (v0, v1) -> three registers holding (bf16(v0) | bf16(v1) << 16) of the hi / mid / lo planes
__device__ __forceinline__ void bx3_split2(float v0, float v1, u32 &h, u32 &m, u32 &l) {
#pragma clang fp contract(off)
    const f32x2v x = {v0, v1};
    h = __builtin_bit_cast(u32, __builtin_convertvector(x, bf16x2v));
    const f32x2v r1 = x - f32x2v{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(u32, __builtin_convertvector(r1, bf16x2v));
    const f32x2v r2 = r1 - f32x2v{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(u32, __builtin_convertvector(r2, bf16x2v));
}

// epilogue of ONE transposed 32-channel tile (channels 32 i ..) of a hidden layer for the wave's P point blocks: bias + folded BN
// (+ ReLU), the exact three-way bf16 split, v_permlane32_swap re-forming: accumulators -> the next layer's fragments Y[p][2 i], Y[p][2 i + 1]
template <int P, bool RELU, int NF>
__device__ __forceinline__ void bx3_tile_epilogue(const Bx3Layer &L, int i, const fx16 (&acc)[P], BxFrag (&Y)[P][NF][3]) {
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    // epilogue: register r = 4 q + t holds channel 32 i + 4 khalf + 8 q + t of point l31
    const int c0 = 32 * i + 4 * khalf;
    float4 bs[4], sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bs[q] = *reinterpret_cast<const float4 *>(L.bias + c0 + 8 * q);
        sc[q] = *reinterpret_cast<const float4 *>(L.scale + c0 + 8 * q);
        sh[q] = *reinterpret_cast<const float4 *>(L.shift + c0 + 8 * q);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        u32 y[4][3][2];                   // [q][plane][channel pair]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // bias + folded BN on packed f32 (each half the same IEEE add / fma as the scalar form)
            f32x2v a01 = {acc[p][4 * q + 0], acc[p][4 * q + 1]}, a23 = {acc[p][4 * q + 2], acc[p][4 * q + 3]};
            a01 = __builtin_elementwise_fma(a01 + f32x2v{bs[q].x, bs[q].y}, f32x2v{sc[q].x, sc[q].y}, f32x2v{sh[q].x, sh[q].y});
            a23 = __builtin_elementwise_fma(a23 + f32x2v{bs[q].z, bs[q].w}, f32x2v{sc[q].z, sc[q].w}, f32x2v{sh[q].z, sh[q].w});
            if (RELU) { a01.x = nmax(a01.x, 0.f); a01.y = nmax(a01.y, 0.f); a23.x = nmax(a23.x, 0.f); a23.y = nmax(a23.y, 0.f); }
#if defined(BX3_KO_SPLIT)      /* timing experiment only (tools/experiments): one conversion instead of the exact three-way split */
            y[q][0][0] = y[q][1][0] = y[q][2][0] = __builtin_bit_cast(u32, __builtin_convertvector(a01, bf16x2v));
            y[q][0][1] = y[q][1][1] = y[q][2][1] = __builtin_bit_cast(u32, __builtin_convertvector(a23, bf16x2v));
#else
            bx3_split2(a01.x, a01.y, y[q][0][0], y[q][1][0], y[q][2][0]);
            bx3_split2(a23.x, a23.y, y[q][0][1], y[q][1][1], y[q][2][1]);
#endif
        }
        // lanes 0..31 hold channel runs 0-3 / 8-11 / 16-19 / 24-27 of the tile, lanes 32..63 the runs 4-7 / 12-15 / 20-23 / 28-31;
        // fragment kb' = 2 i wants channels 0..7 in the lower and 8..15 in the upper lanes: swap(upper's q0, lower's q1); same for q2 / q3
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                auto s01 = __builtin_amdgcn_permlane32_swap(y[0][pl][w], y[1][pl][w], false, false);
                auto s23 = __builtin_amdgcn_permlane32_swap(y[2][pl][w], y[3][pl][w], false, false);
                y[0][pl][w] = s01[0]; y[1][pl][w] = s01[1];
                y[2][pl][w] = s23[0]; y[3][pl][w] = s23[1];
            }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            Y[p][2 * i][pl] = BxFrag{{y[0][pl][0], y[0][pl][1], y[1][pl][0], y[1][pl][1]}};
            Y[p][2 * i + 1][pl] = BxFrag{{y[2][pl][0], y[2][pl][1], y[3][pl][0], y[3][pl][1]}};
        }
    }
}

// hidden layer, transposed: X[P][KB][3] (K = 16 KB channels) -> Y[P][N / 16][3].  init[p] != nullptr: the accumulators of point
// block p start from the f32 row init[p][0:N] of THIS lane's point (the first layer's per-point partial sums over the feature
// channels, see ancsh_sa_module_fused_partial) instead of zero.
template <int KB, int N, int P, bool RELU = true>
__device__ __forceinline__ void bx3_hidden(const Bx3Layer &L, const BxFrag (&X)[P][KB][3], BxFrag (&Y)[P][N / 16][3],
                                           const float *const (&init)[P]) {
    constexpr int TM = N / 32;
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    const uint4 *Wp = L.w + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        fx16 acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (init[p]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(init[p] + 32 * i + 4 * khalf + 8 * q);
                    acc[p][4 * q] = v.x; acc[p][4 * q + 1] = v.y; acc[p][4 * q + 2] = v.z; acc[p][4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
            }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const uint4 *wf = Wp + (size_t)((kb * TM + i) * 3) * 64;
            const bfx8 W[3] = {__builtin_bit_cast(bfx8, wf[0]), __builtin_bit_cast(bfx8, wf[64]), __builtin_bit_cast(bfx8, wf[128])};
            // weights are the A operand here: products W_a * X_b, smallest first (mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi); the
            // point blocks' accumulators alternate so that no MFMA waits for the one issued just before it
            constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[TA[t]], bx_as(X[p][kb][TB[t]]), acc[p], 0, 0, 0);
        }
        bx3_tile_epilogue<P, RELU, N / 16>(L, i, acc, Y);
    }
}


// The same hidden layer with the k-block loop OUTSIDE and all N / 32 output tiles' accumulators live (N = 128: 4 x P x 16 VGPRs): the
// weight fragments of k-block kb + 1 (N / 32 x 3 x 16 bytes per lane) are requested before the 6 x P x N / 32 MFMAs of k-block kb are
// issued, so the L2 latency of the weight stream hides under ~1500 clocks of matrix work instead of being paid once per output tile
// (bx3_hidden's order: one wave per SIMD, no spare registers for the compiler to hoist the loads -- measured on the tail chain: 159 us
// against 42 us of pure MFMA issue per wave).  X is dead when the epilogue forms Y, so Y may take X's registers; NOT for a layer whose
// input must survive it (that one keeps bx3_hidden's order).
template <int KB, int N, int P, bool RELU = true>
__device__ __forceinline__ void bx3_hidden_kouter(const Bx3Layer &L, const BxFrag (&X)[P][KB][3], BxFrag (&Y)[P][N / 16][3]) {
    constexpr int TM = N / 32;
    const int lane = threadIdx.x & 63;
    const uint4 *Wp = L.w + lane;
    fx16 acc[P][TM];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][i][r] = 0.f;
    uint4 w[2][TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[0][i][pl] = Wp[(size_t)(i * 3 + pl) * 64];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#if defined(BX3_KO_WLOAD)      /* timing experiment only: no weight stream after the layer's first k-block */
                    w[(kb + 1) & 1][i][pl] = w[kb & 1][i][pl];
#else
                    w[(kb + 1) & 1][i][pl] = Wp[(size_t)(((kb + 1) * TM + i) * 3 + pl) * 64];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};      // (weight plane, activation plane), smallest products first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, w[kb & 1][i][TA[t]]), bx_as(X[p][kb][TB[t]]), acc[p][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        fx16 a[P];
#pragma unroll
        for (int p = 0; p < P; ++p) a[p] = acc[p][i];
        bx3_tile_epilogue<P, RELU, N / 16>(L, i, a, Y);
    }
}

}  // namespace ancsh
