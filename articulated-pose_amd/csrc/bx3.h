// bx3.h -- building blocks of the SPLIT-16 kernels: f32 operands are split into a few 16-bit terms and an f32 product is emulated by a
// few 16-bit MFMA products with f32 accumulation.  Two schemes (template parameter S of everything below):
//
//   Bf16x3   x = hi + mid + lo, three bf16 terms (8 + 8 + 8 significant bits, the split is EXACT); six v_mfma_f32_32x32x16_bf16
//            products per f32 product (terms below 2^-24 of the product dropped): each product reproduced to f32's own precision
//            at 6/16 of the f32 MFMA's pipe time.  Range = f32's.  (sa_bf16x3.hip has the full statement.)
//   F16x2    x = hi + mid * 2^-11, two f16 terms (11 + 11 bits; mid = f16((x - hi) * 2^11) is kept SCALED so that it stays a normal
//            number wherever hi is), THREE v_mfma_f32_32x32x16_f16 products into TWO accumulators
//                 A0 += a_hi w_hi,     A1 += a_hi w_mid + a_mid w_hi,     result = A0 + 2^-11 A1
//            (a_mid w_mid, 2^-22 of the product, dropped): ~22 bits per operand instead of 24 -- each product to ~7e-7 relative instead
//            of 6e-8 -- at 3/16 of the f32 MFMA's pipe time, two thirds of the operand registers and weight bytes, a shorter split.
//            Range = f16's: an activation or weight beyond +-65504 becomes +-inf, which reaches the cloud's outputs as inf / NaN in all but
//            contrived cases (a lone -inf product that a ReLU turns into 0 where the true sum was positive is the exception) -- the scheme
//            is for networks whose activations stay far inside that range (ANCSH: unit-diagonal clouds, batch-normalised layers: O(1..100));
//            f16 subnormals are exact on this hardware (v_cvt_pk_f16_f32 rounds to nearest even into them, the f16 MFMA does not flush
//            them: tools/f16_probe.hip on an MI355X).
//
// Shared by the fused set-abstraction levels (sa_bf16x3.hip) and the per-point tail chain (tail_bf16x3.hip).  OPT-IN experiment code: f32
// is the graded arithmetic.
#pragma once
#include "common.h"

namespace ancsh {

typedef float fx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 hfx8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

struct Bx3Layer {
    const uint4 *w;             // packed fragments: [(kb * TN + j) * S::NP + plane][64 lanes] x 16 bytes
    const float *bias, *scale, *shift;
};

// eight 16-bit values of one operand plane: point (or output channel) lane & 31, k = 16 kb + 8 (lane >> 5) + 0..7
struct BxFrag {
    u32 r[4];
};

__device__ __forceinline__ void bx3_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Bf16x3 {
    static constexpr int NP = 3, NACC = 1, NPROD = 6;
    // products in issue order, smallest first: (weight plane, activation plane) -> accumulator
    static constexpr int PW[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0}, PC[6] = {0, 0, 0, 0, 0, 0};
    __device__ static __forceinline__ fx16 mfma(const uint4 &a, const uint4 &b, const fx16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
    }
    // (v0, v1) -> one register per plane holding (term(v0) | term(v1) << 16)
    __device__ static __forceinline__ void split2(float v0, float v1, u32 (&pl)[3]) {
#pragma clang fp contract(off)
        const f32x2v x = {v0, v1};
        pl[0] = __builtin_bit_cast(u32, __builtin_convertvector(x, bf16x2v));
        const f32x2v r1 = x - f32x2v{__uint_as_float(pl[0] << 16), __uint_as_float(pl[0] & 0xffff0000u)};
        pl[1] = __builtin_bit_cast(u32, __builtin_convertvector(r1, bf16x2v));
        const f32x2v r2 = r1 - f32x2v{__uint_as_float(pl[1] << 16), __uint_as_float(pl[1] & 0xffff0000u)};
        pl[2] = __builtin_bit_cast(u32, __builtin_convertvector(r2, bf16x2v));
    }
    __device__ static __forceinline__ void split1(float x, unsigned short (&pl)[3]) {
#pragma clang fp contract(off)
        const auto bf = [](float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); };
        const auto fl = [](unsigned short h) { return __uint_as_float(((unsigned)h) << 16); };
        pl[0] = bf(x);
        const float r1 = x - fl(pl[0]);
        pl[1] = bf(r1);
        pl[2] = bf(r1 - fl(pl[1]));
    }
    // folded-BN value of the accumulator registers r, r + 1: acc * scale + (bias * scale + shift); scl = scale * 2^-11 is unused here
    __device__ static __forceinline__ f32x2v bn2(const fx16 (&acc)[1], int r, f32x2v sc, f32x2v scl, f32x2v shf) {
        return __builtin_elementwise_fma(f32x2v{acc[0][r], acc[0][r + 1]}, sc, shf);
    }
};

struct F16x2 {
    static constexpr int NP = 2, NACC = 2, NPROD = 3;
    static constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0}, PC[3] = {1, 1, 0};
    static constexpr float UP = 2048.f, DOWN = 1.f / 2048.f;      // 2^11: the scale the mid plane is kept at
    __device__ static __forceinline__ fx16 mfma(const uint4 &a, const uint4 &b, const fx16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfx8, a), __builtin_bit_cast(hfx8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split2(float v0, float v1, u32 (&pl)[2]) {
#pragma clang fp contract(off)
        const f32x2v x = {v0, v1};
        const f16x2v h = __builtin_convertvector(x, f16x2v);                      // v_cvt_pk_f16_f32: round to nearest even
        pl[0] = __builtin_bit_cast(u32, h);
        // (x - hi) * 2^11 = fma(hi, -2^11, x * 2^11): exact (x - hi is exact, so is the power-of-two scale); written so that the f16 halves
        // feed the fma directly (v_fma_mix_f32) instead of being converted back first
        const f32x2v xs = x * f32x2v{UP, UP};
        const f32x2v r = {__builtin_fmaf((float)h[0], -UP, xs[0]), __builtin_fmaf((float)h[1], -UP, xs[1])};
        pl[1] = __builtin_bit_cast(u32, __builtin_convertvector(r, f16x2v));
    }
    __device__ static __forceinline__ void split1(float x, unsigned short (&pl)[2]) {
#pragma clang fp contract(off)
        const _Float16 h = (_Float16)x;
        pl[0] = __builtin_bit_cast(unsigned short, h);
        pl[1] = __builtin_bit_cast(unsigned short, (_Float16)((x - (float)h) * UP));
    }
    // (A0 + 2^-11 A1) * scale + shf = A0 * scale + (A1 * (scale * 2^-11) + shf): the combination of the two accumulators costs nothing extra
    __device__ static __forceinline__ f32x2v bn2(const fx16 (&acc)[2], int r, f32x2v sc, f32x2v scl, f32x2v shf) {
        return __builtin_elementwise_fma(f32x2v{acc[0][r], acc[0][r + 1]}, sc, __builtin_elementwise_fma(f32x2v{acc[1][r], acc[1][r + 1]}, scl, shf));
    }
};

template <class S>
__device__ __forceinline__ uint4 bx_u4(const BxFrag &f) { return __builtin_bit_cast(uint4, f); }

// ---- register-resident layers ------------------------------------------------------------------------------------------------
// The 8 halves a lane holds of an activation fragment -- point l31, channels 16 kb + 8 (lane >> 5) + 0..7 -- are the SAME registers
// whether the tile is used as the A operand (points as rows) or as the B operand (points as columns).  A hidden layer is therefore
// computed TRANSPOSED, D^T = W^T X^T (weights as A, activations as B): its accumulator then holds, per lane, point l31 and the
// output channels 32 i + 4 (lane >> 5) + 8 q + 0..3 (q = 0..3) -- four runs of four CONSECUTIVE channels, i.e. after bias / BN /
// ReLU, the split and the packed conversion, halves of the next layer's fragments; one v_permlane32_swap per register pair exchanges
// the runs the two lane halves owe each other.  The activations of the whole MLP never leave the registers: no LDS tile, no
// scattered 2-byte stores, no barrier -- a wave owns 64 points (two point blocks, so every weight fragment feeds two MFMA sets: the
// weight stream from L2 is what bounds these kernels otherwise).

// epilogue of ONE transposed 32-channel tile (channels 32 i ..) of a hidden layer for the wave's P point blocks: combine the scheme's
// accumulators, bias + folded BN (+ ReLU), the split, v_permlane32_swap re-forming: -> the next layer's fragments Y[p][2 i], Y[p][2 i + 1]
// the epilogue constants of one transposed tile as loaded: this lane's 4 x 4 channels of bias / scale / shift.  Loaded one tile AHEAD of
// their use (bx3_epi_load before the k loop's last MFMAs; bx3_tile_epilogue requests tile i + 1's while it works on tile i): at one wave
// per SIMD every tile's epilogue used to start with an exposed L2 round trip.
struct EpiRaw {
    float4 b[4], s[4], h[4];
};
__device__ __forceinline__ EpiRaw bx3_epi_load(const Bx3Layer &L, int i) {
    const int c0 = 32 * i + 4 * ((threadIdx.x & 63) >> 5);
    EpiRaw r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r.b[q] = *reinterpret_cast<const float4 *>(L.bias + c0 + 8 * q);
        r.s[q] = *reinterpret_cast<const float4 *>(L.scale + c0 + 8 * q);
        r.h[q] = *reinterpret_cast<const float4 *>(L.shift + c0 + 8 * q);
    }
    return r;
}

// raw: in = tile i's constants (bx3_epi_load), out = tile i + 1's when i + 1 < TM
template <class S, int P, bool RELU, int NF, int TM>
__device__ __forceinline__ void bx3_tile_epilogue(const Bx3Layer &L, int i, EpiRaw &raw, const fx16 (&acc)[P][S::NACC], BxFrag (&Y)[P][NF][S::NP]) {
    // register r = 4 q + t holds channel 32 i + 4 khalf + 8 q + t of point l31
    // bias folded into the shift ONCE per tile, shared by the P point blocks: (v + b) * sc + sh = v * sc + (b * sc + sh) -- VALU time is
    // matrix-pipe time for these kernels (tools/mfma_bf16_ub.hip: 16-bit MFMA and VALU issue strictly one after the other)
    f32x2v sc[4][2], scl[4][2], shf[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = raw.b[q], s4 = raw.s[q], h4 = raw.h[q];
        sc[q][0] = f32x2v{s4.x, s4.y}; sc[q][1] = f32x2v{s4.z, s4.w};
        shf[q][0] = __builtin_elementwise_fma(f32x2v{b4.x, b4.y}, sc[q][0], f32x2v{h4.x, h4.y});
        shf[q][1] = __builtin_elementwise_fma(f32x2v{b4.z, b4.w}, sc[q][1], f32x2v{h4.z, h4.w});
        scl[q][0] = sc[q][0] * f32x2v{1.f / 2048.f, 1.f / 2048.f}; scl[q][1] = sc[q][1] * f32x2v{1.f / 2048.f, 1.f / 2048.f};      // (F16x2 only; dead code otherwise)
    }
    if (i + 1 < TM) raw = bx3_epi_load(L, i + 1);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        u32 y[4][S::NP][2];                   // [q][plane][channel pair]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x2v a01 = S::bn2(acc[p], 4 * q, sc[q][0], scl[q][0], shf[q][0]), a23 = S::bn2(acc[p], 4 * q + 2, sc[q][1], scl[q][1], shf[q][1]);
            if (RELU) { a01.x = nmax(a01.x, 0.f); a01.y = nmax(a01.y, 0.f); a23.x = nmax(a23.x, 0.f); a23.y = nmax(a23.y, 0.f); }
            u32 s01[S::NP], s23[S::NP];
#if defined(BX3_KO_SPLIT)      /* timing experiment only (tools/experiments): one conversion instead of the split */
            for (int pl = 0; pl < S::NP; ++pl) { s01[pl] = __builtin_bit_cast(u32, __builtin_convertvector(a01, bf16x2v)); s23[pl] = __builtin_bit_cast(u32, __builtin_convertvector(a23, bf16x2v)); }
#else
            S::split2(a01.x, a01.y, s01);
            S::split2(a23.x, a23.y, s23);
#endif
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) { y[q][pl][0] = s01[pl]; y[q][pl][1] = s23[pl]; }
        }
        // lanes 0..31 hold channel runs 0-3 / 8-11 / 16-19 / 24-27 of the tile, lanes 32..63 the runs 4-7 / 12-15 / 20-23 / 28-31;
        // fragment kb' = 2 i wants channels 0..7 in the lower and 8..15 in the upper lanes: swap(upper's q0, lower's q1); same for q2 / q3
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                auto s01 = __builtin_amdgcn_permlane32_swap(y[0][pl][w], y[1][pl][w], false, false);
                auto s23 = __builtin_amdgcn_permlane32_swap(y[2][pl][w], y[3][pl][w], false, false);
                y[0][pl][w] = s01[0]; y[1][pl][w] = s01[1];
                y[2][pl][w] = s23[0]; y[3][pl][w] = s23[1];
            }
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl) {
            Y[p][2 * i][pl] = BxFrag{{y[0][pl][0], y[0][pl][1], y[1][pl][0], y[1][pl][1]}};
            Y[p][2 * i + 1][pl] = BxFrag{{y[2][pl][0], y[2][pl][1], y[3][pl][0], y[3][pl][1]}};
        }
    }
}

template <class S, int P>
__device__ __forceinline__ void bx3_zero(fx16 (&acc)[P][S::NACC]) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int a = 0; a < S::NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;
}

// hidden layer, transposed, OUTPUT-TILE loop outside: X[P][KB][NP] (K = 16 KB channels) -> Y[P][N / 16][NP].  init[p] != nullptr: the
// accumulators of point block p start from the f32 row init[p][0:N] of THIS lane's point (the first layer's per-point partial sums over the
// feature channels, see ancsh_sa_module_fused_partial) instead of zero.  Never holds more than X + Y: the order for two waves per SIMD
// (the other wave hides the weight stream's latency) and for a layer whose input must survive it.
template <class S, int KB, int N, int P, bool RELU = true>
__device__ __forceinline__ void bx3_hidden(const Bx3Layer &L, const BxFrag (&X)[P][KB][S::NP], BxFrag (&Y)[P][N / 16][S::NP],
                                           const float *const (&init)[P]) {
    constexpr int TM = N / 32;
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    const uint4 *Wp = L.w + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        EpiRaw raw = bx3_epi_load(L, i);              // flies under this tile's k loop
        fx16 acc[P][S::NACC];
        bx3_zero<S, P>(acc);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (init[p]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(init[p] + 32 * i + 4 * khalf + 8 * q);
                    acc[p][0][4 * q] = v.x; acc[p][0][4 * q + 1] = v.y; acc[p][0][4 * q + 2] = v.z; acc[p][0][4 * q + 3] = v.w;
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const uint4 *wf = Wp + (size_t)((kb * TM + i) * S::NP) * 64;
            uint4 W[S::NP];
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) W[pl] = wf[pl * 64];
            // weights are the A operand here; the point blocks' accumulators alternate so that no MFMA waits for the one issued just before it
#pragma unroll
            for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p][S::PC[t]] = S::mfma(W[S::PW[t]], bx_u4<S>(X[p][kb][S::PA[t]]), acc[p][S::PC[t]]);
        }
        bx3_tile_epilogue<S, P, RELU, N / 16, 0>(L, i, raw, acc, Y);      // TM = 0: no look-ahead, the next tile loads its own before its k loop
    }
}

// The same hidden layer with the k-block loop OUTSIDE and all N / 32 output tiles' accumulators live: the weight fragments of k-block
// kb + 1 (N / 32 x NP x 16 bytes per lane) are requested before the NPROD x P x N / 32 MFMAs of k-block kb are issued, so the L2 latency
// of the weight stream hides under the matrix work instead of being paid once per output tile (bx3_hidden's order: with one wave per
// SIMD and no spare registers for the compiler to hoist the loads -- measured on the bf16x3 tail chain: 159 us against 42 us of pure
// MFMA issue per wave, 118 us with this order).  X is dead when the epilogue forms Y, so Y may take X's registers; NOT for a layer
// whose input must survive it.
template <class S, int KB, int N, int P, bool RELU = true>
__device__ __forceinline__ void bx3_hidden_kouter(const Bx3Layer &L, const BxFrag (&X)[P][KB][S::NP], BxFrag (&Y)[P][N / 16][S::NP]) {
    constexpr int TM = N / 32;
    const int lane = threadIdx.x & 63;
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
                for (int pl = 0; pl < S::NP; ++pl)
#if defined(BX3_KO_WLOAD)      /* timing experiment only: no weight stream after the layer's first k-block */
                    w[(kb + 1) & 1][i][pl] = w[kb & 1][i][pl];
#else
                    w[(kb + 1) & 1][i][pl] = Wp[(size_t)(((kb + 1) * TM + i) * S::NP + pl) * 64];
#endif
        }
        if (kb == KB - 1) raw = bx3_epi_load(L, 0);     // the first tile's epilogue constants fly under the last k-block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[i][p][S::PC[t]] = S::mfma(w[kb & 1][i][S::PW[t]], bx_u4<S>(X[p][kb][S::PA[t]]), acc[i][p][S::PC[t]]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) bx3_tile_epilogue<S, P, RELU, N / 16, TM>(L, i, raw, acc[i], Y);
}


// The FIRST layer of a set-abstraction level (one k-block: the three centred coordinates) with everything it waits for requested up front:
// every output tile's weight fragments and -- PARTIAL levels -- every tile's accumulator start (the per-point partial sums, gathered by
// idx) before the first MFMA.  In bx3_hidden's order each of the N / 32 tiles paid its own L2 round trips for 6 MFMAs of work
// (kernel trace of the F16x2 feature level: four serial load / wait-for-all groups before the second layer starts, ~5 us per wave).
template <class S, int N, int P, bool RELU = true>
__device__ __forceinline__ void bx3_first_kouter(const Bx3Layer &L, const BxFrag (&X)[P][1][S::NP], BxFrag (&Y)[P][N / 16][S::NP],
                                                 const float *const (&init)[P]) {
    constexpr int TM = N / 32;
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    const uint4 *Wp = L.w + lane;
    uint4 w[TM][S::NP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl) w[i][pl] = Wp[(size_t)(i * S::NP + pl) * 64];
    EpiRaw raw = bx3_epi_load(L, 0);
    fx16 acc[TM][P][S::NACC];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bx3_zero<S, P>(acc[i]);
#pragma unroll
        for (int p = 0; p < P; ++p)
            if (init[p]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(init[p] + 32 * i + 4 * khalf + 8 * q);
                    acc[i][p][0][4 * q] = v.x; acc[i][p][0][4 * q + 1] = v.y; acc[i][p][0][4 * q + 2] = v.z; acc[i][p][0][4 * q + 3] = v.w;
                }
            }
    }
#pragma unroll
    for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc[i][p][S::PC[t]] = S::mfma(w[i][S::PW[t]], bx_u4<S>(X[p][0][S::PA[t]]), acc[i][p][S::PC[t]]);
#pragma unroll
    for (int i = 0; i < TM; ++i) bx3_tile_epilogue<S, P, RELU, N / 16, TM>(L, i, raw, acc[i], Y);
}

}  // namespace ancsh
