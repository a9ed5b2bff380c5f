// sa_bf16x3.hip -- EXPERIMENT (opt-in, never the default): the fused set-abstraction body of sa_fused.hip with every f32 product
// emulated on the 16-bit matrix pipe.  Two split schemes share every line below (template parameter S, bx3.h): Bf16x3 -- described
// here, the form rounds 3-5 had -- and F16x2 (round 6: two f16 terms per operand, three products into two accumulators; bx3.h).
//
// f32 MFMA runs at the vector rate on gfx950 (v_mfma_f32_32x32x2_f32: 64 clocks for 4096 FLOP); v_mfma_f32_32x32x16_bf16 does 32768
// FLOP in 32 clocks, 16x the rate.  An f32 value splits EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significant
// bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), every subtraction exact in f32), a product of two bf16 values
// is exact in f32, and the six products of weight 2^0 .. 2^-16
//      a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid
// drop only terms below 2^-24 of the product: each f32 product is reproduced to f32's own precision, at 6/16 of the f32 MFMA's
// pipe time.  What is NOT reproduced is the ORDER of the additions: the instruction sums 16 products per accumulator update in its
// own internal order, where the reference path (sa_fused.hip, the CPU oracle) adds one k after the other -- results agree with
// the f32 path to f32 summation noise (~1e-7 relative per layer), not bit for bit.  Hence an experiment: f32 stays the graded
// arithmetic.  tests/test_bf16x3_gpu.py reports max |diff|, part-label flips and the kernel time.
//
// Layout: see "register-resident" below -- the activations of a neighbourhood's whole MLP stay in one wave's registers.  (A first
// version kept the three bf16 planes of a 32-row tile in LDS like sa_fused.hip does for f32: 195 us for the feature-less level
// against 207 us in f32 -- the scattered 2-byte stores of the split outputs and the exposed LDS round trips ate the matrix gain --
// and 531 us with the full 131-channel first layer of the second level; removed.)
#include "bx3.h"

namespace ancsh {

// last layer, normal orientation (activations as the A operand), pooled over all P * 32 points of the wave: pm[j] (lanes 0..31) = max.
// Output-tile loop outside (two waves per SIMD).
template <class S, int KB, int N, int P>
__device__ __forceinline__ void bx3_pooled(const Bx3Layer &L, const BxFrag (&X)[P][KB][S::NP], float (&pm)[N / 32]) {
    constexpr int TN = N / 32;
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const uint4 *Wp = L.w + lane;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        fx16 acc[P][S::NACC];
        bx3_zero<S, P>(acc);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const uint4 *wf = Wp + (size_t)((kb * TN + j) * S::NP) * 64;
            uint4 W[S::NP];
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) W[pl] = wf[pl * 64];
#pragma unroll
            for (int t = 0; t < S::NPROD; ++t)          // (activation plane PW[t], weight plane PA[t]): the same set of products, roles swapped
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p][S::PC[t]] = S::mfma(bx_u4<S>(X[p][kb][S::PW[t]]), W[S::PA[t]], acc[p][S::PC[t]]);
        }
        const int col = j * 32 + l31;
        const float sc = L.scale[col], shf = __builtin_fmaf(L.bias[col], sc, L.shift[col]);      // bias folded into the shift
        const f32x2v sc2 = {sc, sc}, scl2 = {sc * (1.f / 2048.f), sc * (1.f / 2048.f)}, shf2 = {shf, shf};
        float mx = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2v v = S::bn2(acc[p], r, sc2, scl2, shf2);
                mx = nmax(nmax(mx, v.x), v.y);                  // one v_maximum3_f32; max starts at 0: the ReLU is implicit
            }
        pm[j] = nmax(mx, __shfl_xor(mx, 32, 64));
    }
}

// bx3_pooled with the k-block loop outside (see bx3_hidden_kouter): the N / 32 output tiles in groups of TNG, a group's accumulators
// live across its k loop, the weight fragments of k-block kb + 1 requested before the MFMAs of k-block kb.
template <class S, int KB, int N, int P, int TNG>
__device__ __forceinline__ void bx3_pooled_kouter(const Bx3Layer &L, const BxFrag (&X)[P][KB][S::NP], float (&pm)[N / 32]) {
    constexpr int TN = N / 32;
    static_assert(TN % TNG == 0, "tile groups");
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const uint4 *Wp = L.w + lane;
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += TNG) {
        fx16 acc[TNG][P][S::NACC];
#pragma unroll
        for (int j = 0; j < TNG; ++j) bx3_zero<S, P>(acc[j]);
        uint4 w[2][TNG][S::NP];
#pragma unroll
        for (int j = 0; j < TNG; ++j)
#pragma unroll
            for (int pl = 0; pl < S::NP; ++pl) w[0][j][pl] = Wp[(size_t)((j0 + j) * S::NP + pl) * 64];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + 1 < KB) {
#pragma unroll
                for (int j = 0; j < TNG; ++j)
#pragma unroll
                    for (int pl = 0; pl < S::NP; ++pl) w[(kb + 1) & 1][j][pl] = Wp[(size_t)(((kb + 1) * TN + j0 + j) * S::NP + pl) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
                for (int j = 0; j < TNG; ++j)
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        acc[j][p][S::PC[t]] = S::mfma(bx_u4<S>(X[p][kb][S::PW[t]]), w[kb & 1][j][S::PA[t]], acc[j][p][S::PC[t]]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < TNG; ++j) {
            const int col = (j0 + j) * 32 + l31;
            const float sc = L.scale[col], shf = __builtin_fmaf(L.bias[col], sc, L.shift[col]);
            const f32x2v sc2 = {sc, sc}, scl2 = {sc * (1.f / 2048.f), sc * (1.f / 2048.f)}, shf2 = {shf, shf};
            float mx = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2v v = S::bn2(acc[j][p], r, sc2, scl2, shf2);
                    mx = nmax(nmax(mx, v.x), v.y);
                }
            pm[j0 + j] = nmax(mx, __shfl_xor(mx, 32, 64));
        }
    }
}

struct Bx3Nets {
    Bx3Layer L[ANCSH_MAX_GROUPS][3];
};

// 3 (+ per-point partial sums of the first layer) -> C1 -> C2 -> C3, a wave per neighbourhood.  PARTIAL = false: a level without
// input features; PARTIAL = true: `partial` (b, n, C1) holds, per source point, the first layer's f32 partial sums over the feature
// channels (computed once per point by ancsh_conv1x1*, features first as everywhere in this library) and the layer here only adds
// the three coordinate products on the 16-bit pipe.
// groups = ngroups * geo_groups neighbourhoods, network-major: neighbourhood g of network g / geo_groups reads the SHARED geometry
// (xyz, new_xyz, idx) of neighbourhood g % geo_groups, its own network's partial rows and parameters, and writes out row g.
template <class S, int C1, int C2, int C3, bool PARTIAL, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void sa_bf16x3_reg_kernel(int n, int m, long groups, long geo_groups, const float *__restrict__ xyz, const float *__restrict__ partial,
                          const float *__restrict__ new_xyz, const int *__restrict__ idx, Bx3Nets NL, float *__restrict__ out) {
    constexpr int P = 2;
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware workgroup -> neighbourhood map (sa_fused.hip): workgroups go round-robin to the 8 XCDs, a source row is gathered by ~16
    // neighbourhoods of ITS cloud: XCD x takes the clouds x, x + 8, ... whole, so its L2 only ever holds 1 / 8 of the clouds' rows
    // (PMC, feature level: 90 MB per launch with the identity map against the f32 kernel's 18)
    long wg = blockIdx.x;
    {
        const long wpc = m / 4, clouds = groups / m;           // workgroups per cloud (four neighbourhoods each)
        if ((clouds & 7) == 0 && wpc * 4 == m) {
            const long xcd = wg & 7, j = wg >> 3;
            wg = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
    const long g = wg * 4 + wave;
    if (g >= groups) return;                                   // no barrier anywhere: a wave may simply leave
    const int net = (int)(g / geo_groups);
    const long gg = g - (long)net * geo_groups;                // the neighbourhood in the shared geometry
    const long b = gg / m;                                     // its cloud
    const long bn = (geo_groups / m) * net + b;                // the cloud's row block in the network-major arrays
    const Bx3Layer &L1 = NL.L[net][0], &L2 = NL.L[net][1], &L3 = NL.L[net][2];
    BxFrag X0[P][1][S::NP];
    const float *init[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int ii = idx[gg * 64 + 32 * p + l31];
        const float *pt = xyz + ((size_t)b * n + ii) * 3, *c = new_xyz + (size_t)gg * 3;
        const float dx = pt[0] - c[0], dy = pt[1] - c[1], dz = pt[2] - c[2];
        u32 s01[S::NP], s2[S::NP];
        S::split2(dx, dy, s01);
        S::split2(dz, 0.f, s2);
        // channels 0..2 live in the lower lanes' elements 0..2; everything else of the 16-channel block is zero
#pragma unroll
        for (int pl = 0; pl < S::NP; ++pl) X0[p][0][pl] = BxFrag{{khalf ? 0u : s01[pl], khalf ? 0u : s2[pl], 0u, 0u}};
        init[p] = PARTIAL ? partial + ((size_t)bn * n + ii) * C1 : nullptr;
    }
    BxFrag X1[P][C1 / 16][S::NP], X2[P][C2 / 16][S::NP];
    const float *const none[P] = {nullptr, nullptr};
#ifndef SA_SPLIT16_FIRST_KOUTER
#define SA_SPLIT16_FIRST_KOUTER 1       /* 0: the first layer in bx3_hidden's tile-by-tile order (tools/experiments) */
#endif
    if (SA_SPLIT16_FIRST_KOUTER && (WAVES == 1 || S::NACC * (C1 / 32) <= 4)) bx3_first_kouter<S, C1, P>(L1, X0, X1, init);
    else bx3_hidden<S, 1, C1, P>(L1, X0, X1, init);
    float pm[C3 / 32];
#ifndef SA_SPLIT16_KOUTER_2W
#define SA_SPLIT16_KOUTER_2W 0          /* tools/experiments: k-block-outer layers for the two-waves-per-SIMD level too */
#endif
    if (WAVES == 1 || SA_SPLIT16_KOUTER_2W) {
        // one wave per SIMD: nothing hides the weight stream's L2 latency unless it is double-buffered under the MFMAs (k-block loop
        // outside; bx3.h): 332 -> 295 us for the bf16x3 feature level.  With two waves per SIMD the other wave already hides it and the
        // longer live ranges cost more than they buy (255 -> 311 us measured for the feature-less level): output-tile-outer order.
        bx3_hidden_kouter<S, C1 / 16, C2, P>(L2, X1, X2);
        bx3_pooled_kouter<S, C2 / 16, C3, P, (C3 / 32 >= 8 ? 4 : 2)>(L3, X2, pm);
    } else {
        bx3_hidden<S, C1 / 16, C2, P>(L2, X1, X2, none);
        bx3_pooled<S, C2 / 16, C3, P>(L3, X2, pm);
    }
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < C3 / 32; ++j) out[(size_t)g * C3 + j * 32 + lane] = pm[j];
    }
}

static int bx3_reg_layers(const float *const *params, Bx3Layer (&L)[3], const char *who) {
    ANCSH_REQUIRE(params, "%s: null parameter table", who);
    for (int i = 0; i < 3; ++i) {
        L[i].w = reinterpret_cast<const uint4 *>(params[4 * i]);
        L[i].bias = params[4 * i + 1]; L[i].scale = params[4 * i + 2]; L[i].shift = params[4 * i + 3];
        ANCSH_REQUIRE(L[i].w && L[i].bias && L[i].scale && L[i].shift, "%s: null layer parameter", who);
        ANCSH_REQUIRE(((((uintptr_t)L[i].w) | (uintptr_t)L[i].bias | (uintptr_t)L[i].scale | (uintptr_t)L[i].shift) & 15) == 0,
                      "%s: parameters must be 16-byte aligned", who);
    }
    return ANCSH_OK;
}

// packed[(((kb * TN + j) * NP + plane) * 64 + lane) * 8 + e] = plane(W[kb*16 + 8*(lane>>5) + e][j*32 + (lane&31)]), 0 past row k-1
template <class S>
__global__ __launch_bounds__(256) void sa_pack_split16_kernel(int k, int n, const float *__restrict__ w, unsigned short *__restrict__ packed,
                                                              long frags) {
    const long f = (long)blockIdx.x * 256 + threadIdx.x;          // one (kb, j, lane, e)
    if (f >= frags) return;
    const int tn = n / 32;
    const int e = (int)(f & 7), lane = (int)((f >> 3) & 63);
    const long kj = f >> 9;
    const int j = (int)(kj % tn), kb = (int)(kj / tn);
    const int kk = kb * 16 + 8 * (lane >> 5) + e, col = j * 32 + (lane & 31);
    unsigned short pl[S::NP];
    S::split1(kk < k ? w[(size_t)kk * n + col] : 0.f, pl);
    const size_t base = ((size_t)(kb * tn + j) * S::NP * 64 + lane) * 8 + e;
#pragma unroll
    for (int q = 0; q < S::NP; ++q) packed[base + (size_t)q * 64 * 8] = pl[q];
}

template <class S>
static long split16_packed_bytes(int k, int n) { return (long)((k + 15) / 16) * (n / 32) * S::NP * 64 * 16; }

template <class S>
static int pack_split16(const char *who, int k, int n, const float *w, void *packed, void *stream) {
    ANCSH_REQUIRE(k > 0 && n > 0 && n % 32 == 0, "%s: k=%d must be positive, n=%d a positive multiple of 32", who, k, n);
    ANCSH_REQUIRE(w && packed, "%s: null pointer", who);
    const long frags = (long)((k + 15) / 16) * (n / 32) * 64 * 8;
    hipLaunchKernelGGL(sa_pack_split16_kernel<S>, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k, n, w,
                       reinterpret_cast<unsigned short *>(packed), frags);
    return check_launch(who);
}

static int bx3_nets(int ngroups, const float *const *params, Bx3Nets &NL, const char *who) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= ANCSH_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, ANCSH_MAX_GROUPS);
    ANCSH_REQUIRE(params, "%s: null parameter table", who);
    for (int g = 0; g < ANCSH_MAX_GROUPS; ++g)
        if (int rc = bx3_reg_layers(params + 12 * (g < ngroups ? g : 0), NL.L[g], who)) return rc;
    return ANCSH_OK;
}

template <class S>
static int sa_split16(const char *who, int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                      const float *new_xyz, const int *idx, const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0, "%s: bad shape b=%d n=%d m=%d", who, b, n, m);
    ANCSH_REQUIRE(nsample == 64, "%s: nsample must be 64 (got %d)", who, nsample);
    ANCSH_REQUIRE(cfeat == 0 && c1 == 64 && c2 == 64 && c3 == 128, "%s: unsupported layer shape (cfeat=%d mlp=[%d,%d,%d]); "
                  "a level with input features goes through the _partial entry point", who, cfeat, c1, c2, c3);
    Bx3Nets NL;
    if (int rc = bx3_nets(ngroups, params, NL, who)) return rc;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && new_xyz && idx && out, "%s: null pointer", who);
    const long geo = (long)b * m, groups = geo * ngroups;
#ifndef SA_SPLIT16_SA1_WAVES
#define SA_SPLIT16_SA1_WAVES 2
#endif
    hipLaunchKernelGGL((sa_bf16x3_reg_kernel<S, 64, 64, 128, false, SA_SPLIT16_SA1_WAVES>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n, m,
                       groups, geo, xyz, (const float *)nullptr, new_xyz, idx, NL, out);
    return check_launch(who);
}

template <class S>
static int sa_partial_split16(const char *who, int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                              const float *partial, const float *new_xyz, const int *idx, const float *const *params, float *out, void *stream) {
    ANCSH_REQUIRE(b >= 0 && n > 0 && m > 0, "%s: bad shape b=%d n=%d m=%d", who, b, n, m);
    ANCSH_REQUIRE(nsample == 64, "%s: nsample must be 64 (got %d)", who, nsample);
    ANCSH_REQUIRE(c1 == 128 && c2 == 128 && c3 == 256, "%s: unsupported layer shape (mlp=[%d,%d,%d])", who, c1, c2, c3);
    Bx3Nets NL;
    if (int rc = bx3_nets(ngroups, params, NL, who)) return rc;
    if (b == 0) return ANCSH_OK;
    ANCSH_REQUIRE(xyz && partial && new_xyz && idx && out, "%s: null pointer", who);
    ANCSH_REQUIRE((((uintptr_t)partial) & 15) == 0, "%s: partial must be 16-byte aligned", who);
    const long geo = (long)b * m, groups = geo * ngroups;
    hipLaunchKernelGGL((sa_bf16x3_reg_kernel<S, 128, 128, 256, true, 1>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n, m,
                       groups, geo, xyz, partial, new_xyz, idx, NL, out);
    return check_launch(who);
}

}  // namespace ancsh

using namespace ancsh;

extern "C" long ancsh_sa_packed_weight_bytes_bf16x3(int k, int n) { return (k <= 0 || n <= 0 || n % 32 != 0) ? -1 : split16_packed_bytes<Bf16x3>(k, n); }
extern "C" long ancsh_sa_packed_weight_bytes_f16x2(int k, int n) { return (k <= 0 || n <= 0 || n % 32 != 0) ? -1 : split16_packed_bytes<F16x2>(k, n); }

extern "C" int ancsh_sa_pack_weights_bf16x3(int k, int n, const float *w, void *packed, void *stream) {
    return pack_split16<Bf16x3>("sa_pack_weights_bf16x3", k, n, w, packed, stream);
}
extern "C" int ancsh_sa_pack_weights_f16x2(int k, int n, const float *w, void *packed, void *stream) {
    return pack_split16<F16x2>("sa_pack_weights_f16x2", k, n, w, packed, stream);
}

// `ngroups` networks on the SAME b clouds in one launch (like ancsh_sa_module_fused_grouped): geometry shared, params = 12 pointers per
// network, out (ngroups * b, m, c3) network-major.
extern "C" int ancsh_sa_module_fused_bf16x3_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                                    const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                                    const float *const *params, float *out, void *stream) {
    return sa_split16<Bf16x3>("sa_module_fused_bf16x3", ngroups, b, n, m, nsample, cfeat, c1, c2, c3, xyz, new_xyz, idx, params, out, stream);
}
extern "C" int ancsh_sa_module_fused_f16x2_grouped(int ngroups, int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3,
                                                   const float *xyz, const float *feats, const float *new_xyz, const int *idx,
                                                   const float *const *params, float *out, void *stream) {
    return sa_split16<F16x2>("sa_module_fused_f16x2", ngroups, b, n, m, nsample, cfeat, c1, c2, c3, xyz, new_xyz, idx, params, out, stream);
}
extern "C" int ancsh_sa_module_fused_bf16x3(int b, int n, int m, int nsample, int cfeat, int c1, int c2, int c3, const float *xyz,
                                            const float *feats, const float *new_xyz, const int *idx, const float *const *params,
                                            float *out, void *stream) {
    return ancsh_sa_module_fused_bf16x3_grouped(1, b, n, m, nsample, cfeat, c1, c2, c3, xyz, feats, new_xyz, idx, params, out, stream);
}

// A level WITH input features, like ancsh_sa_module_fused_partial: `partial` (ngroups * b, n, c1) = the first layer's raw f32 partial sums
// over the feature channels per source point; params[12 g] = the packed kernel rows 0..2 of network g's first layer.
extern "C" int ancsh_sa_module_fused_partial_bf16x3_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                                            const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                                            const float *const *params, float *out, void *stream) {
    return sa_partial_split16<Bf16x3>("sa_module_fused_partial_bf16x3", ngroups, b, n, m, nsample, c1, c2, c3, xyz, partial, new_xyz, idx, params, out, stream);
}
extern "C" int ancsh_sa_module_fused_partial_f16x2_grouped(int ngroups, int b, int n, int m, int nsample, int c1, int c2, int c3,
                                                           const float *xyz, const float *partial, const float *new_xyz, const int *idx,
                                                           const float *const *params, float *out, void *stream) {
    return sa_partial_split16<F16x2>("sa_module_fused_partial_f16x2", ngroups, b, n, m, nsample, c1, c2, c3, xyz, partial, new_xyz, idx, params, out, stream);
}
extern "C" int ancsh_sa_module_fused_partial_bf16x3(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz,
                                                    const float *partial, const float *new_xyz, const int *idx,
                                                    const float *const *params, float *out, void *stream) {
    return ancsh_sa_module_fused_partial_bf16x3_grouped(1, b, n, m, nsample, c1, c2, c3, xyz, partial, new_xyz, idx, params, out, stream);
}
