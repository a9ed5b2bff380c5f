// mlp.hip -- shared per-point MLP layer (1x1 conv + bias + inference BN + ReLU [+ max-pool]) on
// the gfx950 matrix cores, exact float32.
//
// Replaces tf_util.conv1d / conv2d(1x1) + batch_norm_for_conv*d + relu + tf.reduce_max as used
// by pointnet_sa_module / pointnet_fp_module / the ANCSH heads
// (pointnet_plusplus/utils/tf_util.py:52-185,512-531; pointnet_util.py:118-134,228-234).
//
// Design (MI355X): y[rows,cout] = epi(x[rows,cin] . w[cin,cout]) with
//   * v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain, so the
//     result equals the CPU restatement's chain exactly (no TF32/bf16 shortcut; 1e-4 parity
//     through ~14 layers needs f32 and gfx950 has no xf32);
//   * 128-row tiles, 4 waves, wave tile up to 64x64 (4 accumulators -> back-to-back MFMAs on
//     independent accumulators keep the 64-cycle pipe full from one wave per SIMD);
//   * A/B tiles staged k-major in LDS (row/col index contiguous => conflict-free ds_read_b32
//     fragment reads: lane -> [k = lane>>5][i = lane&31]); next tile's global loads are issued
//     before the MFMA block of the current one (register prefetch);
//   * epilogue fused in registers: bias, folded BN (one fmaf), ReLU, and -- for SA layers -- the
//     max over each 64/128-row neighbourhood, so the (rows, cout) activation of the last SA
//     layer never reaches HBM.
#include "common.h"

namespace ancsh {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
#ifndef CONV_SMALL_TILE_BELOW
#define CONV_SMALL_TILE_BELOW 512
#endif

template <int WM, int WN, int TM, int TN, bool VEC_A, bool VEC_B>
__global__ __launch_bounds__(256) void conv1x1_kernel(long rows, int cin, int cout, const float *__restrict__ x, int ldx,
                                                      const float *__restrict__ w, const float *__restrict__ bias,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      int act, float *__restrict__ y, int ldy, int pool,
                                                      const float *__restrict__ acc_init, int init_rows) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float lds[BK * LDA + BK * LDB];
    float *As = lds, *Bs = lds + BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long row0 = (long)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;

    // ---- global -> register staging maps --------------------------------------------------
    constexpr int A_ITEMS = VEC_A ? BM / 64 : BM / 16;   // float4 (4 k) or scalar per thread
    constexpr int B_VEC_PER_ROW = BN / 4;
    constexpr int B_ITEMS_V = (BK * B_VEC_PER_ROW + 255) / 256;
    constexpr int B_ITEMS_S = (BK * BN) / 256;
    float4 ra[VEC_A ? A_ITEMS : 1];
    float rs[VEC_A ? 1 : A_ITEMS];
    float4 rb[VEC_B ? B_ITEMS_V : 1];
    float rbs[VEC_B ? 1 : B_ITEMS_S];

    auto load_tiles = [&](int k0) {
        if constexpr (VEC_A) {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int r = tid / 4 + 64 * i, k = k0 + (tid % 4) * 4;
                const long row = row0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows && k < cin) {
                    const float *p = x + (size_t)row * ldx + k;
                    if (k + 3 < cin) v = *reinterpret_cast<const float4 *>(p);
                    else { v.x = p[0]; if (k + 1 < cin) v.y = p[1]; if (k + 2 < cin) v.z = p[2]; }
                }
                ra[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int r = tid / 16 + 16 * i, k = k0 + tid % 16;
                const long row = row0 + r;
                rs[i] = (row < rows && k < cin) ? x[(size_t)row * ldx + k] : 0.f;
            }
        }
        if constexpr (VEC_B) {
#pragma unroll
            for (int i = 0; i < B_ITEMS_V; ++i) {
                const int e = tid + 256 * i;
                const int k = k0 + e / B_VEC_PER_ROW, c = col0 + (e % B_VEC_PER_ROW) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < BK * B_VEC_PER_ROW && k < cin && c < cout) v = *reinterpret_cast<const float4 *>(w + (size_t)k * cout + c);
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_ITEMS_S; ++i) {
                const int e = tid + 256 * i;
                const int k = k0 + e / BN, c = col0 + e % BN;
                rbs[i] = (k < cin && c < cout) ? w[(size_t)k * cout + c] : 0.f;
            }
        }
    };
    auto store_tiles = [&]() {
        if constexpr (VEC_A) {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int r = tid / 4 + 64 * i, k = (tid % 4) * 4;
                As[(k + 0) * LDA + r] = ra[i].x;
                As[(k + 1) * LDA + r] = ra[i].y;
                As[(k + 2) * LDA + r] = ra[i].z;
                As[(k + 3) * LDA + r] = ra[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int r = tid / 16 + 16 * i, k = tid % 16;
                As[k * LDA + r] = rs[i];
            }
        }
        if constexpr (VEC_B) {
#pragma unroll
            for (int i = 0; i < B_ITEMS_V; ++i) {
                const int e = tid + 256 * i;
                if (e < BK * B_VEC_PER_ROW)
                    *reinterpret_cast<float4 *>(&Bs[(e / B_VEC_PER_ROW) * LDB + (e % B_VEC_PER_ROW) * 4]) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_ITEMS_S; ++i) {
                const int e = tid + 256 * i;
                Bs[(e / BN) * LDB + e % BN] = rbs[i];
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (cin + BK - 1) / BK;
    const int khalf = lane >> 5, l31 = lane & 31;
    if (acc_init) {
        // continue a k-ordered chain started elsewhere: the accumulator of (row, col) starts from
        // acc_init[row / init_rows][col] (the MFMA C operand), e.g. the part of the dot product that is the same for every
        // row of a group (FP module fed by a single interpolation source: pointnet_util.py:218-229 with ndataset2 == 1)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    acc[i][j][r] = (col < cout && row < rows) ? acc_init[(size_t)(row / init_rows) * cout + col] : 0.f;
                }
        }
    }
    const float *Af = As + khalf * LDA + wm * TM * 32 + l31;
    const float *Bf = Bs + khalf * LDB + wn * TN * 32 + l31;

    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        // all BK/2 k-steps of the tile, fully unrolled (rows of the tile past cin are zero in both operands, so the tail adds
        // exact zeros): fragments of k-step s+1 are read while the MFMAs of k-step s issue; sched_barrier keeps that order
        // (a rolled loop read its fragments, waited lgkmcnt(0) and only then issued its MFMAs, every k-step)
        float a[2][TM], bq[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = Af[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[0][j] = Bf[j * 32];
#pragma unroll
        for (int s2 = 0; s2 < BK / 2; ++s2) {
            if (s2 + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(s2 + 1) & 1][i] = Af[(2 * s2 + 2) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bq[(s2 + 1) & 1][j] = Bf[(2 * s2 + 2) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2 & 1][i], bq[s2 & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: bias -> folded BN -> activation (-> max-pool) ---------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + wn * TN * 32 + j * 32 + l31;
        const bool cok = col < cout;
        const bool ep = cok && act != ANCSH_ACT_RAW;                 // raw accumulators: bias / scale / shift may be NULL
        const float bs = ep ? bias[col] : 0.f, sc = ep ? scale[col] : 0.f, sh = ep ? shift[col] : 0.f;
        float pmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                float v = act == ANCSH_ACT_RAW ? acc[i][j][r] : __builtin_fmaf(acc[i][j][r] + bs, sc, sh);
                if (act == ANCSH_ACT_RELU) v = nmax(v, 0.f);
                if (pool == 0) {
                    if (cok && row < rows) y[(size_t)row * ldy + col] = v;
                } else if (row < rows) {
                    pmax = nmax(pmax, v);
                }
            }
        }
        if (pool != 0) {
            // rows of this wave = TM*32 consecutive rows; combine the two lane halves first
            pmax = nmax(pmax, __shfl_xor(pmax, 32, 64));
            if constexpr (TM == 2) {
                if (pool == 64) {
                    const long g = (row0 + wm * 64) / 64;
                    if (cok && khalf == 0 && row0 + wm * 64 < rows) y[(size_t)g * ldy + col] = pmax;
                } else {   // pool == 128 == BM: combine the WM = 2 wave rows through LDS
                    float *red = lds;   // tiles are dead after the final barrier of the k loop
                    if (wm == 1 && khalf == 0) red[wn * TN * 32 + j * 32 + l31] = pmax;
                    __syncthreads();
                    if (wm == 0 && khalf == 0 && cok && row0 < rows)
                        y[(size_t)(row0 / 128) * ldy + col] = nmax(pmax, red[wn * TN * 32 + j * 32 + l31]);
                    __syncthreads();
                }
            }
        }
    }
}

__global__ void group_max_kernel(long groups, int nsample, int c, const float *__restrict__ x, float *__restrict__ y) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= groups * c) return;
    long g = e / c;
    int o = (int)(e - g * c);
    const float *p = x + (size_t)g * nsample * c + o;
    float mx = p[0];
    for (int s = 1; s < nsample; ++s) mx = nmax(mx, p[(size_t)s * c]);
    y[e] = mx;
}

template <int WM, int WN, int TM, int TN>
static void launch_cfg(bool va, bool vb, dim3 grid, hipStream_t st, long rows, int cin, int cout, const float *x, int ldx,
                       const float *w, const float *bias, const float *scale, const float *shift, int act, float *y,
                       int ldy, int pool, const float *acc_init, int init_rows) {
#define ANCSH_GO(VA, VB)                                                                                              \
    hipLaunchKernelGGL((conv1x1_kernel<WM, WN, TM, TN, VA, VB>), grid, dim3(256), 0, st, rows, cin, cout, x, ldx, w, bias, \
                       scale, shift, act, y, ldy, pool, acc_init, init_rows)
    if (va && vb) ANCSH_GO(true, true);
    else if (va) ANCSH_GO(true, false);
    else if (vb) ANCSH_GO(false, true);
    else ANCSH_GO(false, false);
#undef ANCSH_GO
}

// Few rows (<= 64: one row per cloud): a (rows x cin) . (cin x cout) product as one k-ordered fmaf chain per output on
// the vector ALU -- the same arithmetic as the MFMA chain (bit-identical), but 1024 dependent v_fma are ~4x shorter in
// latency than 512 dependent MFMAs and a 32-row problem cannot fill the matrix pipe anyway.  Raw accumulators out.
// rows_per_group: rows [g * rows_per_group, ...) use the kernel G.wp[g] (a grouped launch: the same layer of several networks).
__global__ __launch_bounds__(64) void conv1x1_few_rows_kernel(int cin, int cout, const float *__restrict__ x, int ldx,
                                                              const float *__restrict__ w, float *__restrict__ y, int ldy,
                                                              int rows_per_group, ConvGroups G) {
    const int col = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y;
    if (col >= cout) return;
    if (G.n > 1) w = G.wp[row / rows_per_group];
    const float *xr = x + (size_t)row * ldx;
    const float *wc = w + col;
    float acc = 0.f;
    // latency-bound: one wave streams its 64 columns of w through registers, U k-rows (32 KiB per wave) per round trip,
    // all issued before the first is consumed (sched_barrier: the scheduler would otherwise sink each load to its use);
    // x[row][k] is uniform (scalar loads)
    constexpr int U = 128;
    const int nb = cin / U;
    for (int bt = 0; bt < nb; ++bt) {
        float wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u] = wc[(size_t)(bt * U + u) * cout];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_fmaf(xr[bt * U + u], wv[u], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int k = nb * U; k < cin; ++k) acc = __builtin_fmaf(xr[k], wc[(size_t)k * cout], acc);
    y[(size_t)row * ldy + col] = acc;
}

static int conv1x1_launch(long rows, int cin, int cout, const float *x, int ldx, const float *w, const float *bias,
                          const float *scale, const float *shift, int act, float *y, int ldy, int pool,
                          const float *acc_init, int init_rows, void *stream) {
    ANCSH_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "conv1x1: bad shape rows=%ld cin=%d cout=%d", rows, cin, cout);
    ANCSH_REQUIRE(ldx >= cin && ldy >= cout, "conv1x1: row strides ldx=%d ldy=%d too small for cin=%d cout=%d", ldx, ldy, cin, cout);
    ANCSH_REQUIRE(act == ANCSH_ACT_NONE || act == ANCSH_ACT_RELU || act == ANCSH_ACT_RAW, "conv1x1: unknown activation %d", act);
    ANCSH_REQUIRE(pool == 0 || pool == 64 || pool == 128, "conv1x1: pool must be 0, 64 or 128 (got %d)", pool);
    ANCSH_REQUIRE(pool == 0 || rows % pool == 0, "conv1x1: rows %ld not a multiple of pool %d", rows, pool);
    ANCSH_REQUIRE(!acc_init || init_rows > 0, "conv1x1: acc_init needs init_rows > 0 (got %d)", init_rows);
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE(x && w && y && (act == ANCSH_ACT_RAW || (bias && scale && shift)), "conv1x1: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (act == ANCSH_ACT_RAW && rows <= 64 && pool == 0 && !acc_init) {
        ConvGroups G1{};
        G1.n = 1;
        hipLaunchKernelGGL(conv1x1_few_rows_kernel, dim3((cout + 63) / 64, (unsigned)rows), dim3(64), 0, st, cin, cout, x, ldx, w, y, ldy,
                           (int)rows, G1);
        return check_launch("conv1x1");
    }
    const bool va = (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const bool vb = (cout % 4 == 0) && ((uintptr_t)w % 16 == 0);
    const unsigned gx = (unsigned)((rows + 127) / 128);
    // few rows (SA3 / FP1 / FP2: 4096..16384 rows): 128x128 tiles would leave most of the 256 CUs idle, so
    // drop to 64x64 tiles (one 32x32 accumulator per wave still issues MFMAs back to back: issue = latency = 64)
    const long big_tiles = (long)gx * ((cout + 127) / 128);
    if (pool == 0 && cout >= 64 && big_tiles < CONV_SMALL_TILE_BELOW) {
        launch_cfg<2, 2, 1, 1>(va, vb, dim3((unsigned)((rows + 63) / 64), (cout + 63) / 64), st, rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, acc_init, init_rows);
    } else if (cout > 64) {
        launch_cfg<2, 2, 2, 2>(va, vb, dim3(gx, (cout + 127) / 128), st, rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, acc_init, init_rows);
    } else if (cout > 32 || pool != 0) {
        launch_cfg<2, 2, 2, 1>(va, vb, dim3(gx, (cout + 63) / 64), st, rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, acc_init, init_rows);
    } else {
        launch_cfg<4, 1, 1, 1>(va, vb, dim3(gx, 1), st, rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, acc_init, init_rows);
    }
    return check_launch("conv1x1");
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_conv1x1(long rows, int cin, int cout, const float *x, int ldx, const float *w, const float *bias,
                             const float *scale, const float *shift, int act, float *y, int ldy, int pool,
                             void *stream) {
    return conv1x1_launch(rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, nullptr, 0, stream);
}

extern "C" int ancsh_conv1x1_ex(long rows, int cin, int cout, const float *x, int ldx, const float *w, const float *bias,
                                const float *scale, const float *shift, int act, float *y, int ldy, int pool,
                                const float *acc_init, int init_rows, void *stream) {
    return conv1x1_launch(rows, cin, cout, x, ldx, w, bias, scale, shift, act, y, ldy, pool, acc_init, init_rows, stream);
}

// `ngroups` equal-shaped layers on stacked rows (group g = rows [g * rows, (g + 1) * rows) of x / y, kernel w[g]).  The few-rows
// raw product (one row per cloud: the single-source FP shortcut) is ONE launch; every other shape is issued group by group.
extern "C" int ancsh_conv1x1_grouped(int ngroups, long rows, int cin, int cout, const float *x, int ldx, const float *const *w,
                                     const float *const *bias, const float *const *scale, const float *const *shift, int act, float *y,
                                     int ldy, int pool, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= CONV_MAX_GROUPS, "conv1x1_grouped: ngroups=%d must be in [1,%d]", ngroups, CONV_MAX_GROUPS);
    ANCSH_REQUIRE(w, "conv1x1_grouped: null parameter table");
    ANCSH_REQUIRE(rows >= 0 && cin > 0 && cout > 0 && ldx >= cin && ldy >= cout, "conv1x1_grouped: bad shape");
    if (act == ANCSH_ACT_RAW && rows * ngroups <= 65535 && rows <= 64 && pool == 0) {
        if (rows == 0) return ANCSH_OK;
        ANCSH_REQUIRE(x && y, "conv1x1_grouped: null pointer");
        ConvGroups G{};
        G.n = ngroups;
        for (int g = 0; g < CONV_MAX_GROUPS; ++g) {
            G.wp[g] = w[g < ngroups ? g : 0];
            ANCSH_REQUIRE(G.wp[g], "conv1x1_grouped: null kernel of group %d", g);
        }
        hipLaunchKernelGGL(conv1x1_few_rows_kernel, dim3((cout + 63) / 64, (unsigned)(rows * ngroups)), dim3(64), 0, (hipStream_t)stream, cin,
                           cout, x, ldx, G.wp[0], y, ldy, (int)rows, G);
        return check_launch("conv1x1_grouped");
    }
    const long yrows = pool ? rows / (pool ? pool : 1) : rows;
    for (int g = 0; g < ngroups; ++g)
        if (int rc = conv1x1_launch(rows, cin, cout, x + (size_t)g * rows * ldx, ldx, w[g], bias ? bias[g] : nullptr, scale ? scale[g] : nullptr,
                                    shift ? shift[g] : nullptr, act, y + (size_t)g * yrows * ldy, ldy, pool, nullptr, 0, stream))
            return rc;
    return ANCSH_OK;
}

extern "C" int ancsh_group_max(long groups, int nsample, int c, const float *x, float *y, void *stream) {
    ANCSH_REQUIRE(groups >= 0 && nsample > 0 && c > 0, "group_max: bad shape");
    if (groups == 0) return ANCSH_OK;
    ANCSH_REQUIRE(x && y, "group_max: null pointer");
    long total = groups * c;
    hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, groups,
                       nsample, c, x, y);
    return check_launch("group_max");
}
