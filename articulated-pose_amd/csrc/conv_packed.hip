// conv_packed.hip -- shared per-point MLP layer for WIDE layers (cout a multiple of 128) with wave-independent execution.
//
// Same operator as ancsh_conv1x1 (tf_util.conv1d / conv2d 1x1 + bias + inference BN + ReLU [+ max over the neighbourhood];
// pointnet_plusplus/utils/tf_util.py:52-185, pointnet_util.py:118-134,228-234) and the same arithmetic (one k-ordered f32
// fmaf chain per output on v_mfma_f32_32x32x2_f32 => identical bits), organised like the fused SA kernels instead of a
// workgroup-tiled GEMM:
//   * a WAVE owns 32 rows x (TN x 32) columns: TN accumulators, no other wave ever touches its data, no barrier in the k loop
//     (the workgroup-tiled kernel in mlp.hip spends two __syncthreads per 16 k and reaches 25-50 % of the matrix peak on the
//     4096..16384-row layers of SA3 / FP2 / FP3);
//   * weights in the pre-packed MFMA fragment order of ancsh_sa_pack_weights: one 16-byte load = the B fragments of four
//     k-steps, L2 -> registers, one slot (4 k-steps = 4*TN MFMAs) ahead;
//   * activations: the wave stages its own 32 rows x 16 k through a private 2 x 2.2 KB LDS tile (coalesced 16-B global loads
//     one chunk ahead in registers -> ds_write -> conflict-free ds_read_b32 fragments, odd row stride);
//   * epilogue in registers; pool = 64 / 128 combines the 2 / 4 waves' row maxima through LDS (the only barrier).
#include "common.h"

namespace ancsh {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int CP_KC = 16;            // k per staged chunk (= 2 packed weight slots)
constexpr int CP_LD = CP_KC + 1;     // LDS row stride (odd)

__device__ __forceinline__ void cp_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float cp_f4(const float4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

template <int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void conv_packed_kernel(long rows, int cin, int cout, const float *__restrict__ x, int ldx, const float *__restrict__ wp,
                        const float *__restrict__ bias, const float *__restrict__ scale, const float *__restrict__ shift, int act,
                        float *__restrict__ y, int ldy, int pool, const float *__restrict__ acc_init, int init_rows, ConvGroups G) {
    CONV_SELECT_GROUP(G, x, y, acc_init, wp, bias, scale, shift)
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 32 * CP_LD + 4 * TN * 32];
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *T = lds + wave * (2 * 32 * CP_LD);                 // this wave's two chunk buffers
    const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
    const int ct0 = blockIdx.y * TN;                          // first 32-column tile of this workgroup
    const int tn_all = cout / 32;                             // packed layout: tiles per slot
    const int nch = (cin + CP_KC - 1) / CP_KC;

    floatx16 acc[TN];
    if (acc_init) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            row = row < rows ? row : rows - 1;                    // rows past the end compute a copy of the last row; never stored
            const unsigned grp = (unsigned)row / (unsigned)init_rows;      // rows < 2^31 (checked by the launcher): 32-bit divide
            const float *ip = acc_init + (size_t)grp * cout + ct0 * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j][r] = ip[j * 32];
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }

    // ---- operand streams ---------------------------------------------------------------------------------------------
    // Every load below is UNCONDITIONAL (addresses clamped, values masked afterwards): a load inside a branch makes the
    // s_waitcnt insertion pass lose count at the join and wait for vmcnt(0) before the next MFMA, which exposes a full L2
    // round trip per chunk.
    // A: lane (row = lane>>1, 8 consecutive k = 8*(lane&1) ..) of a 32 x 16 chunk as two 16-byte loads.  Rows past the end
    // re-read the last row (their results are never stored); k >= cin is zeroed after the load (ldx % 4 == 0, so a 16-byte
    // load that starts below ldx stays inside the row's allocation).
    const int arow = lane >> 1, ak = (lane & 1) * 8;
    const long arow_g = row0 + arow < rows ? row0 + arow : rows - 1;
    const float *ag = x + (size_t)arow_g * ldx;
    auto a_fetch = [&](float4 (&v)[2], int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = c * CP_KC + ak + 4 * i;
            const int kc = k < ldx ? k : 0;
            float4 t = *reinterpret_cast<const float4 *>(ag + kc);
            t.x = k < cin ? t.x : 0.f; t.y = k + 1 < cin ? t.y : 0.f; t.z = k + 2 < cin ? t.z : 0.f; t.w = k + 3 < cin ? t.w : 0.f;
            v[i] = t;
        }
    };
    auto a_store = [&](const float4 (&v)[2], int buf) {
        float *d = T + buf * (32 * CP_LD) + arow * CP_LD + ak;
#pragma unroll
        for (int i = 0; i < 2; ++i) { d[4 * i] = v[i].x; d[4 * i + 1] = v[i].y; d[4 * i + 2] = v[i].z; d[4 * i + 3] = v[i].w; }
    };
    // B: packed[((slot*tn_all + tile)*64 + lane)*4 + q]; a slot past the end re-reads the last one (its activations are zero)
    const float4 *bg = reinterpret_cast<const float4 *>(wp) + (size_t)ct0 * 64 + lane;
    const int nslot = ((cin + 1) / 2 + 3) / 4;
    auto b_fetch = [&](float4 (&b)[TN], int slot) {
        const int sl = slot < nslot ? slot : nslot - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = bg[((size_t)sl * tn_all + j) * 64];
    };

    // Register pipeline, statically indexed (the chunk loop advances two chunks = four weight slots per trip):
    //   weights: ring of 4 slots, fetched PFB slots ahead (PFB = 1 when a slot is 32 MFMAs, 3 when it is only 8..16);
    //   activations: chunk c is in LDS buffer c&1, chunks c+1 and c+2 are in flight / in registers (two register sets).
    // A trailing odd chunk runs one all-zero chunk (operands beyond cin are zero): exact, at most 1/nch extra MFMAs.
    constexpr int PFB = TN >= 8 ? 1 : 3;
    float4 areg[2][2], bw[4][TN];
    a_fetch(areg[0], 0);
#pragma unroll
    for (int sl = 0; sl < PFB; ++sl) b_fetch(bw[sl], sl);
    a_store(areg[0], 0);
    a_fetch(areg[1], 1);
    a_fetch(areg[0], 2);
    auto half_chunk = [&](const float *Af, int slot_ring, int slot, int s0) {
        // four k-steps s0..s0+3 of the chunk at Af with the weights of ring slot `slot_ring`; prefetches slot + PFB
        b_fetch(bw[(slot_ring + PFB) & 3], slot + PFB);
        __builtin_amdgcn_sched_barrier(0);
        float a_cur = Af[2 * s0], a_nxt = Af[2 * s0 + 2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, cp_f4(bw[slot_ring][j], s), acc[j], 0, 0, 0);
            a_cur = a_nxt;
            if (s + 2 < 4) a_nxt = Af[2 * (s0 + s + 2)];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c < nch; c += 2) {
        // ---- even chunk c (LDS buffer 0); areg[1] = chunk c+1, areg[0] = chunk c+2 ----
        const float *Af0 = T + l31 * CP_LD + khalf;
        cp_wave_fence();
        half_chunk(Af0, 0, 2 * c, 0);
        a_store(areg[1], 1);                      // chunk c+1 -> buffer 1 (its last reader was chunk c-1)
        a_fetch(areg[1], c + 3);
        half_chunk(Af0, 1, 2 * c + 1, 4);
        // ---- odd chunk c+1 (LDS buffer 1); areg[0] = chunk c+2, areg[1] = chunk c+3 ----
        const float *Af1 = Af0 + 32 * CP_LD;
        cp_wave_fence();
        half_chunk(Af1, 2, 2 * c + 2, 0);
        a_store(areg[0], 0);                      // chunk c+2 -> buffer 0
        a_fetch(areg[0], c + 4);
        half_chunk(Af1, 3, 2 * c + 3, 4);
    }

    // ---- epilogue: bias -> folded BN -> activation (-> max over the neighbourhood) ----------------------------------------
    // RAW is the same expression with bias 0, scale 1, shift 0 (acc + 0 and fmaf(acc, 1, 0) are exact; a k-ordered chain that
    // starts from +0 never yields -0), "no activation" clamps at -inf: one straight-line body for every mode
    const bool raw = act == ANCSH_ACT_RAW;
    const float lo = act == ANCSH_ACT_RELU ? 0.f : -INFINITY;
    float *red = lds + 4 * 2 * 32 * CP_LD;       // [4 waves][TN*32] row maxima
    if (pool == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = (ct0 + j) * 32 + l31;
            const float bs = raw ? 0.f : bias[col], sc = raw ? 1.f : scale[col], sh = raw ? 0.f : shift[col];
            float *yc = y + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const float v = nmax(__builtin_fmaf(acc[j][r] + bs, sc, sh), lo);
                if (row < rows) yc[(size_t)row * ldy] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = (ct0 + j) * 32 + l31;
        const float bs = raw ? 0.f : bias[col], sc = raw ? 1.f : scale[col], sh = raw ? 0.f : shift[col];
        float pmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) pmax = nmax(pmax, nmax(__builtin_fmaf(acc[j][r] + bs, sc, sh), lo));   // rows % pool == 0: all rows exist
        pmax = nmax(pmax, __shfl_xor(pmax, 32, 64));
        if (khalf == 0) red[wave * (TN * 32) + j * 32 + l31] = pmax;
    }
    __syncthreads();
    const int wpg = pool / 32;                               // waves per neighbourhood: 2 or 4
    if (wave % wpg == 0 && khalf == 0 && row0 < rows) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float m = red[wave * (TN * 32) + j * 32 + l31];
            for (int w = 1; w < wpg; ++w) m = nmax(m, red[(wave + w) * (TN * 32) + j * 32 + l31]);
            y[(size_t)(row0 / pool) * ldy + (ct0 + j) * 32 + l31] = m;
        }
    }
}

}  // namespace ancsh

using namespace ancsh;

static int conv_packed_launch(const char *who, int ngroups, long rows, int cin, int cout, const float *x, int ldx, const float *const *w_packed,
                              const float *const *bias, const float *const *scale, const float *const *shift, int act, float *y, int ldy,
                              int pool, const float *acc_init, int init_rows, void *stream) {
    ANCSH_REQUIRE(ngroups >= 1 && ngroups <= CONV_MAX_GROUPS, "%s: ngroups=%d must be in [1,%d]", who, ngroups, CONV_MAX_GROUPS);
    ANCSH_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "%s: bad shape rows=%ld cin=%d cout=%d", who, rows, cin, cout);
    ANCSH_REQUIRE(cout % 64 == 0, "%s: cout %d is not a multiple of 64 (use ancsh_conv1x1)", who, cout);
    ANCSH_REQUIRE(ldx >= cin && ldy >= cout, "%s: row strides ldx=%d ldy=%d too small for cin=%d cout=%d", who, ldx, ldy, cin, cout);
    ANCSH_REQUIRE(act == ANCSH_ACT_NONE || act == ANCSH_ACT_RELU || act == ANCSH_ACT_RAW, "%s: unknown activation %d", who, act);
    ANCSH_REQUIRE(pool == 0 || pool == 64 || pool == 128, "%s: pool must be 0, 64 or 128 (got %d)", who, pool);
    ANCSH_REQUIRE(pool == 0 || rows % pool == 0, "%s: rows %ld not a multiple of pool %d", who, rows, pool);
    ANCSH_REQUIRE(!acc_init || init_rows > 0, "%s: acc_init needs init_rows > 0 (got %d)", who, init_rows);
    ANCSH_REQUIRE(!acc_init || ngroups == 1 || rows % init_rows == 0, "%s: rows %ld per group not a multiple of init_rows %d", who, rows, init_rows);
    ANCSH_REQUIRE(rows < (1L << 31), "%s: rows %ld >= 2^31", who, rows);
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE(x && w_packed && y && (act == ANCSH_ACT_RAW || (bias && scale && shift)), "%s: null pointer", who);
    ConvGroups G{};
    G.n = ngroups;
    G.x_stride = rows * (long)ldx;
    G.y_stride = (pool ? rows / pool : rows) * (long)ldy;
    G.init_stride = acc_init ? (rows / init_rows) * (long)cout : 0;
    for (int g = 0; g < CONV_MAX_GROUPS; ++g) {
        const int s = g < ngroups ? g : 0;
        ANCSH_REQUIRE(w_packed[s] && (act == ANCSH_ACT_RAW || (bias[s] && scale[s] && shift[s])), "%s: null parameter pointer of group %d", who, s);
        G.wp[g] = w_packed[s];
        G.bias[g] = act == ANCSH_ACT_RAW ? nullptr : bias[s]; G.scale[g] = act == ANCSH_ACT_RAW ? nullptr : scale[s]; G.shift[g] = act == ANCSH_ACT_RAW ? nullptr : shift[s];
    }
    hipStream_t st = (hipStream_t)stream;
    // the backbone's small layers (128 / 256 / 259 / 384 input channels, no pooling): whole input tile in LDS, see conv_rowtile.hip
    if (pool == 0 && conv_rowtile_launch(rows, cin, cout, x, ldx, G.wp[0], G.bias[0], G.scale[0], G.shift[0], act, y, ldy, acc_init, init_rows, G, st))
        return check_launch(who);
    ANCSH_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x % 16) == 0, "%s: x must be 16-byte aligned with ldx %% 4 == 0 (ldx=%d)", who, ldx);
    const unsigned gx = (unsigned)((rows + 127) / 128);
    // column tiles per wave: the widest that still gives ~2 waves per SIMD (2048 waves); narrow problems take TN = 2 so that
    // a launch is not a single round of long serial k loops
    const long row_waves = (rows + 31) / 32 * ngroups;
    int tn = 2;
    if (cout % 256 == 0 && row_waves * (cout / 256) >= 2048) tn = 8;
    else if (cout % 128 == 0 && row_waves * (cout / 128) >= 2048) tn = 4;
#define ANCSH_CP_GO(TNV)                                                                                                        \
    hipLaunchKernelGGL(conv_packed_kernel<TNV>, dim3(gx, cout / (32 * TNV), ngroups), dim3(256), 0, st, rows, cin, cout, x, ldx, G.wp[0], G.bias[0], \
                       G.scale[0], G.shift[0], act, y, ldy, pool, acc_init, init_rows, G)
    if (tn == 8) ANCSH_CP_GO(8);
    else if (tn == 4) ANCSH_CP_GO(4);
    else ANCSH_CP_GO(2);
#undef ANCSH_CP_GO
    return check_launch(who);
}

extern "C" int ancsh_conv1x1_packed(long rows, int cin, int cout, const float *x, int ldx, const float *w_packed,
                                    const float *bias, const float *scale, const float *shift, int act, float *y, int ldy,
                                    int pool, const float *acc_init, int init_rows, void *stream) {
    return conv_packed_launch("conv1x1_packed", 1, rows, cin, cout, x, ldx, &w_packed, &bias, &scale, &shift, act, y, ldy, pool, acc_init,
                              init_rows, stream);
}

extern "C" int ancsh_conv1x1_packed_grouped(int ngroups, long rows, int cin, int cout, const float *x, int ldx,
                                            const float *const *w_packed, const float *const *bias, const float *const *scale,
                                            const float *const *shift, int act, float *y, int ldy, int pool, const float *acc_init,
                                            int init_rows, void *stream) {
    ANCSH_REQUIRE(w_packed && (act == ANCSH_ACT_RAW || (bias && scale && shift)), "conv1x1_packed_grouped: null parameter table");
    static const float *const none[CONV_MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    const bool raw = act == ANCSH_ACT_RAW;
    return conv_packed_launch("conv1x1_packed_grouped", ngroups, rows, cin, cout, x, ldx, w_packed, raw ? none : bias, raw ? none : scale,
                              raw ? none : shift, act, y, ldy, pool, acc_init, init_rows, stream);
}
