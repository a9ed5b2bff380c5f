// sa_regchain.h -- the first set-abstraction level (3 -> C1 -> C2 -> C3 on 64-row neighbourhoods, no input features) with the activations
// of its hidden layers in REGISTERS (round 5).  Included by sa_fused.hip inside namespace ancsh, after SaGroupLayers and the SA_STAMP
// macros.
//
// v_mfma_f32_32x32x2_f32 leaves its 32x32 output tile with the COLUMN index across the lanes and the ROW index in the 16 registers x 2
// lane halves; both of its input operands want their non-k index across the lanes and k = 2s + (lane half).  sa_body (the general
// kernel, still SA2's) therefore writes every layer's output to an LDS tile and reads it back transposed.  Here a hidden layer is
// computed TRANSPOSED -- weights as the A operand, activations as B (the packed weight layout serves both roles unchanged; a product
// a*b is the same bits either way round, the k order is the same) -- so its output tile has the neighbourhood's ROW across the lanes,
// which is exactly what the next layer's activation operand needs, and the CHANNEL in the register index: register r of lane half h
// holds channel 32 t + (r & 3) + 8 (r >> 2) + 4 h.  The next layer's k-step s wants channel 2s in the lower half and 2s + 1 in the
// upper: two v_permlane32_swap_b32 per 8 channels put them there, after which k-steps 0..3 of the 8-channel group q read registers
// 4q + {0, 2, 1, 3}.  The last layer runs in the standard orientation (activations as A) so that the rows it pools over are in the
// registers.  No LDS tile, no fence between the layers, no LDS reads inside the k loops; LDS only holds a 1.5 KB copy of the hidden
// layers' bias / scale / shift per wave (their channel is register-indexed now: read as one float4 per 4 channels of the lane's half).
// Bit-identical to sa_body (same fmaf chains): 410 -> 400 us per launch of two networks x 32 clouds (0.814 -> 0.833 of the f32 matrix
// peak); the same scheme for SA2 (32-row tiles, one weight fragment per MFMA, 238-256 VGPRs) was built, is bit-identical too and is
// SLOWER (421-475 vs 401 us): sa_body keeps that level.  DESIGN.md section 8.
#pragma once

typedef unsigned rc_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rc_swap(float &x, float &y) {     // x.upper <-> y.lower
    const rc_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r.x); y = __uint_as_float(r.y);
}

// bias + folded BN + ReLU on a TRANSPOSED tile (register r of lane half h = channel tile*32 + (r&3) + 8(r>>2) + 4h), then the swaps
template <int NT>
__device__ __forceinline__ void rc_epilogue(floatx16 (&a)[NT], const float *__restrict__ Lc, int C, int tile, int khalf) {
    typedef float ep_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = tile * 32 + 8 * q + 4 * khalf;
        const float4 b4 = *reinterpret_cast<const float4 *>(Lc + c0);
        const float4 s4 = *reinterpret_cast<const float4 *>(Lc + C + c0);
        const float4 t4 = *reinterpret_cast<const float4 *>(Lc + 2 * C + c0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ep_f2 v0 = {a[nt][4 * q], a[nt][4 * q + 1]}, v1 = {a[nt][4 * q + 2], a[nt][4 * q + 3]};
            v0 = __builtin_elementwise_fma(v0 + ep_f2{b4.x, b4.y}, ep_f2{s4.x, s4.y}, ep_f2{t4.x, t4.y});
            v1 = __builtin_elementwise_fma(v1 + ep_f2{b4.z, b4.w}, ep_f2{s4.z, s4.w}, ep_f2{t4.z, t4.w});
            float e0 = fmaxf(v0.x, 0.f), e1 = fmaxf(v0.y, 0.f), e2 = fmaxf(v1.x, 0.f), e3 = fmaxf(v1.y, 0.f);
            rc_swap(e0, e1);              // e0 = (ch 0 | ch 1), e1 = (ch 4 | ch 5)
            rc_swap(e2, e3);              // e2 = (ch 2 | ch 3), e3 = (ch 6 | ch 7)
            a[nt][4 * q] = e0; a[nt][4 * q + 1] = e1; a[nt][4 * q + 2] = e2; a[nt][4 * q + 3] = e3;
        }
    }
}

// k-step ks (0..3) of the 8-channel group q of a swapped tile lives in register 4q + {0, 2, 1, 3}[ks]
__device__ __forceinline__ constexpr int rc_reg(int q, int ks) { return 4 * q + (ks == 0 ? 0 : ks == 1 ? 2 : ks == 2 ? 1 : 3); }

template <int C1, int C2, int C3>
__device__ __forceinline__ void sa1_rc_body(int n, int m, long groups, int bgeo, const float *__restrict__ xyz,
                                            const float *__restrict__ new_xyz, const int *__restrict__ idx, const SaGroupLayers &GL,
                                            float *__restrict__ out) {
    constexpr int NT = 2;                                  // 32-row tiles of the neighbourhood
    constexpr int T1 = C1 / 32, T2 = C2 / 32, T3 = C3 / 32;
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long wg = blockIdx.x;
    {
        const long wpc = m / 4, clouds = groups / m;
        if ((clouds & 7) == 0 && wpc * 4 == m) {
            const long xcd = wg & 7, j = wg >> 3;
            wg = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
    const long g = wg * 4 + wave;
    const bool live = g < groups;
    const long cloud = (live ? g : groups - 1) / m;
    const int grp = (int)(cloud / bgeo);
    const long cg = cloud - (long)grp * bgeo;
    const long gg = cg * m + ((live ? g : groups - 1) - cloud * m);
    const SaLayer L1 = GL.L[grp][0], L2 = GL.L[grp][1], L3 = GL.L[grp][2];
#ifdef SA_STAMPS
    const unsigned long long t0_ = __builtin_readcyclecounter();
    unsigned st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const float4 *W1 = reinterpret_cast<const float4 *>(L1.w) + lane;
    const float4 *W2 = reinterpret_cast<const float4 *>(L2.w) + lane;
    const float4 *W3 = reinterpret_cast<const float4 *>(L3.w) + lane;
    // the hidden layers' bias / scale / shift: a per-wave LDS copy (read as float4 per lane half in the epilogues)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Lc1 = smem + wave * (3 * (C1 + C2)), *Lc2 = Lc1 + 3 * C1;
    for (int e = lane; e < C1; e += 64) { Lc1[e] = L1.bias[e]; Lc1[C1 + e] = L1.scale[e]; Lc1[2 * C1 + e] = L1.shift[e]; }
    for (int e = lane; e < C2; e += 64) { Lc2[e] = L2.bias[e]; Lc2[C2 + e] = L2.scale[e]; Lc2[2 * C2 + e] = L2.shift[e]; }
    // ---- layer 1 (K = 3: k-steps (0, 1) and (2, pad)), transposed ----
    float4 w1[T1];
#pragma unroll
    for (int j = 0; j < T1; ++j) w1[j] = W1[(size_t)j * 64];
    float bx[NT][2];
    const float *c = new_xyz + (size_t)gg * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ii = idx[gg * 64 + nt * 32 + l31];
        const float *p = xyz + ((size_t)cg * n + ii) * 3;
        const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
        bx[nt][0] = khalf ? dy : dx;
        bx[nt][1] = khalf ? 0.f : dz;
    }
    float4 w2[3][T2];                                        // layer 2's weights: slot ring of three (two ahead)
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int j = 0; j < T2; ++j) w2[s0][j] = W2[(size_t)(s0 * T2 + j) * 64];
    SA_STAMP(0);
    floatx16 a1[T1][NT];
#pragma unroll
    for (int j = 0; j < T1; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[j][nt][r] = 0.f;
            a1[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j].x, bx[nt][0], a1[j][nt], 0, 0, 0);
            a1[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j].y, bx[nt][1], a1[j][nt], 0, 0, 0);
        }
    SA_STAMP(1);
    wave_lds_fence();                                        // the wave's copy of the constants is complete
#pragma unroll
    for (int j = 0; j < T1; ++j) rc_epilogue<NT>(a1[j], Lc1, C1, j, khalf);
    SA_STAMP(2);
    // ---- layer 2 (K = C1), transposed ----
    constexpr int NS2 = C1 / 8;                              // slots of four k-steps
    floatx16 a2[T2][NT];
#pragma unroll
    for (int j = 0; j < T2; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[j][nt][r] = 0.f;
#pragma unroll
    for (int slot = 0; slot < NS2; ++slot) {
        if (slot + 2 < NS2) {
#pragma unroll
            for (int j = 0; j < T2; ++j) w2[(slot + 2) % 3][j] = W2[(size_t)((slot + 2) * T2 + j) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < T2; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    a2[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(w2[slot % 3][j], ks), a1[slot / 4][nt][rc_reg(slot % 4, ks)], a2[j][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    SA_STAMP(3);
    float4 w3p[T3];                                          // layer 3's first weights fly under layer 2's epilogue
#pragma unroll
    for (int j = 0; j < T3; ++j) w3p[j] = W3[(size_t)j * 64];
#pragma unroll
    for (int j = 0; j < T2; ++j) rc_epilogue<NT>(a2[j], Lc2, C2, j, khalf);
    SA_STAMP(4);
    // ---- layer 3 (K = C2), standard orientation: the rows it pools over are in the registers ----
    constexpr int NS3 = C2 / 8;
    float4 w3[2][T3];                                        // 32 MFMAs per slot: one slot ahead is 2048 clocks
#pragma unroll
    for (int j = 0; j < T3; ++j) w3[0][j] = w3p[j];
    floatx16 a3[NT][T3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < T3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a3[nt][j][r] = 0.f;
    float ep3[3][T3], pm[T3];
#pragma unroll
    for (int slot = 0; slot < NS3; ++slot) {
        if (slot + 1 < NS3) {
#pragma unroll
            for (int j = 0; j < T3; ++j) w3[(slot + 1) & 1][j] = W3[(size_t)((slot + 1) * T3 + j) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < T3; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    a3[nt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[slot / 4][nt][rc_reg(slot % 4, ks)], f4_get(w3[slot & 1][j], ks), a3[nt][j], 0, 0, 0);
        if (slot == NS3 - 2) ep_load<C3>(L3, ep3);
        __builtin_amdgcn_sched_barrier(0);
    }
    SA_STAMP(5);
    epilogue<C3, C3 + 1, true, NT>(nullptr, a3, ep3, pm);    // the pooled epilogue of wave_mlp.h: it never touches its tile pointer
    SA_STAMP(6);
    if (lane < 32 && live) {
#pragma unroll
        for (int j = 0; j < T3; ++j) out[(size_t)g * C3 + j * 32 + lane] = pm[j];
    }
#ifdef SA_STAMPS
    sa_write_stamps(out + (size_t)g * C3, t0_, st_, live && lane == 0);
#endif
}
