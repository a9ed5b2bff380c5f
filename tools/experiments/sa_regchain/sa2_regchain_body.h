// SA2 on per-point partial sums of its first layer (C1 values per source point): two waves per neighbourhood (32 rows each); the partial
// sums are loaded straight into the transposed accumulator layout (a float4 per 4 channels of the lane's half), the coordinates continue
// the chain, layer 3 runs in RC_H3B passes over its column tiles, the two waves' maxima meet in LDS.
#ifndef RC_H3B
#define RC_H3B 2
#endif
template <int C1, int C2, int C3>
__device__ __forceinline__ void sa2_rc_body(int n, int m, long groups, int bgeo, const float *__restrict__ xyz,
                                            const float *__restrict__ partial, const float *__restrict__ new_xyz,
                                            const int *__restrict__ idx, const SaGroupLayers &GL, float *__restrict__ out) {
    constexpr int NT = 1;
    constexpr int T1 = C1 / 32, T2 = C2 / 32, T3 = C3 / 32;
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long wg = blockIdx.x;
    {
        const long wpc = m / 2, clouds = groups / m;
        if ((clouds & 7) == 0 && wpc * 2 == m) {
            const long xcd = wg & 7, j = wg >> 3;
            wg = (xcd + 8 * (j / wpc)) * wpc + j % wpc;
        }
    }
    const long g = wg * 2 + (wave >> 1);
    const int half = wave & 1;
    const bool live = g < groups;
    const long cloud = (live ? g : groups - 1) / m;
    const int grp = (int)(cloud / bgeo);
    const long cg = cloud - (long)grp * bgeo;
    const long gg = cg * m + ((live ? g : groups - 1) - cloud * m);
    const SaLayer L1 = GL.L[grp][0], L2 = GL.L[grp][1], L3 = GL.L[grp][2];
    const float4 *W1 = reinterpret_cast<const float4 *>(L1.w) + lane;
    const float4 *W2 = reinterpret_cast<const float4 *>(L2.w) + lane;
    const float4 *W3 = reinterpret_cast<const float4 *>(L3.w) + lane;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LCW = 3 * (C1 + C2);                       // floats of LDS per wave
    float *Lc1 = smem + wave * LCW, *Lc2 = Lc1 + 3 * C1;
    for (int e = lane; e < C1; e += 64) { Lc1[e] = L1.bias[e]; Lc1[C1 + e] = L1.scale[e]; Lc1[2 * C1 + e] = L1.shift[e]; }
    for (int e = lane; e < C2; e += 64) { Lc2[e] = L2.bias[e]; Lc2[C2 + e] = L2.scale[e]; Lc2[2 * C2 + e] = L2.shift[e]; }
    // ---- gather: the row's partial sums as the accumulators of layer 1, its centred coordinates as the B operand ----
    const int ii = idx[gg * 64 + half * 32 + l31];
    float4 w1[T1];
#pragma unroll
    for (int j = 0; j < T1; ++j) w1[j] = W1[(size_t)j * 64];
    const float *prow = partial + ((size_t)cloud * n + ii) * C1 + 4 * khalf;
    floatx16 a1[T1][NT];
#pragma unroll
    for (int j = 0; j < T1; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(prow + j * 32 + 8 * q);
            a1[j][0][4 * q] = v.x; a1[j][0][4 * q + 1] = v.y; a1[j][0][4 * q + 2] = v.z; a1[j][0][4 * q + 3] = v.w;
        }
    const float *p = xyz + ((size_t)cg * n + ii) * 3;
    const float *c = new_xyz + (size_t)gg * 3;
    const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
    const float bx0 = khalf ? dy : dx, bx1 = khalf ? 0.f : dz;
    float4 w2[3][T2];
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int j = 0; j < T2; ++j) w2[s0][j] = W2[(size_t)(s0 * T2 + j) * 64];
#pragma unroll
    for (int j = 0; j < T1; ++j) {
        a1[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j].x, bx0, a1[j][0], 0, 0, 0);
        a1[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[j].y, bx1, a1[j][0], 0, 0, 0);
    }
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < T1; ++j) rc_epilogue<NT>(a1[j], Lc1, C1, j, khalf);
    // ---- layer 2, transposed ----
    constexpr int NS2 = C1 / 8;
    floatx16 a2[T2][NT];
#pragma unroll
    for (int j = 0; j < T2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[j][0][r] = 0.f;
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
                a2[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(w2[slot % 3][j], ks), a1[slot / 4][0][rc_reg(slot % 4, ks)], a2[j][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int H3 = RC_H3B, TH = T3 / H3;
    float4 w3p[2][TH];                                       // layer 3's first two slots fly under layer 2's epilogue
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int j = 0; j < TH; ++j) w3p[s0][j] = W3[(size_t)(s0 * T3 + j) * 64];
#pragma unroll
    for (int j = 0; j < T2; ++j) rc_epilogue<NT>(a2[j], Lc2, C2, j, khalf);
    // ---- layer 3, standard orientation, H3 passes over the column tiles as ONE stream of slots (the weight ring runs across the pass
    //      boundary: the next pass's first weights are in flight during the previous pass's last slots and its pooled epilogue) ----
    constexpr int NS3 = C2 / 8;
    float pm[T3];
    float4 w3[3][TH];
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int j = 0; j < TH; ++j) w3[s0][j] = w3p[s0][j];
    floatx16 a3[NT][TH];
#pragma unroll
    for (int j = 0; j < TH; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[0][j][r] = 0.f;
    float ep3[3][TH];
#pragma unroll
    for (int s = 0; s < H3 * NS3; ++s) {
        const int h = s / NS3, slot = s % NS3;
        if (s + 2 < H3 * NS3) {
            const int h2 = (s + 2) / NS3, slot2 = (s + 2) % NS3;
#pragma unroll
            for (int j = 0; j < TH; ++j) w3[(s + 2) % 3][j] = W3[(size_t)(slot2 * T3 + h2 * TH + j) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < TH; ++j)
                a3[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[slot / 4][0][rc_reg(slot % 4, ks)], f4_get(w3[s % 3][j], ks), a3[0][j], 0, 0, 0);
        if (slot == NS3 - 2) {
            SaLayer L3h = L3;
            L3h.bias += h * TH * 32; L3h.scale += h * TH * 32; L3h.shift += h * TH * 32;
            ep_load<TH * 32>(L3h, ep3);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (slot == NS3 - 1) {
            float pmh[TH];
            epilogue<TH * 32, TH * 32 + 1, true, NT>(nullptr, a3, ep3, pmh);
#pragma unroll
            for (int j = 0; j < TH; ++j) {
                pm[h * TH + j] = pmh[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) a3[0][j][r] = 0.f;
            }
        }
    }
    // ---- the two row halves of the neighbourhood: the odd wave hands its maxima to the even one through its (now free) LDS block ----
    wave_lds_fence();
    if (half && lane < 32) {
#pragma unroll
        for (int j = 0; j < T3; ++j) Lc1[j * 32 + lane] = pm[j];
    }
    __syncthreads();
    if (!half && lane < 32 && live) {
        const float *O = Lc1 + LCW;
#pragma unroll
        for (int j = 0; j < T3; ++j) out[(size_t)g * C3 + j * 32 + lane] = fmaxf(pm[j], O[j * 32 + lane]);
    }
}
