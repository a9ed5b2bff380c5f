// wave_mlp.h -- the wave-private shared-MLP building blocks of the fused set-abstraction kernels (sa_fused.hip) and of the
// per-point layer chain (chain.hip): one wave owns 32*RT rows in a private LDS tile and runs whole 1x1-conv layers on them with
// v_mfma_f32_32x32x2_f32, weights streaming from L2 in the packed fragment order, no workgroup synchronisation.
#pragma once
#include "common.h"

namespace ancsh {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct SaLayer {
    const float *w, *bias, *scale, *shift;
    int ncol;                   // valid output columns (bias / scale / shift hold this many entries; the packed weights are zero beyond)
    int wstride;                // WSTRIDE kernels only: float4 per packed slot of the WHOLE layer (64 * its column tiles), w already
                                // advanced to this wave's first tile -- a wave that computes a column slice of a wider layer
};

// everything this wave wrote to its LDS tile is visible to all of its lanes, and the compiler keeps the order
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Weights are read in the PACKED order written by ancsh_sa_pack_weights: one 16-byte load hands a lane the MFMA B
// fragments of four consecutive k-steps of one 32-column tile,
//     packed[((slot*TN + j)*64 + lane)*4 + q] = W[2*(4*slot + q) + (lane>>5)][j*32 + (lane&31)]      (0 past row K-1)
// so a wave-instruction reads 1 KiB of consecutive memory.  (With the plain [K][N] layout every lane needs single dwords
// 128 B apart; the texture path then spends ~20 cycles per 256-byte wave-load and the kernel is bound by weight-load ISSUE,
// not by the matrix pipe: measured 332 vs 223 us for SA1 with / without those loads.)
template <int K, int N>
struct LayerCfg {
    static constexpr int TN = N / 32;                   // 32-column accumulators per wave
    static constexpr int NK = (K + 1) / 2;              // MFMA k-steps (two k values each)
    static constexpr int NS = (NK + 3) / 4;             // packed weight slots (4 k-steps each)
    static constexpr int DW = TN >= 8 ? 1 : 2;          // weight prefetch distance in slots (a slot = 4*TN MFMAs = 256*TN cycles)
};

template <int K, int N, bool WSTRIDE = false>
__device__ __forceinline__ void w_load(const SaLayer &L, float4 (&b)[N / 32], int slot) {   // slot: compile-time after unrolling
    constexpr int TN = N / 32;
    const float4 *Wp = reinterpret_cast<const float4 *>(L.w) + (threadIdx.x & 63);
    if (slot < LayerCfg<K, N>::NS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = WSTRIDE ? Wp[(size_t)slot * L.wstride + j * 64] : Wp[(size_t)(slot * TN + j) * 64];
    }
}

template <int K, int N, bool WSTRIDE = false>
__device__ __forceinline__ void w_prologue(const SaLayer &L, float4 (&bw)[LayerCfg<K, N>::DW + 1][N / 32]) {
#pragma unroll
    for (int s = 0; s < LayerCfg<K, N>::DW; ++s) w_load<K, N, WSTRIDE>(L, bw[s], s);
}

template <int N>
__device__ __forceinline__ void ep_load(const SaLayer &L, float (&ep)[3][N / 32]) {
    const int l31 = threadIdx.x & 31;
#pragma unroll
    for (int j = 0; j < N / 32; ++j) {
        const int c = j * 32 + l31 < L.ncol ? j * 32 + l31 : L.ncol - 1;       // unconditional loads, index clamped
        ep[0][j] = L.bias[c]; ep[1][j] = L.scale[c]; ep[2][j] = L.shift[c];
    }
}

__device__ __forceinline__ float f4_get(const float4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// The k loop of one layer over the wave's private tile T[32][LD] (row-major, odd LD >= K + 1, column K readable and
// finite when K is odd): weights DW slots ahead (the first DW slots were issued by the caller, before the previous
// layer's epilogue or the gather), activations DA k-steps ahead, every load issued in the shadow of the MFMAs; the
// epilogue constants of this lane's columns are fetched a few k-steps before the end.
// A_OFF: first input column of the layer inside the tile.  INIT = 1: the accumulators start from T[row][0:N] instead of zero -- the
// k-ordered chain CONTINUES a partial sum that is already in the tile (ancsh_sa_module_fused_partial); INIT = 2: the caller has
// filled them.  WSTRIDE: see SaLayer::wstride.
template <int K, int N, int LD, int RT, int A_OFF = 0, int INIT = 0, bool WSTRIDE = false>
__device__ __forceinline__ void mfma_loop(const float *__restrict__ T, const SaLayer &L, float4 (&bw)[LayerCfg<K, N>::DW + 1][N / 32],
                                          floatx16 (&acc)[RT][N / 32], float (&ep)[3][N / 32]) {
    using C = LayerCfg<K, N>;
    constexpr int TN = C::TN, NK = C::NK, DW = C::DW;
    constexpr int DA = (RT * TN >= 4) ? 2 : 8 / (RT * TN);      // activation (LDS) prefetch distance in k-steps: >= 8 MFMAs
    constexpr int EP_AT = NK > 6 ? NK - 6 : 0;
    static_assert(LD % 2 == 1 && LD >= A_OFF + K + 1, "tile stride");
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    const float *Af = T + l31 * LD + khalf + A_OFF;
    float aw[DA + 1][RT];
    wave_lds_fence();                             // the tile (gather or the previous layer's epilogue) is complete
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)          // MFMA C layout: register r of lane (khalf, l31) = row (r&3) + 8*(r>>2) + 4*khalf, column l31
                if (INIT != 2) acc[i][j][r] = INIT == 1 ? T[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * LD + j * 32 + l31] : 0.f;
#pragma unroll
    for (int s = 0; s < DA; ++s)
        if (s < NK) {
#pragma unroll
            for (int i = 0; i < RT; ++i) aw[s][i] = Af[i * 32 * LD + 2 * s];
        }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int slot = s >> 2, q = s & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s % (DA + 1)][i], f4_get(bw[slot % (DW + 1)][j], q), acc[i][j], 0, 0, 0);
        if (q == 0) w_load<K, N, WSTRIDE>(L, bw[(slot + DW) % (DW + 1)], slot + DW);
        if (s + DA < NK) {
#pragma unroll
            for (int i = 0; i < RT; ++i) aw[(s + DA) % (DA + 1)][i] = Af[i * 32 * LD + 2 * (s + DA)];
        }
        if (s == EP_AT) ep_load<N>(L, ep);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// POOL = false: T[32*RT][0:N] = relu(bn(acc + b)) in place;  POOL = true: pm[j] (lanes 0..31) = max over the wave's 32*RT rows
template <int N, int LD, bool POOL, int RT, bool RELU = true>
__device__ __forceinline__ void epilogue(float *__restrict__ T, const floatx16 (&acc)[RT][N / 32], const float (&ep)[3][N / 32],
                                         float (&pm)[N / 32]) {
    static_assert(RELU || !POOL, "the pooled epilogue starts its maximum at 0");
    static_assert(POOL || LD >= N, "tile stride");
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    wave_lds_fence();     // every read of T by the k loop has completed (its value fed an MFMA already issued)
#pragma unroll
    for (int j = 0; j < N / 32; ++j) {
        const int col = j * 32 + l31;
        float m = 0.f;    // post-ReLU values are >= 0
        // bias + folded BN on PACKED f32 (v_pk_add_f32 / v_pk_fma_f32: two accumulator registers per instruction, each half the
        // same IEEE add / fma as the scalar form): the epilogue's VALU instructions take matrix-pipe time (f32 VALU and f32 MFMA
        // share the lanes), 3 per value before, 2 now
        typedef float ep_f2 __attribute__((ext_vector_type(2)));
        const ep_f2 b2 = {ep[0][j], ep[0][j]}, s2 = {ep[1][j], ep[1][j]}, t2 = {ep[2][j], ep[2][j]};
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                a = __builtin_elementwise_fma(a + b2, s2, t2);
                if (POOL) {
                    m = nmax(nmax(m, a.x), a.y);              // one v_maximum3_f32 (NaN-propagating): the running maximum starts at 0, so the ReLU is implicit
                } else {
                    const float v0 = RELU ? nmax(a.x, 0.f) : a.x, v1 = RELU ? nmax(a.y, 0.f) : a.y;
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;      // r even: rows row, row + 1
                    T[row * LD + col] = v0;
                    T[(row + 1) * LD + col] = v1;
                }
            }
        if (POOL) pm[j] = nmax(m, __shfl_xor(m, 32, 64));
    }
}

// the same values written to global memory instead: out[(row0 + row) * ld + col] for col < ncol, row0 + row < rows
template <int N, int RT, bool RELU>
__device__ __forceinline__ void epilogue_global(float *__restrict__ out, int ld, int ncol, long row0, long rows,
                                                const floatx16 (&acc)[RT][N / 32], const float (&ep)[3][N / 32]) {
    const int lane = threadIdx.x & 63, khalf = lane >> 5, l31 = lane & 31;
    typedef float ep_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < N / 32; ++j) {
        const int col = j * 32 + l31;
        const ep_f2 b2 = {ep[0][j], ep[0][j]}, s2 = {ep[1][j], ep[1][j]}, t2 = {ep[2][j], ep[2][j]};
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                a = __builtin_elementwise_fma(a + b2, s2, t2);
                const float v0 = RELU ? nmax(a.x, 0.f) : a.x, v1 = RELU ? nmax(a.y, 0.f) : a.y;
                const long row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (col < ncol && row < rows) out[(size_t)row * ld + col] = v0;
                if (col < ncol && row + 1 < rows) out[(size_t)(row + 1) * ld + col] = v1;
            }
    }
}

}  // namespace ancsh
