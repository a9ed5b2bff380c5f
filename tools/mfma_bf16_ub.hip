// micro-benchmark: what does v_mfma_f32_32x32x16_bf16 sustain on gfx950, alone and next to the VALU / load work of the bf16x3 kernels?
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_ub.hip -o mfma_bf16_ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float fx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

// V: 0 pure (NACC accumulators, constant operands)
//    1: + per 6 x NACC MFMAs one "epilogue slice": VW packed VALU instructions on independent registers (the split's instruction mix)
//    2: + per 6 x NACC MFMAs three 16-byte global loads (the weight stream) consumed by the next block
template <int NACC, int V, int VW>
__global__ __launch_bounds__(256) void k_mfma(int iters, const uint4 *__restrict__ g, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    fx16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    uint4 a = make_uint4(lane, lane + 1, lane + 2, lane + 3), b[3];
    b[0] = b[1] = b[2] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)lane + i, 1.5f};
    const uint4 *gp = g + lane;
    for (int it = 0; it < iters; ++it) {
        uint4 nb[3];
        if (V == 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q) nb[q] = gp[(size_t)(((it & 255) * 3 + q)) * 64];
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, b[t % 3]), __builtin_bit_cast(bfx8, a), acc[j], 0, 0, 0);
        if (V == 1) {
#pragma unroll
            for (int w = 0; w < VW; ++w) {
                const int i = w & 7;
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v[i], bf2));
                v[i] = v[i] - f2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)} + f2{1.f, 1.f};
            }
        }
        if (V == 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = nb[q];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int V, int VW>
void run(int wg_per_cu, int iters, const uint4 *g, float *out) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma<NACC, V, VW>), dim3(grid), dim3(256), 0, 0, 8, g, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<NACC, V, VW>), dim3(grid), dim3(256), 0, 0, iters, g, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)grid * 4 * iters * 6 * NACC;
    printf("NACC %d V %d VW %2d waves/SIMD %d : %8.1f us  %7.1f TF/s (bf16)  %5.1f clk/MFMA at 2.4 GHz\n", NACC, V, VW, wg_per_cu, ms * 1e3, mf * 32768.0 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)iters * 6 * NACC) / wg_per_cu);
}

int main() {
    uint4 *g; float *out; hipMalloc(&g, 256 * 3 * 64 * 16); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemset(g, 0x3f, 256 * 3 * 64 * 16);
    const int it = 2048;
    for (int w = 1; w <= 2; ++w) {
        run<8, 0, 0>(w, it, g, out); run<4, 0, 0>(w, it * 2, g, out); run<2, 0, 0>(w, it * 4, g, out); run<1, 0, 0>(w, it * 8, g, out);
        run<8, 1, 8>(w, it, g, out); run<8, 1, 24>(w, it, g, out); run<8, 1, 48>(w, it, g, out);
        run<8, 2, 0>(w, it, g, out); run<2, 2, 0>(w, it * 4, g, out);
    }
    return 0;
}
