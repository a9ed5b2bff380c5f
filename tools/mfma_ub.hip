// micro-benchmark: what sustains v_mfma_f32_32x32x2_f32 at peak?   hipcc --offload-arch=gfx950 -O3 mfma_ub.hip -o mfma_ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

// V: 0 pure (NACC accumulators, constant operands)   1: + v_mov of A between k-steps   2: + LDS read of A per k-step
//    3: + global float4 B load per 4 k-steps          4: 2 + 3
template <int NACC, int V>
__global__ __launch_bounds__(256) void k_mfma(int iters, const float *__restrict__ g, float *__restrict__ out) {
    __shared__ float lds[4 * 32 * 17 * 2];
    const int lane = threadIdx.x & 63;
    floatx16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = (float)lane, b[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) b[j] = (float)(lane + j);
    for (int e = threadIdx.x; e < 4 * 32 * 17 * 2; e += 256) lds[e] = (float)e;
    __syncthreads();
    const float *Af = lds + (threadIdx.x >> 6) * (32 * 17 * 2) + (lane & 31) * 17 + (lane >> 5);
    const float4 *bg = reinterpret_cast<const float4 *>(g) + lane;
    float4 bv[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) bv[j] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int it = 0; it < iters; ++it) {
        if (V == 3 || V == 4) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) bv[j] = bg[(size_t)((it & 63) * NACC + j) * 64];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float av = a;
            if (V == 2 || V == 4) av = Af[2 * s + (it & 1) * 8];
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const float bb = (V == 3 || V == 4) ? (s == 0 ? bv[j].x : s == 1 ? bv[j].y : s == 2 ? bv[j].z : bv[j].w) : b[j];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb, acc[j], 0, 0, 0);
            }
            if (V == 1) { a = a + 1.0f; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int V>
void run(int wg_per_cu, int iters, const float *g, float *out) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma<NACC, V>), dim3(grid), dim3(256), 0, 0, 8, g, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<NACC, V>), dim3(grid), dim3(256), 0, 0, iters, g, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 4 * NACC * 4096.0;
    printf("NACC %d V %d waves/SIMD %d : %8.1f us  %6.1f TF/s\n", NACC, V, wg_per_cu, ms * 1e3, flops / ms / 1e9);
}

int main() {
    float *g, *out; hipMalloc(&g, 64 * 8 * 64 * 16 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemset(g, 0, 64 * 8 * 64 * 16 * 4);
    const int it = 4096;
    for (int w = 1; w <= 2; ++w) {
        run<8, 0>(w, it, g, out); run<8, 1>(w, it, g, out); run<8, 2>(w, it, g, out); run<8, 3>(w, it, g, out); run<8, 4>(w, it, g, out);
        run<2, 0>(w, it * 4, g, out); run<2, 4>(w, it * 4, g, out); run<4, 0>(w, it * 2, g, out); run<4, 4>(w, it * 2, g, out);
    }
    run<4, 0>(4, it * 2, g, out); run<4, 4>(4, it * 2, g, out); run<2, 0>(4, it * 4, g, out); run<2, 4>(4, it * 4, g, out);
    return 0;
}
