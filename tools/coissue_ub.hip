// micro-benchmark: do MFMA waves and VALU waves sharing a SIMD co-issue?   hipcc --offload-arch=gfx950 -O3 coissue_ub.hip -o coissue_ub
// A 512-thread workgroup per CU: waves 0-3 (one per SIMD) run back-to-back v_mfma_f32_32x32x2_f32 on 8 accumulators, waves 4-7
// (one per SIMD) run dependent-free packed-f32 FMAs.  MODE 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            floatx16 acc[8];
            for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            float a = (float)lane, b = 1.f + lane;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
            for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        }
    } else if (MODE & 2) {
        f32x2 v[16];
        for (int j = 0; j < 16; ++j) v[j] = f32x2{(float)(lane + j), 1.f};
        const f32x2 m = f32x2{1.0001f, 0.9999f}, c = f32x2{1e-6f, -1e-6f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)           // 16 x 16 = 256 packed FMAs per iteration ~ the 2048 cycles of 32 MFMAs
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
        }
        for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(int iters, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, 8, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int it = 20000;
    const float m1 = run<1>(it, out), m2 = run<2>(it, out), m3 = run<3>(it, out);
    const double mf = 256.0 * 4 * it * 32 * 4096.0, vf = 256.0 * 4 * 64 * it * 256 * 4.0;   // flops
    printf("MFMA waves alone  : %8.2f ms  %6.1f TF/s\n", m1, mf / m1 / 1e9);
    printf("VALU waves alone  : %8.2f ms  %6.1f TF/s (packed f32 FMA)\n", m2, vf / m2 / 1e9);
    printf("both, same SIMDs  : %8.2f ms  (sum of the two alone: %.2f ms, max: %.2f ms)\n", m3, m1 + m2, m1 > m2 ? m1 : m2);
    return 0;
}
