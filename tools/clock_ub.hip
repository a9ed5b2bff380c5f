// What does s_memtime count, and how do two matrix waves share a SIMD?   hipcc --offload-arch=gfx950 -O3 clock_ub.hip -o clock_ub
// ticks/ns under (0) a sleeping kernel, (1) a pure f32 MFMA loop, (2) MFMA + LDS + global operand feeds; workgroup 0 reports its own
// lifetime in ticks.  Findings (profiles/r02_clock_microbench.txt): s_memtime is the shader clock (64.01 ticks per 32x32x2 MFMA);
// with two workgroups per CU the kernel takes twice as long but workgroup 0's lifetime does not grow: the older wave owns the matrix
// pipe and the younger one runs afterwards (oldest-first issue), the waves do not interleave; the first launches after idle run at
// 2.23 -> 2.29 -> 2.35 ticks/ns (clock still ramping), a long run at 2.36.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(256) void k(int iters, const float *__restrict__ g, float *__restrict__ out, unsigned long long *ticks) {
    __shared__ float lds[4 * 32 * 17 * 2];
    const int lane = threadIdx.x & 63;
    floatx16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int e = threadIdx.x; e < 4 * 32 * 17 * 2; e += 256) lds[e] = (float)e;
    __syncthreads();
    const float *Af = lds + (threadIdx.x >> 6) * (32 * 17 * 2) + (lane & 31) * 17 + (lane >> 5);
    const float4 *bg = reinterpret_cast<const float4 *>(g) + lane;
    float4 bv[8];
    for (int j = 0; j < 8; ++j) bv[j] = make_float4(1.f, 2.f, 3.f, 4.f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { __builtin_amdgcn_s_sleep(127); continue; }
        if (V == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = bg[(size_t)((it & 63) * 8 + j) * 64];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float av = (float)lane;
            if (V == 2) av = Af[2 * s + (it & 1) * 8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float bb = s == 0 ? bv[j].x : s == 1 ? bv[j].y : s == 2 ? bv[j].z : bv[j].w;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
template <int V> void run(const char *name, int grid, int iters, const float *g, float *out, unsigned long long *ticks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, 8, g, out, ticks); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, iters, g, out, ticks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double tf = V ? (double)grid * 4 * iters * 32 * 4096.0 / ms / 1e9 : 0.0;
        printf("%-28s grid %4d: %9.1f us  %12llu ticks  %.3f ticks/ns  %.3f ticks per MFMA  %.1f TF/s\n", name, grid, ms * 1e3, t, t / (ms * 1e6),
               V ? (double)t / (iters * 32.0) : 0.0, tf);
    }
}
int main() {
    float *g, *out; unsigned long long *ticks;
    hipMalloc(&g, 64 * 8 * 64 * 16 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&ticks, 8);
    hipMemset(g, 0, 64 * 8 * 64 * 16 * 4);
    run<0>("sleep", 1, 20000, g, out, ticks);
    run<1>("pure MFMA, 1 workgroup", 1, 4096, g, out, ticks);
    run<1>("pure MFMA, whole chip", 256, 4096, g, out, ticks);
    run<1>("pure MFMA, chip x2 waves", 512, 4096, g, out, ticks);
    run<2>("MFMA+LDS+L2, 1 workgroup", 1, 4096, g, out, ticks);
    run<2>("MFMA+LDS+L2, whole chip", 256, 4096, g, out, ticks);
    run<2>("MFMA+LDS+L2, chip x2 waves", 512, 4096, g, out, ticks);
    run<2>("MFMA+LDS+L2, x2, long", 512, 65536, g, out, ticks);
    return 0;
}
