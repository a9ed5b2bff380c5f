// Issue rate of packed / plain f32 vector instructions on gfx950 (8 independent chains per lane, 4 waves per SIMD and more).
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate      -> profiles/r04_packed_f32_rate.txt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x2 acc[8];
    float sacc[16];
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{a + i, b + i};
    for (int i = 0; i < 16; ++i) sacc[i] = a + i;
    const f32x2 m = {a, b}, c = {b, a};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], m, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) sacc[i] = __builtin_fmaf(sacc[i], a, b);
        } else if (MODE == 2) {   // packed mul
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = acc[i] * m;
        } else {                  // packed fma with an SGPR-pair operand
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(m), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    for (int i = 0; i < 16; ++i) s += sacc[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wgs) {
    float *d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<wgs, 256>>>(d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<wgs, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)wgs * 4 * iters * 32;       // wave-instructions
    printf("%-28s wgs %5d  %.3f ms  -> %.2f SIMD-clocks per wave-instruction at 2.4 GHz (1024 SIMDs)\n", name, wgs, ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
}
int main() {
    for (int wgs : {256, 1024, 2048}) {
        run<0>("v_pk_fma_f32 (vgpr)", wgs);
        run<1>("v_fma_f32", wgs);
        run<2>("v_pk_mul_f32", wgs);
        run<3>("v_pk_fma_f32 (sgpr pair)", wgs);
    }
    return 0;
}
