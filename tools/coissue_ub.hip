// micro-benchmark: do MFMA waves and VALU waves sharing a SIMD co-issue?   hipcc --offload-arch=gfx950 -O3 coissue_ub.hip -o coissue_ub
// A 512-thread workgroup per CU: waves 0-3 (one per SIMD) run back-to-back v_mfma_f32_32x32x2_f32 on 8 accumulators, waves 4-7
// (one per SIMD) run dependence-free VALU work of one KIND.  MODE 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
// Round 1 only measured KIND 0 (packed f32 FMA: the microarch guide lists packed f32 beside MFMAs as an anti-lever).  Round 5 adds the
// kinds the pose fit could be written in: plain v_fma_f32, the verifier's sub / mul / add + compare-and-count sequence in plain f32,
// integer VALU, v_fma_f64 (the LM fit) -- every VALU instruction is inline asm so that the SLP vectoriser cannot re-pack it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { PK_FMA = 0, FMA32 = 1, CMPCNT = 2, INT32 = 3, FMA64 = 4, PK_MUL_CLAMP = 5, NKIND = 6 };
static const char *kind_name[NKIND] = {"v_pk_fma_f32", "v_fma_f32", "v_sub/v_mul/v_add + v_cmp/v_addc (plain f32 verifier)",
                                       "v_add_u32 / v_and_b32 / v_mad_u32_u24", "v_fma_f64", "v_pk_mul/v_pk_add/v_pk_fma clamp (scoring kernel mix)"};
// VALU instructions per inner trip (16 chains), used for the rate column
static const int kind_ops[NKIND] = {16, 16, 16 * 6, 16 * 3, 16, 16 * 3};

template <int KIND>
__device__ __forceinline__ float valu_body(int iters, int lane) {
    float s = 0.f;
    if (KIND == PK_FMA) {
        f32x2 v[16];
        for (int j = 0; j < 16; ++j) v[j] = f32x2{(float)(lane + j), 1.f};
        const f32x2 m = f32x2{1.0001f, 0.9999f}, c = f32x2{1e-6f, -1e-6f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
    } else if (KIND == FMA32) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = (float)(lane + j);
        const float m = 1.0001f, c = 1e-6f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        for (int j = 0; j < 16; ++j) s += v[j];
    } else if (KIND == CMPCNT) {
        // per "point": d = t - x (v_sub), q = d*d (v_mul), q2 = q + e (v_add), compare with th, count through carry (v_addc)
        float x[16]; unsigned cnt[16];
        for (int j = 0; j < 16; ++j) { x[j] = 0.001f * (lane + j); cnt[j] = 0; }
        const float t = 0.37f, e = 0.003f, th = 0.01f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float d, q;
                    asm volatile("v_sub_f32 %0, %2, %3\n\tv_mul_f32 %1, %0, %0\n\tv_add_f32 %1, %1, %4\n\tv_mul_f32 %0, %0, %5"
                                 : "=&v"(d), "=&v"(q) : "v"(t), "v"(x[j]), "v"(e), "v"(1.0001f));
                    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt[j]) : "v"(q), "v"(th) : "vcc");
                    x[j] = d;
                }
        for (int j = 0; j < 16; ++j) s += (float)cnt[j] + x[j];
    } else if (KIND == INT32) {
        unsigned v[16];
        for (int j = 0; j < 16; ++j) v[j] = lane * 31 + j;
        const unsigned a = 0x9e3779b9u, b = 0x00ffffffu;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    asm volatile("v_add_u32 %0, %0, %1\n\tv_and_b32 %0, %0, %2\n\tv_mad_u32_u24 %0, %0, 3, %1" : "+v"(v[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 16; ++j) s += (float)v[j];
    } else if (KIND == FMA64) {
        double v[16];
        for (int j = 0; j < 16; ++j) v[j] = (double)(lane + j);
        const double m = 1.0001, c = 1e-6;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        for (int j = 0; j < 16; ++j) s += (float)v[j];
    } else {
        f32x2 v[16], c[16];
        for (int j = 0; j < 16; ++j) { v[j] = f32x2{0.001f * (lane + j), 1.f}; c[j] = f32x2{0.f, 0.f}; }
        const f32x2 m = f32x2{1.0001f, 0.9999f}, th = f32x2{0.01f, 0.01f}, big = f32x2{0x1p126f, 0x1p126f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    f32x2 q;
                    asm volatile("v_pk_mul_f32 %1, %0, %0\n\tv_pk_add_f32 %1, %3, %1 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_fma_f32 %2, %1, %4, %2 clamp\n\tv_pk_mul_f32 %0, %0, %5"
                                 : "+v"(v[j]), "=&v"(q), "+v"(c[j]) : "v"(th), "v"(big), "v"(m));
                }
        for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y + c[j].x + c[j].y;
    }
    return s;
}

template <int MODE, int KIND>
__global__ __launch_bounds__(512) void k(int iters, int iters_valu, float *out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
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
        s = valu_body<KIND>(iters_valu, lane);
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(int iters, int iters_valu, float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, 8, 8, out); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, iters, iters_valu, out);     // clock ramp: the timed launch runs loaded
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, iters, iters_valu, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int KIND>
void report(float *out, float m1, int it) {
    // the VALU trip count is chosen so that the VALU waves alone take about as long as the MFMA waves alone
    const float probe = run<2, KIND>(0, 2000, out);
    const int itv = (int)(2000.0 * m1 / probe);
    const float m2 = run<2, KIND>(0, itv, out);
    const float m3 = run<3, KIND>(it, itv, out);
    const double inst_per_wave = (double)itv * 16 * kind_ops[KIND];
    printf("%-58s alone %7.2f ms (%5.2f clk/wave-inst @2.4GHz) | MFMA alone %7.2f | both %7.2f | sum %7.2f max %7.2f -> overlap %.2f\n",
           kind_name[KIND], m2, m2 * 1e-3 * 2.4e9 / inst_per_wave, m1, m3, m1 + m2, m1 > m2 ? m1 : m2,
           (m1 + m2 - m3) / (m1 < m2 ? m1 : m2));
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int it = 20000;
    const float m1 = run<1, 0>(it, 0, out);
    const double mf = 256.0 * 4 * it * 32 * 4096.0;
    printf("MFMA waves alone (v_mfma_f32_32x32x2_f32, 8 accumulators, 1 wave/SIMD): %8.2f ms  %6.1f TF/s\n", m1, mf / m1 / 1e9);
    printf("overlap = (sum - both) / min(alone): 1.00 = the shorter one is completely hidden, 0.00 = strictly serial\n");
    report<PK_FMA>(out, m1, it);
    report<FMA32>(out, m1, it);
    report<CMPCNT>(out, m1, it);
    report<INT32>(out, m1, it);
    report<FMA64>(out, m1, it);
    report<PK_MUL_CLAMP>(out, m1, it);
    return 0;
}
