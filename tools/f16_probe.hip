// one-off hardware probe (round 6): conversion and denormal behaviour of the f16 MFMA path used by the F16x2 split scheme (csrc/bx3.h)
//   hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o f16_probe && ./f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float fx16 __attribute__((ext_vector_type(16)));
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// out[0..]: D = A * B with A[i][k] = a (lane pattern), B = b: check denormal inputs and plain values
__global__ void probe(const float *av, const float *bv, float *out) {
    const int lane = threadIdx.x;
    // A operand: row = lane&31, k = 8*(lane>>5)+e ; set A[row][k] = av[0] for k == 0 else 0 ; B[k][col] = bv[0] for k == 0 else 0
    hx8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    if (lane < 32) { a[0] = (_Float16)av[0]; b[0] = (_Float16)bv[0]; }
    fx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = acc[0];
        // conversion behaviour
        f2 x = {av[1], av[2]};
        h2 h = __builtin_convertvector(x, h2);
        out[1] = (float)h[0]; out[2] = (float)h[1];
        out[3] = (float)(_Float16)av[0];
    }
}
int main() {
    float *a, *b, *o; hipMalloc(&a, 64); hipMalloc(&b, 64); hipMalloc(&o, 64);
    struct { float a0, b0, x1, x2; const char *what; } cases[] = {
        {1.5f, 2.0f, 1.0f + 1.0f / 2048.f, 1.0f + 3.0f / 2048.f, "plain 1.5*2; RNE ties at 2^-11"},
        {3.0e-6f, 1024.f, 3.0e-6f, 6.0e-8f, "denormal f16 input a=3e-6 (sub-normal), b=1024"},
        {6.0e-8f, 65504.f, 1e-9f, 70000.f, "smallest subnormal * max; cvt of 1e-9 and 70000"},
    };
    for (auto &c : cases) {
        float ha[3] = {c.a0, c.x1, c.x2}, hb[1] = {c.b0}, ho[4];
        hipMemcpy(a, ha, 12, hipMemcpyHostToDevice); hipMemcpy(b, hb, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, o);
        hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        printf("%s\n  mfma = %.9g (exact product of the f16 roundings: %.9g)\n  cvt(%.9g) = %.9g  cvt(%.9g) = %.9g  cvt(a0) = %.9g\n", c.what, ho[0],
               (double)(float)(_Float16)c.a0 * (double)(float)(_Float16)c.b0, c.x1, ho[1], c.x2, ho[2], ho[3]);
    }
    return 0;
}
