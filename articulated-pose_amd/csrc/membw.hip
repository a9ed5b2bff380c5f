// membw.hip -- the achievable-HBM yardstick next to the 8.0 TB/s datasheet figure: a plain float4 copy (16 B per lane, grid-stride,
// every byte read once and written once).  bench.py runs it over buffers far beyond the 256 MiB Infinity Cache and reports the
// op-level fractions against BOTH numbers (MI355X_MICROARCH.md quotes 6.29 TB/s for this pattern).
#include "common.h"

namespace ancsh {

__global__ __launch_bounds__(256) void hbm_copy_kernel(long n16, const float4 *__restrict__ src, float4 *__restrict__ dst) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x); e < n16; e += stride) {
        // four independent 16-byte loads in flight per lane before the first store
        const long e1 = e + stride / 4, e2 = e + stride / 2, e3 = e + 3 * (stride / 4);
        const float4 a = src[e];
        const float4 b = e1 < n16 ? src[e1] : a;
        const float4 c = e2 < n16 ? src[e2] : a;
        const float4 d = e3 < n16 ? src[e3] : a;
        dst[e] = a;
        if (e1 < n16) dst[e1] = b;
        if (e2 < n16) dst[e2] = c;
        if (e3 < n16) dst[e3] = d;
    }
}

}  // namespace ancsh

extern "C" int ancsh_hbm_copy(long nbytes, const void *src, void *dst, void *stream) {
    using namespace ancsh;
    ANCSH_REQUIRE(nbytes >= 0 && nbytes % 16 == 0, "hbm_copy: nbytes %ld must be a non-negative multiple of 16", nbytes);
    if (nbytes == 0) return ANCSH_OK;
    ANCSH_REQUIRE(src && dst && (((uintptr_t)src | (uintptr_t)dst) % 16) == 0, "hbm_copy: 16-byte aligned non-null buffers");
    const long n16 = nbytes / 16;
    long blocks = (n16 + 1023) / 1024;
    if (blocks > 256L * 32) blocks = 256L * 32;
    hipLaunchKernelGGL(hbm_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n16, (const float4 *)src, (float4 *)dst);
    return check_launch("hbm_copy");
}
