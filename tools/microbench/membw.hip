// membw.hip -- NOT part of the product library (libancsh_hip.so): a measurement aid of bench.py, built on its own into
// tools/microbench/libyardstick.so by tools/microbench/Makefile (__graft_entry__.build() runs it).  The achievable-HBM yardstick next to the 8.0 TB/s datasheet figure: a plain float4 copy (16 B per lane, every byte
// read once and written once).  bench.py runs it over buffers far beyond the 256 MiB Infinity Cache and reports the op-level
// fractions against BOTH numbers (MI355X_MICROARCH.md quotes 6.29 TB/s for this pattern).  Three forms were measured on 1 GiB
// (profiles/r04_hbm_copy_variants.txt): ONE element per thread with a grid as large as the buffer 6.2 TB/s -- the default --,
// grid-stride with four loads in flight per lane 4.7-4.8, the same with non-temporal loads / stores 4.8-4.9 (torch's copy_ 4.7):
// the workgroup dispatcher strides better than the program does.  ANCSH_COPY_VARIANT = 1 / 2 selects the other two.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace {

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

// one element per thread, a grid as large as the buffer (the hardware's workgroup dispatcher does the striding)
__global__ __launch_bounds__(256) void hbm_copy_flat_kernel(long n16, const float4 *__restrict__ src, float4 *__restrict__ dst) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < n16) dst[e] = src[e];
}

// streaming (non-temporal) loads and stores, four in flight per lane
__global__ __launch_bounds__(256) void hbm_copy_nt_kernel(long n16, const float4 *__restrict__ src, float4 *__restrict__ dst) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v *s = reinterpret_cast<const f4v *>(src);
    f4v *d = reinterpret_cast<f4v *>(dst);
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n16; e += 4 * stride) {
        f4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(s + (e + u * stride < n16 ? e + u * stride : e));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e + u * stride < n16) __builtin_nontemporal_store(v[u], d + e + u * stride);
    }
}

}  // namespace

// 0 = launched; -1 = bad argument or launch error (reason on stderr)
extern "C" int yardstick_hbm_copy(long nbytes, const void *src, void *dst, void *stream) {
    auto bad = [](const char *why) { fprintf(stderr, "yardstick_hbm_copy: %s\n", why); return -1; };
    if (nbytes < 0 || nbytes % 16 != 0) return bad("nbytes must be a non-negative multiple of 16");
    if (nbytes == 0) return 0;
    if (!src || !dst || (((uintptr_t)src | (uintptr_t)dst) % 16) != 0) return bad("16-byte aligned non-null buffers");
    const long n16 = nbytes / 16;
    long blocks = (n16 + 1023) / 1024;
    if (blocks > 256L * 32) blocks = 256L * 32;
    static const int variant = [] { const char *e = getenv("ANCSH_COPY_VARIANT"); return e ? atoi(e) : 0; }();
    if ((n16 + 255) / 256 >= (1L << 31)) return bad("nbytes too large for one launch");
    if (variant == 1)
        hipLaunchKernelGGL(hbm_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n16, (const float4 *)src, (float4 *)dst);
    else if (variant == 2)
        hipLaunchKernelGGL(hbm_copy_nt_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n16, (const float4 *)src, (float4 *)dst);
    else
        hipLaunchKernelGGL(hbm_copy_flat_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n16, (const float4 *)src, (float4 *)dst);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : bad(hipGetErrorString(e));
}
