// metrics.hip -- evaluation-side kernels.
//
// iou_3d (lib/d3_utils.py:40-69, called per part by evaluation/compute_miou.py:212-225): two oriented boxes (8 corners each),
// a nres^3 grid over their joint axis-aligned bounds, an inside-box test per grid point and box, IoU = |both| / |either|
// (1 when the union is empty).  The reference builds the 125 000 grid points with itertools.product and tests them with
// numpy, ~25 ms per pair; here ONE WORKGROUP per pair strides over the grid in registers (no point is ever materialised),
// counts by ballot + popcount and reduces 4 waves through LDS.  float64 like the reference: grid coordinates are
// numpy.linspace's (start + i*step, the last one exactly stop), the projections up.u are sums of three products in
// (x, y, z) order and the bounds np.dot(u, u).
#include "common.h"

namespace ancsh {

struct BoxFrame {
    double o[3], u1[3], u2[3], u3[3], d1, d2, d3;
};

__device__ __forceinline__ void box_frame(const double *bb, BoxFrame &f) {      // bb: 8 x 3 corners, reference's order
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f.o[c] = bb[4 * 3 + c];
        f.u1[c] = bb[5 * 3 + c] - bb[4 * 3 + c];
        f.u2[c] = bb[7 * 3 + c] - bb[4 * 3 + c];
        f.u3[c] = bb[0 * 3 + c] - bb[4 * 3 + c];
    }
    f.d1 = f.u1[0] * f.u1[0] + f.u1[1] * f.u1[1] + f.u1[2] * f.u1[2];
    f.d2 = f.u2[0] * f.u2[0] + f.u2[1] * f.u2[1] + f.u2[2] * f.u2[2];
    f.d3 = f.u3[0] * f.u3[0] + f.u3[1] * f.u3[1] + f.u3[2] * f.u3[2];
}

__device__ __forceinline__ bool inside(const BoxFrame &f, double x, double y, double z) {
    const double ux = x - f.o[0], uy = y - f.o[1], uz = z - f.o[2];
    const double p1 = ux * f.u1[0] + uy * f.u1[1] + uz * f.u1[2];
    const double p2 = ux * f.u2[0] + uy * f.u2[1] + uz * f.u2[2];
    const double p3 = ux * f.u3[0] + uy * f.u3[1] + uz * f.u3[2];
    return (p1 > 0.0) & (p1 < f.d1) & (p2 > 0.0) & (p2 < f.d2) & (p3 > 0.0) & (p3 < f.d3);
}

__global__ __launch_bounds__(256) void iou_3d_kernel(int nres, const double *__restrict__ bbox1, const double *__restrict__ bbox2,
                                                     double *__restrict__ iou, long *__restrict__ counts) {
#pragma clang fp contract(off)
    __shared__ int red[2][4];
    const int pair = blockIdx.x;
    const double *b1 = bbox1 + (size_t)pair * 24, *b2 = bbox2 + (size_t)pair * 24;
    BoxFrame f1, f2;
    box_frame(b1, f1);
    box_frame(b2, f2);
    double lo[3], hi[3], step[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double mn = b1[c], mx = b1[c];
        for (int k = 0; k < 8; ++k) {
            mn = fmin(mn, fmin(b1[k * 3 + c], b2[k * 3 + c]));
            mx = fmax(mx, fmax(b1[k * 3 + c], b2[k * 3 + c]));
        }
        lo[c] = mn; hi[c] = mx;
        step[c] = (mx - mn) / (double)(nres - 1);          // numpy.linspace: step = delta / div
    }
    const long total = (long)nres * nres * nres;
    int both = 0, either = 0;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int iz = (int)(e % nres), iy = (int)((e / nres) % nres), ix = (int)(e / ((long)nres * nres));
        // linspace: y = arange(num) * step + start, then y[-1] = stop
        const double x = ix == nres - 1 ? hi[0] : (double)ix * step[0] + lo[0];
        const double y = iy == nres - 1 ? hi[1] : (double)iy * step[1] + lo[1];
        const double z = iz == nres - 1 ? hi[2] : (double)iz * step[2] + lo[2];
        const bool i1 = inside(f1, x, y, z), i2 = inside(f2, x, y, z);
        both += (i1 & i2) ? 1 : 0;
        either += (i1 | i2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { both += __shfl_xor(both, o, 64); either += __shfl_xor(either, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = both; red[1][wave] = either; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long I = (long)red[0][0] + red[0][1] + red[0][2] + red[0][3], U = (long)red[1][0] + red[1][1] + red[1][2] + red[1][3];
        iou[pair] = U == 0 ? 1.0 : (double)I / (double)U;
        if (counts) { counts[pair * 2] = I; counts[pair * 2 + 1] = U; }
    }
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_iou_3d(int npairs, int nres, const double *bbox1, const double *bbox2, double *iou, long *counts, void *stream) {
    ANCSH_REQUIRE(npairs >= 0 && nres >= 2 && nres <= 1024, "iou_3d: npairs=%d nres=%d (2..1024)", npairs, nres);
    if (npairs == 0) return ANCSH_OK;
    ANCSH_REQUIRE(bbox1 && bbox2 && iou, "iou_3d: null pointer");
    hipLaunchKernelGGL(iou_3d_kernel, dim3(npairs), dim3(256), 0, (hipStream_t)stream, nres, bbox1, bbox2, iou, counts);
    return check_launch("iou_3d");
}
