// common.h -- shared helpers for the gfx950 kernels of libancsh_hip.so (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/ancsh_hip.h"

namespace ancsh {

// NaN-PROPAGATING maximum (IEEE 754-2019 maximum; v_maximum3_f32 on gfx950, one instruction like v_max_f32): the ReLU and the
// max-pooling of every shared-MLP kernel use it, so a non-finite input coordinate or feature poisons every output that depends
// on it instead of being silently dropped by maxNum semantics (fmaxf(NaN, 0) = 0).  For finite operands it IS fmaxf.
__device__ __forceinline__ float nmax(float a, float b) { return __builtin_elementwise_maximum(a, b); }


void set_error(const char *fmt, ...);

// A GROUPED layer launch: `n` equal-shaped layers with their own parameters in ONE launch -- group g (= blockIdx.z) works on rows
// [g * rows, (g + 1) * rows) of x / y (the same layer of several networks evaluated on stacked activations).  n == 1 is the plain call.
constexpr int CONV_MAX_GROUPS = ANCSH_MAX_GROUPS;
struct ConvGroups {
    int n;
    long x_stride, y_stride, init_stride;          // floats between consecutive groups' first rows of x / y / acc_init
    const float *wp[CONV_MAX_GROUPS], *bias[CONV_MAX_GROUPS], *scale[CONV_MAX_GROUPS], *shift[CONV_MAX_GROUPS];
};
#define CONV_SELECT_GROUP(G, x, y, acc_init, wp, bias, scale, shift)                       \
    if ((G).n > 1) {                                                                      \
        const int g_ = blockIdx.z;                                                        \
        x += (size_t)g_ * (G).x_stride;                                                   \
        y += (size_t)g_ * (G).y_stride;                                                   \
        if (acc_init) acc_init += (size_t)g_ * (G).init_stride;                           \
        wp = (G).wp[g_]; bias = (G).bias[g_]; scale = (G).scale[g_]; shift = (G).shift[g_]; \
    }

// conv_rowtile.hip: the small-layer schedule behind ancsh_conv1x1_packed (true = launched)
bool conv_rowtile_launch(long rows, int cin, int cout, const float *x, int ldx, const float *wp, const float *bias,
                         const float *scale, const float *shift, int act, float *y, int ldy, const float *acc_init, int init_rows,
                         const ConvGroups &G, hipStream_t st);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ANCSH_EHIP;
    }
    return ANCSH_OK;
}

#define ANCSH_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            ancsh::set_error(__VA_ARGS__); \
            return ANCSH_EINVAL;          \
        }                                 \
    } while (0)

// ---- wave64 cross-lane helpers (DPP: no LDS round trip) ---------------------------------
// dpp_ctrl encodings (LLVM AMDGPU): quad_perm = p0|p1<<2|p2<<4|p3<<6; row_shr:n = 0x110+n;
// row_mirror 0x140; row_half_mirror 0x141; row_bcast15 0x142; row_bcast31 0x143.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// max over the 64 lanes of a wave, result uniform (max is idempotent, so butterfly-with-mirrors
// leaves every 16-lane row uniform after 4 DPP steps; rows are combined through SGPRs).
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141>(v));  // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));  // row_mirror
    float a = readlane_f(v, 0), b = readlane_f(v, 16), c = readlane_f(v, 32), d = readlane_f(v, 48);
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

}  // namespace ancsh
