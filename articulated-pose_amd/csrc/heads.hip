// heads.hip -- ANCSH head activations + global-NOCS composition, one pass over the per-point
// logits (lib/architecture.py:124-159): softmax(W), sigmoid(nocs, confi, scale, heatmap),
// tanh(trans, unitvec, axis), softmax(joint_cls), gocs = nocs * repeat(scale,3) + trans.
// The reference runs these as ~12 separate TF elementwise ops; here one thread per point reads
// its logits row once and writes every output tensor (HBM-bound: 4*(ld + 11 + 11K) B/point).
#include "common.h"

namespace ancsh {

__device__ __forceinline__ float sigmoidf_(float v) { return __fdiv_rn(1.0f, 1.0f + expf(-v)); }

template <int K>
__device__ __forceinline__ void softmax_store(const float *in, float *out) {
    float mx = in[0];
#pragma unroll
    for (int o = 1; o < K; ++o) mx = fmaxf(mx, in[o]);
    float e[K], s = 0.f;
#pragma unroll
    for (int o = 0; o < K; ++o) { e[o] = expf(in[o] - mx); s += e[o]; }
#pragma unroll
    for (int o = 0; o < K; ++o) out[o] = __fdiv_rn(e[o], s);
}

template <int K>
__global__ __launch_bounds__(256) void head_act_kernel(long rows, int mixed, const float *__restrict__ logits, int ld,
                                                       float *__restrict__ W, float *__restrict__ nocs,
                                                       float *__restrict__ confi, float *__restrict__ heatmap,
                                                       float *__restrict__ unitvec, float *__restrict__ axis,
                                                       float *__restrict__ joint_cls, float *__restrict__ gocs,
                                                       float *__restrict__ scale, float *__restrict__ trans) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *in = logits + (size_t)r * ld;
    constexpr int NW = K, NN = 3 * K;
    const int o_w = 0, o_n = NW, o_s = NW + NN, o_t = o_s + (mixed ? K : 0), o_c = o_t + (mixed ? NN : 0);
    const int o_ax = o_c + 1, o_uv = o_ax + 3, o_hm = o_uv + 3, o_jc = o_hm + 1;

    float buf[NN];
    float wv[K];
#pragma unroll
    for (int o = 0; o < K; ++o) wv[o] = in[o_w + o];
    if (W) softmax_store<K>(wv, W + (size_t)r * K);
    float nv[NN];
#pragma unroll
    for (int o = 0; o < NN; ++o) nv[o] = sigmoidf_(in[o_n + o]);
    if (nocs)
#pragma unroll
        for (int o = 0; o < NN; ++o) nocs[(size_t)r * NN + o] = nv[o];
    if (mixed) {
        float sv[K];
#pragma unroll
        for (int o = 0; o < K; ++o) sv[o] = sigmoidf_(in[o_s + o]);
#pragma unroll
        for (int o = 0; o < NN; ++o) buf[o] = tanhf(in[o_t + o]);
        if (scale)
#pragma unroll
            for (int o = 0; o < K; ++o) scale[(size_t)r * K + o] = sv[o];
        if (trans)
#pragma unroll
            for (int o = 0; o < NN; ++o) trans[(size_t)r * NN + o] = buf[o];
        if (gocs)
#pragma unroll
            for (int o = 0; o < NN; ++o) gocs[(size_t)r * NN + o] = nv[o] * sv[o / 3] + buf[o];   // mul then add (two TF ops)
    }
    if (confi) confi[r] = sigmoidf_(in[o_c]);
    if (axis)
#pragma unroll
        for (int o = 0; o < 3; ++o) axis[(size_t)r * 3 + o] = tanhf(in[o_ax + o]);
    if (unitvec)
#pragma unroll
        for (int o = 0; o < 3; ++o) unitvec[(size_t)r * 3 + o] = tanhf(in[o_uv + o]);
    if (heatmap) heatmap[r] = sigmoidf_(in[o_hm]);
    if (joint_cls) {
        float jv[3] = {in[o_jc], in[o_jc + 1], in[o_jc + 2]};
        softmax_store<3>(jv, joint_cls + (size_t)r * 3);
    }
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_head_activations(long rows, int K, int mixed_pred, const float *logits, int ld, float *W,
                                      float *nocs, float *confi, float *heatmap, float *unitvec, float *axis,
                                      float *joint_cls, float *gocs, float *scale, float *trans, void *stream) {
    ANCSH_REQUIRE(rows >= 0 && K >= 1 && K <= 8, "head_activations: K=%d outside 1..8", K);
    const int need = (mixed_pred ? 8 * K : 4 * K) + 11;
    ANCSH_REQUIRE(ld >= need, "head_activations: ld %d < %d logits per point", ld, need);
    if (rows == 0) return ANCSH_OK;
    ANCSH_REQUIRE(logits, "head_activations: null logits");
    dim3 grid((unsigned)((rows + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define ANCSH_HA(KK)                                                                                                   \
    case KK:                                                                                                           \
        hipLaunchKernelGGL(head_act_kernel<KK>, grid, dim3(256), 0, st, rows, mixed_pred, logits, ld, W, nocs, confi,  \
                           heatmap, unitvec, axis, joint_cls, gocs, scale, trans);                                      \
        break;
    switch (K) {
        ANCSH_HA(1) ANCSH_HA(2) ANCSH_HA(3) ANCSH_HA(4) ANCSH_HA(5) ANCSH_HA(6) ANCSH_HA(7) ANCSH_HA(8)
    }
#undef ANCSH_HA
    return check_launch("head_activations");
}
