// input.hip -- the sampling step in front of the network, on the GPU (gfx950).
//
// Reference: lib/dataset.py:290-351 (create_unit_data_from_hdf5, after the per-part arrays are concatenated): a cloud
// with fewer raw points than num_points is TILED (np.concatenate([arr] * tile_n), :290-317), a random permutation picks
// num_points rows (:341-351), the coordinates are scaled by the category's norm_factor (:346), the part label becomes a
// one-hot mask_array (:357) and joint_cls_mask = (joint_cls > 0) (:353-355) -- one numpy fancy-index per array per
// cloud on the host.  Here a whole ragged batch is ONE launch: row i of cloud b reads raw row perm[b][i] % n_raw[b]
// (the tiled array is never materialised: tiled[t] == raw[t % n_raw]) and writes every output.
// HBM-bound: 4*nchan B read + 4*(nchan + n_parts + 1) B written per sampled point; the raw rows of one cloud
// (<= a few hundred KB) are L2-resident while its num_points rows are gathered.
#include "common.h"

namespace ancsh {

// one thread per (sampled row, channel group): lanes of a wave cover consecutive channels of consecutive rows, so the
// output streams are written coalesced; the gathered source row is 4*nchan contiguous bytes.
__global__ __launch_bounds__(256) void input_sample_kernel(int num_points, int nchan, const float *__restrict__ rows,
                                                           const int *__restrict__ offsets, const int *__restrict__ perm,
                                                           const float *__restrict__ norm_factor, int cls_col, int jcls_col,
                                                           int n_parts, float *__restrict__ P, float *__restrict__ chan_out,
                                                           float *__restrict__ mask_array, float *__restrict__ joint_cls_mask) {
    const int b = blockIdx.y;
    const int r0 = offsets[b];
    const int n_raw = offsets[b + 1] - r0;
    if (n_raw <= 0) return;                                // an empty cloud leaves its outputs untouched
    const float nf = norm_factor[b];
    const int cout = nchan - 3;                            // channels besides xyz
    const int width = nchan + n_parts + 1;                 // work items per sampled row: nchan copies, n_parts mask entries, 1 joint mask
    const long total = (long)num_points * width;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / width), k = (int)(e - (long)i * width);
        const size_t o = (size_t)b * num_points + i;
        const int t = perm[o];                             // index into the (virtually) tiled cloud
        const float *src = rows + (size_t)(r0 + t % n_raw) * nchan;
        if (k < 3) {
            P[o * 3 + k] = src[k] * nf;
        } else if (k < nchan) {
            chan_out[o * cout + (k - 3)] = src[k];
        } else if (k < nchan + n_parts) {
            const int lab = (int)(signed char)(int)src[cls_col];      // astype(np.int8) like the reference (:357)
            mask_array[o * n_parts + (k - nchan)] = (lab == k - nchan || lab + n_parts == k - nchan) ? 1.f : 0.f;
        } else {
            joint_cls_mask[o] = (jcls_col >= 0 && src[jcls_col] > 0.f) ? 1.f : 0.f;
        }
    }
}

}  // namespace ancsh

using namespace ancsh;

extern "C" int ancsh_input_sample(int nclouds, int num_points, int nchan, const float *rows, const int *offsets, const int *perm,
                                  const float *norm_factor, int cls_col, int jcls_col, int n_parts, float *P, float *chan_out,
                                  float *mask_array, float *joint_cls_mask, void *stream) {
    ANCSH_REQUIRE(nclouds >= 0 && num_points > 0, "input_sample: bad shape nclouds=%d num_points=%d", nclouds, num_points);
    ANCSH_REQUIRE(nchan >= 3, "input_sample: rows need at least the 3 coordinate channels (nchan=%d)", nchan);
    ANCSH_REQUIRE(cls_col >= 3 && cls_col < nchan, "input_sample: cls_col=%d must name a channel in [3,%d)", cls_col, nchan);
    ANCSH_REQUIRE(jcls_col < nchan && (jcls_col < 0 || jcls_col >= 3), "input_sample: jcls_col=%d out of range", jcls_col);
    ANCSH_REQUIRE(n_parts > 0 && n_parts <= 64, "input_sample: n_parts=%d must be in [1,64]", n_parts);
    ANCSH_REQUIRE(nclouds <= 65535, "input_sample: %d clouds exceed the 65535-cloud grid range; split the batch", nclouds);
    if (nclouds == 0) return ANCSH_OK;
    ANCSH_REQUIRE(rows && offsets && perm && norm_factor && P && mask_array && joint_cls_mask && (nchan == 3 || chan_out),
                  "input_sample: null pointer");
    const long total = (long)num_points * (nchan + n_parts + 1);
    long bx = (total + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(input_sample_kernel, dim3((unsigned)bx, nclouds), dim3(256), 0, (hipStream_t)stream, num_points, nchan, rows,
                       offsets, perm, norm_factor, cls_col, jcls_col, n_parts, P, chan_out, mask_array, joint_cls_mask);
    return check_launch("input_sample");
}
