// ref_shim.cpp -- ORACLE-SIDE glue (test infrastructure, not product code).
//
// extern "C" entry points over the REFERENCE's own launcher functions, whose sources are
// compiled where they lie under /root/reference by oracle/Makefile (target `ref`) into
// oracle/_ref/libancsh_ref_gfx950.so.  No reference source is copied: this file only declares
// the launchers' prototypes (ops/sampling/tf_sampling_g.cu:196-208,
// ops/grouping/tf_grouping_g.cu:125-136) and forwards to them.  The reference launches on the
// default stream and never checks errors, so each shim synchronises and returns the HIP status.
#include <hip/hip_runtime.h>

void farthestpointsamplingLauncher(int b, int n, int m, const float *inp, float *temp, int *out);
void gatherpointLauncher(int b, int n, int m, const float *inp, const int *idx, float *out);
void probsampleLauncher(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out);
void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float *xyz1,
                            const float *xyz2, int *idx, int *pts_cnt);
void groupPointLauncher(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                        float *out);
void selectionSortLauncher(int b, int n, int m, int k, const float *dist, int *outi, float *out);

static int done() {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    return (int)e;
}

extern "C" {
// temp must hold 32*n floats (ops/sampling/tf_sampling.cpp:115)
int ref_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out) {
    farthestpointsamplingLauncher(b, n, m, inp, temp, out);
    return done();
}
// temp: b*n floats (ops/sampling/tf_sampling.cpp:86)
int ref_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out) {
    probsampleLauncher(b, n, m, inp_p, inp_r, temp, out);
    return done();
}
int ref_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out) {
    gatherpointLauncher(b, n, m, inp, idx, out);
    return done();
}
int ref_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                         const float *xyz2, int *idx, int *pts_cnt) {
    queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt);
    return done();
}
int ref_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out) {
    selectionSortLauncher(b, n, m, k, dist, outi, out);
    return done();
}
int ref_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out) {
    groupPointLauncher(b, n, c, m, nsample, points, idx, out);
    return done();
}
}
