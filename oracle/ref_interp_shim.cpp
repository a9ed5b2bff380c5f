// ref_interp_shim.cpp -- ORACLE-SIDE glue (test infrastructure, not product code).
//
// extern "C" entry points over the REFERENCE's own host loops threenn_cpu / threeinterpolate_cpu
// (ops/3d_interpolation/tf_interpolate.cpp:60-103 and :107-127).  That file cannot be compiled as
// a whole here (it includes TensorFlow headers for the op shells around the loops), but the two
// loops use no TensorFlow type: oracle/Makefile (target `ref`) streams exactly lines 60-127 of
// the file WHERE IT LIES into g++ (stdin, nothing is written to disk or committed) with the
// reference's own flags (-std=c++11 -O2, tf_interpolate_compile.sh:5) and links this shim, which
// only declares the two prototypes and forwards to them.
void threenn_cpu(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx);
void threeinterpolate_cpu(int b, int m, int c, int n, const float *points, const int *idx,
                          const float *weight, float *out);

extern "C" {
void ref_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx) {
    threenn_cpu(b, n, m, xyz1, xyz2, dist, idx);
}
void ref_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                           const float *weight, float *out) {
    threeinterpolate_cpu(b, m, c, n, points, idx, weight, out);
}
}
