// pose_math.h -- double-precision device math for the pose-fit kernels (gfx950).
//
// Restates, per thread / per workgroup, the numerical building blocks of the reference's pose fit:
//   rotate_pts  (lib/d3_utils.py:206-220)  Kabsch rotation: the reference takes a LAPACK 3x3 SVD and
//               flips the last singular direction when det(U)det(Vh) < 0.  That matrix is the proper
//               rotation maximising tr(R^T M); here it is obtained WITHOUT an SVD as the eigenvector of
//               the largest eigenvalue of Horn's symmetric 4x4 matrix (cyclic Jacobi, fully unrolled,
//               registers only) -- no branches on singular-value order, reflection handled for free.
//   scale_pts   (lib/d3_utils.py:237-246)  <A,b>/(<A,A>+1e-6) over ALL ordered point pairs.
//   rotate_points_with_rotvec (lib/d3_utils.py:150-163)  Rodrigues formula.
//   scipy.optimize.least_squares(method='lm') = MINPACK lmdif/lmpar (third-party, call sites
//               evaluation/parallel_ancsh_pose.py:151-155): same control flow, forward-difference
//               Jacobian with MINPACK's step rule, but carried on the 6x6 normal matrix A = J^T J and
//               g = J^T f (lmpar's quantities are all expressible through A and g when diag = 1), so a
//               whole solve lives in registers and the m-row Jacobian is never materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// The pose math is float64 with no bit-level contract to mirror (parity bar 1e-4), so let the compiler fuse
// multiply-adds here: it halves the instruction count of the LM loop (the library is otherwise built with
// -ffp-contract=off).  The float32 verifier in pose.hip switches contraction off again locally.
#pragma clang fp contract(fast)

namespace ancsh {
namespace pose {

#define PM_INL __device__ __forceinline__
#define PM_CALL __device__ __attribute__((noinline))

// ---- cheap reciprocals (f64 division / sqrt are 20+ instruction sequences on gfx950) -----------------
// v_rcp_f64 / v_rsq_f64 seed + two Newton steps: full double precision to ~1 ulp, 5 / 9 instructions.
PM_INL double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
PM_INL double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}

// sin/cos for the rotation angles of the fit (|x| stays O(pi); valid to ~1 ulp for |x| < 1e4): two-term
// Cody-Waite reduction by pi/2 + the fdlibm kernel polynomials.  A fraction of the size of the generic
// library sincos (no Payne-Hanek path), which matters because the LM loop must stay I-cache resident.
PM_INL void sincos_compact(double x, double &sn, double &cs) {
    const double k = rint(x * 6.36619772367581382433e-01);
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double s = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                     z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double c = 1.0 - 0.5 * z + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                     z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const int q = (int)k & 3;
    sn = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
    cs = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

// ---- Horn / Kabsch -------------------------------------------------------------------------------
// M[a*3+b] = sum_i tgt_i[a] * src_i[b] (the reference's M = target^T source).  Returns unit
// quaternion (w,x,y,z), w >= 0, of the rotation R (src -> tgt) maximising tr(R^T M).
struct Quat { double w, x, y, z; };
PM_INL void horn_quat_impl(const double M[9], double q[4]) {
    // S[a][b] = sum src_a tgt_b = M[b][a]
    const double Sxx = M[0], Sxy = M[3], Sxz = M[6];
    const double Syx = M[1], Syy = M[4], Syz = M[7];
    const double Szx = M[2], Szy = M[5], Szz = M[8];
    double a[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, dg = 1e-300;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            dg += a[p][p] * a[p][p];
#pragma unroll
            for (int r = p + 1; r < 4; ++r) off += a[p][r] * a[p][r];
        }
        if (off < 1e-30 * dg) break;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                const double apq = a[p][r];
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-150 * (fabs(a[p][p]) + fabs(a[r][r]))) {
                    // Jacobi rotation: t = sgn(th)/(|th| + sqrt(th^2+1)), th = (aqq-app)/(2 apq); c = 1/sqrt(t^2+1)
                    // (reciprocal / rsqrt seeds + Newton instead of IEEE div/sqrt sequences)
                    const double th = (a[r][r] - a[p][p]) * fast_rcp(2.0 * apq);
                    const double h2 = th * th + 1.0;
                    const double t = (th >= 0.0 ? 1.0 : -1.0) * fast_rcp(fabs(th) + h2 * fast_rsqrt(h2));
                    const double c = fast_rsqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double akp = a[k][p], akq = a[k][r];
                        a[k][p] = c * akp - s * akq;
                        a[k][r] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double apk = a[p][k], aqk = a[r][k];
                        a[p][k] = c * apk - s * aqk;
                        a[r][k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double vkp = v[k][p], vkq = v[k][r];
                        v[k][p] = c * vkp - s * vkq;
                        v[k][r] = s * vkp + c * vkq;
                    }
                }
            }
        }
    }
    // column of the largest eigenvalue, selected without dynamic register indexing
    double best = a[0][0];
    double qw = v[0][0], qx = v[1][0], qy = v[2][0], qz = v[3][0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const bool take = a[i][i] > best;
        best = take ? a[i][i] : best;
        qw = take ? v[0][i] : qw;
        qx = take ? v[1][i] : qx;
        qy = take ? v[2][i] : qy;
        qz = take ? v[3][i] : qz;
    }
    double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    nrm = (qw < 0.0 ? -1.0 : 1.0) / nrm;
    q[0] = qw * nrm; q[1] = qx * nrm; q[2] = qy * nrm; q[3] = qz * nrm;
}

PM_CALL Quat horn_quat_call(double m0, double m1, double m2, double m3, double m4, double m5, double m6, double m7, double m8) {
    const double M[9] = {m0, m1, m2, m3, m4, m5, m6, m7, m8};
    double q[4];
    horn_quat_impl(M, q);
    Quat r = {q[0], q[1], q[2], q[3]};
    return r;
}
PM_INL void horn_quat(const double M[9], double q[4]) {
    const Quat r = horn_quat_call(M[0], M[1], M[2], M[3], M[4], M[5], M[6], M[7], M[8]);
    q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
}

// ---- fast Horn for the per-hypothesis 3-point fits (960k of them per batch) ------------------------------------------
// Same quaternion as horn_quat, found without the Jacobi sweeps: the largest root of the characteristic quartic of N by
// Newton from an upper bound (monotone, quadratic: Theobald's QCP, Acta Cryst. A61 (2005) 478), then the eigenvector as a
// column of adj(N - lambda I) -- the column of the largest diagonal cofactor, i.e. of the largest quaternion component.
// ~350 flops against ~4000.  Rank-deficient M (a sample with a repeated or collinear point: the optimal rotation is then
// a one-parameter family and the reference's SVD returns an arbitrary member of it) has a double top root and a vanishing
// adjugate; those take the shortest-arc rotation between the dominant directions instead, also a member of the family.
PM_INL double det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

PM_INL void horn_quat_fast(const double M[9], double q[4]) {
    const double Sxx = M[0], Sxy = M[3], Sxz = M[6];
    const double Syx = M[1], Syy = M[4], Syz = M[7];
    const double Szx = M[2], Szy = M[5], Szz = M[8];
    const double n00 = Sxx + Syy + Szz, n01 = Syz - Szy, n02 = Szx - Sxz, n03 = Sxy - Syx;
    const double n11 = Sxx - Syy - Szz, n12 = Sxy + Syx, n13 = Szx + Sxz;
    const double n22 = -Sxx + Syy - Szz, n23 = Syz + Szy;
    const double n33 = -Sxx - Syy + Szz;
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) fro += M[i] * M[i];
    // lambda^4 + c2 lambda^2 + c1 lambda + c0  (N is traceless: c2 = -2 |M|_F^2, c1 = -8 det M, c0 = det N)
    const double c2 = -2.0 * fro;
    const double c1 = -8.0 * det3(M[0], M[1], M[2], M[3], M[4], M[5], M[6], M[7], M[8]);
    const double c0 = n00 * det3(n11, n12, n13, n12, n22, n23, n13, n23, n33) - n01 * det3(n01, n12, n13, n02, n22, n23, n03, n23, n33) +
                      n02 * det3(n01, n11, n13, n02, n12, n23, n03, n13, n33) - n03 * det3(n01, n11, n12, n02, n12, n22, n03, n13, n23);
    double lam = sqrt(3.0 * fro);             // >= sigma1 + sigma2 + sigma3 >= lambda_max
    for (int it = 0; it < 40; ++it) {
        const double l2 = lam * lam;
        const double f = (l2 + c2) * l2 + c1 * lam + c0;
        const double df = (4.0 * l2 + 2.0 * c2) * lam + c1;
        if (!(df > 0.0)) break;
        const double step = f * fast_rcp(df);
        lam -= step;
        if (fabs(step) <= 1e-15 * lam) break;
    }
    const double b00 = n00 - lam, b11 = n11 - lam, b22 = n22 - lam, b33 = n33 - lam;
    // cofactors of the symmetric B = N - lambda I (adj B = C): diagonal, then the off-diagonals
    const double C00 = det3(b11, n12, n13, n12, b22, n23, n13, n23, b33);
    const double C11 = det3(b00, n02, n03, n02, b22, n23, n03, n23, b33);
    const double C22 = det3(b00, n01, n03, n01, b11, n13, n03, n13, b33);
    const double C33 = det3(b00, n01, n02, n01, b11, n12, n02, n12, b22);
    const double C01 = -det3(n01, n12, n13, n02, b22, n23, n03, n23, b33);
    const double C02 = det3(n01, b11, n13, n02, n12, n23, n03, n13, b33);
    const double C03 = -det3(n01, b11, n12, n02, n12, b22, n03, n13, n23);
    const double C12 = -det3(b00, n01, n03, n02, n12, n23, n03, n13, b33);
    const double C13 = det3(b00, n01, n02, n02, n12, b22, n03, n13, n23);
    const double C23 = -det3(b00, n01, n02, n01, b11, n12, n03, n13, n23);
    const double a0 = fabs(C00), a1 = fabs(C11), a2 = fabs(C22), a3 = fabs(C33);
    double best = a0, qw = C00, qx = C01, qy = C02, qz = C03;
    if (a1 > best) { best = a1; qw = C01; qx = C11; qy = C12; qz = C13; }
    if (a2 > best) { best = a2; qw = C02; qx = C12; qy = C22; qz = C23; }
    if (a3 > best) { best = a3; qw = C03; qx = C13; qy = C23; qz = C33; }
    const double scale3 = fro * sqrt(fro);
    if (!(best > 1e-9 * scale3)) {
        // rank <= 1: M ~ a b^T (a: target direction, b: source direction); shortest arc taking b to a
        double mx = 0.0;
        int bi = 0, bj = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (fabs(M[i * 3 + j]) > mx) { mx = fabs(M[i * 3 + j]); bi = i; bj = j; }
        if (!(mx > 0.0)) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; return; }
        double a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            a[t] = bj == 0 ? M[t * 3] : (bj == 1 ? M[t * 3 + 1] : M[t * 3 + 2]);
            b[t] = bi == 0 ? M[t] : (bi == 1 ? M[3 + t] : M[6 + t]);
        }
        const double na = fast_rsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = fast_rsqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
#pragma unroll
        for (int t = 0; t < 3; ++t) { a[t] *= na; b[t] *= nb; }
        const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        if (d > -1.0 + 1e-12) {
            qw = 1.0 + d; qx = b[1] * a[2] - b[2] * a[1]; qy = b[2] * a[0] - b[0] * a[2]; qz = b[0] * a[1] - b[1] * a[0];
        } else {                                                   // opposite directions: half turn about any axis normal to b
            const bool ux = fabs(b[0]) < 0.9;
            const double e0 = ux ? 1.0 : 0.0, e1 = ux ? 0.0 : 1.0;
            qw = 0.0; qx = b[1] * 0.0 - b[2] * e1; qy = b[2] * e0 - b[0] * 0.0; qz = b[0] * e1 - b[1] * e0;
        }
    }
    double nrm = fast_rsqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    nrm = qw < 0.0 ? -nrm : nrm;
    q[0] = qw * nrm; q[1] = qx * nrm; q[2] = qy * nrm; q[3] = qz * nrm;
}

PM_INL void quat_to_mat(const double q[4], double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// scipy Rotation.as_rotvec: angle = 2*atan2(|q_xyz|, q_w) (w >= 0), rotvec = angle/sin(angle/2) * q_xyz
PM_INL void quat_to_rotvec(const double q[4], double rv[3]) {
    const double sn = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double ang = 2.0 * atan2(sn, q[0]);
    double sc;
    if (ang <= 1e-3) {
        const double a2 = ang * ang;
        sc = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
    } else {
        sc = ang / sin(0.5 * ang);
    }
    rv[0] = sc * q[1]; rv[1] = sc * q[2]; rv[2] = sc * q[3];
}

// scipy Rotation.from_rotvec(...).as_matrix() == Rodrigues' rotation matrix
PM_INL void rotvec_to_mat(const double rv[3], double R[9]) {
    const double th = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    double q[4];
    double sc;
    if (th <= 1e-3) {
        const double a2 = th * th;
        sc = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
    } else {
        sc = sin(0.5 * th) / th;
    }
    q[0] = cos(0.5 * th); q[1] = sc * rv[0]; q[2] = sc * rv[1]; q[3] = sc * rv[2];
    quat_to_mat(q, R);
}

// Rodrigues rotation prepared once per rotation vector (lib/d3_utils.py:150-163):
// out = cos*p + sin*(v x p) + (p.v)(1-cos) v,  v = rv/|rv| (0 when |rv| = 0).
struct Rod {
    double c, s, vx, vy, vz;
};
PM_CALL Rod rod_prepare(double rx, double ry, double rz) {
    Rod r;
    const double th = sqrt(rx * rx + ry * ry + rz * rz);
    const double inv = th > 0.0 ? fast_rcp(th) : 0.0;
    r.vx = rx * inv; r.vy = ry * inv; r.vz = rz * inv;
    sincos_compact(th, r.s, r.c);
    return r;
}
PM_INL void rod_apply(const Rod &r, double px, double py, double pz, double &ox, double &oy, double &oz) {
    const double d = (px * r.vx + py * r.vy + pz * r.vz) * (1.0 - r.c);
    ox = r.c * px + r.s * (r.vy * pz - r.vz * py) + d * r.vx;
    oy = r.c * py + r.s * (r.vz * px - r.vx * pz) + d * r.vy;
    oz = r.c * pz + r.s * (r.vx * py - r.vy * px) + d * r.vz;
}

// ---- 6x6 symmetric positive-definite solves (normal equations of the LM step) -------------------
// Symmetric matrices are stored packed, lower triangle row by row: S(i,j) = i(i+1)/2 + j, i >= j (21 doubles).
#define PM_S(i, j) ((i) >= (j) ? (i) * ((i) + 1) / 2 + (j) : (j) * ((j) + 1) / 2 + (i))

// Cholesky of (A + par I): L lower (packed), with the RECIPROCAL of each diagonal entry stored on the
// diagonal (the solves then need no division).  Returns false when a pivot is not positive.
PM_INL bool chol6(const double A[21], double par, double L[21]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = A[PM_S(j, j)] + par;
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[PM_S(j, k)] * L[PM_S(j, k)];
        ok = ok && (s > 0.0);
        const double r = fast_rsqrt(s);
        L[PM_S(j, j)] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[PM_S(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[PM_S(i, k)] * L[PM_S(j, k)];
            L[PM_S(i, j)] = t * r;
        }
    }
    return ok;
}
PM_INL void chol6_solve(const double L[21], const double b[6], double x[6]) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[PM_S(i, k)] * y[k];
        y[i] = s * L[PM_S(i, i)];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[PM_S(k, i)] * x[k];
        x[i] = s * L[PM_S(i, i)];
    }
}
PM_INL double norm6(const double v[6]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}
PM_INL double dot6(const double a[6], const double b[6]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += a[i] * b[i];
    return s;
}

// MINPACK lmpar on the normal matrix (diag = 1): find par >= 0 and z with (A + par I) z = g and
// | ||z|| - delta | <= 0.1 delta (or par = 0 if the Gauss-Newton step is inside the trust region).
// One Cholesky site and one pair of solves serve both the Gauss-Newton probe (it = 0, par = 0) and the
// Newton iterations on par (it >= 1).
PM_INL void lmpar6(const double A[21], const double g[6], double delta, double &par, double z[6], int *nchol = nullptr) {
    const double dwarf = 2.2250738585072014e-308;
    const double gn = norm6(g);
    double parl = 0.0, paru = 0.0, fp = 0.0, cur = 0.0;
    for (int it = 0;; ++it) {
        if (it > 0 && cur == 0.0) cur = fmax(dwarf, 0.001 * paru);
        double L[21];
        const bool ok = chol6(A, cur, L);
        if (nchol) ++*nchol;
        double dxnorm, wy = 1.0;
        if (ok) {
            chol6_solve(L, g, z);
            dxnorm = norm6(z);
            double w[6], y[6];
            const double rd = fast_rcp(dxnorm);
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = z[i] * rd;
            chol6_solve(L, w, y);
            wy = dot6(w, y);                     // w^T (A + par I)^-1 w
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) z[i] = 0.0;
            dxnorm = INFINITY;
        }
        const double temp = fp;
        fp = dxnorm - delta;
        if (it == 0) {
            if (fp <= 0.1 * delta) { par = 0.0; return; }
            if (ok) parl = (fp / delta) / wy;
            paru = gn / delta;
            if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
            cur = fmin(fmax(par, parl), paru);
            if (cur == 0.0) cur = gn / dxnorm;
            continue;
        }
        if (!ok) { cur = fmax(2.0 * cur, 1e-300); if (it < 10) continue; break; }
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || it == 10) break;
        const double parc = (fp / delta) / wy;
        if (fp > 0.0) parl = fmax(parl, cur);
        if (fp < 0.0) paru = fmin(paru, cur);
        cur = fmax(parl, cur + parc);
    }
    par = cur;
}

// MINPACK lmdif driver (mode 2: diag = 1; factor 100; forward differences with epsfcn = EPS).
// Problem P supplies, for the 6-vector x,
//     double cost(const double x[6])                    -> sum of squared residuals
//     void   normal(const double x[6], double A[21], double g[6])  -> J^T J (packed), J^T f with MINPACK's
//                                                           forward-difference J at x
// Both may be workgroup-cooperative as long as every calling thread receives identical results.
// The driver is written as a resumable state machine (begin + one trip of MINPACK's inner loop per call) so that a
// wave of independent thread-local problems can hand a finished lane its next problem between trips instead of idling
// until the slowest lane of the wave converges (ransac_joint_lm_kernel); lmdif6 below is the plain loop over it.
struct Lm6 {
    double x[6], A[21], g[6];
    double fnorm, par, delta, xnorm, gnorm;
    int nfev, info, iter;
    bool need_normal;
#ifdef LM_COUNT               // diagnostic build only (scratch/lm_count.py): trips of the loop body and Cholesky factorisations of a fit
    int trips, nchol;
#endif
};

template <class P>
PM_INL void lm6_begin(P &prob, Lm6 &s) {
    s.fnorm = sqrt(prob.cost(s.x));
    s.nfev = 1; s.info = 0; s.iter = 1;
    s.par = 0.0; s.delta = 0.0; s.xnorm = 0.0; s.gnorm = 0.0;
    s.need_normal = true;
#ifdef LM_COUNT
    s.trips = s.nchol = 0;
#endif
}

// one pass of the lmdif loop body; returns true when the run has terminated (s.info set)
template <class P>
PM_INL bool lm6_trip(P &prob, Lm6 &s, double ftol, double xtol, double gtol, int maxfev) {
    const double epsmch = 2.220446049250313e-16, factor = 100.0;
    if (s.need_normal) {          // outer iteration of lmdif: new Jacobian at the accepted point
        prob.normal(s.x, s.A, s.g);
        s.nfev += 6;
        if (s.iter == 1) {
            s.xnorm = norm6(s.x);
            s.delta = factor * s.xnorm;
            if (s.delta == 0.0) s.delta = factor;
        }
        s.gnorm = 0.0;
        if (s.fnorm != 0.0) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (s.A[PM_S(j, j)] != 0.0) s.gnorm = fmax(s.gnorm, fabs(s.g[j] / s.fnorm) * fast_rsqrt(s.A[PM_S(j, j)]));
        }
        if (s.gnorm <= gtol) { s.info = 4; return true; }
        s.need_normal = false;
    }
    double z[6], xn[6];
#ifdef LM_COUNT
    ++s.trips;
    lmpar6(s.A, s.g, s.delta, s.par, z, &s.nchol);
#else
    lmpar6(s.A, s.g, s.delta, s.par, z);
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = s.x[i] - z[i];
    const double pnorm = norm6(z);
    if (s.iter == 1) s.delta = fmin(s.delta, pnorm);
    const double fnorm1 = sqrt(prob.cost(xn));
    ++s.nfev;
    double actred = -1.0;
    if (0.1 * fnorm1 < s.fnorm) { const double r = fnorm1 / s.fnorm; actred = 1.0 - r * r; }
    double pAp = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sv = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) sv += s.A[PM_S(i, j)] * z[j];
        pAp += z[i] * sv;
    }
    const double rf = 1.0 / s.fnorm;
    const double temp1 = sqrt(fmax(pAp, 0.0)) * rf, temp2 = sqrt(s.par) * pnorm * rf;
    const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
    const double dirder = -(temp1 * temp1 + temp2 * temp2);
    const double ratio = prered != 0.0 ? actred / prered : 0.0;
    if (ratio <= 0.25) {
        double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= s.fnorm || temp < 0.1) temp = 0.1;
        s.delta = temp * fmin(s.delta, pnorm / 0.1);
        s.par = s.par / temp;
    } else if (s.par == 0.0 || ratio >= 0.75) {
        s.delta = pnorm / 0.5;
        s.par = 0.5 * s.par;
    }
    if (ratio >= 1e-4) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.x[i] = xn[i];
        s.xnorm = norm6(s.x);
        s.fnorm = fnorm1;
        ++s.iter;
        s.need_normal = true;
    }
    const bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
    if (small) s.info = 1;
    if (s.delta <= xtol * s.xnorm) s.info = 2;
    if (small && s.info == 2) s.info = 3;
    if (s.info != 0) return true;
    if (s.nfev >= maxfev) s.info = 5;
    if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) s.info = 6;
    if (s.delta <= epsmch * s.xnorm) s.info = 7;
    if (s.gnorm <= epsmch) s.info = 8;
    return s.info != 0;
}

template <class P>
PM_INL int lmdif6(P &prob, double x[6], double ftol, double xtol, double gtol, int maxfev, int *nfev_out) {
    Lm6 s;
#pragma unroll
    for (int i = 0; i < 6; ++i) s.x[i] = x[i];
    lm6_begin(prob, s);
    while (!lm6_trip(prob, s, ftol, xtol, gtol, maxfev)) {}
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = s.x[i];
    if (nfev_out) *nfev_out = s.nfev;
    return s.info;
}

PM_INL double fd_step(double xj) {
    const double eps = 1.4901161193847656e-08;   // sqrt(max(epsfcn = EPS, epsmch))
    const double h = eps * fabs(xj);
    return h == 0.0 ? eps : h;
}

#undef PM_INL
#undef PM_CALL
}  // namespace pose
}  // namespace ancsh
