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

namespace ancsh {
namespace pose {

#define PM_INL __device__ __forceinline__

// ---- Horn / Kabsch -------------------------------------------------------------------------------
// M[a*3+b] = sum_i tgt_i[a] * src_i[b] (the reference's M = target^T source).  Returns unit
// quaternion (w,x,y,z), w >= 0, of the rotation R (src -> tgt) maximising tr(R^T M).
PM_INL void horn_quat(const double M[9], double q[4]) {
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
        if (off < 1e-32 * dg) break;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                const double apq = a[p][r];
                if (fabs(apq) > 1e-300) {
                    const double th = (a[r][r] - a[p][p]) / (2.0 * apq);
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
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
PM_INL Rod rod_prepare(double rx, double ry, double rz) {
    Rod r;
    const double th = sqrt(rx * rx + ry * ry + rz * rz);
    const double inv = th > 0.0 ? 1.0 / th : 0.0;
    r.vx = rx * inv; r.vy = ry * inv; r.vz = rz * inv;
    sincos(th, &r.s, &r.c);
    return r;
}
PM_INL void rod_apply(const Rod &r, double px, double py, double pz, double &ox, double &oy, double &oz) {
    const double d = (px * r.vx + py * r.vy + pz * r.vz) * (1.0 - r.c);
    ox = r.c * px + r.s * (r.vy * pz - r.vz * py) + d * r.vx;
    oy = r.c * py + r.s * (r.vz * px - r.vx * pz) + d * r.vy;
    oz = r.c * pz + r.s * (r.vx * py - r.vy * px) + d * r.vz;
}

// ---- 6x6 symmetric positive-definite solves (normal equations of the LM step) -------------------
// A stored full row-major 6x6.  Returns false when a pivot is not positive.
PM_INL bool chol6(const double A[36], double par, double L[36]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j] + (i == j ? par : 0.0);
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i * 6 + i] = sqrt(s);
            } else {
                L[i * 6 + j] = s / L[j * 6 + j];
            }
        }
    }
    return true;
}
PM_INL void chol6_solve(const double L[36], const double b[6], double x[6]) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
}
PM_INL double norm6(const double v[6]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}

// MINPACK lmpar on the normal matrix (diag = 1): find par >= 0 and z with (A + par I) z = g and
// | ||z|| - delta | <= 0.1 delta (or par = 0 if the Gauss-Newton step is inside the trust region).
PM_INL void lmpar6(const double A[36], const double g[6], double delta, double &par, double z[6]) {
    const double dwarf = 2.2250738585072014e-308;
    double L[36];
    const bool ok = chol6(A, 0.0, L);
    double dxnorm;
    if (ok) {
        chol6_solve(L, g, z);
        dxnorm = norm6(z);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = 0.0;
        dxnorm = INFINITY;
    }
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0.0; return; }
    double parl = 0.0;
    if (ok) {
        double w[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = z[i] / dxnorm;
        chol6_solve(L, w, y);
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) t += w[i] * y[i];
        parl = (fp / delta) / t;
    }
    const double gn = norm6(g);
    double paru = gn / delta;
    if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gn / dxnorm;
    for (int it = 1;; ++it) {
        if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
        if (!chol6(A, par, L)) { par = fmax(2.0 * par, 1e-300); if (it < 10) continue; break; }
        chol6_solve(L, g, z);
        dxnorm = norm6(z);
        const double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || it == 10) break;
        double w[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = z[i] / dxnorm;
        chol6_solve(L, w, y);
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) t += w[i] * y[i];
        const double parc = (fp / delta) / t;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// MINPACK lmdif driver (mode 2: diag = 1; factor 100; forward differences with epsfcn = EPS).
// Problem P supplies, for the 6-vector x,
//     double cost(const double x[6])                    -> sum of squared residuals
//     void   normal(const double x[6], double A[36], double g[6])  -> J^T J, J^T f with MINPACK's
//                                                           forward-difference J at x
// Both may be workgroup-cooperative as long as every calling thread receives identical results.
template <class P>
PM_INL int lmdif6(P &prob, double x[6], double ftol, double xtol, double gtol, int maxfev, int *nfev_out) {
    const double epsmch = 2.220446049250313e-16, factor = 100.0;
    double fnorm = sqrt(prob.cost(x));
    int nfev = 1, info = 0, iter = 1;
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    double A[36], g[6];
    for (;;) {
        prob.normal(x, A, g);
        nfev += 6;
        if (iter == 1) {
            xnorm = norm6(x);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        double gnorm = 0.0;
        if (fnorm != 0.0) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (A[j * 6 + j] != 0.0) gnorm = fmax(gnorm, fabs(g[j] / fnorm) / sqrt(A[j * 6 + j]));
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (;;) {
            double z[6], xn[6];
            lmpar6(A, g, delta, par, z);
#pragma unroll
            for (int i = 0; i < 6; ++i) xn[i] = x[i] - z[i];
            const double pnorm = norm6(z);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnorm1 = sqrt(prob.cost(xn));
            ++nfev;
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) { const double r = fnorm1 / fnorm; actred = 1.0 - r * r; }
            double pAp = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) s += A[i * 6 + j] * z[j];
                pAp += z[i] * s;
            }
            const double temp1 = sqrt(fmax(pAp, 0.0)) / fnorm, temp2 = sqrt(par) * pnorm / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            const double ratio = prered != 0.0 ? actred / prered : 0.0;
            if (ratio <= 0.25) {
                double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = xn[i];
                xnorm = norm6(x);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (small) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (small && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= epsmch * xnorm) info = 7;
            if (gnorm <= epsmch) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    if (nfev_out) *nfev_out = nfev;
    return info;
}

PM_INL double fd_step(double xj) {
    const double eps = 1.4901161193847656e-08;   // sqrt(max(epsfcn = EPS, epsmch))
    const double h = eps * fabs(xj);
    return h == 0.0 ? eps : h;
}

// Accumulate one residual triple into A, g.  j0: first parameter this residual depends on (0 or 3);
// a[c][p] = d f_c / d x_{j0+p} (3x3), f[c] the residual, w its multiplicity.
PM_INL void acc_block(double A[36], double g[6], int j0, const double a[3][3], const double f[3], double w) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) s += a[c][p] * a[c][r];
            A[(j0 + p) * 6 + j0 + r] += w * s;
        }
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) s += a[c][p] * f[c];
        g[j0 + p] += w * s;
    }
}

#undef PM_INL
}  // namespace pose
}  // namespace ancsh
