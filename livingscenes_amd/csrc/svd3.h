// svd3.h -- 3x3 SVD / Kabsch rotation in registers (shared by match.hip and icp.hip)
#pragma once
#include "ls_common.h"

namespace ls {

// 3x3 SVD by one-sided (Hestenes) Jacobi in fp64: A V = U S.  Returns the rotation
// R = V diag(1,1,det(V U^T)) U^T of pose_estimation.py:90-94 written through the two dominant singular pairs:
// R = v_a u_a^T + v_b u_b^T + (v_a x v_b)(u_a x u_b)^T  (identical, and well defined for rank-2 covariances).
// Return code (LS_KABSCH_* in the header): 0 = rank >= 2 (the rotation is unique); 1 = rank 1: the least-squares rotation is
// any rotation taking u_a to v_a -- the one of smallest angle is returned (torch.svd succeeds here too and returns one member
// of the same family, pose_estimation.py:79-94); 2 = zero covariance: identity (LAPACK returns U = V = I); 3 = non-finite
// input: identity -- the only case in which the reference's try/except branch (:79-88) is taken.
__device__ __forceinline__ int kabsch_rotation(const double H[9], float R[9]) {
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { A[i][j] = H[i * 3 + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
                off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
                if (fabs(ga) <= 1e-18 * sqrt(al * be) || ga == 0.0) continue;
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double sg[3];
    for (int j = 0; j < 3; ++j) sg[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    int lo = 0;
    if (sg[1] < sg[lo]) lo = 1;
    if (sg[2] < sg[lo]) lo = 2;
    const int a = (lo + 1) % 3, b = (lo + 2) % 3;
    auto identity = [&]() { for (int e = 0; e < 9; ++e) R[e] = (e % 4 == 0) ? 1.f : 0.f; };
    if (!isfinite(sg[0]) || !isfinite(sg[1]) || !isfinite(sg[2])) { identity(); return 3; }
    auto rank1 = [&](int c) {   // smallest rotation taking u = A[:,c]/sg to v = V[:,c]  (Rodrigues about u x v)
        double u[3], v[3];
        for (int i = 0; i < 3; ++i) { u[i] = A[i][c] / sg[c]; v[i] = V[i][c]; }
        const double w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const double cs = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
        if (cs < -1.0 + 1e-12) {   // u = -v: half turn about any axis orthogonal to u
            double ax[3] = {0, 0, 0};
            int k = fabs(u[0]) < fabs(u[1]) ? (fabs(u[0]) < fabs(u[2]) ? 0 : 2) : (fabs(u[1]) < fabs(u[2]) ? 1 : 2);
            ax[k] = 1.0;
            double d = ax[0] * u[0] + ax[1] * u[1] + ax[2] * u[2], nn = 0;
            for (int i = 0; i < 3; ++i) { ax[i] -= d * u[i]; nn += ax[i] * ax[i]; }
            nn = sqrt(nn);
            for (int i = 0; i < 3; ++i) ax[i] /= nn;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)(2.0 * ax[i] * ax[j] - (i == j ? 1.0 : 0.0));
            return;
        }
        const double k1 = 1.0 / (1.0 + cs);
        const double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = 0;
                for (int l = 0; l < 3; ++l) kk += K[i][l] * K[l][j];
                R[i * 3 + j] = (float)((i == j ? 1.0 : 0.0) + K[i][j] + k1 * kk);
            }
    };
    if (!(sg[a] > 1e-150) || !(sg[b] > 1e-150)) {   // rank < 2
        const int hi = sg[a] >= sg[b] ? a : b;
        if (!(sg[hi] > 1e-150)) { identity(); return 2; }
        rank1(hi);
        return 1;
    }
    double ua[3], ub[3], va[3], vb[3];
    for (int i = 0; i < 3; ++i) { ua[i] = A[i][a] / sg[a]; ub[i] = A[i][b] / sg[b]; va[i] = V[i][a]; vb[i] = V[i][b]; }
    // re-orthogonalise u_b against u_a (exact in theory; guards the near-degenerate case)
    double dot = ua[0] * ub[0] + ua[1] * ub[1] + ua[2] * ub[2];
    for (int i = 0; i < 3; ++i) ub[i] -= dot * ua[i];
    double nb = sqrt(ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2]);
    if (!(nb > 1e-150)) { rank1(a); return 1; }
    for (int i = 0; i < 3; ++i) ub[i] /= nb;
    const double uc[3] = {ua[1] * ub[2] - ua[2] * ub[1], ua[2] * ub[0] - ua[0] * ub[2], ua[0] * ub[1] - ua[1] * ub[0]};
    const double vc[3] = {va[1] * vb[2] - va[2] * vb[1], va[2] * vb[0] - va[0] * vb[2], va[0] * vb[1] - va[1] * vb[0]};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)(va[i] * ua[j] + vb[i] * ub[j] + vc[i] * uc[j]);
    return 0;
}


}  // namespace ls
