// cluster_rot.hpp - the rotation part of a cluster centre (Markley's quaternion mean, modules/pose.py:112-147): the
// eigenvector of the largest eigenvalue of sum w q q^T / sum w by repeated squaring in float64, written as a rotation matrix.
// Shared by cluster.hip (midas_cluster_centers) and loop.hip (the loop step computes it beside the annealing).
#pragma once
#include "midas_internal.hpp"

namespace midas {

// The eigenvector of the largest eigenvalue of a symmetric positive semi-definite 4x4 (float64) by repeated squaring:
// B <- B B (rescaled by a power of two, exact) squares every eigenvalue ratio, so after k rounds B is the outer product of the
// wanted vector with itself up to (lambda_2 / lambda_1)^(2^k); the loop ends when trace(B B) = |B|_F^2 reaches trace(B)^2 to
// 1e-12 (then the ratio is below 5e-13 - the vector is rounded to float32 afterwards) or after 44 rounds (gaps below 1e-11 of
// the top eigenvalue: a degenerate mean, arbitrary under any method).  One round is ten 4-term dot products that do not depend
// on each other - ~0.15 us for one lane, three to six rounds for a gathered cluster.  (Rounds 2 - 4 ran a cyclic Jacobi here:
// two square roots and three divisions per rotation, one after the other, six rotations a sweep, four sweeps: 11 - 12 us on
// the one lane a cluster has - the floor of the loop step's annealing launch whenever clusters exist.)
MD void top_eigvec4(const double* A10, double& qx, double& qy, double& qz, double& qw) {
    double b00 = A10[0], b01 = A10[1], b02 = A10[2], b03 = A10[3], b11 = A10[4], b12 = A10[5], b13 = A10[6], b22 = A10[7], b23 = A10[8],
           b33 = A10[9];
    double tb = (b00 + b11) + (b22 + b33);
#pragma unroll 1
    for (int it = 0; it < 44; ++it) {
        const double c00 = (b00 * b00 + b01 * b01) + (b02 * b02 + b03 * b03);
        const double c01 = (b00 * b01 + b01 * b11) + (b02 * b12 + b03 * b13);
        const double c02 = (b00 * b02 + b01 * b12) + (b02 * b22 + b03 * b23);
        const double c03 = (b00 * b03 + b01 * b13) + (b02 * b23 + b03 * b33);
        const double c11 = (b01 * b01 + b11 * b11) + (b12 * b12 + b13 * b13);
        const double c12 = (b01 * b02 + b11 * b12) + (b12 * b22 + b13 * b23);
        const double c13 = (b01 * b03 + b11 * b13) + (b12 * b23 + b13 * b33);
        const double c22 = (b02 * b02 + b12 * b12) + (b22 * b22 + b23 * b23);
        const double c23 = (b02 * b03 + b12 * b13) + (b22 * b23 + b23 * b33);
        const double c33 = (b03 * b03 + b13 * b13) + (b23 * b23 + b33 * b33);
        const double tc = (c00 + c11) + (c22 + c33);
        const bool done = tc >= (1.0 - 1e-12) * (tb * tb);  // (false on NaN: the full count, NaN out)
        int ex;
        (void)__builtin_frexp(tc, &ex);
        const double sc = __builtin_ldexp(1.0, -ex);  // exact rescaling: trace in [0.5, 1)
        b00 = c00 * sc; b01 = c01 * sc; b02 = c02 * sc; b03 = c03 * sc; b11 = c11 * sc; b12 = c12 * sc; b13 = c13 * sc;
        b22 = c22 * sc; b23 = c23 * sc; b33 = c33 * sc;
        tb = tc * sc;
        if (done || !(tc > 0.0)) break;
    }
    // the column with the largest diagonal entry (the best conditioned one; first on ties)
    double top = b00;
    qx = b00; qy = b01; qz = b02; qw = b03;
    if (b11 > top) { top = b11; qx = b01; qy = b11; qz = b12; qw = b13; }
    if (b22 > top) { top = b22; qx = b02; qy = b12; qz = b22; qw = b23; }
    if (b33 > top) { top = b33; qx = b03; qy = b13; qz = b23; qw = b33; }
}

// A10 = upper triangle of the normalised moment matrix (rows x, y, z, w); out = row-major 4x4 pose: the nine rotation entries
// are written, stride 4
MD void cluster_rotation_write(const double* A10, float* out) {
    double qx, qy, qz, qw;
    top_eigvec4(A10, qx, qy, qz, qw);
    if (qw < 0.0) { qx = -qx; qy = -qy; qz = -qz; qw = -qw; }  // :139
    const double n = __builtin_sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx /= n; qy /= n; qz /= n; qw /= n;
    out[0] = (float)(1.0 - 2.0 * (qy * qy + qz * qz)); out[1] = (float)(2.0 * (qx * qy - qz * qw)); out[2] = (float)(2.0 * (qx * qz + qy * qw));
    out[4] = (float)(2.0 * (qx * qy + qz * qw)); out[5] = (float)(1.0 - 2.0 * (qx * qx + qz * qz)); out[6] = (float)(2.0 * (qy * qz - qx * qw));
    out[8] = (float)(2.0 * (qx * qz - qy * qw)); out[9] = (float)(2.0 * (qy * qz + qx * qw)); out[10] = (float)(1.0 - 2.0 * (qx * qx + qy * qy));
}

// ---- the moments of a cluster and what is made of them (cluster.hip's finishing kernels)
constexpr int CL_MOM = 36;  // moments per cluster, see the enum
enum : int {
    M_SW = 0,      // sum w
    M_CNT = 1,     // members
    M_WMAX = 2,    // max w (float32 values)
    M_WMIN = 3,    // min w
    M_QQW = 4,     // 10: upper triangle of sum w q q^T, q = (x, y, z, w)
    M_QQ1 = 14,    // 10: the same with w = 1
    M_TW = 24,     // 3: sum w t
    M_T1 = 27,     // 3: sum t
    M_TTW = 30,    // 3: sum w t^2
    M_TT1 = 33,    // 3: sum t^2
};

// One cluster from its CL_MOM summed moments s_m (one thread): member count, centre row `out` (16 floats), spreads `sd` (3 floats).
// rot_out (10 doubles; loop step): the normalised moment matrix goes there and the rotation entries of the centre are left to
// whoever solves it (cluster_rotation_write, beside the annealing); nullptr: solved here.
MD void cluster_close(const double* s_m, float* out, float* sd, int64_t* count, double* rot_out) {
    if (count) *count = (int64_t)s_m[M_CNT];
    if (s_m[M_CNT] == 0.0) {  // empty cluster (the caller passed a label nobody carries): NaN like a 0/0 mean
        for (int i = 0; i < 16; ++i) out[i] = NAN;
        for (int i = 0; i < 3; ++i) sd[i] = NAN;
        return;
    }
    // torch.isclose(max - min, 0): |d| <= atol (1e-8), in the float32 arithmetic of the reference
    const float d = (float)s_m[M_WMAX] - (float)s_m[M_WMIN];
    const bool flat = __builtin_fabsf(d) <= 1e-8f;
    const int oq = flat ? M_QQ1 : M_QQW, ot = flat ? M_T1 : M_TW, ott = flat ? M_TT1 : M_TTW;
    const double sw = flat ? s_m[M_CNT] : s_m[M_SW];
    double A10[10];
    for (int k = 0; k < 10; ++k) A10[k] = s_m[oq + k] / sw;
    if (rot_out) {
        for (int k = 0; k < 10; ++k) rot_out[k] = A10[k];
    } else {
        cluster_rotation_write(A10, out);
    }
    float mean[3];
    for (int i = 0; i < 3; ++i) mean[i] = (float)(s_m[ot + i] / sw);
    out[3] = mean[0]; out[7] = mean[1]; out[11] = mean[2];
    out[12] = 0.f; out[13] = 0.f; out[14] = 0.f; out[15] = 1.f;
    // sum w (t - m)^2 / sum w with m the float32 centre, from the moments
    for (int i = 0; i < 3; ++i) {
        const double m = (double)mean[i];
        double var = (s_m[ott + i] - 2.0 * m * s_m[ot + i] + m * m * sw) / sw;
        var = var < 0.0 ? 0.0 : var;
        sd[i] = (float)__builtin_sqrt(var);
    }
}

}  // namespace midas
