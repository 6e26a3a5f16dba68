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

}  // namespace midas
