// cluster_rot.hpp - the rotation part of a cluster centre (Markley's quaternion mean, modules/pose.py:112-147): the
// eigenvector of the largest eigenvalue of sum w q q^T / sum w by cyclic Jacobi in float64, written as a rotation matrix.
// Shared by cluster.hip (midas_cluster_centers) and loop.hip (the loop step computes it beside the annealing).
#pragma once
#include "midas_internal.hpp"

namespace midas {

// cyclic Jacobi on a symmetric 4x4 (float64): A -> diag, V = eigenvectors (columns).
// Every index is a compile-time constant (the loops over the matrix are fully unrolled, the eigenvector is picked with
// selects): A and V live in registers.  With indexed arrays they were 272 bytes of scratch memory per lane and every element a
// memory operation - the loop step's annealing kernel spent 30 - 50 us in here (one lane per cluster, beside the selection).
MD void jacobi4(double A[4][4], double V[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
#ifndef MIDAS_JACOBI_SWEEPS
#define MIDAS_JACOBI_SWEEPS 32
#endif
#pragma unroll 1
    for (int sweep = 0; sweep < MIDAS_JACOBI_SWEEPS; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dia += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        }
        // converged when the off-diagonal mass is below rounding of the diagonal (the eigenvector error is of the order
        // sqrt(off) / gap: 1e-15 here, far below the float32 the result is rounded to)
        if (off < 1e-40 || off < 1e-30 * dia) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                if (__builtin_fabs(A[p][q]) < 1e-300 || A[p][q] * A[p][q] < 1e-34 * dia) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
                const double c = 1.0 / __builtin_sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// A10 = upper triangle of the normalised moment matrix (rows x, y, z, w); out = row-major 4x4 pose: the nine rotation entries
// are written, stride 4
MD void cluster_rotation_write(const double* A10, float* out) {
    double A[4][4], V[4][4];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) { A[i][j] = A[j][i] = A10[k]; ++k; }
    }
    jacobi4(A, V);
    // the column of the largest eigenvalue (first one on ties), by selects
    double top = A[0][0], qx = V[0][0], qy = V[1][0], qz = V[2][0], qw = V[3][0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const bool up = A[i][i] > top;
        top = up ? A[i][i] : top;
        qx = up ? V[0][i] : qx; qy = up ? V[1][i] : qy; qz = up ? V[2][i] : qz; qw = up ? V[3][i] : qw;
    }
    if (qw < 0.0) { qx = -qx; qy = -qy; qz = -qz; qw = -qw; }  // :139
    const double n = __builtin_sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx /= n; qy /= n; qz /= n; qw /= n;
    out[0] = (float)(1.0 - 2.0 * (qy * qy + qz * qz)); out[1] = (float)(2.0 * (qx * qy - qz * qw)); out[2] = (float)(2.0 * (qx * qz + qy * qw));
    out[4] = (float)(2.0 * (qx * qy + qz * qw)); out[5] = (float)(1.0 - 2.0 * (qx * qx + qz * qz)); out[6] = (float)(2.0 * (qy * qz - qx * qw));
    out[8] = (float)(2.0 * (qx * qz - qy * qw)); out[9] = (float)(2.0 * (qy * qz + qx * qw)); out[10] = (float)(1.0 - 2.0 * (qx * qx + qy * qy));
}

}  // namespace midas
