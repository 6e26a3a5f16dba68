/*
 * midas_oracle.c - CPU ORACLE for the MidasTouch particle-filter hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (midastouch_amd/) never imports, links
 * or calls anything under oracle/ and fails loudly when its HIP library is missing.
 *
 * It restates, in scalar C, the algorithm of the reference's per-step filter loop
 * (reference = facebookresearch/MidasTouch, paths relative to /root/reference/midastouch):
 *
 *   mo_propagate        modules/particle_filter.py:319-345 (add_noise_to_odom) + :370-375 (motionModel
 *                       compose) + modules/pose.py:215-269 (euler_angles_to_matrix "ZYX")
 *   mo_se3_feature      tactile_tree/tactile_tree.py:73-77 (R3_SE3) + modules/pose.py:19-23
 *                       (theseus SO3.log_map - third-party, absent from the checkout, unpinned)
 *   mo_nn6              tactile_tree/tactile_tree.py:43-58 (SE3_NN: pynanoflann 0.0.9 exact 1-NN, L2)
 *   mo_knn6             the same call with n_neighbors = k
 *   mo_nn3              modules/particle_filter.py:386-391 (sklearn KDTree.query k=1)
 *   mo_score*           modules/particle_filter.py:455-457 (cosine_similarity, eps 1e-8)
 *   mo_softmax          modules/particle_filter.py:459-468 (isclose guard + Softmax(dim=0))
 *   mo_cdf / searches   modules/particle_filter.py:237-261,295-303 (normalise, multinomial == inverse-CDF
 *                       lower_bound on float64; low_var two-pointer loop == upper_bound)
 *   mo_rmse             modules/particle_filter.py:472-496, modules/pose.py:178-208
 *   mo_dbscan           modules/particle_filter.py:208-217 (sklearn DBSCAN on the translations - third-party,
 *                       pinned by fixture G9 written by the reference's own cluster_particles)
 *
 * Pinning: oracle/oracle.py wraps these functions; tests/test_oracle_golden.py checks them against
 * the golden fixtures in tests/golden/ that tools/gen_goldens.py produced by running the real
 * reference functions (G1-G8, G10).  Pieces whose reference implementation lives in third-party
 * packages that are not in the checkout (theseus SO3.log_map, pynanoflann tie order) are pinned
 * against scipy (Rotation.as_rotvec, cKDTree) instead and are "parity unpinned" w.r.t. the reference.
 *
 * ARITHMETIC SPEC.  float32 work that decides an index (the propagated pose, the 6-d feature, the
 * NN distance) is written with explicit fmaf() chains and self-contained polynomial sin/cos/atan2/log
 * so that the HIP kernels - which restate the same spec independently - agree bit for bit; the
 * float64 reductions use the fixed blocked order documented at mo_blocked_scan.  Build with
 * -ffp-contract=off (see oracle/Makefile) so the compiler adds no contraction of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* float32 elementary functions (spec)                                                        */
/* ------------------------------------------------------------------------------------------ */

/* sin and cos of a (radians): Cody-Waite reduction by pi/2 in three fma steps, degree-7/8
 * minimax polynomials on [-pi/4, pi/4] (Cephes sinf/cosf coefficients). */
MO_API void mo_sincosf(float a, float* s_out, float* c_out) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float PIO2_HI = 1.5703125f;
    const float PIO2_MED = 4.837512969970703125e-4f;
    const float PIO2_LO = 7.54978995489188e-8f;
    float k = rintf(a * TWO_OVER_PI);
    float r = fmaf(-k, PIO2_HI, a);
    r = fmaf(-k, PIO2_MED, r);
    r = fmaf(-k, PIO2_LO, r);
    float z = r * r;
    /* sin(r) = r + r*z*(S1 + z*(S2 + z*S3)) */
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sr = fmaf(ps * z, r, r);
    /* cos(r) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    float s, c;
    switch (q) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    *s_out = s;
    *c_out = c;
}

/* atan2(y, x): octant reduction + Cephes atanf polynomial. */
MO_API float mo_atan2f(float y, float x) {
    const float PI = 3.14159274101257324f;
    const float PIO2 = 1.57079637050628662f;
    const float PIO4 = 0.785398163397448310f;
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float t = (mx == 0.0f) ? 0.0f : mn / mx; /* in [0,1] */
    float y0 = 0.0f;
    if (t > 0.4142135623730950f) { /* tan(pi/8) */
        y0 = PIO4;
        t = (t - 1.0f) / (t + 1.0f);
    }
    float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = y0 + fmaf(p * z, t, t);
    if (ay > ax) r = PIO2 - r;
    if (x < 0.0f) r = PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* natural log of a positive normal float (Cephes logf). */
MO_API float mo_logf(float x) {
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int e = (int)((bits >> 23) & 0xff) - 126; /* x = m * 2^e, m in [0.5,1) */
    bits = (bits & 0x807fffffu) | 0x3f000000u;
    float m;
    memcpy(&m, &bits, 4);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float yv = p * m * z;
    float fe = (float)e;
    yv = fmaf(-2.12194440e-4f, fe, yv);
    yv = fmaf(-0.5f, z, yv);
    float r = m + yv;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* small matrix helpers (spec: k-ordered fma chains)                                          */
/* ------------------------------------------------------------------------------------------ */
static void mat3_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = A[i * 3 + 0] * B[0 * 3 + j];
            acc = fmaf(A[i * 3 + 1], B[1 * 3 + j], acc);
            acc = fmaf(A[i * 3 + 2], B[2 * 3 + j], acc);
            C[i * 3 + j] = acc;
        }
}

static void mat4_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = A[i * 4 + 0] * B[0 * 4 + j];
            acc = fmaf(A[i * 4 + 1], B[1 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 2], B[2 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 3], B[3 * 4 + j], acc);
            C[i * 4 + j] = acc;
        }
}

/* R = Rz(a0) Ry(a1) Rx(a2) with a = deg2rad(rot_deg); pose.py:215-269 + torch.deg2rad. */
MO_API void mo_euler_zyx_deg(const float* rot_deg, float* R) {
    const float RAD_PER_DEG = 0.017453292519943295f;
    float sz, cz, sy, cy, sx, cx;
    mo_sincosf(rot_deg[0] * RAD_PER_DEG, &sz, &cz);
    mo_sincosf(rot_deg[1] * RAD_PER_DEG, &sy, &cy);
    mo_sincosf(rot_deg[2] * RAD_PER_DEG, &sx, &cx);
    const float Rz[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
    const float Ry[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
    const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
    float M[9];
    mat3_mul(Rz, Ry, M);
    mat3_mul(M, Rx, R);
}

/* radians variant used for the G7 golden (euler_angles_to_matrix takes radians). */
MO_API void mo_euler_zyx_rad(int64_t n, const float* ang, float* R) {
    for (int64_t i = 0; i < n; ++i) {
        float sz, cz, sy, cy, sx, cx;
        mo_sincosf(ang[i * 3 + 0], &sz, &cz);
        mo_sincosf(ang[i * 3 + 1], &sy, &cy);
        mo_sincosf(ang[i * 3 + 2], &sx, &cx);
        const float Rz[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
        const float Ry[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
        const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
        float M[9];
        mat3_mul(Rz, Ry, M);
        mat3_mul(M, Rx, R + i * 9);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* propagate: P' = P @ (odom @ Tn(noise))                                                     */
/* ------------------------------------------------------------------------------------------ */
MO_API void mo_propagate(int64_t N, const float* poses_in, const float* odom16, const float* tn,
                         const float* rot_deg, float* poses_out) {
    for (int64_t n = 0; n < N; ++n) {
        float Rn[9], Tn[16], NO[16];
        mo_euler_zyx_deg(rot_deg + n * 3, Rn);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Tn[i * 4 + j] = Rn[i * 3 + j];
            Tn[i * 4 + 3] = tn[n * 3 + i];
        }
        Tn[12] = 0.f; Tn[13] = 0.f; Tn[14] = 0.f; Tn[15] = 1.f;
        mat4_mul(odom16, Tn, NO);
        mat4_mul(poses_in + n * 16, NO, poses_out + n * 16);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 device-mode random streams (spec shared with the HIP kernels)                */
/* ------------------------------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

MO_API void mo_philox_raw(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

/* one Box-Muller pair from two 32-bit words */
static void box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
    float u1 = fmaf((float)(a >> 9), 1.1920928955078125e-7f /*2^-23*/, 5.9604644775390625e-8f /*2^-24*/);
    float u2 = (float)(b >> 8) * 5.9604644775390625e-8f; /* [0,1) */
    float r = sqrtf(-2.0f * mo_logf(u1));
    float s, c;
    mo_sincosf(6.28318530717958648f * u2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

/* tn[n] = z(0..2) * std_t ; rot[n] = z(3..5) * std_r ; counters (n, step, stream, 0), key = seed */
MO_API void mo_philox_noise(int64_t N, uint64_t seed, uint64_t step, float std_t, float std_r, float* tn,
                            float* rot) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t n = 0; n < N; ++n) {
        uint32_t a[4], b[4];
        philox4x32_10((uint32_t)n, (uint32_t)step, 0u, (uint32_t)((uint64_t)n >> 32), k0, k1, a);
        philox4x32_10((uint32_t)n, (uint32_t)step, 1u, (uint32_t)((uint64_t)n >> 32), k0, k1, b);
        float z[6];
        box_muller(a[0], a[1], &z[0], &z[1]);
        box_muller(a[2], a[3], &z[2], &z[3]);
        box_muller(b[0], b[1], &z[4], &z[5]);
        for (int j = 0; j < 3; ++j) {
            tn[n * 3 + j] = z[j] * std_t;
            rot[n * 3 + j] = z[3 + j] * std_r;
        }
    }
}

/* 53-bit uniforms in [0,1) for resample slot i: counter (i>>1, step, 2, 0), words 2*(i&1).. */
MO_API void mo_philox_uniform64(int64_t N, uint64_t seed, uint64_t step, double* u) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t i = 0; i < N; ++i) {
        uint32_t w[4];
        uint64_t c = (uint64_t)i >> 1;
        philox4x32_10((uint32_t)c, (uint32_t)step, 2u, (uint32_t)(c >> 32), k0, k1, w);
        uint32_t hi = w[2 * (i & 1)], lo = w[2 * (i & 1) + 1];
        uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
        u[i] = (double)m * 1.1102230246251565e-16; /* 2^-53 */
    }
}

/* float32 uniform in [0,1) for the systematic offset: counter (0, step, 3, 0) word 0 */
MO_API float mo_philox_uniform32(uint64_t seed, uint64_t step) {
    uint32_t w[4];
    philox4x32_10(0u, (uint32_t)step, 3u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return (float)(w[0] >> 8) * 5.9604644775390625e-8f;
}

/* ------------------------------------------------------------------------------------------ */
/* SO(3) log map and the 6-d pose feature                                                     */
/* ------------------------------------------------------------------------------------------ */
/* R given as the upper-left 3x3 of a row-major 4x4 (stride 4). */
static void so3_log_stride4(const float* P, float* w) {
    const float R00 = P[0], R01 = P[1], R02 = P[2];
    const float R10 = P[4], R11 = P[5], R12 = P[6];
    const float R20 = P[8], R21 = P[9], R22 = P[10];
    float ax = 0.5f * (R21 - R12);
    float ay = 0.5f * (R02 - R20);
    float az = 0.5f * (R10 - R01);
    float c = 0.5f * ((R00 + R11) + R22 - 1.0f);
    c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
    float s2 = fmaf(az, az, fmaf(ay, ay, ax * ax));
    float s = sqrtf(s2);
    float theta = mo_atan2f(s, c);
    if (1.0f + c <= 1e-2f) {
        /* near pi: axis from the dominant column of (R+R^T)/2 - c I */
        int major = 0;
        if (R11 > R00 && R11 > R22) major = 1;
        if (R22 > R00 && R22 > R11) major = 2;
        float v[3];
        const float* Rm = P; /* row-major 4x4 */
        for (int j = 0; j < 3; ++j) v[j] = 0.5f * (Rm[major * 4 + j] + Rm[j * 4 + major]);
        v[major] -= c;
        float nv = sqrtf(fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
        float sa = (major == 0) ? ax : (major == 1 ? ay : az);
        float sign = (sa < 0.0f) ? -1.0f : 1.0f;
        float k = theta * sign;
        for (int j = 0; j < 3; ++j) w[j] = (v[j] / nv) * k;
        return;
    }
    float scale;
    if (theta < 5e-3f)
        scale = fmaf(s2, 0.16666667163372040f, 1.0f);
    else
        scale = theta / s;
    w[0] = ax * scale;
    w[1] = ay * scale;
    w[2] = az * scale;
}

MO_API void mo_so3_log(int64_t N, const float* poses, float* out3) {
    for (int64_t n = 0; n < N; ++n) so3_log_stride4(poses + n * 16, out3 + n * 3);
}

/* f = [ (1-w) t , w log(R) ], w = 0.01  (R3_SE3, tactile_tree.py:73-77) */
MO_API void mo_se3_feature(int64_t N, const float* poses, float wt, float wr, float* feat6) {
    for (int64_t n = 0; n < N; ++n) {
        const float* P = poses + n * 16;
        float w[3];
        so3_log_stride4(P, w);
        feat6[n * 6 + 0] = wt * P[3];
        feat6[n * 6 + 1] = wt * P[7];
        feat6[n * 6 + 2] = wt * P[11];
        feat6[n * 6 + 3] = wr * w[0];
        feat6[n * 6 + 4] = wr * w[1];
        feat6[n * 6 + 5] = wr * w[2];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* exact k nearest neighbours by (distance, index) (tactile_tree.py:50-52 with n_neighbors = k)  */
/* ------------------------------------------------------------------------------------------ */
MO_API void mo_knn6(int64_t N, int64_t K, int32_t k, const float* q6, const float* c6, int32_t* idx, float* d2out) {
    for (int64_t n = 0; n < N; ++n) {
        const float* q = q6 + n * 6;
        int32_t* bi = idx + n * k;
        float* bd = d2out + n * k;
        int32_t have = 0;
        for (int64_t j = 0; j < K; ++j) {
            const float* p = c6 + j * 6;
            float d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
            float d3 = q[3] - p[3], d4 = q[4] - p[4], d5 = q[5] - p[5];
            float d = d0 * d0;
            d = fmaf(d1, d1, d);
            d = fmaf(d2, d2, d);
            d = fmaf(d3, d3, d);
            d = fmaf(d4, d4, d);
            d = fmaf(d5, d5, d);
            if (!(d == d)) continue;
            int32_t pos = have;
            if (have == k) {
                if (!(d < bd[k - 1])) continue; /* j ascends: an equal distance never displaces an earlier index */
                pos = k - 1;
            } else {
                ++have;
            }
            while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
            bd[pos] = d;
            bi[pos] = (int32_t)j;
        }
        for (int32_t r = have; r < k; ++r) { bd[r] = INFINITY; bi[r] = 0x7fffffff; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* exact nearest neighbour (brute force; ties -> smallest index)                              */
/* ------------------------------------------------------------------------------------------ */
MO_API void mo_nn6(int64_t N, int64_t K, const float* q6, const float* c6, int32_t* idx, float* d2out) {
    for (int64_t n = 0; n < N; ++n) {
        const float* q = q6 + n * 6;
        float best = INFINITY;
        int32_t bi = 0;
        for (int64_t k = 0; k < K; ++k) {
            const float* p = c6 + k * 6;
            float d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
            float d3 = q[3] - p[3], d4 = q[4] - p[4], d5 = q[5] - p[5];
            float d = d0 * d0;
            d = fmaf(d1, d1, d);
            d = fmaf(d2, d2, d);
            d = fmaf(d3, d3, d);
            d = fmaf(d4, d4, d);
            d = fmaf(d5, d5, d);
            if (d < best) { best = d; bi = (int32_t)k; }
        }
        idx[n] = bi;
        if (d2out) d2out[n] = best;
    }
}

/* distance (float64) from each particle translation to the nearest mesh vertex */
MO_API void mo_nn3(int64_t N, int64_t M, const float* poses, const double* verts, double* dist) {
    for (int64_t n = 0; n < N; ++n) {
        double x = (double)poses[n * 16 + 3], y = (double)poses[n * 16 + 7], z = (double)poses[n * 16 + 11];
        double best = INFINITY;
        for (int64_t m = 0; m < M; ++m) {
            double dx = x - verts[m * 3], dy = y - verts[m * 3 + 1], dz = z - verts[m * 3 + 2];
            double d = dx * dx;
            d = fma(dy, dy, d);
            d = fma(dz, dz, d);
            if (d < best) best = d;
        }
        dist[n] = sqrt(best);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cosine scores of one tactile code against every codebook row                               */
/* ------------------------------------------------------------------------------------------ */
/* s_k = <e, C_k> / (max(|e|,eps) * max(|C_k|,eps)), eps = 1e-8, float64 accumulation (cosine_similarity,
 * modules/particle_filter.py:455-457).  SUMMATION ORDER (spec; csrc/score_body.hpp score_wave / score_claimed_rows and
 * csrc/score.hip k_score_generic state the same): the D products are dealt to 16 partial sums - for D in {128, 256,
 * 512, 1024} partial s takes elements 64 j + 4 s + c (j ascending, c = 0..3), for any other D elements s, s + 16, ... -
 * each a sequential fma chain from +0.0; the 16 partials are then added as a balanced tree in the order of an
 * xor-butterfly (offsets 8, 4, 2, 1).  |e|^2 and |C_k|^2 are summed the same way.  With the order fixed the scores -
 * and through the softmax numerators the resample CDF - are bit-identical between this oracle and the kernels. */
static double mo_quarter_tree(const double* a) {
    double b[16], c[16], d[16];
    for (int s = 0; s < 16; ++s) b[s] = a[s] + a[s ^ 8];
    for (int s = 0; s < 16; ++s) c[s] = b[s] + b[s ^ 4];
    for (int s = 0; s < 16; ++s) d[s] = c[s] + c[s ^ 2];
    return d[0] + d[1];
}
static int mo_score_reg_layout(int64_t D) { return D == 128 || D == 256 || D == 512 || D == 1024; }
/* element index of the t-th term of partial s, or -1 past the end */
static int64_t mo_score_elem(int64_t D, int reg, int s, int64_t t) {
    const int64_t d = reg ? 64 * (t / 4) + 4 * s + (t % 4) : s + 16 * t;
    return d < D ? d : -1;
}
#define MO_SCORE_BODY(ROWTYPE)                                                                      \
    const int reg = mo_score_reg_layout(D);                                                         \
    const int64_t nt = reg ? D / 16 : (D + 15) / 16;                                                \
    double pe[16];                                                                                  \
    for (int s = 0; s < 16; ++s) {                                                                  \
        double acc = 0.0;                                                                           \
        for (int64_t t = 0; t < nt; ++t) {                                                          \
            const int64_t d = mo_score_elem(D, reg, s, t);                                          \
            if (d >= 0) acc = fma(code[d], code[d], acc);                                           \
        }                                                                                           \
        pe[s] = acc;                                                                                \
    }                                                                                               \
    double ne = sqrt(mo_quarter_tree(pe));                                                          \
    if (ne < 1e-8) ne = 1e-8;                                                                       \
    for (int64_t k = 0; k < K; ++k) {                                                               \
        const ROWTYPE* row = emb + k * D;                                                           \
        double pd[16], pn[16];                                                                      \
        for (int s = 0; s < 16; ++s) {                                                              \
            double dot = 0.0, nr = 0.0;                                                             \
            for (int64_t t = 0; t < nt; ++t) {                                                      \
                const int64_t d = mo_score_elem(D, reg, s, t);                                      \
                if (d >= 0) {                                                                       \
                    const double v = (double)row[d];                                                \
                    dot = fma(v, code[d], dot);                                                     \
                    nr = fma(v, v, nr);                                                             \
                }                                                                                   \
            }                                                                                       \
            pd[s] = dot;                                                                            \
            pn[s] = nr;                                                                             \
        }                                                                                           \
        double nr = sqrt(mo_quarter_tree(pn));                                                      \
        if (nr < 1e-8) nr = 1e-8;                                                                   \
        scores[k] = mo_quarter_tree(pd) / (ne * nr);                                                \
    }

MO_API void mo_score_f32(int64_t K, int64_t D, const float* emb, const double* code, double* scores) {
    MO_SCORE_BODY(float)
}

MO_API void mo_score_f64(int64_t K, int64_t D, const double* emb, const double* code, double* scores) {
    MO_SCORE_BODY(double)
}

/*
 * Batched scoring spec (the MFMA kernel k_score_mfma): the code is rounded to float32 and the dot product is
 * a float32 fma chain in the order of the matrix-core schedule - D in chunks of 16; inside a chunk the four
 * components s of a lane's float4, and for each the four k-slots g of v_mfma_f32_16x16x4_f32:
 * d = 16c + 4g + s.  The division by the float64 norms is as in mo_score_f32.
 */
MO_API void mo_score_batch_f32(int64_t K, int64_t D, int64_t B, const float* emb, const double* codes, double* scores) {
    for (int64_t b = 0; b < B; ++b) {
        const double* code = codes + b * D;
        double ne = 0.0;
        for (int64_t j = 0; j < D; ++j) ne = fma(code[j], code[j], ne);
        ne = sqrt(ne);
        if (ne < 1e-8) ne = 1e-8;
        for (int64_t k = 0; k < K; ++k) {
            const float* row = emb + k * D;
            float acc = 0.0f;
            double nr = 0.0;
            for (int64_t c = 0; c < D; c += 16)
                for (int s2 = 0; s2 < 4; ++s2)
                    for (int g = 0; g < 4; ++g) {
                        int64_t d = c + 4 * g + s2;
                        acc = fmaf(row[d], (float)code[d], acc);
                    }
            for (int64_t j = 0; j < D; ++j) nr += (double)row[j] * (double)row[j];
            nr = sqrt(nr);
            if (nr < 1e-8) nr = 1e-8;
            scores[b * K + k] = (double)acc / (ne * nr);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* float64 blocked scan (the summation-order spec)                                            */
/* ------------------------------------------------------------------------------------------ */
/*
 * Three-level fixed order.  16 consecutive values form a chunk, 16 chunks a group (256 values),
 * 16 groups a block (4096 values).  Every accumulator starts at +0.0 and adds in index order:
 *   local_i  = inclusive sum inside the chunk
 *   TP_c     = exclusive sum of the chunk totals inside the group
 *   GP_g     = exclusive sum of the group totals inside the block (group total = sum of its chunk totals)
 *   BP_b     = exclusive sum of the block totals            (block total = sum of its group totals)
 *   prefix_i = BP_b + (GP_g + (TP_c + local_i))
 * The grand total is the sum of the block totals in order (== prefix_{N-1}).  Missing trailing
 * elements count as +0.0.
 */
#define MO_CHUNK 16
#define MO_GROUP 16
#define MO_BLOCK 16

MO_API double mo_blocked_scan(int64_t N, const double* w, double* prefix) {
    const int64_t GRP = (int64_t)MO_CHUNK * MO_GROUP, BLK = GRP * MO_BLOCK;
    double BP = 0.0;
    for (int64_t b0 = 0; b0 < N; b0 += BLK) {
        double GP = 0.0;
        for (int64_t g0 = b0; g0 < b0 + BLK && g0 < N; g0 += GRP) {
            double TP = 0.0;
            for (int64_t c0 = g0; c0 < g0 + GRP && c0 < N; c0 += MO_CHUNK) {
                double local = 0.0;
                for (int64_t i = c0; i < c0 + MO_CHUNK && i < N; ++i) {
                    local = local + w[i];
                    if (prefix) prefix[i] = BP + (GP + (TP + local));
                }
                TP = TP + local;
            }
            GP = GP + TP;
        }
        BP = BP + GP;
    }
    return BP;
}

/* ------------------------------------------------------------------------------------------ */
/* float64 exponential (spec)                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* The softmax numerators exp(x - shift) (modules/particle_filter.py:466-468, torch Softmax) decide the
 * resample CDF and with it the indices, so the exponential is a SPEC function like the float32 ones
 * above (a math library's exp differs between libm and the device's in the last place):
 *   k = rint(x / ln 2); r = x - k ln2_hi - k ln2_lo (two fma steps, fdlibm's split of ln 2);
 *   exp(r) = Horner of the degree-13 Taylor polynomial in fma steps (|r| <= 0.3466: truncation < 5e-18);
 *   result = p * 2^(k >> 1) * 2^(k - (k >> 1))  (two exact power-of-two factors: one rounding, also
 *   in the subnormal range).  x > 709.78... -> +inf, x < -745.14 -> 0, NaN -> NaN.
 * Within 1 ulp of the correctly rounded value on the arguments the path uses (checked against libm in
 * tests/test_oracle_math.py). */
static double mo_pow2i(int k) {  /* 2^k for -1022 <= k <= 1023 */
    union { uint64_t u; double d; } v;
    v.u = (uint64_t)(k + 1023) << 52;
    return v.d;
}
MO_API double mo_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double kf = rint(x * INV_LN2);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.6059043836821613e-10;          /* 1/13! */
    p = fma(p, r, 2.08767569878681e-09);        /* 1/12! */
    p = fma(p, r, 2.505210838544172e-08);       /* 1/11! */
    p = fma(p, r, 2.755731922398589e-07);       /* 1/10! */
    p = fma(p, r, 2.7557319223985893e-06);      /* 1/9!  */
    p = fma(p, r, 2.48015873015873e-05);        /* 1/8!  */
    p = fma(p, r, 1.984126984126984e-04);       /* 1/7!  */
    p = fma(p, r, 1.388888888888889e-03);       /* 1/6!  */
    p = fma(p, r, 8.333333333333333e-03);       /* 1/5!  */
    p = fma(p, r, 4.1666666666666664e-02);      /* 1/4!  */
    p = fma(p, r, 1.6666666666666666e-01);      /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int k = (int)kf, k1 = k >> 1, k2 = k - k1;
    return (p * mo_pow2i(k1)) * mo_pow2i(k2);
}
MO_API void mo_exp_vec(int64_t N, const double* x, double shift, double* out) {
    for (int64_t i = 0; i < N; ++i) out[i] = mo_exp(x[i] - shift);
}

/*
 * weights -> softmax.  get_similarity tail (particle_filter.py:459-468):
 * if |max-min| <= 1e-8 (torch.isclose(.., 0) with default atol) or !softmax -> copy x;
 * else w = mo_exp(x - max) / blocked_sum(mo_exp(x - max)).
 * returns 1 when softmax was applied.
 */
MO_API int mo_softmax(int64_t N, const double* x, int softmax, double* w) {
    if (N == 0) return 0;
    double mx = x[0], mn = x[0];
    for (int64_t i = 1; i < N; ++i) {
        if (x[i] > mx) mx = x[i];
        if (x[i] < mn) mn = x[i];
    }
    if (!softmax || fabs(mx - mn) <= 1e-8) {
        if (w != x) memcpy(w, x, (size_t)N * sizeof(double));
        return 0;
    }
    for (int64_t i = 0; i < N; ++i) w[i] = mo_exp(x[i] - mx);
    double S = mo_blocked_scan(N, w, NULL);
    for (int64_t i = 0; i < N; ++i) w[i] = w[i] / S;
    return 1;
}

/*
 * cdf_i = prefix_i / total, cdf_{N-1} := 1.   status: 0 ok, 1 all weights zero, 2 NaN present
 * (the two cases in which resampler returns its input unchanged, particle_filter.py:240-241).
 */
MO_API int mo_cdf(int64_t N, const double* w, double* cdf) {
    if (N == 0) return 1;
    double total = mo_blocked_scan(N, w, cdf);
    if (isnan(total)) return 2;
    for (int64_t i = 0; i < N; ++i) if (isnan(w[i])) return 2;
    if (total == 0.0) return 1;
    for (int64_t i = 0; i < N; ++i) cdf[i] = cdf[i] / total;
    cdf[N - 1] = 1.0;
    return 0;
}

/* multinomial with replacement == first j with cdf_j >= u (ATen binary search, cum_prob < u -> right) */
MO_API void mo_search_lower(int64_t N, const double* cdf, int64_t M, const double* u, int32_t* idx) {
    for (int64_t i = 0; i < M; ++i) {
        int64_t lo = 0, hi = N;
        while (hi - lo > 0) {
            int64_t mid = lo + (hi - lo) / 2;
            if (cdf[mid] < u[i]) lo = mid + 1; else hi = mid;
        }
        idx[i] = (int32_t)(lo < N ? lo : N - 1);
    }
}

/* low-variance: loc_i = fmod(i/M + off, 1), off = float32(u32 / M); first j with cdf_j > loc_i */
MO_API void mo_search_systematic(int64_t N, const double* cdf, int64_t M, float u32, int32_t* idx) {
    float off = u32 / (float)M;
    for (int64_t i = 0; i < M; ++i) {
        double loc = fmod((double)i / (double)M + (double)off, 1.0);
        int64_t lo = 0, hi = N;
        while (hi - lo > 0) {
            int64_t mid = lo + (hi - lo) / 2;
            if (cdf[mid] <= loc) lo = mid + 1; else hi = mid;
        }
        idx[i] = (int32_t)(lo < N ? lo : N - 1);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* torch's CPU generator stream (at::mt19937)                                                 */
/* ------------------------------------------------------------------------------------------ */
/* The reference's draws come from torch's default CPU generator: torch.manual_seed(s) seeds MT19937 with the low 32
 * bits of s (ATen/core/MT19937RNGEngine.h); torch.multinomial(w.double(), N, True) - what WeightedRandomSampler calls,
 * modules/particle_filter.py:245 - and torch.rand(N, dtype=float64) both take, per value, two 32-bit outputs as
 * ((hi << 32 | lo) & (2^53 - 1)) * 2^-53 (at::uniform_real_distribution<double>).  Restated from the published algorithm
 * (Matsumoto & Nishimura 1998); pinned against torch.rand itself in tests/test_torch_stream.py (torch is importable on
 * every box).  mo_mt19937_rand64 skips `skip_words` outputs after seeding, then writes N uniforms. */
typedef struct { uint32_t mt[624]; int pos; } mo_mt;
static void mo_mt_seed(mo_mt* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int j = 1; j < 624; ++j) g->mt[j] = 1812433253u * (g->mt[j - 1] ^ (g->mt[j - 1] >> 30)) + (uint32_t)j;
    g->pos = 624;
}
static uint32_t mo_mt_next(mo_mt* g) {
    if (g->pos >= 624) {
        uint32_t* mt = g->mt;
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->pos = 0;
    }
    uint32_t y = g->mt[g->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
MO_API void mo_mt19937_rand64(uint64_t seed, int64_t skip_words, int64_t N, double* out) {
    mo_mt g;
    mo_mt_seed(&g, (uint32_t)(seed & 0xffffffffu));
    for (int64_t i = 0; i < skip_words; ++i) (void)mo_mt_next(&g);
    for (int64_t i = 0; i < N; ++i) {
        const uint64_t hi = mo_mt_next(&g), lo = mo_mt_next(&g);
        out[i] = (double)(((hi << 32) | lo) & ((1ull << 53) - 1ull)) * 1.1102230246251565e-16;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* particle_rmse                                                                              */
/* ------------------------------------------------------------------------------------------ */
MO_API void mo_rmse(int64_t N, const float* poses, const float* gt16, double* out2) {
    double st = 0.0, sr = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        const float* P = poses + n * 16;
        float dx = gt16[3] - P[3], dy = gt16[7] - P[7], dz = gt16[11] - P[11];
        float e2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        /* trace(R_gt R_n^T) = sum_ij Rgt_ij Rn_ij */
        float tr = 0.0f;
        for (int i = 0; i < 3; ++i) {
            float acc = gt16[i * 4] * P[i * 4];
            acc = fmaf(gt16[i * 4 + 1], P[i * 4 + 1], acc);
            acc = fmaf(gt16[i * 4 + 2], P[i * 4 + 2], acc);
            tr += acc;
        }
        float ang = acosf((tr - 1.0f) * 0.5f) * 57.2957795130823209f;
        if (isnan(ang)) ang = 0.0f;
        if (ang > 180.0f) ang -= 360.0f;
        if (ang < -180.0f) ang += 360.0f;
        st += (double)e2;
        sr += (double)ang * (double)ang;
    }
    out2[0] = sqrt(st / (double)N);
    out2[1] = sqrt(sr / (double)N);
}

/* rows gather helper for the oracle step */
MO_API void mo_gather_rows(int64_t M, const int32_t* idx, const void* src, void* dst, int64_t row_bytes) {
    for (int64_t i = 0; i < M; ++i)
        memcpy((char*)dst + i * row_bytes, (const char*)src + (int64_t)idx[i] * row_bytes, (size_t)row_bytes);
}

/* ------------------------------------------------------------------------------------------ */
/* cluster_particles: DBSCAN labels                                                           */
/* ------------------------------------------------------------------------------------------ */
/* modules/particle_filter.py:208-217: sklearn.cluster.DBSCAN(eps, min_samples).fit(X).labels_ on the particle
 * translations X (N,3) float32 (scikit-learn is third-party; its published algorithm, sklearn/cluster/_dbscan.py +
 * _dbscan_inner.pyx, restated):
 *   - neighbourhood of i = { j : |x_i - x_j|^2 <= eps^2 }, i included; the KD-tree sklearn builds works on float64
 *     copies and compares the reduced distance d = ((dx*dx) + dy*dy) + dz*dz (accumulated in that order) with eps*eps;
 *   - core <=> |neighbourhood| >= min_samples;
 *   - clusters = connected components of the core points under "within eps", numbered in the order the scan over
 *     i = 0, 1, ... meets their first core point; a non-core point within eps of core points takes the number of the
 *     first cluster that reaches it = the smallest number among them (cluster c is expanded completely before
 *     c + 1 starts); everything else is noise (-1).
 * O(N^2): meant for the sizes the parity tests use. */
static int64_t uf_find(int64_t* p, int64_t i) {
    while (p[i] != i) { p[i] = p[p[i]]; i = p[i]; }
    return i;
}

MO_API int32_t mo_dbscan(int64_t N, const float* X, double eps, int64_t min_samples, int32_t* labels) {
    const double r2 = eps * eps;
    int64_t* parent = (int64_t*)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    int32_t* number = (int32_t*)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    uint8_t* core = (uint8_t*)calloc((size_t)(N > 0 ? N : 1), 1);
    for (int64_t i = 0; i < N; ++i) {
        int64_t cnt = 0;
        const double xi = X[3 * i], yi = X[3 * i + 1], zi = X[3 * i + 2];
        for (int64_t j = 0; j < N; ++j) {
            const double dx = xi - (double)X[3 * j], dy = yi - (double)X[3 * j + 1], dz = zi - (double)X[3 * j + 2];
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            cnt += d <= r2;
        }
        core[i] = cnt >= min_samples;
        parent[i] = i;
    }
    for (int64_t i = 0; i < N; ++i) {
        if (!core[i]) continue;
        const double xi = X[3 * i], yi = X[3 * i + 1], zi = X[3 * i + 2];
        for (int64_t j = i + 1; j < N; ++j) {
            if (!core[j]) continue;
            const double dx = xi - (double)X[3 * j], dy = yi - (double)X[3 * j + 1], dz = zi - (double)X[3 * j + 2];
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            if (d <= r2) {
                int64_t a = uf_find(parent, i), b = uf_find(parent, j);
                if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }  /* root = smallest index */
            }
        }
    }
    int32_t ncl = 0;
    for (int64_t i = 0; i < N; ++i) number[i] = (core[i] && uf_find(parent, i) == i) ? ncl++ : -1;
    for (int64_t i = 0; i < N; ++i) {
        if (core[i]) { labels[i] = number[uf_find(parent, i)]; continue; }
        int32_t best = -1;
        const double xi = X[3 * i], yi = X[3 * i + 1], zi = X[3 * i + 2];
        for (int64_t j = 0; j < N; ++j) {
            if (!core[j]) continue;
            const double dx = xi - (double)X[3 * j], dy = yi - (double)X[3 * j + 1], dz = zi - (double)X[3 * j + 2];
            double d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            if (d <= r2) {
                const int32_t c = number[uf_find(parent, j)];
                if (best < 0 || c < best) best = c;
            }
        }
        labels[i] = best;
    }
    free(parent); free(number); free(core);
    return ncl;
}
