// midas_math.hpp - device-side arithmetic spec for the particle-filter kernels (gfx950).
//
// float32 work that decides an index (propagated pose -> 6-d feature -> nearest codebook entry) is
// written as explicit fma chains with self-contained polynomial sin/cos/atan2/log, so the result
// is a pure function of IEEE-754 operations and does not depend on a math library.  The kernels
// are compiled with -ffp-contract=off; every fused operation below is spelled __builtin_fmaf.
// DESIGN.md "Arithmetic spec" is the normative text; this file implements it for the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace midas {

#define MD __device__ __forceinline__

MD float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// sin/cos: Cody-Waite reduction by pi/2 (three fma steps) + minimax polynomials on [-pi/4, pi/4].
MD void sincos_spec(float a, float& s, float& c) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float PIO2_HI = 1.5703125f;
    const float PIO2_MED = 4.837512969970703125e-4f;
    const float PIO2_LO = 7.54978995489188e-8f;
    float k = __builtin_rintf(a * TWO_OVER_PI);
    float r = fmaf_(-k, PIO2_HI, a);
    r = fmaf_(-k, PIO2_MED, r);
    r = fmaf_(-k, PIO2_LO, r);
    float z = r * r;
    float ps = fmaf_(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf_(ps, z, -1.6666654611e-1f);
    float sr = fmaf_(ps * z, r, r);
    float pc = fmaf_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf_(pc, z, 4.166664568298827e-2f);
    float cr = fmaf_(pc * z, z, fmaf_(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

MD float atan2_spec(float y, float x) {
    const float PI = 3.14159274101257324f;
    const float PIO2 = 1.57079637050628662f;
    const float PIO4 = 0.785398163397448310f;
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float t = (mx == 0.0f) ? 0.0f : mn / mx;
    float y0 = 0.0f;
    if (t > 0.4142135623730950f) {
        y0 = PIO4;
        t = (t - 1.0f) / (t + 1.0f);
    }
    float z = t * t;
    float p = fmaf_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf_(p, z, 1.99777106478e-1f);
    p = fmaf_(p, z, -3.33329491539e-1f);
    float r = y0 + fmaf_(p * z, t, t);
    if (ay > ax) r = PIO2 - r;
    if (x < 0.0f) r = PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

MD float log_spec(float x) {
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xff) - 126;
    float m = __uint_as_float((bits & 0x807fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = fmaf_(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf_(p, m, 1.1676998740e-1f);
    p = fmaf_(p, m, -1.2420140846e-1f);
    p = fmaf_(p, m, 1.4249322787e-1f);
    p = fmaf_(p, m, -1.6668057665e-1f);
    p = fmaf_(p, m, 2.0000714765e-1f);
    p = fmaf_(p, m, -2.4999993993e-1f);
    p = fmaf_(p, m, 3.3333331174e-1f);
    float yv = p * m * z;
    float fe = (float)e;
    yv = fmaf_(-2.12194440e-4f, fe, yv);
    yv = fmaf_(-0.5f, z, yv);
    float r = m + yv;
    r = fmaf_(0.693359375f, fe, r);
    return r;
}

// float64 exponential of the softmax numerators (spec: oracle/midas_oracle.c mo_exp states the same operations):
// k = rint(x / ln 2), two-step fma reduction by fdlibm's split of ln 2, degree-13 Taylor polynomial in Horner fma steps,
// two exact power-of-two factors (one rounding, also for subnormal results).  A math library's exp is not bit-identical
// between libm and the device; the numerators decide the resample CDF and with it the indices.
MD double pow2i_spec(int k) { return __longlong_as_double((long long)((uint64_t)(k + 1023) << 52)); }
MD double exp_spec(double x) {
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    // clamped argument for the arithmetic (the special cases are selected at the end: no branches)
    const double xc = x > 709.782712893384 ? 709.782712893384 : (x < -745.2 ? -745.2 : x);
    const double kf = __builtin_rint(xc * INV_LN2);
    double r = fma_(-kf, LN2_HI, xc);
    r = fma_(-kf, LN2_LO, r);
    double p = 1.6059043836821613e-10;
    p = fma_(p, r, 2.08767569878681e-09);
    p = fma_(p, r, 2.505210838544172e-08);
    p = fma_(p, r, 2.755731922398589e-07);
    p = fma_(p, r, 2.7557319223985893e-06);
    p = fma_(p, r, 2.48015873015873e-05);
    p = fma_(p, r, 1.984126984126984e-04);
    p = fma_(p, r, 1.388888888888889e-03);
    p = fma_(p, r, 8.333333333333333e-03);
    p = fma_(p, r, 4.1666666666666664e-02);
    p = fma_(p, r, 1.6666666666666666e-01);
    p = fma_(p, r, 0.5);
    p = fma_(p, r, 1.0);
    p = fma_(p, r, 1.0);
    const int k = (int)kf, k1 = k >> 1, k2 = k - k1;
    double v = (p * pow2i_spec(k1)) * pow2i_spec(k2);
    v = x > 709.782712893384 ? (double)INFINITY : v;
    v = x < -745.2 ? 0.0 : v;
    return x != x ? x : v;
}

// ---- Philox4x32-10 ---------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

MD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

MD void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    float u1 = fmaf_((float)(a >> 9), 1.1920928955078125e-7f, 5.9604644775390625e-8f);
    float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    float r = __builtin_sqrtf(-2.0f * log_spec(u1));
    float s, c;
    sincos_spec(6.28318530717958648f * u2, s, c);
    z0 = r * c;
    z1 = r * s;
}

// six N(0,1) draws of particle n at step `step`: counters (n, step, {0,1}, n>>32)
MD void philox_normals6(uint64_t n, uint64_t seed, uint64_t step, float* z) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    u32x4 a = philox4x32_10((uint32_t)n, (uint32_t)step, 0u, (uint32_t)(n >> 32), k0, k1);
    u32x4 b = philox4x32_10((uint32_t)n, (uint32_t)step, 1u, (uint32_t)(n >> 32), k0, k1);
    box_muller(a.x, a.y, z[0], z[1]);
    box_muller(a.z, a.w, z[2], z[3]);
    box_muller(b.x, b.y, z[4], z[5]);
}

// 53-bit uniform of resample slot i: counter (i>>1, step, 2, (i>>1)>>32), words 2*(i&1)..
MD double philox_uniform53(uint64_t i, uint64_t seed, uint64_t step) {
    uint64_t c = i >> 1;
    u32x4 w = philox4x32_10((uint32_t)c, (uint32_t)step, 2u, (uint32_t)(c >> 32), (uint32_t)seed,
                            (uint32_t)(seed >> 32));
    uint32_t hi = (i & 1) ? w.z : w.x, lo = (i & 1) ? w.w : w.y;
    uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)m * 1.1102230246251565e-16;
}

MD float philox_uniform24(uint64_t seed, uint64_t step) {
    u32x4 w = philox4x32_10(0u, (uint32_t)step, 3u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(w.x >> 8) * 5.9604644775390625e-8f;
}

// ---- small matrices (row-major), k-ordered fma chains -----------------------------------------
MD void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = A[i * 3 + 0] * B[0 * 3 + j];
            acc = fmaf_(A[i * 3 + 1], B[1 * 3 + j], acc);
            acc = fmaf_(A[i * 3 + 2], B[2 * 3 + j], acc);
            C[i * 3 + j] = acc;
        }
}

MD void mat4_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = A[i * 4 + 0] * B[0 * 4 + j];
            acc = fmaf_(A[i * 4 + 1], B[1 * 4 + j], acc);
            acc = fmaf_(A[i * 4 + 2], B[2 * 4 + j], acc);
            acc = fmaf_(A[i * 4 + 3], B[3 * 4 + j], acc);
            C[i * 4 + j] = acc;
        }
}

// Tn = [Rz(a0) Ry(a1) Rx(a2), tn; 0 0 0 1], a = deg2rad(rot_deg)
MD void noise_transform(const float* tn, const float* rot_deg, float* Tn) {
    const float RAD_PER_DEG = 0.017453292519943295f;
    float sz, cz, sy, cy, sx, cx;
    sincos_spec(rot_deg[0] * RAD_PER_DEG, sz, cz);
    sincos_spec(rot_deg[1] * RAD_PER_DEG, sy, cy);
    sincos_spec(rot_deg[2] * RAD_PER_DEG, sx, cx);
    const float Rz[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
    const float Ry[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
    const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
    float M[9], R[9];
    mat3_mul(Rz, Ry, M);
    mat3_mul(M, Rx, R);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Tn[i * 4 + j] = R[i * 3 + j];
        Tn[i * 4 + 3] = tn[i];
    }
    Tn[12] = 0.f; Tn[13] = 0.f; Tn[14] = 0.f; Tn[15] = 1.f;
}

// SO(3) log of the upper-left 3x3 of a row-major 4x4
MD void so3_log(const float* P, float* w) {
    const float R00 = P[0], R01 = P[1], R02 = P[2];
    const float R10 = P[4], R11 = P[5], R12 = P[6];
    const float R20 = P[8], R21 = P[9], R22 = P[10];
    float ax = 0.5f * (R21 - R12);
    float ay = 0.5f * (R02 - R20);
    float az = 0.5f * (R10 - R01);
    float c = 0.5f * ((R00 + R11) + R22 - 1.0f);
    c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
    float s2 = fmaf_(az, az, fmaf_(ay, ay, ax * ax));
    float s = __builtin_sqrtf(s2);
    float theta = atan2_spec(s, c);
    if (1.0f + c <= 1e-2f) {
        int major = 0;
        if (R11 > R00 && R11 > R22) major = 1;
        if (R22 > R00 && R22 > R11) major = 2;
        float v0, v1, v2, sa;
        if (major == 0) {
            v0 = 0.5f * (R00 + R00) - c; v1 = 0.5f * (R01 + R10); v2 = 0.5f * (R02 + R20); sa = ax;
        } else if (major == 1) {
            v0 = 0.5f * (R10 + R01); v1 = 0.5f * (R11 + R11) - c; v2 = 0.5f * (R12 + R21); sa = ay;
        } else {
            v0 = 0.5f * (R20 + R02); v1 = 0.5f * (R21 + R12); v2 = 0.5f * (R22 + R22) - c; sa = az;
        }
        float nv = __builtin_sqrtf(fmaf_(v2, v2, fmaf_(v1, v1, v0 * v0)));
        float k = theta * ((sa < 0.0f) ? -1.0f : 1.0f);
        w[0] = (v0 / nv) * k;
        w[1] = (v1 / nv) * k;
        w[2] = (v2 / nv) * k;
        return;
    }
    float scale = (theta < 5e-3f) ? fmaf_(s2, 0.16666667163372040f, 1.0f) : theta / s;
    w[0] = ax * scale;
    w[1] = ay * scale;
    w[2] = az * scale;
}

MD void se3_feature(const float* P, float wt, float wr, float* f) {
    float w[3];
    so3_log(P, w);
    f[0] = wt * P[3];
    f[1] = wt * P[7];
    f[2] = wt * P[11];
    f[3] = wr * w[0];
    f[4] = wr * w[1];
    f[5] = wr * w[2];
}

// ---- wave-wide maximum / minimum / integer sum by DPP moves (no LDS round trips) ----------------------------------------------
// quad permutes (lane^1, lane^2), row_half_mirror, row_mirror: every lane of a 16-lane row holds the row's result; row_bcast15
// and row_bcast31 carry it up the rows; lane 63 holds the wave's, read back for everybody.  Order-insensitive operations only
// (floating-point SUMS keep their shuffle butterflies: their order is part of the arithmetic spec).  A `__shfl_xor` butterfly of a
// double is twelve dependent LDS-crossbar trips - 1.4 us of the tail kernel's 7.9 per workgroup.
template <int CTRL, int ROW_MASK = 0xf>
MD uint32_t dpp_move(uint32_t own, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)own, (int)v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK = 0xf>
MD double dpp_move(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_move<CTRL, ROW_MASK>((uint32_t)b, (uint32_t)b), hi = dpp_move<CTRL, ROW_MASK>((uint32_t)(b >> 32), (uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// lane J of the own 16-lane row, to every lane of the row (row_newbcast)
template <int J>
MD double row_bcast(double v) { return dpp_move<0x150 + J>(v); }
MD double wave_bcast63(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
#define MIDAS_DPP_REDUCE(v, OP)                                                        \
    { auto t_ = dpp_move<0xB1>(v); v = OP(v, t_); }   /* lane ^ 1 */                    \
    { auto t_ = dpp_move<0x4E>(v); v = OP(v, t_); }   /* lane ^ 2 */                    \
    { auto t_ = dpp_move<0x141>(v); v = OP(v, t_); }  /* 7 - lane within the octet */   \
    { auto t_ = dpp_move<0x140>(v); v = OP(v, t_); }  /* 15 - lane within the row */    \
    { auto t_ = dpp_move<0x142, 0xa>(v); v = OP(v, t_); }  /* lane 15 of the row below, rows 1 and 3 */ \
    { auto t_ = dpp_move<0x143, 0xc>(v); v = OP(v, t_); }  /* lane 31, rows 2 and 3 */
MD double dpp_max_(double a, double b) { return b > a ? b : a; }
MD double dpp_min_(double a, double b) { return b < a ? b : a; }
MD double wave_max_dpp(double v) {  // NaN never wins a comparison (as in the shuffle form): callers flag NaN separately
    MIDAS_DPP_REDUCE(v, dpp_max_)
    return wave_bcast63(v);
}
MD double wave_min_dpp(double v) {
    MIDAS_DPP_REDUCE(v, dpp_min_)
    return wave_bcast63(v);
}
// the same for order-preserving 64-bit keys (the small-set selection's extrema: a `__shfl_xor` butterfly of two 64-bit values is
// 24 trips through the LDS crossbar, one after the other - 1.6 us with sixteen waves on one compute unit)
template <int CTRL, int ROW_MASK = 0xf>
MD uint64_t dpp_move(uint64_t b) {
    const uint32_t lo = dpp_move<CTRL, ROW_MASK>((uint32_t)b, (uint32_t)b), hi = dpp_move<CTRL, ROW_MASK>((uint32_t)(b >> 32), (uint32_t)(b >> 32));
    return ((uint64_t)hi << 32) | lo;
}
MD uint64_t dpp_umax_(uint64_t a, uint64_t b) { return b > a ? b : a; }
MD uint64_t dpp_umin_(uint64_t a, uint64_t b) { return b < a ? b : a; }
MD uint64_t wave_bcast63(uint64_t b) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}
MD uint64_t wave_umax64_dpp(uint64_t v) {
    MIDAS_DPP_REDUCE(v, dpp_umax_)
    return wave_bcast63(v);
}
MD uint64_t wave_umin64_dpp(uint64_t v) {
    MIDAS_DPP_REDUCE(v, dpp_umin_)
    return wave_bcast63(v);
}
MD int wave_isum_dpp(int v) {
    // (a sum is not idempotent: the mirrors add the OTHER half's total to lanes that all hold their own half's - still every
    // element once; the row broadcasts add whole rows)
    uint32_t u = (uint32_t)v;
    u += dpp_move<0xB1>(u, u);
    u += dpp_move<0x4E>(u, u);
    u += dpp_move<0x141>(u, u);
    u += dpp_move<0x140>(u, u);
    u += dpp_move<0x142, 0xa>(0u, u);
    u += dpp_move<0x143, 0xc>(0u, u);
    return (int)(uint32_t)__builtin_amdgcn_readlane((int)u, 63);
}
// ---- floating-point sums in the shuffle butterflies' ORDER, without the shuffles ------------------------------------------------
// `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)` is part of the arithmetic spec (the oracle restates it); each `__shfl_xor` of a
// double is two trips through the LDS crossbar.  The same additions with register moves only:
//   o = 32, 16: v_permlane32_swap / v_permlane16_swap (gfx950) of the value with itself - the two results are {own row | own row}
//     and {partner row | partner row} in some order, so their sum is own + partner on every lane (addition commutes bit for bit);
//   o = 8: row_ror:8 IS lane ^ 8 inside a 16-lane row; after it the row's values repeat with period 8, and then a rotation by
//     4 reads a lane with the same value as lane ^ 4 ((i +- 4) mod 8 = (i mod 8) ^ 4); likewise 2 and 1.
// quarter_sum_ordered: the last four steps alone (score_body.hpp's 16-lane tree).  midas_selftest_wave_sums compares both forms bit for bit.
MD double double_of(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); }
MD double quarter_sum_ordered(double v) {
    v += dpp_move<0x128>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x122>(v);
    v += dpp_move<0x121>(v);
    return v;
}
MD double wave_sum_ordered(double v) {
    {
        const uint64_t b = (uint64_t)__double_as_longlong(v);
        const auto l = __builtin_amdgcn_permlane32_swap((uint32_t)b, (uint32_t)b, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
        v = double_of(l[0], h[0]) + double_of(l[1], h[1]);
    }
    {
        const uint64_t b = (uint64_t)__double_as_longlong(v);
        const auto l = __builtin_amdgcn_permlane16_swap((uint32_t)b, (uint32_t)b, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
        v = double_of(l[0], h[0]) + double_of(l[1], h[1]);
    }
    return quarter_sum_ordered(v);
}
// inclusive prefix sum over the wave's lanes: row shifts by 1, 2, 4, 8 (lanes shifted in from outside the row add zero), then the
// totals of the rows below by the two row broadcasts
MD int wave_iscan_dpp(int v) {
    uint32_t u = (uint32_t)v;
    u += dpp_move<0x111>(0u, u);
    u += dpp_move<0x112>(0u, u);
    u += dpp_move<0x114>(0u, u);
    u += dpp_move<0x118>(0u, u);
    u += dpp_move<0x142, 0xa>(0u, u);
    u += dpp_move<0x143, 0xc>(0u, u);
    return (int)u;
}

}  // namespace midas
