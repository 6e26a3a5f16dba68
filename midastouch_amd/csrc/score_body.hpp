// score_body.hpp - the per-wave body of the codebook scoring (K1), shared by k_score_reg (score.hip) and
// the fused front kernel of the step (particles.hip).  A wave owns four consecutive rows (one per quarter-wave).
#pragma once
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

constexpr double COS_EPS = 1e-8;

MD double quarter_reduce(double v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

template <typename T>
struct Vec4;
template <>
struct Vec4<float> { using type = float4; };
template <>
struct Vec4<double> { using type = double4; };

// MODE 0: scores = <e,row> / (max(|e|,eps) * norms[row]);  MODE 1: norms[row] = max(|row|, eps)
template <typename T, int NJ, int MODE>
MD void score_wave(const T* __restrict__ emb, const double* __restrict__ norms, const double* __restrict__ code,
                   double* __restrict__ out, int64_t K, int64_t wave) {
    constexpr int D = NJ * 64;
    const int lane = threadIdx.x & 63;
    const int s = lane & 15;
    const int64_t row = wave * 4 + (lane >> 4);
    double e[NJ * 4];
    double ne2 = 0.0;
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const double2* p = reinterpret_cast<const double2*>(code + j * 64 + s * 4);
            double2 a = p[0], b = p[1];
            e[j * 4 + 0] = a.x; e[j * 4 + 1] = a.y; e[j * 4 + 2] = b.x; e[j * 4 + 3] = b.y;
        }
#pragma unroll
        for (int i = 0; i < NJ * 4; ++i) ne2 = fma_(e[i], e[i], ne2);
        ne2 = quarter_reduce(ne2);
    }
    const bool live = row < K;
    const T* r = emb + (live ? row : 0) * (int64_t)D + s * 4;
    using V = typename Vec4<T>::type;
    V v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const V*>(r + j * 64);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double x0 = (double)v[j].x, x1 = (double)v[j].y, x2 = (double)v[j].z, x3 = (double)v[j].w;
        if (MODE == 0) {
            acc = fma_(x0, e[j * 4 + 0], acc);
            acc = fma_(x1, e[j * 4 + 1], acc);
            acc = fma_(x2, e[j * 4 + 2], acc);
            acc = fma_(x3, e[j * 4 + 3], acc);
        } else {
            acc = fma_(x0, x0, acc);
            acc = fma_(x1, x1, acc);
            acc = fma_(x2, x2, acc);
            acc = fma_(x3, x3, acc);
        }
    }
    acc = quarter_reduce(acc);
    if (live && s == 0) {
        if (MODE == 0) {
            double ne = __builtin_sqrt(ne2);
            ne = ne < COS_EPS ? COS_EPS : ne;
            out[row] = acc / (ne * norms[row]);
        } else {
            double nr = __builtin_sqrt(acc);
            out[row] = nr < COS_EPS ? COS_EPS : nr;
        }
    }
}

}  // namespace midas
