// score.hip - K1: cosine score of a tactile code against every codebook row (gfx950).
//
// Replaces the reference's per-particle cosine over a gathered (N, D) float64 matrix
// (modules/particle_filter.py:455-457 after tactile_tree.py:54-58): cos(e, C[idx[n]]) only depends on
// idx[n], so the codebook is scored once per frame (K rows) and particles gather the scalar.
//
// HBM-bound streaming GEMV.  Layout: K x D row-major (float32 when the embeddings are float32 casts,
// float64 otherwise).  A quarter-wave (16 lanes) owns one row: lane s reads the 16-byte pieces
// [64 j + 4 s, +4) for j = 0..D/64-1, so every wave-level load covers four 256-byte row segments.
// The code e sits in registers as float64 (products of float32 values are exact in float64, the
// accumulation is float64 like the reference's), reduced over the 16 lanes with an xor butterfly.
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
__global__ __launch_bounds__(256) void k_score_reg(const T* __restrict__ emb, const double* __restrict__ norms,
                                                   const double* __restrict__ code, double* __restrict__ out,
                                                   int64_t K) {
    constexpr int D = NJ * 64;
    const int lane = threadIdx.x & 63;
    const int s = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
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

// generic D: one quarter-wave per row, scalar strided loads (fallback for unusual D)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_score_generic(const T* __restrict__ emb, const double* __restrict__ norms,
                                                       const double* __restrict__ code, double* __restrict__ out,
                                                       int64_t K, int D) {
    const int lane = threadIdx.x & 63;
    const int s = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < K;
    const T* r = emb + (live ? row : 0) * (int64_t)D;
    double acc = 0.0, ne2 = 0.0;
    for (int j = s; j < D; j += 16) {
        double x = (double)r[j];
        if (MODE == 0) {
            double ev = code[j];
            acc = fma_(x, ev, acc);
            ne2 = fma_(ev, ev, ne2);
        } else {
            acc = fma_(x, x, acc);
        }
    }
    acc = quarter_reduce(acc);
    ne2 = quarter_reduce(ne2);
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

template <typename T, int MODE>
static int dispatch(midas_ctx* ctx, int64_t K, int32_t D, const T* emb, const double* norms, const double* code,
                    double* out) {
    if (K == 0) return MIDAS_OK;
    dim3 grid((unsigned)ceil_div(K, 16)), block(256);
    const bool aligned = ((uintptr_t)emb % 16 == 0) && (MODE == 1 || (uintptr_t)code % 16 == 0);
    if (aligned && D == 512) {
        hipLaunchKernelGGL((k_score_reg<T, 8, MODE>), grid, block, 0, ctx->stream, emb, norms, code, out, K);
    } else if (aligned && D == 256) {
        hipLaunchKernelGGL((k_score_reg<T, 4, MODE>), grid, block, 0, ctx->stream, emb, norms, code, out, K);
    } else if (aligned && D == 128) {
        hipLaunchKernelGGL((k_score_reg<T, 2, MODE>), grid, block, 0, ctx->stream, emb, norms, code, out, K);
    } else if (aligned && D == 1024) {
        hipLaunchKernelGGL((k_score_reg<T, 16, MODE>), grid, block, 0, ctx->stream, emb, norms, code, out, K);
    } else {
        hipLaunchKernelGGL((k_score_generic<T, MODE>), grid, block, 0, ctx->stream, emb, norms, code, out, K, D);
    }
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_row_norms(midas_ctx* ctx, int64_t K, int32_t D, const void* emb, int32_t dtype, double* norms) {
    if (dtype == MIDAS_F32) return dispatch<float, 1>(ctx, K, D, (const float*)emb, nullptr, nullptr, norms);
    return dispatch<double, 1>(ctx, K, D, (const double*)emb, nullptr, nullptr, norms);
}

int launch_score(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes, double* scores) {
    for (int32_t b = 0; b < B; ++b) {
        int rc;
        if (cb->dtype == MIDAS_F32)
            rc = dispatch<float, 0>(ctx, cb->K, cb->D, (const float*)cb->emb, cb->norms, codes + (int64_t)b * cb->D,
                                    scores + (int64_t)b * cb->K);
        else
            rc = dispatch<double, 0>(ctx, cb->K, cb->D, (const double*)cb->emb, cb->norms,
                                     codes + (int64_t)b * cb->D, scores + (int64_t)b * cb->K);
        if (rc) return rc;
    }
    return MIDAS_OK;
}

}  // namespace midas
