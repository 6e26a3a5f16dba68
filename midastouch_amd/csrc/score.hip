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
#include "score_body.hpp"

namespace midas {

template <typename T, int NJ, int MODE>
__global__ __launch_bounds__(256) void k_score_reg(const T* __restrict__ emb, const double* __restrict__ norms,
                                                   const double* __restrict__ code, double* __restrict__ out,
                                                   int64_t K) {
    score_wave<T, NJ, MODE>(emb, norms, code, out, K, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
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

// ================================================================================================
// Batched scoring on the matrix cores: scores[b][k] for B tactile codes in ONE pass over the codebook
// ================================================================================================
// S = C (K x D) . E^T (D x B) as a float32 GEMM on v_mfma_f32_16x16x4_f32 (exact f32 fma chains at the f32
// vector rate; bf16/fp8 would lose the bits the softmax weights need).  Tile: a wave owns 16 codebook rows
// (MFMA M) x up to 64 codes (4 N-tiles of 16, 16 accumulator registers); the reduction runs over D in
// chunks of 16: each lane fetches ONE float4 of its row (A operand: lane (g = l>>4, i = l&15) holds
// C[row0+i][16c+4g .. +3], so a wave-level load reads 16 rows x 64 contiguous bytes) and feeds its four
// components to four consecutive MFMAs; the codes sit in LDS as float32 with the same (g, n) ownership.
// Accumulation order (the spec the oracle restates, mo_score_batch_f32): for c, for s in 0..3, for g in
// 0..3: acc = fmaf(C[k][16c+4g+s], E[b][16c+4g+s], acc).  Epilogue: float64 division by the norms.
// HBM: K*D*4 bytes once (102 MB at c2) vs B times for the GEMV loop; MFMA: 2*K*D*B flop.
constexpr int MF_ROWS_PER_WAVE = 16;
#ifndef MIDAS_MF_WAVES
#define MIDAS_MF_WAVES 16
#endif
#ifndef MIDAS_MF_DC
#define MIDAS_MF_DC 512
#endif
constexpr int MF_WAVES = MIDAS_MF_WAVES;   // waves per workgroup: they share one staged copy of the codes
constexpr int MF_CODES = 64;   // codes per pass (4 N-tiles)
constexpr int MF_DC = MIDAS_MF_DC;         // D-chunk staged in LDS
constexpr int MF_PAD = 4;      // floats of padding per staged code row (bank spread)
#ifndef MIDAS_MF_PF
#define MIDAS_MF_PF 2
#endif
constexpr int MF_PF = MIDAS_MF_PF;         // row pieces in flight per lane and queue

using f32x4 = __attribute__((ext_vector_type(4))) float;

// codes (B x D float64) -> float32 rows padded to a multiple of 64 codes (zeros), row stride D, and max(|e_b|, eps) of
// every code in float64 (sequential fma chain per quarter-wave lane, as in the single-code kernel).  One quarter-wave
// per code; the loads of a lane are issued eight at a time (a rolled loop would wait for each one).
__global__ __launch_bounds__(256) void k_codes_prepare(const double* __restrict__ codes, float* __restrict__ out,
                                                       double* __restrict__ norms, int B, int Bpad, int D) {
    const int lane = threadIdx.x & 63, s = lane & 15;
    const int b = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    if (b >= Bpad) return;
    const bool real = b < B;
    const double* c = codes + (int64_t)(real ? b : 0) * D;
    float* o = out + (int64_t)b * D;
    double acc = 0.0;
    for (int j0 = s; j0 < D; j0 += 16 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = j0 + 16 * k;
            v[k] = c[j < D ? j : D - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = j0 + 16 * k;
            if (j < D) {
                o[j] = real ? (float)v[k] : 0.0f;
                acc = fma_(v[k], v[k], acc);
            }
        }
    }
    acc = quarter_reduce(acc);
    if (real && s == 0) { const double n = __builtin_sqrt(acc); norms[b] = n < COS_EPS ? COS_EPS : n; }
}

__global__ __launch_bounds__(64 * MF_WAVES) void k_score_mfma(const float* __restrict__ emb, const double* __restrict__ norms,
                                                     const float* __restrict__ codes32, const double* __restrict__ code_norms,
                                                     double* __restrict__ out, int64_t K, int D, int B, int b0) {
    extern __shared__ __attribute__((aligned(16))) float s_e[];  // [MF_CODES][dc + MF_PAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int64_t row0 = ((int64_t)blockIdx.x * MF_WAVES + wave) * MF_ROWS_PER_WAVE;
    const int nb = B - b0 < MF_CODES ? B - b0 : MF_CODES;  // codes of this pass
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t row = row0 + i < K ? row0 + i : K - 1;  // clamp: surplus rows are computed and dropped
    const float* arow = emb + row * (int64_t)D;
    for (int d0 = 0; d0 < D; d0 += MF_DC) {
        const int dc = D - d0 < MF_DC ? D - d0 : MF_DC;
        const int ld = dc + MF_PAD;
        __syncthreads();
        // stage this D-chunk of the 64 codes: eight 16-byte pieces per thread and round trip (a rolled copy loop waits
        // for every load before it issues the next one)
        const int q4 = dc / 4, total = MF_CODES * q4;
        for (int base = 0; base < total; base += 8 * 64 * MF_WAVES) {
            float4 v[8];
            int off[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = base + k * 64 * MF_WAVES + (int)threadIdx.x;
                const int ic = idx < total ? idx : total - 1;
                const int b = ic / q4, d = (ic - b * q4) * 4;
                off[k] = idx < total ? b * ld + d : -1;
                v[k] = *reinterpret_cast<const float4*>(&codes32[(int64_t)(b0 + b) * D + d0 + d]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (off[k] >= 0) *reinterpret_cast<float4*>(&s_e[off[k]]) = v[k];
        }
        __syncthreads();
        // The row pieces travel MF_PF steps ahead of the multiplies.  Two register queues alternate: a piece is fetched
        // into the queue that is NOT being multiplied from - fetched into the registers the MFMAs of the step still read,
        // the compiler parks it in a temporary and waits for it on the spot (s_waitcnt vmcnt(0) in every step).
        auto piece = [&](int c) {
            const int cc = c < dc ? c : dc - 16;
            return *reinterpret_cast<const float4*>(arow + d0 + cc + 4 * g);
        };
        auto step = [&](const float4& a, int c) {
            const float* eb = reinterpret_cast<const float*>(__builtin_assume_aligned(s_e, 16));
            float4 e[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) e[t] = *reinterpret_cast<const float4*>(&eb[(16 * t + i) * ld + c + 4 * g]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, e[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, e[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, e[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, e[t].w, acc[t], 0, 0, 0);
        };
        float4 qa[MF_PF], qb[MF_PF];
#pragma unroll
        for (int p = 0; p < MF_PF; ++p) qa[p] = piece(16 * p);
        for (int c0 = 0; c0 < dc; c0 += 32 * MF_PF) {
#pragma unroll
            for (int p = 0; p < MF_PF; ++p) {
                const int c = c0 + 16 * p;
                qb[p] = piece(c + 16 * MF_PF);
                if (c < dc) step(qa[p], c);  // wave-uniform
            }
#pragma unroll
            for (int p = 0; p < MF_PF; ++p) {
                const int c = c0 + 16 * (MF_PF + p);
                qa[p] = piece(c + 16 * MF_PF);
                if (c < dc) step(qb[p], c);
            }
        }
    }
    // D[row = 4g + r][code = i] in register r of lane (g, i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int b = 16 * t + i;
        if (b < nb) {
            const double ne = code_norms[b0 + b];
            const int64_t k0 = row0 + 4 * g;  // the lane's four rows are consecutive: one 32-byte run of the code's scores
            double* o = out + (int64_t)(b0 + b) * K + k0;
            if (k0 + 3 < K && ((uintptr_t)o & 15) == 0) {
                double2 lo, hi;
                lo.x = (double)acc[t][0] / (ne * norms[k0]); lo.y = (double)acc[t][1] / (ne * norms[k0 + 1]);
                hi.x = (double)acc[t][2] / (ne * norms[k0 + 2]); hi.y = (double)acc[t][3] / (ne * norms[k0 + 3]);
                reinterpret_cast<double2*>(o)[0] = lo;
                reinterpret_cast<double2*>(o)[1] = hi;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + r < K) o[r] = (double)acc[t][r] / (ne * norms[k0 + r]);
            }
        }
    }
}

int launch_score_batch(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes, double* scores) {
    if (cb->dtype != MIDAS_F32 || cb->D % 16 != 0 || (uintptr_t)cb->emb % 16 != 0)
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "midas_score_batch", "needs float32 embeddings with D % 16 == 0");
    const int D = cb->D;
    const int Bpad = (int)ceil_div(B, MF_CODES) * MF_CODES;
    void *cn, *c32;
    int rc = midas_scratch(ctx, (size_t)B * sizeof(double), &cn);
    if (rc) return rc;
    if ((rc = midas_scratch(ctx, (size_t)Bpad * D * sizeof(float), &c32))) return rc;
    hipLaunchKernelGGL(k_codes_prepare, dim3((unsigned)ceil_div(Bpad, 16)), dim3(256), 0, ctx->stream, codes, (float*)c32,
                       (double*)cn, B, Bpad, D);
    const int dc = D < MF_DC ? D : MF_DC;
    const size_t lds = (size_t)MF_CODES * (dc + MF_PAD) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        MIDAS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_score_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const unsigned grid = (unsigned)ceil_div(cb->K, MF_ROWS_PER_WAVE * MF_WAVES);
    for (int b0 = 0; b0 < B; b0 += MF_CODES) {
        hipLaunchKernelGGL(k_score_mfma, dim3(grid), dim3(64 * MF_WAVES), lds, ctx->stream, (const float*)cb->emb, cb->norms,
                           (const float*)c32, (const double*)cn, scores, cb->K, D, B, b0);
    }
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(score, k_codes_prepare)

}  // namespace midas
