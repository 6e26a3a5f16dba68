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
// Round 4: persistent workgroups, one per CU, sixteen waves each (four per SIMD); the 64 codes sit in LDS whole (132 KB at
// D = 512), staged once per workgroup.  The schedule is static and balanced in units of one N-tile (16 rows x 16 codes x D):
// with G = ceil(K / 16) row groups and S = 4 x grid SIMDs, the first floor(G / S) S groups go whole (four tiles, the row
// pieces fetched once) to SIMD g mod S, wave slot (g / S) mod 4; the G mod S groups left over are cut into their four
// tiles and dealt round-robin, one tile a turn (K = 50 000: 3125 groups on 1024 SIMDs = three whole groups per SIMD and 53
// groups left: as whole groups they would give 53 SIMDs a fourth round - 27.3 us at the matrix rate; as 212 single tiles the
// longest SIMD has 13 tiles - 22.2 us).  Round 3 had 196 workgroups of 256 rows on 256 CUs, two 1 KB row pieces in flight per
// wave (47 us, 70 TFLOP/s); a wave now keeps MF_PF = 4 + 4 pieces in flight and twelve to sixteen waves per CU are live.
constexpr int MF_ROWS_PER_WAVE = 16;
// Waves per workgroup.  Rounds 4 - 5 ran twelve (three slots a SIMD, 170 registers a wave: 46.0 us + the prepare launch).  With the
// prepare folded into the staging (round 6) sixteen are better: the 64 codes are then ONE pass of 64 quarter-waves - twelve waves
// are 48, and the sixteen codes left over were a second round trip of a quarter of the workgroup while the rest waited at the
// barrier (staged at 8.7 us, tools/mf_clocks.py) - and the fourth slot takes the tile-units.  K 50k x D 512 x B 64: 46.6 -> 44.1 us
// (70.3 -> 74.3 TFLOP/s), D 256: 32.6 -> 30.1, B 128: 92.1 -> 86.8; K 500k: 381.6 -> 387.6 (127 registers, no spills).
#ifndef MIDAS_MF_WAVES
#define MIDAS_MF_WAVES 16
#endif
constexpr int MF_WAVES = MIDAS_MF_WAVES;
constexpr int MF_CODES = 64;   // codes per pass (4 N-tiles)

#ifndef MIDAS_MF_PF
#define MIDAS_MF_PF 4
#endif
constexpr int MF_PF = MIDAS_MF_PF;         // row pieces in flight per lane and queue (two queues)

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

// one unit of the schedule: rows [row0, row0 + 16) x the NT N-tiles starting at tile t0, over the whole D.
// begin() sends the first burst of row pieces and the epilogue's norms on their way (the kernel calls it for a wave's first unit
// BEFORE the codes are staged: the two latencies overlap), run() multiplies, finish() divides and stores.
// (eb / ld: the staged codes in LDS, or - CODES_LDS false, D too large for the CU's LDS - the float32 code rows in memory)
#ifndef MIDAS_MF_DBG
#define MIDAS_MF_DBG 0  // profiling builds only (tools/ab_score.sh): 1 no row fetches, 2 no LDS reads, 4 no epilogue, 8 fetch-shape probe - wrong scores
#endif
// staged piece idx = (t nc + c) 64 + 16 g + i (see k_score_mfma): the code and the first column it holds
MD int mf_code_of(int idx, int nc) { return 16 * ((idx >> 6) / nc) + (idx & 15); }
MD int mf_d_of(int idx, int nc) { return 16 * ((idx >> 6) % nc) + 4 * ((idx >> 4) & 3); }
template <int NT, bool CODES_LDS>
struct MfUnit {
    f32x4 acc[NT];
    float4 qa[MF_PF], qb[MF_PF];
    double nr[4], cn[NT];
    const float* arow;
    const float* arow_base;  // (MIDAS_MF_DBG & 8 only)
    int64_t row0, K_;
    int t0, nc;

    MD float4 piece(int c) const {
        if (MIDAS_MF_DBG & 1) { float v = (float)c; asm volatile("" : "+v"(v)); return make_float4(v, v, v, v); }
        if (MIDAS_MF_DBG & 8) {  // fetch-shape probe (wrong operands): a load = 4 rows x 256 contiguous bytes, as the GEMV stream reads
            const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
            const int cc = c < nc ? c : nc - 1;
            const int64_t r = row0 + 4 * (cc & 3) + g;
            return *reinterpret_cast<const float4*>(arow_base + (r < K_ ? r : K_ - 1) * (int64_t)(16 * nc) + 64 * (cc >> 2) + 4 * i);
        }
        return *reinterpret_cast<const float4*>(arow + 16 * (c < nc ? c : nc - 1));
    }
    MD void begin(const float* __restrict__ emb, const double* __restrict__ norms, const double* __restrict__ code_norms, int64_t K,
                  int D, int b0, int nb, int64_t row0_, int t0_) {
        const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
        row0 = row0_; t0 = t0_; nc = D / 16;
        arow_base = emb; K_ = K;
        const int64_t row = row0 + i < K ? row0 + i : K - 1;  // clamp: surplus rows are computed and dropped
        arow = emb + row * (int64_t)D + 4 * g;
#pragma unroll
        for (int p = 0; p < MF_PF; ++p) qa[p] = piece(p);
        // the epilogue's divisors travel now (fetched in the epilogue they were a round trip at the end of every unit)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int64_t k = row0 + 4 * g + r; nr[r] = norms[k < K ? k : K - 1]; }
        if (code_norms) set_cn(code_norms, b0, nb);  // (nullptr: the norms are being computed by this workgroup - set_cn() behind its barrier)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the NEXT unit of the same tiles: its first burst of row pieces and its divisors leave before this unit's divisions (finish()
    // reads neither the row pointer nor the piece queue); adopt() makes it the current unit behind them
    MD void prefetch_next(const float* __restrict__ emb, const double* __restrict__ norms, int64_t K, int D, int64_t row0n, double (&nrn)[4]) {
        const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
        const int64_t row = row0n + i < K ? row0n + i : K - 1;
        arow = emb + row * (int64_t)D + 4 * g;
#pragma unroll
        for (int p = 0; p < MF_PF; ++p) qa[p] = piece(p);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int64_t k = row0n + 4 * g + r; nrn[r] = norms[k < K ? k : K - 1]; }
    }
    MD void adopt(int64_t row0n, const double (&nrn)[4]) {
        row0 = row0n;
#pragma unroll
        for (int r = 0; r < 4; ++r) nr[r] = nrn[r];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    MD void set_cn(const double* __restrict__ code_norms, int b0, int nb) {
        const int i = threadIdx.x & 15;
#pragma unroll
        for (int t = 0; t < NT; ++t) { const int b = 16 * (t0 + t) + i; cn[t] = code_norms[b0 + (b < nb ? b : 0)]; }
    }
    // the codes' operands of step c: four 16-byte LDS reads (one per N-tile)
    MD void load_e(float4 (&e)[NT], const float* __restrict__ eb, int ld, int c) const {
        const int cc = c < nc ? c : nc - 1;  // (the step behind the last one re-reads it: unconditional reads, nobody uses them)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (MIDAS_MF_DBG & 2) { float v = (float)(cc + t); asm volatile("" : "+v"(v)); e[t] = make_float4(v, v, v, v); }
            else if (CODES_LDS) e[t] = *reinterpret_cast<const float4*>(eb + (t * nc + cc) * 256);  // (operand order: see k_score_mfma's staging)
            else e[t] = *reinterpret_cast<const float4*>(eb + 16 * t * ld + 16 * cc);
        }
    }
    MD void mul(const float4& a, const float4 (&e)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, e[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, e[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, e[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, e[t].w, acc[t], 0, 0, 0);
    }
    MD void run(const float* __restrict__ codes, int ld) {
        const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
        const float* eb = CODES_LDS ? codes + (t0 * nc * 64 + lane) * 4 : codes + (16 * t0 + i) * ld + 4 * g;
        int c0 = 0;
        // Whole rounds of 2 MF_PF steps, straight-line code (a condition around a step makes the compiler's wait counting give
        // up at the merge: s_waitcnt vmcnt(0) once per round).  The row pieces travel MF_PF .. 2 MF_PF steps ahead of the
        // multiplies in two register queues that alternate (fetched into the registers the MFMAs of a step still read, a piece
        // is parked in a temporary and waited for on the spot); the pieces of the next half round leave as one burst.
        // The codes' LDS reads run ONE STEP ahead of the multiplies, in two register sets that alternate (phase stamps of the
        // form that read a step's operands in front of its own multiplies: the matrix pipe serves the oldest wave first, so one
        // wave at a time runs and its LDS latency - ~200 cycles a step against 512 cycles of multiplies - is exposed: 9.8 us a
        // unit against 6.6 at the matrix rate).  Instruction order pinned with sched_group_barrier: per step the NEXT step's
        // reads, then this step's multiplies; the burst in front of a half round.
        float4 e0[NT], e1[NT];
        load_e(e0, eb, ld, 0);
        static_assert(MF_PF % 2 == 0, "the two operand sets alternate step by step");
        for (; c0 + 2 * MF_PF <= nc; c0 += 2 * MF_PF) {
#pragma unroll
            for (int p = 0; p < MF_PF; ++p) qb[p] = piece(c0 + p + MF_PF);
#pragma unroll
            for (int p = 0; p < MF_PF; p += 2) {
                load_e(e1, eb, ld, c0 + p + 1);
                mul(qa[p], e0);
                load_e(e0, eb, ld, c0 + p + 2);
                mul(qa[p + 1], e1);
            }
            if constexpr (CODES_LDS) {
                __builtin_amdgcn_sched_group_barrier(0x020, MF_PF, 0);
#pragma unroll
                for (int p = 0; p < MF_PF; ++p) {
                    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
                }
            }
#pragma unroll
            for (int p = 0; p < MF_PF; ++p) qa[p] = piece(c0 + p + 2 * MF_PF);
#pragma unroll
            for (int p = 0; p < MF_PF; p += 2) {
                load_e(e1, eb, ld, c0 + MF_PF + p + 1);
                mul(qb[p], e0);
                load_e(e0, eb, ld, c0 + MF_PF + p + 2);
                mul(qb[p + 1], e1);
            }
            if constexpr (CODES_LDS) {
                __builtin_amdgcn_sched_group_barrier(0x020, MF_PF, 0);
#pragma unroll
                for (int p = 0; p < MF_PF; ++p) {
                    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
                }
            }
        }
        if (c0 < nc) {  // D / 16 not a multiple of 2 MF_PF: the last, partial round (wave-uniform conditions; operands read in place)
#pragma unroll
            for (int p = 0; p < MF_PF; ++p) {
                qb[p] = piece(c0 + p + MF_PF);
                if (c0 + p < nc) { load_e(e0, eb, ld, c0 + p); mul(qa[p], e0); }
            }
#pragma unroll
            for (int p = 0; p < MF_PF; ++p)
                if (c0 + MF_PF + p < nc) { load_e(e0, eb, ld, c0 + MF_PF + p); mul(qb[p], e0); }
        }
    }
    // D[row = 4g + r][code = i] in register r of lane (g, i)
    MD void finish(double* __restrict__ out, int64_t K, int b0, int nb) {
        const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
        if (MIDAS_MF_DBG & 4) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
            if (sum == 12345.678f) out[0] = (double)sum;
            return;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int b = 16 * (t0 + t) + i;
            if (b < nb) {
                const double ne = cn[t];
                const int64_t k0 = row0 + 4 * g;  // the lane's four rows are consecutive: one 32-byte run of the code's scores
                double* o = out + (int64_t)(b0 + b) * K + k0;
                if (k0 + 3 < K && ((uintptr_t)o & 15) == 0) {
                    double2 lo, hi;
                    lo.x = (double)acc[t][0] / (ne * nr[0]); lo.y = (double)acc[t][1] / (ne * nr[1]);
                    hi.x = (double)acc[t][2] / (ne * nr[2]); hi.y = (double)acc[t][3] / (ne * nr[3]);
                    reinterpret_cast<double2*>(o)[0] = lo;
                    reinterpret_cast<double2*>(o)[1] = hi;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + r < K) o[r] = (double)acc[t][r] / (ne * nr[r]);
                }
            }
        }
    }
};

#if MIDAS_MF_DBG & 16  // profiling build: wall-clock stamps (100 MHz) of every wave's phases, tools/mf_clocks.py
__device__ long long g_mf_clk[256 * MF_WAVES * 8];
extern "C" __attribute__((visibility("default"))) int midas_debug_mf_clocks(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_clk), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#define MF_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_mf_clk[(blockIdx.x * MF_WAVES + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(k)
#endif

// FOLD (round 6, with CODES_LDS): the workgroup converts the float64 codes itself while it stages them and forms their norms (the
// sums of k_codes_prepare in its order: a quarter-wave per code, lane s over the columns s, s + 16, .., then the 16-lane tree) -
// no k_codes_prepare launch in front, no float32 copy of the codes in memory; codes64 = the caller's B x D float64 codes.
template <bool CODES_LDS, bool FOLD = false>
__global__ __launch_bounds__(64 * MF_WAVES) void k_score_mfma(const float* __restrict__ emb, const double* __restrict__ norms,
                                                     const float* __restrict__ codes32, const double* __restrict__ code_norms,
                                                     double* __restrict__ out, int64_t K, int D, int B, int b0,
                                                     const double* __restrict__ codes64 = nullptr) {
    // The 64 codes in LDS in the ORDER THE OPERAND READS TAKE THEM: sixteen-byte piece ((t nc + c) 64 + 16 g + i) = code 16 t + i,
    // columns 16 c + 4 g .. + 3 (t: N-tile, c: step of 16 columns, lane (g, i) of the wave) - a step's operand read of one N-tile is
    // 64 consecutive pieces, lane l the l-th: no bank is asked twice (rows of D + 4 floats put lanes (g, i) and (g + 1, i - 1) on one
    // bank: 1.65 M conflict cycles against 3.56 M LDS cycles per launch, profiles/r04_d_pmc_score_mfma.txt), and the staging store of
    // piece idx goes to piece idx - conflict-free as well.
    extern __shared__ __attribute__((aligned(16))) float s_e[];  // [4 N-tiles][D / 16 steps][64 lanes] x 4 floats
    const int wave = threadIdx.x >> 6;
    const int nb = B - b0 < MF_CODES ? B - b0 : MF_CODES;  // codes of this pass
    const int ld = D;
    const float* eb = CODES_LDS ? s_e : codes32 + (int64_t)b0 * D;
    // ---- the schedule (see above): this wave = slot `slot` of SIMD `simd` ----
    constexpr int SL = MF_WAVES / 4;
    const int64_t S = (int64_t)gridDim.x * 4, simd = (int64_t)blockIdx.x * 4 + (wave & 3);
    const int slot = wave >> 2;
    const int64_t G = (K + MF_ROWS_PER_WAVE - 1) / MF_ROWS_PER_WAVE;
    const int64_t rounds = G / S, Gw = rounds * S;   // whole groups: `rounds` per SIMD
    // The wave's leftover tile-unit, if it has one (one N-tile a turn: tile-unit q = 4 (group - Gw) + tile, to
    // the slot that gets the matrix pipe LAST - the pipe serves the oldest wave first - and runs there FIRST: a tile-unit is a
    // chain of fetches with little to multiply, 10 us that ended 6 us behind everything else when it ran after the wave's group).
    const int64_t Q = 4 * (G - Gw);
    MF_STAMP(0);
    float4 stage[(CODES_LDS && !FOLD) ? 11 : 1];
    const int q4 = D / 4, total = MF_CODES * q4, ncq = D / 16;
    constexpr int NT_ = 64 * MF_WAVES;
    double* s_cn = reinterpret_cast<double*>(s_e + (size_t)MF_CODES * D);  // FOLD: the 64 code norms behind the staged codes
    constexpr int FOLD_MAXC = 40;  // columns per lane: D <= 640
    double fv[FOLD ? FOLD_MAXC : 1];
    constexpr int NQ = NT_ / 16;  // quarter-waves of the workgroup: pass p of the staging gives quarter-wave q the code q + NQ p
    if constexpr (CODES_LDS && FOLD) {
        const int qw = (int)threadIdx.x >> 4, s = (int)threadIdx.x & 15;
        const int b = qw < MF_CODES ? qw : MF_CODES - 1;
        const double* c = codes64 + (int64_t)(b0 + (b < nb ? b : 0)) * D + s;
#pragma unroll
        for (int k = 0; k < FOLD_MAXC; ++k) fv[k] = c[16 * (k < ncq ? k : ncq - 1)];
    }
    if constexpr (CODES_LDS && !FOLD) {
        // The 64 codes (padded with zero rows by k_codes_prepare) are requested FIRST: loads come back in order, so the first
        // burst of row pieces (memory) in front of them made the staging wait for it (6.8 us to the barrier); behind them
        // it travels while the codes are written to LDS.  Eleven 16-byte pieces per thread at D = 512 (more rounds beyond).
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int idx = k * NT_ + (int)threadIdx.x;
            const int ic = idx < total ? idx : total - 1;
            const int b = mf_code_of(ic, ncq), d = mf_d_of(ic, ncq);
            stage[k] = *reinterpret_cast<const float4*>(&codes32[(int64_t)(b0 + b) * D + d]);
        }
    }
    MfUnit<4, CODES_LDS> u4;
    const bool first4 = slot < rounds;
    if (first4) u4.begin(emb, norms, FOLD ? nullptr : code_norms, K, D, b0, nb, ((int64_t)slot * S + simd) * MF_ROWS_PER_WAVE, 0);
    if constexpr (CODES_LDS && FOLD) {
        const int qw0 = (int)threadIdx.x >> 4, s = (int)threadIdx.x & 15;
#pragma unroll
        for (int k = 0; k < FOLD_MAXC; ++k) asm volatile("" : "+v"(fv[k]));  // (pinned: requested above, in front of the first row burst)
        MF_STAMP(5);  // the codes' float64 values are there
        for (int pass = 0; pass * NQ < MF_CODES; ++pass) {
            const int b = qw0 + NQ * pass;
            if (b >= MF_CODES) break;  // (with twelve waves: quarter-waves 16 .. 47 have no second code)
            if (pass >= 1) {
                const double* c = codes64 + (int64_t)(b0 + (b < nb ? b : 0)) * D + s;
#pragma unroll
                for (int k = 0; k < FOLD_MAXC; ++k) fv[k] = c[16 * (k < ncq ? k : ncq - 1)];
            }
            const bool real = b < nb;
            double acc2 = 0.0;
            // piece ((t nc + k) 64 + 16 g + i) holds code 16 t + i, columns 16 k + 4 g .. + 3: column s + 16 k is component s & 3 of g = s >> 2
            float* o = s_e + ((size_t)((b >> 4) * ncq) * 64 + 16 * (s >> 2) + (b & 15)) * 4 + (s & 3);
#pragma unroll
            for (int k = 0; k < FOLD_MAXC; ++k)
                if (k < ncq) {
                    o[(size_t)k * 256] = real ? (float)fv[k] : 0.0f;
                    acc2 = fma_(fv[k], fv[k], acc2);
                }
            acc2 = quarter_reduce(acc2);
            if (s == 0) { const double n = __builtin_sqrt(acc2); s_cn[b] = real ? (n < COS_EPS ? COS_EPS : n) : 1.0; }
        }
        MF_STAMP(6);  // converted, stored, norms formed: at the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (first4) u4.set_cn(s_cn - b0, b0, nb);
    }
    if constexpr (CODES_LDS && !FOLD) {
        float4* s4 = reinterpret_cast<float4*>(s_e);
#pragma unroll
        for (int k = 0; k < 11; ++k)  // (pinned here: the compiler otherwise sinks each load into its store's condition)
            asm volatile("" : "+v"(stage[k].x), "+v"(stage[k].y), "+v"(stage[k].z), "+v"(stage[k].w));
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int idx = k * NT_ + (int)threadIdx.x;
            if (idx < total) s4[idx] = stage[k];
        }
        for (int base = 11 * NT_; base < total; base += 6 * NT_) {  // D > 528: the rest, six pieces a round
            float4 v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int idx = base + k * NT_ + (int)threadIdx.x;
                const int ic = idx < total ? idx : total - 1;
                const int b = mf_code_of(ic, ncq), d = mf_d_of(ic, ncq);
                v[k] = *reinterpret_cast<const float4*>(&codes32[(int64_t)(b0 + b) * D + d]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int idx = base + k * NT_ + (int)threadIdx.x;
                if (idx < total) s4[idx] = v[k];
            }
        }
        // the barrier orders the LDS writes only: __syncthreads() would also drain the wave's loads (s_waitcnt vmcnt(0) in front of
        // s_barrier) - the first burst of row pieces and the divisors, which are meant to travel across it (3 us of the 5 to here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    MF_STAMP(1);
    __builtin_amdgcn_s_setprio(1);
    // (dealt CU by CU first - tile-unit q to CU q mod grid, SIMD (q / grid) mod 4 - so that 212 of them are one extra tile on 212
    // different CUs, not four on each of 53)
    for (int64_t q = (int64_t)blockIdx.x + (int64_t)gridDim.x * (wave & 3); q < Q; q += S) {
        if (SL - 1 - (int)((q / S) % SL) != slot) continue;
        MfUnit<1, CODES_LDS> u1;
        u1.begin(emb, norms, FOLD ? (const double*)(s_cn - b0) : code_norms, K, D, b0, nb, (Gw + (q >> 2)) * MF_ROWS_PER_WAVE, (int)(q & 3));
        u1.run(eb, ld);
        u1.finish(out, K, b0, nb);
    }
    for (int64_t r = slot; r < rounds; r += SL) {
        u4.run(eb, ld);
        MF_STAMP(2);
        const bool more = r + SL < rounds;  // (the wave's next unit: same tiles, same code norms)
        const int64_t row0n = ((r + SL) * S + simd) * MF_ROWS_PER_WAVE;
        double nrn[4] = {0.0, 0.0, 0.0, 0.0};
        if (more) u4.prefetch_next(emb, norms, K, D, row0n, nrn);
        u4.finish(out, K, b0, nb);
        if (more) u4.adopt(row0n, nrn);
        MF_STAMP(3);
    }
    MF_STAMP(4);
}

int launch_score_batch(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes, double* scores) {
    if (cb->dtype != MIDAS_F32 || cb->D % 16 != 0 || (uintptr_t)cb->emb % 16 != 0)
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "midas_score_batch", "needs float32 embeddings with D % 16 == 0");
    const int D = cb->D;
    // the 64 staged codes fit the CU's 160 KB of LDS up to D = 640; beyond (D = 1024) the waves read the float32 code rows from
    // memory (256 KB: cache-resident) - the same arithmetic, slower
    const bool codes_lds = (size_t)MF_CODES * D * sizeof(float) <= 160 * 1024;
    // folded staging (the workgroups convert the float64 codes and form their norms themselves: no k_codes_prepare launch) where the
    // 64 norms fit behind the staged codes and a lane's columns fit its registers (D <= 624: D = 128 .. 512)
    // (K 50k x D 512 x B 64: 52.3 -> 46.6 us per call, 62.7 -> 70.3 TFLOP/s; the k_codes_prepare form stays for D = 640 and D > 640)
#ifdef MIDAS_MF_NOFOLD  // (A/B builds: the k_codes_prepare form at every D)
    const bool fold = false;
#else
    const bool fold = codes_lds && D <= 624 && (uintptr_t)codes % 8 == 0;
#endif
    const size_t lds = codes_lds ? (size_t)MF_CODES * D * sizeof(float) + (fold ? MF_CODES * sizeof(double) : 0) : 0;
    const int Bpad = (int)ceil_div(B, MF_CODES) * MF_CODES;
    void *cn = nullptr, *c32 = nullptr;
    int rc;
    if (!fold) {
        if ((rc = midas_scratch(ctx, (size_t)B * sizeof(double), &cn))) return rc;
        if ((rc = midas_scratch(ctx, (size_t)Bpad * D * sizeof(float), &c32))) return rc;
        hipLaunchKernelGGL(k_codes_prepare, dim3((unsigned)ceil_div(Bpad, 16)), dim3(256), 0, ctx->stream, codes, (float*)c32,
                           (double*)cn, B, Bpad, D);
    }
    // per device (the same rule as launch_presort, particles.hip): a second GPU's context must not inherit the first one's
    // dynamic-LDS limit and CU count
    constexpr int MAXDEV = 64;
    static bool attr_set[MAXDEV] = {};
    static int ncu_dev[MAXDEV] = {};
    const int di = ctx->device >= 0 && ctx->device < MAXDEV ? ctx->device : 0;
    if (!attr_set[di] || ctx->device != di) {
        MIDAS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_score_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MIDAS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_score_mfma<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipDeviceProp_t prop;
        ncu_dev[di] = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        attr_set[di] = true;
    }
    const int ncu = ncu_dev[di];
    // one persistent workgroup per CU (fewer when the codebook has fewer row groups than that many SIMDs)
    const int64_t G = ceil_div(cb->K, MF_ROWS_PER_WAVE);
    const unsigned grid = (unsigned)(G < (int64_t)ncu * 4 ? ceil_div(G, 4) : ncu);
    for (int b0 = 0; b0 < B; b0 += MF_CODES) {
        if (fold)
            hipLaunchKernelGGL((k_score_mfma<true, true>), dim3(grid), dim3(64 * MF_WAVES), lds, ctx->stream, (const float*)cb->emb, cb->norms,
                               (const float*)nullptr, (const double*)nullptr, scores, cb->K, D, B, b0, codes);
        else if (codes_lds)
            hipLaunchKernelGGL(k_score_mfma<true>, dim3(grid), dim3(64 * MF_WAVES), lds, ctx->stream, (const float*)cb->emb, cb->norms,
                               (const float*)c32, (const double*)cn, scores, cb->K, D, B, b0, (const double*)nullptr);
        else
            hipLaunchKernelGGL(k_score_mfma<false>, dim3(grid), dim3(64 * MF_WAVES), 0, ctx->stream, (const float*)cb->emb, cb->norms,
                               (const float*)c32, (const double*)cn, scores, cb->K, D, B, b0, (const double*)nullptr);
    }
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(score, k_codes_prepare)

}  // namespace midas
