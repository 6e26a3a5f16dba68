// selfsim.hip - the codebook's self-similarity on the matrix cores, for the single-touch evaluation.
//
// eval/single_touch_test.py:35-73 (`top_n_error`): C = pairwise cosine similarity of the K embeddings (sklearn's
// cosine_similarity: a K x K x D GEMM - 1.3 TFLOP at K = 50k, D = 256 - the only dense GEMM in the reference), diagonal
// zeroed, per row the n = 25 best and the smallest pose distance among them.  The K x K matrix (20 GB in float64) never
// exists here: row blocks of R queries are multiplied against all K entries by k_selfsim_mfma into an R x K float32 panel
// of raw dot products, which k_topn_pose_error<dots> consumes (normalisation by the float64 row norms, diagonal, selection).
//
// k_selfsim_mfma: S[i][j] = <E_i, E_j> as exact float32 fma chains on v_mfma_f32_16x16x4_f32, in the accumulation order
// the batched scorer fixed (score.hip k_score_mfma, oracle mo_score_batch_f32): for c (16 d-values), for s in 0..3, for
// g in 0..3: acc = fmaf(E_j[16c+4g+s], E_i[16c+4g+s], acc) - so a panel entry equals midas_score_batch's dot bit for bit.
//   * workgroup tile 128 queries x 128 entries, four waves (2 x 2) of 64 x 64 = 16 accumulator tiles (64 registers) each:
//     128 MFMAs (4096 issue cycles per SIMD) per 32-wide D-chunk against 16 ds_read_b128 - the matrix pipe is the bound;
//   * both operands staged through LDS in 32-wide D-chunks, double-buffered (the next chunk's eight 16-byte loads per
//     thread are in flight while the current one is multiplied), rows padded to 36 floats: a quarter-wave's 16-byte reads
//     (row i, columns 4g .. 4g+3) and the staging stores both spread over all 32 banks;
//   * two workgroups per CU (74 KB of LDS each): one's staging barrier is covered by the other's MFMAs;
//   * XCD-aware tile order: the workgroups of one XCD (every eighth by linear id) walk the panel in patches of 8 x 8
//     tiles, whose 2 x 8 x 128 rows (2 MB at D = 256... 4 MB at D = 512) stay in that XCD's L2 - 64 tiles read them 8 times each;
//   * epilogue: a lane holds four consecutive entries j of one query i (operand A = entry rows): one 16-byte store per
//     accumulator tile, no guards (the panel is padded to whole tiles).
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

using ss_f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SS_T = 128;            // tile edge (queries and entries)
constexpr int SS_DC = 32;            // D-chunk staged per step
#ifndef MIDAS_SS_PAD
#define MIDAS_SS_PAD 4
#endif
#ifndef MIDAS_SS_PIPE
#define MIDAS_SS_PIPE 1  // the second half-chunk's LDS reads under the first half's MFMAs (sched_group_barrier): +1 TFLOP/s
#endif
#ifndef MIDAS_SS_PRIO
#define MIDAS_SS_PRIO 1  // wave priority while it multiplies: 103 -> 109 TFLOP/s at panels of 4096 rows (the staging wave of the other workgroup no longer takes issue slots from it)
#endif
constexpr int SS_LD = SS_DC + MIDAS_SS_PAD;  // LDS row stride in floats
constexpr int SS_PATCH = 8;          // tiles per patch edge

__global__ __launch_bounds__(256, 2) void k_selfsim_mfma(const float* __restrict__ emb, int64_t K, int D, int64_t i0, int tiles_i, int tiles_j,
                                                         float* __restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float s_a[2][SS_T * SS_LD];  // entry rows j (MFMA operand A)
    __shared__ __attribute__((aligned(16))) float s_b[2][SS_T * SS_LD];  // query rows i (operand B)
    // ---- which tile: XCD x = linear id mod 8 walks its own sequence of 8 x 8 patches ----
    const unsigned bid = blockIdx.x, xcd = bid & 7u, l = bid >> 3;
    const int patches_j = (tiles_j + SS_PATCH - 1) / SS_PATCH;
    const unsigned patch = (l / (SS_PATCH * SS_PATCH)) * 8u + xcd, w_in = l % (SS_PATCH * SS_PATCH);
    const int ti_ = (int)(patch / patches_j) * SS_PATCH + (int)(w_in / SS_PATCH);
    const int tj_ = (int)(patch % patches_j) * SS_PATCH + (int)(w_in % SS_PATCH);
    if (ti_ >= tiles_i || tj_ >= tiles_j) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, i = lane & 15;
    const int wi = wave >> 1, wj = wave & 1;
    const int64_t ibase = i0 + (int64_t)ti_ * SS_T, jbase = (int64_t)tj_ * SS_T;
    // ---- staging: per chunk a thread moves one 16-byte piece of four rows of each operand; the eight threads of a row fetch its
    // whole 128-byte line (a vector-cache lookup per LINE, not per piece: rows fetched as half lines by thread pairs kept the
    // load path as busy as the matrix pipe) ----
    const int srow = t >> 3, spiece = t & 7;
    const float* ga[4];
    const float* gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // surplus rows (beyond K): clamped, computed, never read
        const int64_t ja = jbase + srow + 32 * q < K ? jbase + srow + 32 * q : K - 1;
        const int64_t ib = ibase + srow + 32 * q < K ? ibase + srow + 32 * q : K - 1;
        ga[q] = emb + ja * (int64_t)D + 4 * spiece;
        gb[q] = emb + ib * (int64_t)D + 4 * spiece;
    }
    const int soff = srow * SS_LD + 4 * spiece;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // staging registers (named: arrays behind lambdas went to scratch memory)
#define SS_FETCH(d0)                                                                                         \
    do {                                                                                                     \
        ra0 = *reinterpret_cast<const float4*>(ga[0] + (d0)); ra1 = *reinterpret_cast<const float4*>(ga[1] + (d0));  \
        ra2 = *reinterpret_cast<const float4*>(ga[2] + (d0)); ra3 = *reinterpret_cast<const float4*>(ga[3] + (d0));  \
        rb0 = *reinterpret_cast<const float4*>(gb[0] + (d0)); rb1 = *reinterpret_cast<const float4*>(gb[1] + (d0));  \
        rb2 = *reinterpret_cast<const float4*>(gb[2] + (d0)); rb3 = *reinterpret_cast<const float4*>(gb[3] + (d0));  \
    } while (0)
#define SS_STAGE(buf)                                                                                        \
    do {                                                                                                     \
        float* da = &s_a[buf][soff];                                                                         \
        float* db = &s_b[buf][soff];                                                                         \
        *reinterpret_cast<float4*>(da) = ra0;                *reinterpret_cast<float4*>(da + 32 * SS_LD) = ra1;  \
        *reinterpret_cast<float4*>(da + 64 * SS_LD) = ra2;   *reinterpret_cast<float4*>(da + 96 * SS_LD) = ra3;  \
        *reinterpret_cast<float4*>(db) = rb0;                *reinterpret_cast<float4*>(db + 32 * SS_LD) = rb1;  \
        *reinterpret_cast<float4*>(db + 64 * SS_LD) = rb2;   *reinterpret_cast<float4*>(db + 96 * SS_LD) = rb3;  \
    } while (0)
    ss_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = ss_f32x4{0.f, 0.f, 0.f, 0.f};
    SS_FETCH(0);
    SS_STAGE(0);
    __syncthreads();
    const int nchunks = D / SS_DC;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const int dn = (c + 1 < nchunks ? c + 1 : c) * SS_DC;  // (the last chunk re-reads itself: unconditional loads)
        SS_FETCH(dn);                                           // in flight during this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);  // (left alone the compiler sinks the loads behind the MFMAs and waits for them there)
        const float* pa = &s_a[buf][(wj * 64 + i) * SS_LD + 4 * g];
        const float* pb = &s_b[buf][(wi * 64 + i) * SS_LD + 4 * g];
#if defined(MIDAS_SS_PRIO) && MIDAS_SS_PRIO
        __builtin_amdgcn_s_setprio(MIDAS_SS_PRIO);
#endif
#pragma unroll
        for (int cc = 0; cc < SS_DC; cc += 16) {
            float4 fa[4], fb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                fa[k] = *reinterpret_cast<const float4*>(pa + k * 16 * SS_LD + cc);
                fb[k] = *reinterpret_cast<const float4*>(pb + k * 16 * SS_LD + cc);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
#if defined(MIDAS_SS_PIPE) && MIDAS_SS_PIPE
        // order inside the chunk: the first half's eight LDS reads, 32 MFMAs, the second half's reads (their latency under the
        // next 32 MFMAs), the remaining 96 MFMAs - left alone all sixteen reads are issued and waited for before the first MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 96, 0);
#endif
#if defined(MIDAS_SS_PRIO) && MIDAS_SS_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // the other buffer: nobody reads it during this chunk.  Unconditional (after the last chunk it stores a copy nobody
        // reads): behind a condition the compiler sinks the loads into the branch, i.e. behind the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        SS_STAGE(buf ^ 1);
        __syncthreads();
    }
#undef SS_FETCH
#undef SS_STAGE
    // ---- epilogue: acc[a][b][r] = S[query ibase + wi*64 + 16 b + i][entry jbase + wj*64 + 16 a + 4 g + r] ----
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float* orow = out + ((int64_t)ti_ * SS_T + wi * 64 + 16 * b + i) * ldo + jbase + wj * 64 + 4 * g;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4*>(orow + 16 * a) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    }
}

// rows [i0, i0 + R) of the self-similarity as raw float32 dot products: panel[(i - i0) * ldo + j], ldo >= ceil(K / 128) * 128,
// panel rows padded to a multiple of 128
int launch_selfsim_panel(midas_ctx* ctx, const midas_codebook* cb, int64_t i0, int64_t R, float* panel, int64_t ldo) {
    const int tiles_i = (int)ceil_div(R, SS_T), tiles_j = (int)ceil_div(cb->K, SS_T);
    const int patches = (int)(ceil_div(tiles_i, SS_PATCH) * ceil_div(tiles_j, SS_PATCH));
    const unsigned grid = (unsigned)(ceil_div(patches, 8) * 8 * SS_PATCH * SS_PATCH);
    hipLaunchKernelGGL(k_selfsim_mfma, dim3(grid), dim3(256), 0, ctx->stream, (const float*)cb->emb, cb->K, (int)cb->D, i0, tiles_i, tiles_j,
                       panel, ldo);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(selfsim, k_selfsim_mfma)

}  // namespace midas
