// score_body.hpp - the per-wave body of the codebook scoring (K1), shared by k_score_reg (score.hip) and
// the fused front kernel of the step (particles.hip).  A wave owns four consecutive rows (one per quarter-wave).
#pragma once
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

constexpr double COS_EPS = 1e-8;

MD double quarter_reduce(double v) { return quarter_sum_ordered(v); }  // (xor 8, 4, 2, 1 inside the 16-lane row, by register moves: midas_math.hpp)

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

// R consecutive quads of rows by one wave (MODE 0), the R x NJ row pieces requested together.  The scoring waves of the fused
// front with one quad each were 125 k one-wave workgroups for c4's 500 k rows; with two quads a wave the dense frame of c4 takes
// 215 us instead of 228 (four or six: no further gain, six spills).  What the fused stream loses against k_score_reg alone
// (172 us) is the particle waves' phase: ~2 TB/s of scattered requests during the launch's first 25 - 30 us leave the stream
// little of the memory system; register caps that give the stream more wave slots in that phase (three / four waves a SIMD)
// change nothing for it and cost the particle waves 5 / 16 %.  Round 6 measured the two remaining ideas on c2's dense front (30.0 us):
// the row pieces as non-temporal loads (so that the stream stops evicting list records from the L2) 33.2 us, the stream's waves at
// s_setprio 1 / 3: 32.3 / 32.7, both together 34.4 - all slower, none kept (tools/ab_dense_stream.sh, gpurun_out r06_ab_dense).
// Same arithmetic per row as score_wave (same lane ownership, same
// summation order): bit-identical scores.
template <typename T, int NJ, int R>
MD void score_wave_multi(const T* __restrict__ emb, const double* __restrict__ norms, const double* __restrict__ code,
                         double* __restrict__ out, int64_t K, int64_t wave0) {
    constexpr int D = NJ * 64;
    const int lane = threadIdx.x & 63;
    const int s = lane & 15;
    using V = typename Vec4<T>::type;
    V v[R][NJ];
    double nr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = (wave0 + r) * 4 + (lane >> 4);
        const T* p = emb + (row < K ? row : 0) * (int64_t)D + s * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[r][j] = *reinterpret_cast<const V*>(p + j * 64);
        nr[r] = norms[row < K ? row : 0];  // (travels with the rows)
    }
    double e[NJ * 4];
    double ne2 = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double2* p = reinterpret_cast<const double2*>(code + j * 64 + s * 4);
        double2 a = p[0], b = p[1];
        e[j * 4 + 0] = a.x; e[j * 4 + 1] = a.y; e[j * 4 + 2] = b.x; e[j * 4 + 3] = b.y;
    }
#pragma unroll
    for (int i = 0; i < NJ * 4; ++i) ne2 = fma_(e[i], e[i], ne2);
    ne2 = quarter_reduce(ne2);
    double ne = __builtin_sqrt(ne2);
    ne = ne < COS_EPS ? COS_EPS : ne;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = (wave0 + r) * 4 + (lane >> 4);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc = fma_((double)v[r][j].x, e[j * 4 + 0], acc);
            acc = fma_((double)v[r][j].y, e[j * 4 + 1], acc);
            acc = fma_((double)v[r][j].z, e[j * 4 + 2], acc);
            acc = fma_((double)v[r][j].w, e[j * 4 + 3], acc);
        }
        acc = quarter_reduce(acc);
        if (row < K && s == 0) out[row] = acc / (ne * nr[r]);
    }
}

// ---- sparse scoring inside the particle kernels ----------------------------------------------------------------
// cos(code, C_k) is only ever read at k = nearest entry of some particle (modules/particle_filter.py:449-457 scores the
// gathered rows; the dense pass over all K rows is this implementation's restructuring of it).  Once the cloud has
// gathered, a few hundred of the K rows are anybody's nearest entry: scoring exactly those reads 10^5 bytes instead of
// 10^8.  `want` = this lane holds a live particle whose nearest entry is `row`.  One lane per distinct row of the wave
// exchanges the row's stamp with the frame's epoch; whoever finds an older stamp is the first of the whole frame to need
// the row and has the wave score it - four rows at a time in score_wave's layout (quarter-wave per row, lane s owns floats
// [64 j + 4 s, + 4) for ascending j, xor butterfly), so every score is bit-identical to the dense kernel's.
// The claim is split in two: claim_rows_issue elects the leaders and requests their rows' stamps (nobody waits for them),
// the caller goes on with work that does not need the scores (the prune), and score_claimed_rows looks at the stamps -
// long since back -, exchanges the old ones and scores the rows this wave was first on; the row's norm travels with the
// row.  (As one step - stamp read, exchange, row fetch, norm fetch, each waiting for the one before - a claiming wave was
// four dependent round trips longer than the others, and the kernel ends with its slowest wave.  Exchanging without the
// look first was measured too: every wave then hammers the same few dozen stamps - front 34 -> 44 us.)
struct RowClaim { bool leader; uint32_t old; };
// lds64: 64 ints of the wave's own LDS (nullable).  With it the leaders are elected through a 64-slot hash table - every
// wanting lane writes its lane number into slot hash(row), the slot's last writer leads its row, and a lane that finds
// another ROW in its slot leads as well (so a row can have two leaders in a wave: both look, at most one wins the exchange -
// the claim is idempotent); three instructions instead of one ballot / readlane / compare round per DISTINCT row, of which
// a wave has ~40 in the steady state and 60 after a wide start (~3000 cycles of a 55 000-cycle wave).
MD RowClaim claim_rows_issue(const SparseScore& sp, bool want, int32_t row, int* lds64 = nullptr) {
    const int lane = threadIdx.x & 63;
    // leaders: the first lane of each distinct row among the wanting lanes (a per-lane look at the stamps first was
    // measured: 64 scattered 4-byte loads per wave cost more than the election they save - 17.7k vs 20.5k steps/s)
    RowClaim c;
    c.leader = false;
    if (lds64) {
        const int slot = (int)(((unsigned)row ^ ((unsigned)row >> 6) ^ ((unsigned)row >> 12)) & 63u);
        if (want) lds64[slot] = lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int w = want ? lds64[slot] : lane;
        const int32_t rw = __shfl(row, w);
        c.leader = want && (w == lane || rw != row);
    } else {
        unsigned long long todo = __ballot(want);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            const int32_t r = __shfl(row, l);
            const unsigned long long same = __ballot(want && row == r);
            c.leader |= lane == l;
            todo &= ~same;
        }
    }
    c.old = sp.epoch;
    if (c.leader) c.old = sp.stamps[row];
    return c;
}

// four rows (one per quarter-wave; mine < 0: none) against the code, in score_wave's layout and summation order
template <int NJ>
MD void score_rows4(const SparseScore& sp, int32_t mine) {
    constexpr int D = NJ * 64;
    const int lane = threadIdx.x & 63, s = lane & 15;
    const bool have = mine >= 0;
    const float* rp = sp.emb + (size_t)(have ? mine : 0) * D + s * 4;
    float4 v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const float4*>(rp + j * 64);
    const double nrm = sp.norms[have ? mine : 0];  // travels with the row
    double acc = 0.0, ne2 = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double2* p = reinterpret_cast<const double2*>(sp.code + j * 64 + s * 4);
        const double2 a = p[0], b = p[1];
        ne2 = fma_(a.x, a.x, ne2); ne2 = fma_(a.y, a.y, ne2); ne2 = fma_(b.x, b.x, ne2); ne2 = fma_(b.y, b.y, ne2);
        acc = fma_((double)v[j].x, a.x, acc);
        acc = fma_((double)v[j].y, a.y, acc);
        acc = fma_((double)v[j].z, b.x, acc);
        acc = fma_((double)v[j].w, b.y, acc);
    }
    ne2 = quarter_reduce(ne2);
    acc = quarter_reduce(acc);
    if (have && s == 0) {
        double ne = __builtin_sqrt(ne2);
        ne = ne < COS_EPS ? COS_EPS : ne;
        sp.scores[mine] = acc / (ne * nrm);
    }
}

// the frame's scoring mode (uniform; see SparseScore::dense_thr): read early by the particle waves, it is a round trip
MD bool scores_dense(const SparseScore& sp) { return sp.list != nullptr && sp.dense_thr > 0 && *sp.list_count > sp.dense_thr; }

template <int NJ>
MD int score_claimed_rows(const SparseScore& sp, const RowClaim& c, int32_t row, bool dense = false) {  // returns the rows this wave scored
    const int lane = threadIdx.x & 63, qd = lane >> 4;
    if (dense) {  // every row is being scored by the streaming waves: mark the rows in use, claim nothing
        if (c.leader && c.old != sp.epoch) sp.stamps[row] = sp.epoch;
        return 0;
    }
    // (after the first waves of a frame nearly every needed row carries the epoch already)
    bool claim = false;
    // Prediction list (sp.pred_tag != 0): the tail of the previous frame stamped the rows that frame used with pred_tag and
    // listed them; streaming waves of THIS launch score the list (score_list_wave), so a leader that finds the tag only
    // confirms the row is in use again (a plain store: every confirmer writes the same value) and nobody claims it.
    const bool predicted = sp.pred_tag != 0u && (c.old & ~PRED_SECOND) == sp.pred_tag;  // (listed: used last frame, or its second chance)
    if (c.leader && predicted) sp.stamps[row] = sp.epoch;
#if defined(MIDAS_CLAIM_PLAIN) && MIDAS_CLAIM_PLAIN
    if (c.leader && c.old != sp.epoch && !predicted) { sp.stamps[row] = sp.epoch; claim = true; }  // profiling: no exchange, duplicates allowed
#else
    if (c.leader && c.old != sp.epoch && !predicted) {
        const uint32_t was = atomicExch(&sp.stamps[row], sp.epoch);
        claim = was != sp.epoch && !(sp.pred_tag != 0u && (was & ~PRED_SECOND) == sp.pred_tag);
    }
#endif
    unsigned long long m = __ballot(claim);
    const int nrows = (int)__builtin_popcountll(m);
    while (m) {
        int32_t mine = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int32_t r = __shfl(row, l);
                mine = qd == q ? r : mine;
            }
        }
        score_rows4<NJ>(sp, mine);
    }
    return nrows;
}

MD int score_claimed_rows_nj(const SparseScore& sp, const RowClaim& c, int32_t row, bool dense = false) {
    switch (sp.nj) {
        case 8: return score_claimed_rows<8>(sp, c, row, dense);
        case 4: return score_claimed_rows<4>(sp, c, row, dense);
        case 2: return score_claimed_rows<2>(sp, c, row, dense);
        default: return score_claimed_rows<16>(sp, c, row, dense);
    }
}

// Streaming wave `wave` of `nstream`: the rows of the prediction list, four per wave-instruction, strided over the waves.
// Two rounds are kept in flight (the rows of a spread cloud are cold: a round is one long trip to memory).
template <int NJ>
MD void score_list_wave(const SparseScore& sp, int wave, int nstream) {
    const int lane = threadIdx.x & 63, qd = lane >> 4;
    int count = *sp.list_count;
    if (sp.dense_thr > 0 && count > sp.dense_thr) {  // the whole codebook, rows in order (SparseScore::dense_thr)
        for (int64_t i = (int64_t)wave * 4; i < sp.K; i += (int64_t)nstream * 4) {
            const int64_t k = i + qd;
            score_rows4<NJ>(sp, k < sp.K ? (int32_t)k : -1);
        }
        return;
    }
    count = count < 0 ? 0 : (count > sp.list_cap ? sp.list_cap : count);
    for (int i = wave * 4; i < count; i += nstream * 4) {
        const int k = i + qd;
        const int32_t mine = k < count ? sp.list[k] : -1;
        score_rows4<NJ>(sp, mine);
    }
}

}  // namespace midas
