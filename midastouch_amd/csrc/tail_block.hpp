// tail_block.hpp - the blocked scan of the summation spec and the per-block part of the step tail (score gather,
// softmax numerators, masked prefix sums) in the chunk-per-thread view, without LDS staging.
#pragma once
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

#ifndef MIDAS_TAIL_EXP_ILP
#define MIDAS_TAIL_EXP_ILP 1
#endif
constexpr double TAIL_ISCLOSE_ATOL = 1e-8;
#ifdef MIDAS_DEBUG_CLOCKS  // phase clocks of one k_tail_a2d workgroup (tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS"; tools/ta_clocks.py)
extern __device__ long long g_ta_clk[16];
#define TA_CLK(k) do { if (blk == 12 && threadIdx.x == 0) g_ta_clk[k] = clock64(); if (threadIdx.x == 0 && (k) == 0 && blk == 0) g_ta_clk[8] = wall_clock64(); if (threadIdx.x == 0 && (k) == 7 && blk == 0) g_ta_clk[9] = wall_clock64(); if (threadIdx.x == 0 && (k) == 0 && blk == 24) g_ta_clk[10] = wall_clock64(); if (threadIdx.x == 0 && (k) == 7 && blk == 24) g_ta_clk[11] = wall_clock64(); } while (0)
#else
#define TA_CLK(k) do { } while (0)
#endif  // torch.isclose default atol (particle_filter.py:460-463)

// Block-local part of the spec scan.  v[16] = this lane's chunk (absent values = +0.0).
// l[j] = GP_g + (TP_c + local_j) (l may be v itself); returns the block total W (identical in every thread).
// Needs 16 doubles of LDS (s_gtot) and contains one __syncthreads().
MD double block_scan(const double* v, double* l, double* s_gtot) {
    const int tid = threadIdx.x;
    __builtin_amdgcn_sched_barrier(0);  // phase walls: work hoisted across them only adds live registers
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) { run = run + v[j]; l[j] = run; }
    const double T = run;
    const int c = tid & 15;
    double TP = 0.0;
    // chunk totals of the own 16-lane row, lane by lane (DPP row broadcasts; `__shfl` was an LDS-crossbar trip each), added in order
#define MIDAS_TP_STEP(J) { const double t_ = row_bcast<J>(T); if (J < c) TP = TP + t_; }
    MIDAS_TP_STEP(0) MIDAS_TP_STEP(1) MIDAS_TP_STEP(2) MIDAS_TP_STEP(3) MIDAS_TP_STEP(4) MIDAS_TP_STEP(5) MIDAS_TP_STEP(6) MIDAS_TP_STEP(7)
    MIDAS_TP_STEP(8) MIDAS_TP_STEP(9) MIDAS_TP_STEP(10) MIDAS_TP_STEP(11) MIDAS_TP_STEP(12) MIDAS_TP_STEP(13) MIDAS_TP_STEP(14) MIDAS_TP_STEP(15)
#undef MIDAS_TP_STEP
    if (c == 15) s_gtot[tid >> 4] = TP + T;
    __syncthreads();
    const int g = tid >> 4;
    double GP = 0.0, W = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double t = s_gtot[j];
        if (j < g) GP = GP + t;
        W = W + t;
    }
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) l[j] = GP + (TP + l[j]);
    __builtin_amdgcn_sched_barrier(0);
    return W;
}

// The block total of block_scan alone (same additions in the same order), for sums whose prefixes nobody reads.
MD double block_total(const double* v, double* s_gtot) {
    __builtin_amdgcn_sched_barrier(0);
    double T = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) T = T + v[j];
    const int c = threadIdx.x & 15;
    double TP = 0.0;
    // chunk totals of the own 16-lane row, lane by lane (DPP row broadcasts; `__shfl` was an LDS-crossbar trip each), added in order
#define MIDAS_TP_STEP(J) { const double t_ = row_bcast<J>(T); if (J < c) TP = TP + t_; }
    MIDAS_TP_STEP(0) MIDAS_TP_STEP(1) MIDAS_TP_STEP(2) MIDAS_TP_STEP(3) MIDAS_TP_STEP(4) MIDAS_TP_STEP(5) MIDAS_TP_STEP(6) MIDAS_TP_STEP(7)
    MIDAS_TP_STEP(8) MIDAS_TP_STEP(9) MIDAS_TP_STEP(10) MIDAS_TP_STEP(11) MIDAS_TP_STEP(12) MIDAS_TP_STEP(13) MIDAS_TP_STEP(14) MIDAS_TP_STEP(15)
#undef MIDAS_TP_STEP
    if (c == 15) s_gtot[threadIdx.x >> 4] = TP + T;
    __syncthreads();
    double W = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) W = W + s_gtot[j];
    __builtin_amdgcn_sched_barrier(0);
    return W;
}

// One 4096-slot block by one 256-thread workgroup, thread t owning the chunk of slots [16 t, 16 t + 16) - the view the
// summation spec is written in, so nothing is transposed: x = scores[nn_idx], e = exp(x - 1), e*valid, block sums,
// block-local prefix, chunk-end and group-end tables, block extrema of x; the raw-score variant only when the block's own
// range is within the isclose tolerance (see k_tail_a2).  Outputs are identical to k_tail_a2's; the per-slot tables are
// written in whole 16-value chunks (the layouts are padded to multiples of 16, values past N are never read).
// padded: the per-slot tables hold a multiple of 16 values, so the chunk that straddles N is stored whole too.  Needs N >= 16.
// s_gtot: 32 doubles (two reductions in flight), s_red: 24 doubles of LDS.  kept_out (every thread): valid slots of the block; nan_out: some masked
// weight of the block is NaN.  s_gh (nullable, TAIL_GUIDE_LDS words of LDS; with tb.guide / tb.guide_raw): the block's guide table is written too.
MD void tail_a_direct(int64_t N, int blk, const double* __restrict__ scores, const int32_t* __restrict__ nn_idx,
                      const uint8_t* __restrict__ valid, int32_t softmax, const TailTables& tb, bool padded, double* s_gtot,
                      double* s_red, int& kept_out, bool& nan_out, uint32_t* s_gh = nullptr) {
    const int t = threadIdx.x;
    TA_CLK(0);
    auto guide_clear = [&]() {  // the guide pass's histogram (each thread its own counters)
        uint4* h4 = reinterpret_cast<uint4*>(s_gh) + t * (GUIDE_BINS / SCAN_TPB / 8);
#pragma unroll
        for (int i = 0; i < GUIDE_BINS / SCAN_TPB / 8; ++i) h4[i] = make_uint4(0u, 0u, 0u, 0u);
    };
    if (s_gh) {  // (in the shadow of the index loads; the scans' barriers stand between this and the first count)
        guide_clear();
        if (t == 0) s_gh[GUIDE_BINS / 2 + 5] = 0u;  // "some weight of the block is negative"
    }
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK, base = bbase + (int64_t)t * SCAN_CHUNK;
    // Sixteen contiguous slots from ONE address (the loads share it and travel together; a clamped index per slot would
    // cost an address register pair each).  A chunk that would run past N starts at N - 16 instead and is shifted below.
    const int64_t cs = base + SCAN_CHUNK <= N ? base : N - SCAN_CHUNK;
    const int32_t* pn = nn_idx + cs;
    const uint8_t* pv = valid + cs;
    int32_t nn[SCAN_CHUNK];
    unsigned okbits = 0;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        nn[j] = pn[j];
        okbits |= pv[j] != 0 ? (1u << j) : 0u;
    }
    if (cs != base) {  // at most one straddling chunk per block (chunks wholly past N keep neutral values)
        const int shift = base < N ? (int)(base - cs) : SCAN_CHUNK;
        for (int s = 0; s < shift; ++s) {
#pragma unroll
            for (int j = 0; j + 1 < SCAN_CHUNK; ++j) nn[j] = nn[j + 1];
            okbits >>= 1;
        }
    }
    TA_CLK(1);
    double v[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) v[j] = scores[nn[j]];
    double mx = -INFINITY, mn = INFINITY;
    bool xnan = false;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const bool in = base + j < N;
        xnan |= in && v[j] != v[j];
        mx = in && v[j] > mx ? v[j] : mx;
        mn = in && v[j] < mn ? v[j] : mn;
    }
    int kept = __popc(okbits);
    TA_CLK(2);
    // block extrema (NaN propagates, as torch.max / torch.min do)
    mx = wave_max_dpp(mx);
    mn = wave_min_dpp(mn);
    kept = wave_isum_dpp(kept);
    const bool wxnan = __any(xnan);
    if ((t & 63) == 0) { s_red[t >> 6] = mx; s_red[4 + (t >> 6)] = mn; s_red[8 + (t >> 6)] = wxnan ? 1.0 : 0.0; s_red[12 + (t >> 6)] = (double)kept; }
    // (no barrier of its own: the four waves' extrema are read behind the barrier inside the first block_total - the decision
    // they feed, softmax or raw scores, is only needed after the first variant; a barrier pair here was 1.6 us of the
    // workgroup's 7.9, tools/ta_clocks.py)
    bool close = false;
    auto block_extrema = [&]() {
        mx = s_red[0]; mn = s_red[4];
        double f = s_red[8], kd = s_red[12];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            mx = s_red[w] > mx ? s_red[w] : mx;
            mn = s_red[4 + w] < mn ? s_red[4 + w] : mn;
            f += s_red[8 + w];
            kd += s_red[12 + w];
        }
        kept_out = (int)kd;
        if (f != 0.0) { mx = NAN; mn = NAN; }
        if (t == 0) { tb.bmax[blk] = mx; tb.bmin[blk] = mn; }
        close = __builtin_fabs(mx - mn) <= TAIL_ISCLOSE_ATOL;  // false on NaN
    };
    const bool need_soft = softmax != 0;
    bool nan = false;
    TA_CLK(3);
    // the own chunk of a per-slot table, as eight 16-byte stores from one address (padded layout: see above)
    auto store_chunk = [&](double* __restrict__ out, const double* val) {
        if (base + SCAN_CHUNK <= N || (padded && base < N)) {
            double2* o2 = reinterpret_cast<double2*>(out + base);
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK / 2; ++j) o2[j] = make_double2(val[2 * j], val[2 * j + 1]);
        } else if (base < N) {
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK; ++j)
                if (base + j < N) out[base + j] = val[j];
        }
    };
    // one variant, in place: val[j] of the own chunk -> sums, prefix, tables (val is consumed)
    bool extrema_read = false;
    auto variant = [&](double* val, double* __restrict__ lp_out, double* __restrict__ gend_out,
                       double* __restrict__ ggend_out, double& W_all, double& W_masked, bool& vnan, guide_t* __restrict__ guide_out) {
#pragma unroll
        for (int j = 0; j < SCAN_CHUNK; ++j) val[j] = base + j < N ? val[j] : 0.0;
        W_all = block_total(val, s_gtot);
        if (!extrema_read) { block_extrema(); extrema_read = true; }  // behind block_total's barrier
        bool negw = false;
        // (the scan's group totals go to the second half of s_gtot: no barrier between the two reductions)
#pragma unroll
        for (int j = 0; j < SCAN_CHUNK; ++j) {
            val[j] = val[j] * ((okbits >> j) & 1u ? 1.0 : 0.0);  // okbits is clear on out-of-range slots
            vnan |= val[j] != val[j];
            negw |= val[j] < 0.0;
        }
        W_masked = block_scan(val, val, s_gtot + 16);
        if (base < N) gend_out[(bbase >> 4) + t] = val[SCAN_CHUNK - 1];              // block-local prefix at the chunk end
        if ((t & 15) == 15) ggend_out[(bbase >> 8) + (t >> 4)] = val[SCAN_CHUNK - 1];  // ... at the end of each 256-slot group
        store_chunk(lp_out, val);
        if (s_gh && guide_out) {
            // guide table (GUIDE_BINS / GUIDE_UNIT, midas_internal.hpp): entry k = number of unit ends left of edge k.  Every thread
            // finds, for the units of its chunk, the first edge beyond the unit's end, k_u = min{k : fl(k q) > end_u} (estimate
            // through the reciprocal, settled on the edges themselves) and counts it into a histogram over the edges in LDS (16-bit
            // counters, two to a word); the entries are the histogram's running sum: each thread sums its GUIDE_BINS / 256
            // consecutive counters, the threads' totals are scanned across the workgroup, and the thread's entries leave in whole
            // 16-byte pieces.  (Runs of entries written unit by unit - scattered 2-byte stores, to memory or LDS - cost the kernel
            // 1.2 us; ends that do not rise - raw weights of mixed sign, NaN - give entries that are valid unit numbers and no
            // more: the table is a hint.)
            constexpr int UPC = SCAN_CHUNK / GUIDE_UNIT;  // units per chunk
            constexpr int EPT = GUIDE_BINS / SCAN_TPB;    // entries per thread
            static_assert(EPT % 8 == 0, "a thread's entries are whole 16-byte pieces");
            const double q = W_masked * GUIDE_WIDTH, rq = (double)GUIDE_BINS * __builtin_amdgcn_rcp(W_masked);
            const int64_t left_n = N - bbase;
            const int nch = left_n >= SCAN_BLOCK ? SCAN_TPB : (int)((left_n + SCAN_CHUNK - 1) >> 4);
            uint4* h4 = reinterpret_cast<uint4*>(s_gh) + t * (EPT / 8);
            int ku[UPC];
#pragma unroll
            for (int j = 0; j < UPC; ++j) {
                const double end_u = val[GUIDE_UNIT * (j + 1) - 1], kf = end_u * rq;
                int k = kf > 0.0 ? (kf < (double)GUIDE_BINS ? (int)kf + 1 : GUIDE_BINS) : 0;
                if (k > 0 && (double)(k - 1) * q > end_u) --k;
                if (k < GUIDE_BINS && (double)k * q <= end_u) ++k;
                ku[j] = k;
            }
            if (t < nch) {
#pragma unroll
                for (int j = 0; j < UPC; ++j)  // (an end at or beyond the last edge is left of no edge that has an entry)
                    if (ku[j] < GUIDE_BINS) atomicAdd(&s_gh[ku[j] >> 1], 1u << (16 * (ku[j] & 1)));
            }
            // raw scores of mixed sign (the reference's sampler refuses them: torch.multinomial raises on a negative probability)
            // give prefix values that do not rise: where a search starts would then decide which of several crossings it
            // returns.  Such a block's table says "no guide" (entries beyond every unit: the front falls back to the table lines).
            if (negw) atomicOr(&s_gh[GUIDE_BINS / 2 + 5], 1u);
            __syncthreads();
            const bool no_guide = s_gh[GUIDE_BINS / 2 + 5] != 0u;
            unsigned c[EPT];
#pragma unroll
            for (int i = 0; i < EPT / 8; ++i) {
                const uint4 w = h4[i];
                c[8 * i + 0] = w.x & 0xFFFFu; c[8 * i + 1] = w.x >> 16; c[8 * i + 2] = w.y & 0xFFFFu; c[8 * i + 3] = w.y >> 16;
                c[8 * i + 4] = w.z & 0xFFFFu; c[8 * i + 5] = w.z >> 16; c[8 * i + 6] = w.w & 0xFFFFu; c[8 * i + 7] = w.w >> 16;
            }
            guide_clear();  // (for the other variant's pass, if there is one: its scans' barriers stand in between)
#pragma unroll
            for (int i = 1; i < EPT; ++i) c[i] += c[i - 1];
            const int incl = wave_iscan_dpp((int)c[EPT - 1]);
            int* s_wt = reinterpret_cast<int*>(s_gh + GUIDE_BINS / 2);
            if ((t & 63) == 63) s_wt[t >> 6] = incl;
            __syncthreads();
            unsigned basec = (unsigned)incl - c[EPT - 1];
            for (int w = 0; w < (t >> 6); ++w) basec += (unsigned)s_wt[w];
            if (t == 0) s_gh[GUIDE_BINS / 2 + 5] = 0u;  // (every thread has read it: behind the barrier before this one)
            const unsigned maxu = no_guide ? 0xFFFFu : (unsigned)(nch * UPC - 1);
            guide_t* g = guide_out + (int64_t)blk * GUIDE_STRIDE;
            uint4* dst4 = reinterpret_cast<uint4*>(g + t * EPT);
            auto ent = [&](int i) { const unsigned e = basec + c[i]; return no_guide ? 0xFFFFu : (e < maxu ? e : maxu); };
#pragma unroll
            for (int i = 0; i < EPT / 8; ++i) {
                uint4 w;
                w.x = ent(8 * i + 0) | (ent(8 * i + 1) << 16); w.y = ent(8 * i + 2) | (ent(8 * i + 3) << 16);
                w.z = ent(8 * i + 4) | (ent(8 * i + 5) << 16); w.w = ent(8 * i + 6) | (ent(8 * i + 7) << 16);
                dst4[i] = w;
            }
            if (t == 0) g[GUIDE_BINS] = (guide_t)maxu;
        }
    };
    double Wa = 0.0, Wm = 0.0;
    if (need_soft) {
#pragma unroll
        for (int j = 0; j < SCAN_CHUNK; ++j) {
            v[j] = exp_spec(v[j] - 1.0);
            // MIDAS_TAIL_EXP_ILP exponentials interleaved: each is a dependent chain of ~25 fma, and the workgroup is alone on its
            // CU (one wave per SIMD) - nothing else hides the latency of the chain
            if (j % MIDAS_TAIL_EXP_ILP == MIDAS_TAIL_EXP_ILP - 1) __builtin_amdgcn_sched_barrier(0);
        }
        TA_CLK(4);
        store_chunk(tb.e, v);
        TA_CLK(5);
        variant(v, tb.lp, tb.gend, tb.ggend, Wa, Wm, nan, tb.guide);
        TA_CLK(6);
        if (t == 0) { tb.bsum_e[blk] = Wa; tb.btot[blk] = Wm; }
    }
    const bool need_raw = !softmax || close;  // (softmax off: `close` is not looked at - the raw variant below reads the extrema)
    if (need_raw) {  // rare (every particle of the block shares one score) or the softmax is off
        if (need_soft) {  // the scores were consumed in place: gather them again
            __syncthreads();  // (both halves of s_gtot are free again before the second variant writes them)
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK; ++j) v[j] = scores[nn[j]];
        }
        store_chunk(tb.x_raw, v);
        bool nan_raw = false;
        variant(v, tb.lp_raw, tb.gend_raw, tb.ggend_raw, Wa, Wm, nan_raw, tb.guide_raw);
        if (t == 0) tb.btot_raw[blk] = Wm;
        if (!need_soft) nan = nan_raw;  // with the softmax on, x NaN <=> e NaN: counted once
    } else if (t == 0) {
        tb.btot_raw[blk] = 0.0;
    }
    nan_out = __syncthreads_or(nan ? 1 : 0) != 0;
    TA_CLK(7);
}

}  // namespace midas
