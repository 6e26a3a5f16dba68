// resample_search.hpp - the slot of a draw inside a 4096-slot summation block, shared by the pipelined front
// (particles.hip, lazy_source) and the owner-side routing of the sharded step (resample.hip, k_shard_route_*).
#pragma once
#include "midas_internal.hpp"
#include <type_traits>

namespace midas {

// lp / gend / ggend: block-local prefix of e*valid per slot / at the chunk ends / at the 256-slot group ends (k_tail_a2's
// tables of this GPU's particles, padded to whole 16-value lines).  b = block index in those arrays, N = their particle
// count, one_slot = index of the slot whose cdf is forced to 1 (the globally last particle) or -1; bp = exclusive
// prefix of the block totals before block b in GLOBAL block order, total = sum of all block totals.  The block was
// chosen on the exact predicate (its end value is not left of the draw, the previous block's is), so the answer lies
// inside it.  Returns the first slot i of the block with not left(cdf_i), cdf_i = (bp + lp_i) / total, where
// left(c) = c < tq (multinomial, lower bound) or c <= tq (systematic, upper bound).
// gend_lds (nullable): the caller's copy of the WHOLE chunk-end table in LDS - the chunk is then found by a binary search there
// (same predicate, same chunk) instead of the two line fetches; only the chunk's own line comes from memory.
// LPT / GT: `const double*` (tables in memory) or lds_cdp (the caller's copies in LDS, typed as such: through a generic pointer the
// reads would be flat instructions).
// guide (nullable; with Wb = the block's masked total as the tail stored it): guide tables of the blocks (GUIDE_BINS,
// midas_internal.hpp) - the bin of the draw's block-local target names a unit of GUIDE_UNIT slots, or two neighbouring units
// whose slots are fetched together; the group-end and chunk-end lines are read only where a bin spans more (runs of pruned
// particles) or the totals are not positive.  Which slots the search starts from changes nothing in the result: the fix-up is
// exact.  mid: see lazy_source (particles.hip).  Measured at N = 100k (rocprofv3 means): front 29.4 -> 26.5 us, the tail's guide pass + 0.5 - 0.9 us.
typedef const __attribute__((address_space(3))) double* lds_cdp;
struct NoMid { MD void operator()() const {} };
template <typename LPT, typename GT, typename MID = NoMid>
MD int64_t search_in_block_t(LPT lp, const double* __restrict__ gend, const double* __restrict__ ggend,
                             int b, int64_t N, int64_t one_slot, double bp, double total, double tq, bool upper, GT gend_lds,
                             const guide_t* __restrict__ guide = nullptr, double Wb = 0.0, MID mid = MID()) {
    // (total < 0: raw weights - the softmax is skipped when every particle has the same score, particle_filter.py:459-468 -
    // of a negative cosine; p = w / sum(w) is positive again (:238) and dividing by the negative total turns the comparison
    // round.  Without the turn the division-free probes point the wrong way and the exact walk below crosses the whole block
    // one dependent load at a time: 100 - 400 us frames of a lost, collapsed cloud at N = 1000, found with the c1 trajectory.)
    // The turn costs nothing per probe: with sg = -1 for a negative total, sg (bp + v) = fma(sg, v, sg bp) is the same sum with
    // the sign flipped (rounding is symmetric; for sg = 1 it IS bp + v), compared against sg tq total on the usual side.
    const double sg = total < 0.0 ? -1.0 : 1.0;
    const double tt = sg * (tq * total), sbp = sg * bp;
    auto left = [&](double v_) { const double c = fma_(sg, v_, sbp); return upper ? (c <= tt) : (c < tt); };  // v_ = block-local prefix
    auto left_exact = [&](double c) { return upper ? (c <= tq) : (c < tq); };
    const int64_t b_lo = (int64_t)b << 12, b_hi = b_lo + SCAN_BLOCK < N ? b_lo + SCAN_BLOCK : N;
    constexpr bool LP_LDS = __is_same(LPT, lds_cdp);
    auto cdfv = [&](int64_t i, double lpv) { return (i == one_slot) ? 1.0 : (bp + lpv) / total; };
    // Last level, U slots from s0 (and, `two`, the U behind them, fetched in the same round trip): position = how many are left of
    // the draw on the division-free comparison, then the exact fix-up - the predicate on cdf_i is monotone in i; the two
    // neighbours of the boundary are normally in registers, otherwise walk (inside the block: its end is exact).
    auto finish = [&](auto uc, int64_t s0, bool two) -> int64_t {
        constexpr int U = decltype(uc)::value;
        double v[U], v2[U];
        auto fetch_to = [&](LPT p, double* d) {
            if constexpr (LP_LDS) {
#pragma unroll
                for (int j = 0; j < U; ++j) d[j] = p[j];
            } else {  // aligned 16-byte pieces
                const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
                for (int j = 0; j < U / 2; ++j) { const double2 w = p2[j]; d[2 * j] = w.x; d[2 * j + 1] = w.y; }
            }
        };
        double v_prev = lp[s0 > b_lo ? s0 - 1 : b_lo];  // the slot before (same block)
        fetch_to(lp + s0, v);
        if (two) {
            fetch_to(lp + s0 + U, v2);
#pragma unroll
            for (int j = 0; j < U; ++j) asm volatile("" : "+v"(v2[j]));
        }
        // pinned to the round trip: left to itself the compiler sinks this load into the fix-up branch below, where it is a
        // dependent round trip of its own
        asm volatile("" : "+v"(v_prev));
        int lim = (int)(b_hi - s0);  // slots of the block from s0 on (values past the end of the data are not counted)
        int pos = 0;
#pragma unroll
        for (int j = 0; j < U; ++j) pos += (j < lim && left(v[j])) ? 1 : 0;
        if (two && pos == U) {  // every slot of the first unit is left of the draw: the boundary is in the second
            v_prev = v[U - 1];
            s0 += U;
            lim -= U;
            pos = 0;
#pragma unroll
            for (int j = 0; j < U; ++j) { v[j] = v2[j]; pos += (j < lim && left(v[j])) ? 1 : 0; }
        }
        int64_t l2 = s0 + pos;
        double vm = v_prev, vp = 0.0;
#pragma unroll
        for (int j = 0; j < U; ++j) { vm = (j == pos - 1) ? v[j] : vm; vp = (j == pos) ? v[j] : vp; }
        bool walk = l2 >= b_hi;
        if (!walk) {
            if (l2 > b_lo) walk |= !left_exact(cdfv(l2 - 1, vm));
            walk |= pos >= U || left_exact(cdfv(l2, vp));
        }
        if (walk) {
            if (l2 >= b_hi) l2 = b_hi - 1;
            while (l2 > b_lo) {
                if (left_exact(cdfv(l2 - 1, lp[l2 - 1]))) break;
                --l2;
            }
            while (l2 < b_hi - 1) {
                if (!left_exact(cdfv(l2, lp[l2]))) break;
                ++l2;
            }
        }
        return l2;
    };
    if constexpr (!LP_LDS) {
        if (guide && Wb > 0.0 && total > 0.0 && Wb < INFINITY) {
            const double x = tq * total - bp, q = Wb * GUIDE_WIDTH;
            const double kf = x * ((double)GUIDE_BINS * __builtin_amdgcn_rcp(Wb));
            int k = kf > 0.0 ? (kf < (double)(GUIDE_BINS - 1) ? (int)kf : GUIDE_BINS - 1) : 0;
            // (the reciprocal is approximate: the bin is settled on the edges the tail used, fl(k q))
            if (k > 0 && x < (double)k * q) --k;
            if (k < GUIDE_BINS - 1 && x >= (double)(k + 1) * q) ++k;
            const guide_t* g = guide + (int64_t)b * GUIDE_STRIDE + k;
            const int uA = g[0], uB = g[1];
            // the caller's independent arithmetic, under the entries' round trip
            __builtin_amdgcn_sched_barrier(0);
            mid();
            __builtin_amdgcn_sched_barrier(0);
            if (uB >= uA && uB - uA <= 1 && (int64_t)uB * GUIDE_UNIT < b_hi - b_lo)
                return finish(std::integral_constant<int, GUIDE_UNIT>(), b_lo + (int64_t)uA * GUIDE_UNIT, uB != uA);
        }
    }
    // Without a guide (or where a bin spans more than two units): three levels, one 128-byte line each (16 prefix values fetched
    // together with eight aligned 16-byte loads) - 256-slot group ends of the block, chunk ends of the group, slots of the chunk.
    double v[SCAN_CHUNK];
    auto fetch16 = [&](const double* __restrict__ p) {
        const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double2 w = p2[j]; v[2 * j] = w.x; v[2 * j + 1] = w.y; }
    };
    const int n_chunks = (int)((b_hi - b_lo + SCAN_CHUNK - 1) >> 4), n_groups = (n_chunks + 15) >> 4;
    int64_t cidx;  // the chunk: first one of the block whose end value is not left of the draw (the block's last at the latest)
    if (gend_lds) {
        int lo = (int)(b_lo >> 4), hi = lo + n_chunks - 1;  // (the last chunk needs no probe)
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (left(gend_lds[mid])) lo = mid + 1; else hi = mid;
        }
        cidx = lo;
    } else {
        fetch16(ggend + (int64_t)b * 16);
        int g = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) g += (j < n_groups && left(v[j])) ? 1 : 0;
        g = g < n_groups ? g : n_groups - 1;
        const int64_t c0 = (b_lo >> 4) + 16 * g;
        fetch16(gend + c0);
        const int n_in_group = n_chunks - 16 * g < 16 ? n_chunks - 16 * g : 16;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) c += (j < n_in_group && left(v[j])) ? 1 : 0;
        c = c < n_in_group ? c : n_in_group - 1;
        cidx = c0 + c;
    }
    return finish(std::integral_constant<int, SCAN_CHUNK>(), cidx << 4, false);
}

MD int64_t search_in_block(const double* __restrict__ lp, const double* __restrict__ gend, const double* __restrict__ ggend,
                           int b, int64_t N, int64_t one_slot, double bp, double total, double tq, bool upper) {
    return search_in_block_t<const double*, const double*>(lp, gend, ggend, b, N, one_slot, bp, total, tq, upper, (const double*)nullptr);
}

}  // namespace midas
