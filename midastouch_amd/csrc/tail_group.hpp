// tail_group.hpp - the step tail (score gather, softmax numerators, masked prefix sums of the summation spec, tables, guide
// tables: what tail_block.hpp's tail_a_direct produces for a 4096-slot block from ONE 256-thread workgroup) spread over SIXTEEN
// waves a block, one per 256-slot group, each on a SIMD of its own: 391 waves instead of 25 workgroups at N = 100k, four
// gathers and four exponentials a lane instead of sixteen.  Same additions in the same order (resample.hip's header):
//   chunk   (16 slots = a quad of lanes, four slots a lane): the running sum walks the quad lane by lane (DPP row_shr:1);
//   group   (16 chunks = the wave): the chunk totals are read lane by lane (v_readlane) and added in order;
//   block   (16 groups = 16 waves, in up to four workgroups): every wave publishes its group's totals as a record of
//           tear-proof 8-byte pairs {v, v ^ key(launch tag)} by agent-scope stores and reads the records of its block with
//           agent-scope loads until every pair carries this launch's key (MI355X_MICROARCH.md "handoff-1to1": ~1 us); the
//           sixteen group totals are then added in order by every wave for itself.
// The waves of a block are dispatched together (consecutive workgroups) and the launcher only takes this form while the whole
// grid is resident, so nobody waits for a wave that cannot start; the wait is bounded all the same (status bit 16).
#pragma once
#include "midas_internal.hpp"
#include "midas_math.hpp"
#include "tail_block.hpp"

namespace midas {

#ifdef MIDAS_DEBUG_CLOCKS  // wall-clock stamps (100 MHz) of the grouped tail, kept in registers and stored when the wave ends: tools/tg_clocks.py
extern __device__ long long g_tg_clk[64];
extern __device__ long long g_tg_w[8192];  // per launch parity and wave / workgroup: start, end
#define TG_CLK(k) do { tg_clk_[k] = wall_clock64(); } while (0)
#define TG_CLK_ARG , long long* tg_clk_
#define TG_CLK_PASS , tg_clk_
#define TG_SPAN(slot, t_) do { if ((threadIdx.x & 63) == 0 && (slot) < 4096) g_tg_w[(a.tag & 1) * 4096 + (slot)] = (t_); } while (0)
#else
#define TG_CLK(k) do { } while (0)
#define TG_CLK_ARG
#define TG_CLK_PASS
#define TG_SPAN(slot, t_) do { } while (0)
#endif

constexpr int TG_GROUP = 256;        // slots of a wave: one group of the summation spec
constexpr int TG_VALUES = 5;         // record: masked total, unmasked total, max x, min x, {kept | flags}
constexpr int TG_REC_WORDS = 16;     // one 128-byte line per (block, round, group)
constexpr int TG_ROUNDS = 2;         // round 0: the first variant (softmax, or raw scores when the softmax is off); round 1: raw scores
                                     // of a block whose own range is within the isclose tolerance
constexpr size_t TG_BLOCK_WORDS = (size_t)TG_ROUNDS * 16 * TG_REC_WORDS;
constexpr long long TG_WAIT_TICKS = 20000000ll;  // 0.2 s of the 100 MHz wall clock

struct TailGroupArgs {
    int64_t N;
    const double* scores;
    const int32_t* nn_idx;
    const uint8_t* valid;
    int32_t softmax;
    TailTables tb;
    bool padded;
    int32_t* status;
    double* flags_out;                // nullable (sharded exchange record)
    unsigned long long* rec;          // [nb x TG_BLOCK_WORDS], zero at allocation
    uint32_t tag;                     // this launch's tag (never 0, never repeated on this buffer)
};

MD uint64_t tg_key(uint32_t tag, int round, int i) {
    return ((((uint64_t)tag << 8) | (uint64_t)((round << 4) | i)) + 1ull) * 0x9E3779B97F4A7C15ull;  // odd multiplier: never 0
}
MD double readlane_d(double v, int l) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return double_of((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l));
}
MD double shfl_d(double v, int l) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return double_of((uint32_t)__shfl((int)(uint32_t)b, l), (uint32_t)__shfl((int)(uint32_t)(b >> 32), l));
}

// agent-scope (sc1: past the non-coherent caches) 16-byte store / five 16-byte loads of a record
typedef unsigned int tg_u32x4 __attribute__((ext_vector_type(4)));
MD void tg_store16(unsigned long long* p, uint64_t lo, uint64_t hi) {
    tg_u32x4 v;
    v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
MD void tg_load80(const unsigned long long* p, uint64_t* w) {
    tg_u32x4 v0, v1, v2, v3, v4;
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %5, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %5, off offset:64 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4) : "v"(p) : "memory");
    const tg_u32x4 v[5] = {v0, v1, v2, v3, v4};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        w[2 * i] = (uint64_t)v[i].x | ((uint64_t)v[i].y << 32);
        w[2 * i + 1] = (uint64_t)v[i].z | ((uint64_t)v[i].w << 32);
    }
}

struct TGBlock {
    double W, Wa, mx, mn;  // block totals (masked, unmasked), block extrema of x (NaN when some x is)
    int kept;
    bool vnan, late;
};

// One variant of one group by one wave.  v[4]: the lane's values (0.0 on slots past N), ok4: valid bits (clear past N).
// g* : the group's extrema / kept count / "some x is NaN" (published with round 0, ignored in round 1).
MD TGBlock tg_variant(const TailGroupArgs& a, int G, int round, const double* v, unsigned ok4, double gmx, double gmn, int gkept,
                      bool gxnan, double* __restrict__ lp_out, double* __restrict__ gend_out, double* __restrict__ ggend_out,
                      guide_t* __restrict__ guide_out, double* s_E TG_CLK_ARG) {
    const int lane = threadIdx.x & 63, c = lane >> 2, k = lane & 3;
    const int blk = G >> 4, g = G & 15;
    const int64_t N = a.N, bbase = (int64_t)blk * SCAN_BLOCK, s0 = (int64_t)G * TG_GROUP + 4 * lane;
    const int64_t left_n = N - bbase;
    const int nch = left_n >= SCAN_BLOCK ? SCAN_TPB : (int)((left_n + SCAN_CHUNK - 1) >> 4);  // chunks of the block that hold a slot
    const int ngb = (nch + 15) >> 4;                                                            // groups of the block that do
    double m[4];
    bool vnan = false, negw = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = v[j] * ((ok4 >> j) & 1u ? 1.0 : 0.0);
        vnan |= m[j] != m[j];
        negw |= m[j] < 0.0;
    }
    vnan = __any(vnan ? 1 : 0) != 0;
    negw = __any(negw ? 1 : 0) != 0;
    // chunk level: the running sum of the chunk's sixteen values, lane by lane through the quad
    double cm = 0.0, ca = 0.0, L[4] = {0.0, 0.0, 0.0, 0.0}, endm = 0.0, enda = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double r0 = cm + m[0], r1 = r0 + m[1], r2 = r1 + m[2], r3 = r2 + m[3];
        const double q3 = (((ca + v[0]) + v[1]) + v[2]) + v[3];
        if (k == s) { L[0] = r0; L[1] = r1; L[2] = r2; L[3] = r3; endm = r3; enda = q3; }
        if (s < 3) {
            const double um = dpp_move<0x111>(r3), ua = dpp_move<0x111>(q3);  // row_shr:1: the lane below
            if (k == s + 1) { cm = um; ca = ua; }
        }
    }
    // group level: chunk totals (held by the last lane of each quad) in order
    double accm = 0.0, acca = 0.0, TP = 0.0;
#pragma unroll
    for (int J = 0; J < 16; ++J) {
        if (c == J) TP = accm;
        accm = accm + readlane_d(endm, 4 * J + 3);
        acca = acca + readlane_d(enda, 4 * J + 3);
    }
    const double tm = accm, ta = acca;  // (uniform)
    TG_CLK(4 + 6 * round);
    // publish the group's record: five 16-byte pairs {v, v ^ key}, one store each (lanes 0 .. 4)
    unsigned long long* rec_b = a.rec + ((size_t)blk * TG_ROUNDS + round) * 16 * TG_REC_WORDS;
    {
        const uint64_t meta = (uint64_t)(unsigned)gkept | ((uint64_t)(gxnan ? 1 : 0) << 16) | ((uint64_t)(vnan ? 1 : 0) << 17) | ((uint64_t)(negw ? 1 : 0) << 18);
        const uint64_t v0 = lane == 0 ? (uint64_t)__double_as_longlong(tm) : lane == 1 ? (uint64_t)__double_as_longlong(ta)
                          : lane == 2 ? (uint64_t)__double_as_longlong(gmx) : lane == 3 ? (uint64_t)__double_as_longlong(gmn) : meta;
        if (lane < TG_VALUES) tg_store16(rec_b + g * TG_REC_WORDS + 2 * lane, v0, v0 ^ tg_key(a.tag, round, lane));
    }
    // read the block's records (lane j: group j) until all of them are this launch's
    uint64_t w[2 * TG_VALUES];
    bool late = false;
    {
        const unsigned long long* rp = rec_b + (lane < ngb ? lane : 0) * TG_REC_WORDS;
        const long long t0 = wall_clock64();
        for (;;) {
            tg_load80(rp, w);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < TG_VALUES; ++i) ok &= (w[2 * i] ^ w[2 * i + 1]) == tg_key(a.tag, round, i);
            if (__all((ok || lane >= ngb) ? 1 : 0)) break;
            if (wall_clock64() - t0 > TG_WAIT_TICKS) { late = true; break; }
        }
    }
    TG_CLK(5 + 6 * round);
    // block level: the sixteen group totals in order (groups past N add +0.0, as their slots would); extrema, kept count and flags
    // are order-free: reduced inside row 0 (lanes 0 .. 15 hold the records)
    const bool have = lane < ngb;
    const double vtm = have ? __longlong_as_double((long long)w[0]) : 0.0, vta = have ? __longlong_as_double((long long)w[2]) : 0.0;
    double mx = have ? __longlong_as_double((long long)w[4]) : -INFINITY, mn = have ? __longlong_as_double((long long)w[6]) : INFINITY;
    uint32_t kp = have ? ((uint32_t)w[8] & 0xFFFFu) : 0u, fl = have ? ((uint32_t)w[8] >> 16) : 0u;
#define MIDAS_ROW_REDUCE(v, OP)                                                     \
    { auto t_ = dpp_move<0xB1>(v); v = OP(v, t_); } { auto t_ = dpp_move<0x4E>(v); v = OP(v, t_); } \
    { auto t_ = dpp_move<0x141>(v); v = OP(v, t_); } { auto t_ = dpp_move<0x140>(v); v = OP(v, t_); }
    MIDAS_ROW_REDUCE(mx, dpp_max_)
    MIDAS_ROW_REDUCE(mn, dpp_min_)
#undef MIDAS_ROW_REDUCE
    kp += dpp_move<0xB1>(kp, kp); kp += dpp_move<0x4E>(kp, kp); kp += dpp_move<0x141>(kp, kp); kp += dpp_move<0x140>(kp, kp);
    fl |= dpp_move<0xB1>(fl, fl); fl |= dpp_move<0x4E>(fl, fl); fl |= dpp_move<0x141>(fl, fl); fl |= dpp_move<0x140>(fl, fl);
    mx = readlane_d(mx, 0);
    mn = readlane_d(mn, 0);
    const int kept = __builtin_amdgcn_readlane((int)kp, 0);
    const unsigned flags = (unsigned)__builtin_amdgcn_readlane((int)fl, 0);
    double GP = 0.0, GPn = 0.0, W = 0.0, Wa = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        GP = j == g ? W : GP;
        W = W + readlane_d(vtm, j);
        GPn = j == g ? W : GPn;
    }
    if (g == 0) {  // (the block's unmasked total is written by the first group's wave alone)
#pragma unroll
        for (int j = 0; j < 16; ++j) Wa = Wa + readlane_d(vta, j);
    }
    if (flags & 1u) { mx = NAN; mn = NAN; }  // torch.max / torch.min propagate NaN
    const bool b_vnan = (flags & 2u) != 0, b_negw = (flags & 4u) != 0;
    // per-slot prefix, chunk-end and group-end tables
    double out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = GP + (TP + L[j]);
    const int64_t npad = (N + SCAN_CHUNK - 1) & ~(int64_t)(SCAN_CHUNK - 1);
    if (s0 + 4 <= N || (a.padded && s0 < npad)) {
        double2* o2 = reinterpret_cast<double2*>(lp_out + s0);
        o2[0] = make_double2(out[0], out[1]);
        o2[1] = make_double2(out[2], out[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s0 + j < N) lp_out[s0 + j] = out[j];
    }
    if (k == 3 && (int64_t)G * TG_GROUP + 16 * c < N) gend_out[(int64_t)G * 16 + c] = out[3];
    if (lane == 63) ggend_out[(int64_t)blk * 16 + g] = out[3];
    if (g == ngb - 1 && lane > g && lane < 16) ggend_out[(int64_t)blk * 16 + lane] = readlane_d(out[3], 63);  // groups without a slot: the block's total
    TG_CLK(6 + 6 * round);
    if (guide_out) {
        // Guide table (midas_internal.hpp GUIDE_BINS): entry k = min(number of the block's unit ends < edge k, units - 1).  The ends rise
        // (no negative weight), so group g's ends lie in [GP_g, GP_g+1] and the entries of the edges in (GP_g, GP_g+1] are decided by
        // this group's ends alone: units of the groups before + own ends below the edge.  Neighbouring waves computed GP_g+1 by the
        // same additions: the ranges tile the table.  The wave of the first group also takes the edges at or below zero, the last one
        // the edges beyond its end.  A block with a negative weight (raw scores of mixed sign) or without a positive finite total has
        // no guide (every entry 0xFFFF: the search falls back to the table lines).
        constexpr int UPG = TG_GROUP / GUIDE_UNIT;  // units of a group
        constexpr int UPC = SCAN_CHUNK / GUIDE_UNIT;
        guide_t* gt = guide_out + (int64_t)blk * GUIDE_STRIDE;
        const bool plain = !b_negw && !b_vnan && W > 0.0 && W < (double)INFINITY;
        const unsigned maxu = plain ? (unsigned)(nch * UPC - 1) : 0xFFFFu;
        if (!plain) {
            const int share = (GUIDE_BINS + ngb - 1) / ngb, k0 = g * share, k1 = g == ngb - 1 ? GUIDE_BINS : (k0 + share < GUIDE_BINS ? k0 + share : GUIDE_BINS);
            for (int kk = k0 + lane; kk < k1; kk += 64) gt[kk] = (guide_t)0xFFFFu;
        } else {
            const double q = W * GUIDE_WIDTH, rq = (double)GUIDE_BINS * __builtin_amdgcn_rcp(W);
            auto first_beyond = [&](double x) {  // min{k : fl(k q) > x}, GUIDE_BINS when there is none: the estimate through the reciprocal is within one
                const double kf = x * rq;
                int kk = kf > 0.0 ? (kf < (double)GUIDE_BINS ? (int)kf + 1 : GUIDE_BINS) : 0;
                if (kk > 0 && (double)(kk - 1) * q > x) --kk;
                if (kk < GUIDE_BINS && (double)kk * q <= x) ++kk;
                return __builtin_amdgcn_readfirstlane(kk);
            };
            const int k_lo = g == 0 ? 0 : first_beyond(GP), k_hi = g == ngb - 1 ? GUIDE_BINS : first_beyond(GPn);
            int nu = UPC * (nch - 16 * g);
            nu = nu > UPG ? UPG : nu;
            // unit u's end is the last value of lane (u + 1) * GUIDE_UNIT / 4 - 1
            constexpr int LPU = GUIDE_UNIT / 4;  // lanes per unit
            if ((lane & (LPU - 1)) == LPU - 1) s_E[lane / LPU] = out[3];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned below = (unsigned)(UPG * g);
            // lower bound over the UPG ends: the first three levels compare against ends every lane can hold (read off their lanes),
            // the rest are dependent LDS reads
            constexpr int Q = UPG / 8;  // ends Q - 1, 2 Q - 1, .. , 8 Q - 1
            double eq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) eq[i] = readlane_d(out[3], (i + 1) * Q * LPU - 1);
            constexpr int BPL = 4;  // bins a lane and pass (a group's share is 128 bins when the weights are even)
            for (int kb = k_lo; kb < k_hi; kb += 64 * BPL) {
                int lo[BPL];
                double edge[BPL];
#pragma unroll
                for (int i = 0; i < BPL; ++i) {
                    edge[i] = (double)(kb + lane + 64 * i) * q;
                    // (branch-free on purpose: bitwise conditions, every read issued - a short-circuit form serialises the LDS reads)
                    int l = ((8 * Q <= nu) & (eq[7] < edge[i])) ? 8 * Q : 0;  // (only with every unit below the edge)
                    const bool h4 = (l == 0) & (4 * Q <= nu) & (eq[3] < edge[i]);
                    l = h4 ? 4 * Q : l;
                    const double e2 = h4 ? eq[5] : eq[1];
                    const bool h2 = (l < 8 * Q) & (l + 2 * Q <= nu) & (e2 < edge[i]);
                    l = h2 ? l + 2 * Q : l;
                    const double e1 = h4 ? (h2 ? eq[6] : eq[4]) : (h2 ? eq[2] : eq[0]);
                    l = ((l < 8 * Q) & (l + Q <= nu) & (e1 < edge[i])) ? l + Q : l;
                    lo[i] = l;
                }
#pragma unroll
                for (int step = Q / 2; step > 0; step >>= 1) {
                    double e[BPL];
#pragma unroll
                    for (int i = 0; i < BPL; ++i) e[i] = s_E[(lo[i] + step - 1) & (2 * UPG - 1)];
#pragma unroll
                    for (int i = 0; i < BPL; ++i) lo[i] += ((lo[i] + step <= nu) & (e[i] < edge[i])) ? step : 0;
                }
#pragma unroll
                for (int i = 0; i < BPL; ++i) {
                    const int kk = kb + lane + 64 * i;
                    const unsigned e = below + (unsigned)lo[i];
                    if (kk < k_hi) gt[kk] = (guide_t)(e < maxu ? e : maxu);
                }
            }
        }
        if (g == ngb - 1 && lane == 0) gt[GUIDE_BINS] = (guide_t)maxu;
    }
    TG_CLK(7 + 6 * round);
    TGBlock r;
    r.W = W; r.Wa = Wa; r.mx = mx; r.mn = mn; r.kept = kept; r.vnan = b_vnan; r.late = late;
    return r;
}

// One 256-slot group by one wave.  s_E: 2 * (256 / GUIDE_UNIT) doubles of LDS of the wave's own.
MD void tail_group_wave(const TailGroupArgs& a, int G, double* s_E) {
    const int lane = threadIdx.x & 63;
    const int blk = G >> 4, g = G & 15;
    const int64_t N = a.N, s0 = (int64_t)G * TG_GROUP + 4 * lane;
    const TailTables& tb = a.tb;
#ifdef MIDAS_DEBUG_CLOCKS
    long long tg_clk_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TG_CLK(0);
    int32_t nn[4] = {0, 0, 0, 0};
    unsigned ok4 = 0, in4 = 0;
    if (s0 + 4 <= N) {
        const int4 n4 = *reinterpret_cast<const int4*>(a.nn_idx + s0);
        const uint32_t vb = *reinterpret_cast<const uint32_t*>(a.valid + s0);
        nn[0] = n4.x; nn[1] = n4.y; nn[2] = n4.z; nn[3] = n4.w;
        ok4 = ((vb & 0xFFu) ? 1u : 0u) | ((vb & 0xFF00u) ? 2u : 0u) | ((vb & 0xFF0000u) ? 4u : 0u) | ((vb & 0xFF000000u) ? 8u : 0u);
        in4 = 0xFu;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s0 + j < N) {
                nn[j] = a.nn_idx[s0 + j];
                ok4 |= a.valid[s0 + j] != 0 ? (1u << j) : 0u;
                in4 |= 1u << j;
            }
    }
    double x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = a.scores[nn[j]];
    TG_CLK(1);
    double mx = -INFINITY, mn = INFINITY;
    bool xnan = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = (in4 >> j) & 1u;
        x[j] = in ? x[j] : 0.0;
        xnan |= in && x[j] != x[j];
        mx = in && x[j] > mx ? x[j] : mx;
        mn = in && x[j] < mn ? x[j] : mn;
    }
    mx = wave_max_dpp(mx);
    mn = wave_min_dpp(mn);
    const int gkept = wave_isum_dpp(__popc(ok4));
    const bool gxnan = __any(xnan ? 1 : 0) != 0;
    TG_CLK(2);
    const int64_t npad = (N + SCAN_CHUNK - 1) & ~(int64_t)(SCAN_CHUNK - 1);
    auto store4 = [&](double* __restrict__ o, const double* val) {
        if (s0 + 4 <= N || (a.padded && s0 < npad)) {
            double2* o2 = reinterpret_cast<double2*>(o + s0);
            o2[0] = make_double2(val[0], val[1]);
            o2[1] = make_double2(val[2], val[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s0 + j < N) o[s0 + j] = val[j];
        }
    };
    const bool need_soft = a.softmax != 0;
    TGBlock r0;
    bool close = false, late = false, nan = false;
    if (need_soft) {
        double e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = exp_spec(x[j] - 1.0);
        TG_CLK(3);
        store4(tb.e, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (in4 >> j) & 1u ? e[j] : 0.0;
        r0 = tg_variant(a, G, 0, e, ok4, mx, mn, gkept, gxnan, tb.lp, tb.gend, tb.ggend, tb.guide, s_E TG_CLK_PASS);
        close = __builtin_fabs(r0.mx - r0.mn) <= TAIL_ISCLOSE_ATOL;  // false on NaN
        late |= r0.late;
        nan = r0.vnan;
        if (g == 0 && lane == 0) { tb.bsum_e[blk] = r0.Wa; tb.btot[blk] = r0.W; }
    }
    const bool need_raw = !need_soft || close;  // rare with the softmax on: every particle of the block shares one score
    if (need_raw) {
        store4(tb.x_raw, x);
        const TGBlock r1 = tg_variant(a, G, need_soft ? 1 : 0, x, ok4, mx, mn, gkept, gxnan, tb.lp_raw, tb.gend_raw, tb.ggend_raw, tb.guide_raw, s_E TG_CLK_PASS);
        late |= r1.late;
        if (!need_soft) { r0 = r1; nan = r1.vnan; }  // with the softmax on, x NaN <=> e NaN: counted once
        if (g == 0 && lane == 0) tb.btot_raw[blk] = r1.W;
    } else if (g == 0 && lane == 0) {
        tb.btot_raw[blk] = 0.0;
    }
    if (g == 0 && lane == 0) {
        tb.bmax[blk] = r0.mx;
        tb.bmin[blk] = r0.mn;
        if (nan) atomicOr(&a.status[0], 2);
        if (r0.kept) atomicAdd(&a.status[1], r0.kept);
        if (a.flags_out) {  // sharded exchange record: NaN marker (any non-zero) and kept count (exact: integers far below 2^53)
            if (nan) atomicAdd(&a.flags_out[0], 1.0);
            if (r0.kept) atomicAdd(&a.flags_out[1], (double)r0.kept);
        }
    }
    if (late && lane == 0) atomicOr(&a.status[0], 16);
#ifdef MIDAS_DEBUG_CLOCKS
    TG_SPAN(2 * G, tg_clk_[0]);
    TG_SPAN(2 * G + 1, wall_clock64());
    if (lane == 0 && (G == 0 || G == 200))
        for (int i = 0; i < 8; ++i) g_tg_clk[(G ? 24 : 8) + i] = tg_clk_[i];
#endif
}

}  // namespace midas
