// loop.hip - the reference's whole loop body on a variable-size particle set (midas_loop_step).
//
// filter/filter.py:150-190 changes the particle count every frame between the measurement update and the resample
// (particle_filter.annealing, modules/particle_filter.py:405-447).  Here the count lives in device memory
// (ctl_i[LOOP_I_N]); every kernel is launched over the capacity of the arrays and reads the live count itself, so a frame
// needs no host round trip:
//
//   FRONT     k_frame_front (particles.hip, live count from the control block; sets up to 16 384: k_front_small) ->
//             k_loop_xe (scores gathered, softmax numerators, per-block sums in the spec order) -> k_loop_weights (S,
//             isclose guard, masked weights, drift re-projection, rmse; loop_weights.hpp - in frames without DBSCAN this
//             work sits at the head of the cluster-moment launch instead, k_loop_weights_moments in cluster.hip)
//   DBSCAN    dbscan.hip
//   ANNEAL    k_loop_cluster_* (cluster.hip) -> sets up to 16 384: k_loop_anneal_small (decision + selection by one
//             workgroup, the centres' rotations by a second); larger: k_loop_decide (labels present, var = mean(stds),
//             the annealing rule in float32 as torch evaluates it) -> k_loop_select x 6 (radix select of the k-th smallest
//             / largest weight, 11-bit digits of the order-preserving 64-bit key) -> k_loop_compact_count / k_loop_compact
//             (the annealed set as an index list: survivors in their order, or everybody + the k best) ->
//             k_loop_sort_chunks / k_loop_sort_rank (the duplicates in topk's output order: weight descending, index ascending)
//   RESAMPLE  k_loop_scan (blocked prefix sums of (e * valid)[src]) -> k_loop_resample (n_set draws, exact inverse
//             search on cdf_i = (BP_b + lp_i) / total, gathers through src)
//
// Spec (oracle/oracle.py OracleLoop): identical to midas_filter_step where the two overlap; ties of the top-k selection go
// to the smaller index (torch's CUDA rule), or - midas_loop_args.topk_ties = MIDAS_TOPK_TIES_ATEN_CPU, the mode that replays a
// seeded run of the reference - to the members ATen's CPU kernel takes (topk_aten.hip).
#include <cstdlib>

#include "midas_internal.hpp"
#include "midas_math.hpp"
#include "cluster_rot.hpp"
#include "tail_block.hpp"
#include "loop_weights.hpp"

namespace midas {

#define LAUNCH_CHECK(ctx) MIDAS_HIP_CHECK(ctx, hipGetLastError())

constexpr int SEL_PASSES = 6;
constexpr int SEL_BINS = 2048;
constexpr int SORT_CHUNK = 2048;  // (key, index) pairs sorted per workgroup in LDS
__device__ const int kSelShift[SEL_PASSES] = {53, 42, 31, 20, 9, 0};
__device__ const int kSelWidth[SEL_PASSES] = {11, 11, 11, 11, 11, 9};

MD double lw_sum(double v) { return wave_sum_ordered(v); }  // (the xor butterfly 32 .. 1 of the spec, by register moves: midas_math.hpp)
MD int lw_isum(int v) { return wave_isum_dpp(v); }  // (integer: any order; DPP moves instead of six LDS-crossbar trips)

// Order-preserving 64-bit key of a weight: a < b <=> key(a) < key(b); -0.0 == +0.0; NaN above everything (torch.topk
// treats NaN as the largest value).
MD uint64_t weight_key(double w) {
    if (w == 0.0) w = 0.0;  // -0.0 -> +0.0
    if (w != w) return ~0ull;
    const uint64_t b = (uint64_t)__double_as_longlong(w);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ---- FRONT, second half ---------------------------------------------------------------------------------------------
// x = scores[nn], e = exp(x - 1) (or x when the softmax is off); per 4096-slot block the sum of e in the spec order,
// the extrema of x, the particles the prune kept, NaN among the scores.
// (round 6) The block's slots are read and written slot-per-lane - a wave's load is 64 consecutive indices, its stores 64
// consecutive values - and the numerators reach the chunk-per-thread layout the summation spec is written in through LDS
// (index i at i + i / 16: the chunk reads are spread over the banks).  With sixteen CONSECUTIVE slots per thread every lane of
// every load and store was a cache line of its own: 64 look-ups an instruction, ~80 instructions a wave - the launch's bound.
MD int lds_chunk_pos(int i) { return i + (i >> 4); }
constexpr int LDS_CHUNK_DOUBLES = SCAN_BLOCK + SCAN_BLOCK / 16;

__global__ __launch_bounds__(256) void k_loop_xe(const int32_t* __restrict__ ctl_i, const double* __restrict__ scores,
                                                 const int32_t* __restrict__ nn_idx, const uint8_t* __restrict__ valid,
                                                 int32_t softmax, int32_t unit, double* __restrict__ x_out, double* __restrict__ e_out,
                                                 double* __restrict__ bsum, double* __restrict__ bmax, double* __restrict__ bmin,
                                                 int32_t* __restrict__ bkept, int32_t* __restrict__ bnan, int32_t cap_n, int64_t K,
                                                 double* __restrict__ clk = nullptr) {
#ifdef MIDAS_ANNEAL_CLOCKS  // phase clocks of workgroup 0, thread 0 (tools/anneal_clocks.py; slots 36 .. 39 of the profiling block)
    const long long xck0 = wall_clock64();
#define XCK(i) do { if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[56 + (i)] += (double)(wall_clock64() - xck0); } while (0)
#else
#define XCK(i) do { } while (0)
#endif
    __shared__ double s_gtot[16];
    __shared__ double s_mx[4], s_mn[4];
    __shared__ int s_k[4], s_f[4];
    __shared__ double s_t[LDS_CHUNK_DOUBLES];
    const int blk = blockIdx.x, t = threadIdx.x;
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK;
    // batched, unconditional loads on clamped indices (a branch around a load makes hipcc wait for each one in turn); clamped to
    // what the launch was sized for, not to the live count - the control block is a trip of its own, the indices and the scores
    // travel beside it (a slot behind the live count holds anything: its row number is clamped too, its values are masked)
    double v[SCAN_CHUNK];
    int32_t nn[SCAN_CHUNK];
    unsigned okbits = 0;
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int64_t i = bbase + (int64_t)k * 256 + t, ic = i < cap_n ? i : cap_n - 1;
        const int32_t r = nn_idx[ic];
        nn[k] = r < 0 ? 0 : ((int64_t)r >= K ? (int32_t)(K - 1) : r);
        okbits |= valid[ic] != 0 ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) v[k] = unit ? 1.0 : scores[nn[k]];
    const int64_t n = ctl_i[LOOP_I_N];
    if (bbase >= n) return;
    double mx = -INFINITY, mn = INFINITY;
    XCK(36);
    int kept = 0;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int64_t i = bbase + (int64_t)k * 256 + t;
        const bool in = i < n;
        const double xv = v[k];
        mx = in && xv > mx ? xv : mx;
        mn = in && xv < mn ? xv : mn;
        kept += in && ((okbits >> k) & 1u) ? 1 : 0;
        nan |= in && xv != xv;
        if (in) x_out[i] = xv;
    }
    XCK(37);  // scores there
    if (softmax) {
#pragma unroll
        for (int k = 0; k < SCAN_CHUNK; ++k) {
            v[k] = exp_spec(v[k] - 1.0);
            // four exponentials interleaved: each is a dependent chain of ~25 fma at ~20 cycles a step, and the workgroup's four
            // waves are alone on their SIMDs
            if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int il = k * 256 + t;
        const bool in = bbase + il < n;
        if (in) e_out[bbase + il] = v[k];
        s_t[lds_chunk_pos(il)] = in ? v[k] : 0.0;
    }
    XCK(38);  // exponentials done, values stored
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) v[j] = s_t[17 * t + j];  // the own chunk: slots 16 t .. 16 t + 15 of the block
    const double W = block_total(v, s_gtot);
    XCK(39);  // block total
    mx = wave_max_dpp(mx);  // (register moves; NaN never wins, as in the shuffle form - flagged beside)
    mn = wave_min_dpp(mn);
    kept = lw_isum(kept);
    const bool wnan = __any(nan);
    if ((t & 63) == 0) { s_mx[t >> 6] = mx; s_mn[t >> 6] = mn; s_k[t >> 6] = kept; s_f[t >> 6] = wnan ? 1 : 0; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; ++w) {
            mx = s_mx[w] > mx ? s_mx[w] : mx;
            mn = s_mn[w] < mn ? s_mn[w] : mn;
            kept += s_k[w];
        }
        const int f = s_f[0] | s_f[1] | s_f[2] | s_f[3];
        bsum[blk] = W;
        bmax[blk] = f ? NAN : mx;  // NaN propagates, as torch.max / torch.min do
        bmin[blk] = f ? NAN : mn;
        bkept[blk] = kept;
        bnan[blk] = f;
    }
    XCK(40);
#ifdef MIDAS_ANNEAL_CLOCKS
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[56 + 41] += 1.0;
#endif
}

// S = blocks summed in order; guard; w = (e or x) / S * valid; every particle back onto its codebook pose when all of them
// were pruned (filter.py:176-179); block 0 finalises the control block and the rmse.  (The arithmetic lives in loop_weights.hpp:
// frames whose cluster moments follow in the same call have it at the head of that launch instead, cluster.hip.)
__global__ __launch_bounds__(256) void k_loop_weights(LoopWeightsArgs a) {
    __shared__ double s_sum[LAZY_MAX_BLOCKS];
    __shared__ double s_red[8], s_ab[8];
    __shared__ int s_ired[8];
    const int blk = blockIdx.x, t = threadIdx.x;
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK;
    // Everything this workgroup reads leaves before the live count is looked at (the control block is a trip of its own): the block
    // results of every LAUNCHED block (the ones behind the live count hold stale values and are masked), and the slots'
    // numerators / masks on indices bounded by the launches' cap.
    const LoopWeightsPre pre = loop_weights_prefetch(a, blk == 0);
    double pe[SCAN_CHUNK], px[SCAN_CHUNK];
    uint8_t pv[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = bbase + (int64_t)j * 256 + t, ic = i < a.grid_n ? i : a.grid_n - 1;
        pe[j] = a.e[ic]; px[j] = a.x[ic]; pv[j] = a.valid[ic];
    }
    const int64_t n = a.ctl_i[LOOP_I_N];
    if (bbase >= n && blk != 0) return;
    const LoopWeightsHead h = loop_weights_head(a, pre, n, s_sum, s_red, s_ired);
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = bbase + (int64_t)j * 256 + t;
        if (i < n) {
            loop_weight_store(a, h, i, pe[j], px[j], pv[j]);
            if (h.drifted) {
                const float4* s4 = reinterpret_cast<const float4*>(a.cb_poses + (size_t)a.nn_idx[i] * 16);
                float4* d4 = reinterpret_cast<float4*>(a.poses_prop + (size_t)i * 16);
                d4[0] = s4[0]; d4[1] = s4[1]; d4[2] = s4[2]; d4[3] = s4[3];
            }
        }
    }
    if (blk != 0) return;
    loop_weights_finalise(a, h, pre, s_ab);
}

// ---- ANNEAL ---------------------------------------------------------------------------------------------------------
// One workgroup: the clusters present (labels ascending) -> compact rows, var = float32 running sum of their stds / count
// (torch.mean(cluster_stds), filter.py:189), then particle_filter.annealing's rule (:413-447) in the float32 arithmetic torch
// uses for `var / self.particle_var`, `1.0 - ratio` and `* N`.  Also clears the select histograms.
// (one thread) -> {mode, k}
MD void loop_decide(int32_t* __restrict__ ctl_i, double* __restrict__ ctl_d, const float* __restrict__ centers_all,
                    const float* __restrict__ stds_all, const int64_t* __restrict__ counts_all, float* __restrict__ centers_out,
                    float* __restrict__ stds_out, int32_t floor_n, int& mode_out, int& k_out) {
    const int32_t n = ctl_i[LOOP_I_N];
    int C = ctl_i[LOOP_I_NCL] + 1;
    if (C > LOOP_MAX_CLUSTERS) { C = LOOP_MAX_CLUSTERS; ctl_i[LOOP_I_ERR] |= 1; }
    int np = 0;
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
        if (counts_all[c] == 0) continue;
        // the translation and the bottom row; the rotation entries of the row come from loop_rotations (second workgroup)
        for (int i = 3; i < 12; i += 4) centers_out[np * 16 + i] = centers_all[c * 16 + i];
        for (int i = 12; i < 16; ++i) centers_out[np * 16 + i] = centers_all[c * 16 + i];
        for (int i = 0; i < 3; ++i) {
            const float s = stds_all[c * 3 + i];
            stds_out[np * 3 + i] = s;
            sum = sum + s;
        }
        ++np;
    }
    const float var = sum / (float)(3 * np);
    ctl_i[LOOP_I_NPRES] = np;
    ctl_d[LOOP_D_VAR] = (double)var;
    int mode = 0, k = 0;
    if (!ctl_i[LOOP_I_VARSET]) {  // first call: remember the variance and the particle count (:413-417)
        ctl_d[LOOP_D_VARPREV] = (double)var;
        ctl_i[LOOP_I_INIT] = n;
        ctl_i[LOOP_I_VARSET] = 1;
    } else if (var == 0.0f) {     // converged to a single pose (:418-420)
    } else {
        const float prev = (float)ctl_d[LOOP_D_VARPREV];
        const float ratio = var / prev;
        ctl_d[LOOP_D_VARPREV] = (double)var;
        if (ratio < 1.0f) {
            const int a = (int)((1.0f - ratio) * (float)n);
            const int b = n - floor_n < 0 ? floor_n - n : n - floor_n;
            k = a < b ? a : b;
            k = k < n / 3 ? k : n / 3;
            mode = k > 0 ? 1 : 0;
        } else if (ratio > 1.0f) {
            const int a = (int)((ratio - 1.0f) * (float)n);
            k = a < n / 3 ? a : n / 3;
            mode = (k + n > ctl_i[LOOP_I_INIT] || k <= 0) ? 0 : 2;
        }
        if (!mode) k = 0;
    }
    ctl_i[LOOP_I_MODE] = mode;
    ctl_i[LOOP_I_K] = k;
    ctl_i[LOOP_I_NSET] = mode == 1 ? n - k : mode == 2 ? n + k : n;
    mode_out = mode;
    k_out = k;
}

// Second workgroup of the annealing kernels: the rotation of every present cluster's centre (top eigenvector of the moment
// matrix k_loop_cluster_finish left in `rot`), written into the compact row loop_decide gives the cluster - the selection
// does not wait for it.  Lane c < 8 of one wave takes cluster slot c.
MD void loop_rotations(const int32_t* __restrict__ ctl_i, const int64_t* __restrict__ counts_all, const double* __restrict__ rot,
                       float* __restrict__ centers_out) {
    int C = ctl_i[LOOP_I_NCL] + 1;
    C = C > LOOP_MAX_CLUSTERS ? LOOP_MAX_CLUSTERS : C;
    const int c = threadIdx.x;
    if (c >= C || counts_all[c] == 0) return;
    int np = 0;
    for (int j = 0; j < c; ++j) np += counts_all[j] != 0 ? 1 : 0;
    double A10[10];
    for (int k = 0; k < 10; ++k) A10[k] = rot[(size_t)c * 10 + k];
    cluster_rotation_write(A10, centers_out + np * 16);
}

__global__ __launch_bounds__(256) void k_loop_decide(int32_t* __restrict__ ctl_i, double* __restrict__ ctl_d,
                                                     const float* __restrict__ centers_all, const float* __restrict__ stds_all,
                                                     const int64_t* __restrict__ counts_all, float* __restrict__ centers_out,
                                                     float* __restrict__ stds_out, uint32_t* __restrict__ hist,
                                                     int32_t* __restrict__ sel_state, int32_t floor_n, const double* __restrict__ rot,
                                                     int32_t frozen = 0) {
    if (blockIdx.x == 1) { loop_rotations(ctl_i, counts_all, rot, centers_out); return; }
    const int t = threadIdx.x;
    for (int i = t; i < SEL_PASSES * SEL_BINS; i += 256) hist[i] = 0u;
    for (int i = t; i < 4 * (SEL_PASSES + 2); i += 256) sel_state[i] = 0;
    __syncthreads();
    if (t != 0) return;
    int mode, k;
    loop_decide(ctl_i, ctl_d, centers_all, stds_all, counts_all, centers_out, stds_out, floor_n, mode, k);
    if (frozen && mode) {  // the caller said annealing cannot act and no selection was launched: the set stays, the frame says so
        ctl_i[LOOP_I_ERR] |= 128;
        ctl_i[LOOP_I_MODE] = 0; ctl_i[LOOP_I_K] = 0; ctl_i[LOOP_I_NSET] = ctl_i[LOOP_I_N];
        k = 0;
    }
    sel_state[0] = 0; sel_state[1] = 0; sel_state[2] = k;  // pass 0: empty prefix, rank k
}

// the selection key of particle i: ascending for a removal (k smallest weights), complemented for a duplication (k largest)
MD uint64_t select_key(const double* __restrict__ w, int64_t i, int mode) {
    const uint64_t key = weight_key(w[i]);
    return mode == 2 ? ~key : key;
}

// (prefix, rank) of pass p from the histogram of pass p - 1 and its own (prefix, rank); every workgroup computes it for
// itself, workgroup 0 also publishes it for the next pass.  state[4 * p + {0, 1, 2}] = prefix hi, prefix lo, rank.
MD void select_advance(const uint32_t* __restrict__ hist, int32_t* __restrict__ state, int p, uint64_t& prefix, int& krem,
                       int* s_scan) {
    const int t = threadIdx.x;
    prefix = ((uint64_t)(uint32_t)state[4 * (p - 1)] << 32) | (uint32_t)state[4 * (p - 1) + 1];
    krem = state[4 * (p - 1) + 2];
    const uint32_t* h = hist + (size_t)(p - 1) * SEL_BINS;
    // 2048 bins, 8 per thread: the bin in which the running count reaches the rank
    uint32_t c[8];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = h[t * 8 + j]; mine += (int)c[j]; }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) s_scan[t >> 6] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) before += s_scan[w];
    __syncthreads();
    if (krem > before && krem <= before + mine) {  // exactly one thread
        int run = before, d = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (krem > run && krem <= run + (int)c[j]) { d = j; break; }
            run += (int)c[j];
        }
        s_scan[4] = t * 8 + d;
        s_scan[5] = krem - run;
    }
    __syncthreads();
    const int digit = s_scan[4];
    krem = s_scan[5];
    prefix = (prefix << kSelWidth[p - 1]) | (uint64_t)digit;
    __syncthreads();
    if (blockIdx.x == 0 && t == 0) {
        state[4 * p] = (int32_t)(uint32_t)(prefix >> 32);
        state[4 * p + 1] = (int32_t)(uint32_t)prefix;
        state[4 * p + 2] = krem;
    }
}

// pass P of the radix select: histogram of digit P over the particles whose key starts with the prefix found so far
template <int P>
__global__ __launch_bounds__(256) void k_loop_select(const int32_t* __restrict__ ctl_i, const double* __restrict__ w,
                                                     uint32_t* __restrict__ hist, int32_t* __restrict__ state) {
    __shared__ uint32_t s_h[SEL_BINS];
    __shared__ int s_scan[8];
    const int mode = ctl_i[LOOP_I_MODE];
    if (!mode) return;
    const int64_t n = ctl_i[LOOP_I_N];
    const int64_t bbase = (int64_t)blockIdx.x * SCAN_BLOCK;
    if (bbase >= n && blockIdx.x != 0) return;
    const int t = threadIdx.x;
    uint64_t prefix = 0;
    int krem = 0;
    if (P > 0) select_advance(hist, state, P, prefix, krem, s_scan);
    if (bbase >= n) return;
    for (int i = t; i < SEL_BINS; i += 256) s_h[i] = 0u;
    __syncthreads();
    const int shift = kSelShift[P], width = kSelWidth[P];
#pragma unroll 4
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = bbase + (int64_t)j * 256 + t;
        if (i < n) {
            const uint64_t key = select_key(w, i, mode);
            const bool match = P == 0 || (key >> (shift + width)) == prefix;
            if (match) atomicAdd(&s_h[(uint32_t)(key >> shift) & ((1u << width) - 1u)], 1u);
        }
    }
    __syncthreads();
    uint32_t* h = hist + (size_t)P * SEL_BINS;
    for (int i = t; i < SEL_BINS; i += 256)
        if (s_h[i]) atomicAdd(&h[i], s_h[i]);
}

// per 4096-slot block: particles whose key is below the threshold key, and equal to it
__global__ __launch_bounds__(256) void k_loop_compact_count(const int32_t* __restrict__ ctl_i, const double* __restrict__ w,
                                                            const uint32_t* __restrict__ hist, int32_t* __restrict__ state,
                                                            int32_t* __restrict__ c_less, int32_t* __restrict__ c_eq) {
    __shared__ int s_scan[8];
    __shared__ int s_a[4], s_b[4];
    const int mode = ctl_i[LOOP_I_MODE];
    if (!mode) return;
    const int64_t n = ctl_i[LOOP_I_N];
    const int64_t bbase = (int64_t)blockIdx.x * SCAN_BLOCK;
    if (bbase >= n && blockIdx.x != 0) return;
    const int t = threadIdx.x;
    uint64_t T;
    int r;
    select_advance(hist, state, SEL_PASSES, T, r, s_scan);
    if (bbase >= n) return;
    int less = 0, eq = 0;
#pragma unroll 4
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = bbase + (int64_t)j * 256 + t;
        if (i < n) {
            const uint64_t key = select_key(w, i, mode);
            less += key < T ? 1 : 0;
            eq += key == T ? 1 : 0;
        }
    }
    less = lw_isum(less);
    eq = lw_isum(eq);
    if ((t & 63) == 0) { s_a[t >> 6] = less; s_b[t >> 6] = eq; }
    __syncthreads();
    if (t == 0) {
        c_less[blockIdx.x] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        c_eq[blockIdx.x] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    }
}

// The annealed set as an index list.  Selected = key < T, or key == T among the first r such particles in index order.
// Removal: src = the unselected particles in their order.  Duplication: src[0, n) = identity, the selected particles go to
// (sel_key, sel_idx) in index order and are put into topk's output order by the two sort kernels.
__global__ __launch_bounds__(256) void k_loop_compact(const int32_t* __restrict__ ctl_i, const double* __restrict__ w,
                                                      const int32_t* __restrict__ state, const int32_t* __restrict__ c_less,
                                                      const int32_t* __restrict__ c_eq, int32_t* __restrict__ src,
                                                      uint64_t* __restrict__ sel_key, int32_t* __restrict__ sel_idx) {
    __shared__ int s_l[LAZY_MAX_BLOCKS], s_e[LAZY_MAX_BLOCKS];
    __shared__ int s_wl[4], s_we[4];
    const int mode = ctl_i[LOOP_I_MODE];
    const int64_t n = ctl_i[LOOP_I_N];
    const int blk = blockIdx.x, t = threadIdx.x;
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK;
    if (bbase >= n) return;
    const int64_t base = bbase + (int64_t)t * SCAN_CHUNK;
    if (!mode) {  // no annealing this frame: the identity
#pragma unroll 4
        for (int j = 0; j < SCAN_CHUNK; ++j) {
            const int64_t i = bbase + (int64_t)j * 256 + t;
            if (i < n) src[i] = (int32_t)i;
        }
        return;
    }
    const uint64_t T = ((uint64_t)(uint32_t)state[4 * SEL_PASSES] << 32) | (uint32_t)state[4 * SEL_PASSES + 1];
    const int r = state[4 * SEL_PASSES + 2];
    for (int i = t; i < blk; i += 256) { s_l[i] = c_less[i]; s_e[i] = c_eq[i]; }
    __syncthreads();
    int less_before = 0, eq_before = 0;
    for (int i = 0; i < blk; ++i) { less_before += s_l[i]; eq_before += s_e[i]; }
    // chunk-per-thread view: flags of the own 16 slots, exclusive counts inside the block
    unsigned lbits = 0, ebits = 0;
    uint64_t keys[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        const bool in = i < n;
        keys[j] = select_key(w, in ? i : n - 1, mode);
        lbits |= (in && keys[j] < T) ? (1u << j) : 0u;
        ebits |= (in && keys[j] == T) ? (1u << j) : 0u;
    }
    const int ml = __popc(lbits), me = __popc(ebits);
    int il = ml, ie = me;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int a = __shfl_up(il, o), b = __shfl_up(ie, o);
        if ((t & 63) >= o) { il += a; ie += b; }
    }
    if ((t & 63) == 63) { s_wl[t >> 6] = il; s_we[t >> 6] = ie; }
    __syncthreads();
    int lb = less_before + il - ml, eb = eq_before + ie - me;
    for (int wv = 0; wv < (t >> 6); ++wv) { lb += s_wl[wv]; eb += s_we[wv]; }
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        if (i >= n) break;
        const bool isl = (lbits >> j) & 1u, ise = (ebits >> j) & 1u;
        const int taken_eq = eb < r ? eb : r;          // equal keys before i that were selected
        const bool selected = isl || (ise && eb < r);
        const int sel_before = lb + taken_eq;
        if (mode == 1) {
            if (!selected) src[i - sel_before] = (int32_t)i;
        } else {
            src[i] = (int32_t)i;
            if (selected) { sel_key[sel_before] = keys[j]; sel_idx[sel_before] = (int32_t)i; }
        }
        lb += isl ? 1 : 0;
        eb += ise ? 1 : 0;
    }
}

// Duplication only: the k selected (key, index) pairs, key = ~weight_key (ascending = weight descending), are sorted by
// (key, index) - torch.topk's sorted output with ties by index.  Chunks of 2048 pairs by a bitonic network in LDS ...
MD bool pair_less(uint64_t ka, int32_t ia, uint64_t kb, int32_t ib) { return ka < kb || (ka == kb && ia < ib); }

__global__ __launch_bounds__(1024) void k_loop_sort_chunks(const int32_t* __restrict__ ctl_i, const uint64_t* __restrict__ key_in,
                                                          const int32_t* __restrict__ idx_in, uint64_t* __restrict__ key_out,
                                                          int32_t* __restrict__ idx_out) {
    __shared__ uint64_t s_k[SORT_CHUNK];
    __shared__ int32_t s_i[SORT_CHUNK];
    if (ctl_i[LOOP_I_MODE] != 2) return;
    const int k = ctl_i[LOOP_I_K];
    const int c0 = blockIdx.x * SORT_CHUNK;
    if (c0 >= k) return;
    const int t = threadIdx.x;
    for (int i = t; i < SORT_CHUNK; i += 1024) {
        const bool in = c0 + i < k;
        s_k[i] = in ? key_in[c0 + i] : ~0ull;
        s_i[i] = in ? idx_in[c0 + i] : 0x7fffffff;  // padding sorts behind every real pair
    }
    __syncthreads();
    for (int size = 2; size <= SORT_CHUNK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = t; p < SORT_CHUNK / 2; p += 1024) {
                const int lo = 2 * p - (p & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t ka = s_k[lo], kb = s_k[hi];
                const int32_t ia = s_i[lo], ib = s_i[hi];
                const bool swap = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
                if (swap) { s_k[lo] = kb; s_k[hi] = ka; s_i[lo] = ib; s_i[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < SORT_CHUNK; i += 1024)
        if (c0 + i < k) { key_out[c0 + i] = s_k[i]; idx_out[c0 + i] = s_i[i]; }
}

// ... then every pair finds its rank: its place in its own chunk plus, by binary search, the pairs of every other chunk
// that precede it (all pairs are distinct).  src[n + rank] = index.
__global__ __launch_bounds__(256) void k_loop_sort_rank(const int32_t* __restrict__ ctl_i, const uint64_t* __restrict__ key_s,
                                                        const int32_t* __restrict__ idx_s, int32_t* __restrict__ src) {
    if (ctl_i[LOOP_I_MODE] != 2) return;
    const int k = ctl_i[LOOP_I_K];
    const int64_t n = ctl_i[LOOP_I_N];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= k) return;
    const uint64_t key = key_s[q];
    const int32_t idx = idx_s[q];
    const int own = q / SORT_CHUNK, nch = (k + SORT_CHUNK - 1) / SORT_CHUNK;
    int rank = q - own * SORT_CHUNK;
    for (int c = 0; c < nch; ++c) {
        if (c == own) continue;
        int lo = c * SORT_CHUNK, hi = lo + SORT_CHUNK < k ? lo + SORT_CHUNK : k;
        const int c0 = lo;
        while (hi > lo) {
            const int mid = lo + ((hi - lo) >> 1);
            if (pair_less(key_s[mid], idx_s[mid], key, idx)) lo = mid + 1; else hi = mid;
        }
        rank += lo - c0;
    }
    src[n + rank] = idx;
}

// ---- ANNEAL for a small set: decide + radix select + compaction + sort by ONE workgroup -----------------------------------
// After a few frames of annealing the set holds ~10^4 particles and the eleven launches above do a few microseconds of work
// each behind ~4.5 us of launch floor.  When the caller knows that n <= LOOP_SMALL_MAX (LoopEngine bounds the count it saw
// last by the largest growth annealing allows per frame) one 1024-thread workgroup does the same steps - sixteen keys per
// thread in registers, LDS histograms, the k duplicates sorted in LDS - with identical results.
constexpr int LOOP_SMALL_MAX = 16384, LOOP_SMALL_PAIRS = 8192;

// exclusive prefix over the 1024 threads of (a, b); totals returned in ta / tb.  s_w: 32 ints of LDS.
MD void small_scan2(int a, int b, int& ea, int& eb, int& ta, int& tb, int* s_w) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int ia = wave_iscan_dpp(a), ib = wave_iscan_dpp(b);  // (inclusive, by register moves)
    __syncthreads();
    if (lane == 63) { s_w[wv] = ia; s_w[16 + wv] = ib; }
    __syncthreads();
    int pa = 0, pb = 0;
    ta = 0; tb = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int x = s_w[w], y = s_w[16 + w];
        if (w < wv) { pa += x; pb += y; }
        ta += x; tb += y;
    }
    ea = pa + ia - a;
    eb = pb + ib - b;
}

#ifdef MIDAS_ANNEAL_CLOCKS
#define ACK(i) do { if (threadIdx.x == 0) ctl_d[56 + (i)] += (double)(wall_clock64() - ck0); } while (0)
#define ctl_d_dbg(d, k) do { (d)[56 + 6] += 1.0; (d)[56 + 8] += (double)(k); } while (0)
#define ctl_d_pass(d) do { (d)[56 + 15] += 1.0; } while (0)
#else
#define ctl_d_pass(d) do { } while (0)
#define ACK(i) do { } while (0)
#define ctl_d_dbg(d, k) do { } while (0)
#endif

// The sixteen weights a thread of the small-set selection owns, as order-preserving keys: particles t, 1024 + t, .., 15 x 1024 + t
// (indices clamped to the live count).  A wave's load is 64 consecutive values; with sixteen CONSECUTIVE particles per thread
// every lane was a cache line of its own - 64 tag look-ups per instruction, 16 k of them in the one compute unit this
// workgroup has: ~5 us that showed as the wait at the first barrier (tools/anneal_clocks.py "minmax end").
MD void small_keys(const double* __restrict__ w, int n, uint64_t* wkey) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = 1024 * j + t;
        wkey[j] = weight_key(w[i < n ? i : n - 1]);
    }
}

// The selection of a small set by all 1024 threads of one workgroup: radix select of the k-th key, compaction in index order, the
// duplicates in topk's order.  s_w[36 .. 39] = {mode, k, n, 0} (the caller's decision, behind a barrier); wkey: small_keys().
// s_key / s_idx: LOOP_SMALL_PAIRS entries of LDS each.
MD void anneal_small_select(int* s_w, const uint64_t* wkey, int32_t* __restrict__ src, int32_t* __restrict__ ctl_i,
                            double* __restrict__ ctl_d, long long ck0, uint64_t* s_key, int32_t* s_idx) {
    __shared__ uint32_t s_h[SEL_BINS];
    __shared__ uint64_t s_small[64];
    __shared__ uint64_t s_ext[32];   // the waves' extrema (an array of its own: no barrier before s_small's other use)
    __shared__ int2 s_ce[256];       // the compaction's (row, wave) counts (likewise: not the histogram's memory)
    __shared__ uint64_t s_T[1];
    const int t = threadIdx.x;
    const int mode = s_w[36], k = s_w[37], n = s_w[38];
    if (!mode) {
        if (n > LOOP_SMALL_MAX && t == 0) { ctl_i[LOOP_I_MODE] = 0; ctl_i[LOOP_I_K] = 0; ctl_i[LOOP_I_NSET] = n; }
        return;  // identity index list: written by k_loop_weights
    }
    // thread t owns the particles 1024 j + t: index order = (row j, wave, lane)
    uint64_t key[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) key[j] = mode == 2 ? ~wkey[j] : wkey[j];  // (select_key)
    const int mine_n = n - t <= 0 ? 0 : (n - t + 1023) >> 10;  // how many of them exist (rows 0 .. mine_n - 1; n <= 16 384)
    // rows that hold a particle at all (the same for every thread): the per-row loops below skip the others on a scalar branch - at
    // n ~ 8k half of what the sixteen waves of this ONE compute unit would issue
    const int rows = (n + 1023) >> 10;
    ACK(1);
    // ---- radix select: the k-th smallest key T and how many of its equals to take (r)
    // The digits start at the highest bit in which the keys DIFFER (weights of one frame share sign and exponent, often the
    // leading mantissa bits too): with the fixed windows of the large-set kernels the first pass put all 16 k keys into one bin -
    // 16 k same-address LDS atomics one after the other, 15 - 24 us of this kernel's 30 - 50 (phase clocks, tools/anneal_clocks.py).
    uint64_t kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < rows && j < mine_n) { kmin = key[j] < kmin ? key[j] : kmin; kmax = key[j] > kmax ? key[j] : kmax; }
    ACK(30);
    kmin = wave_umin64_dpp(kmin);  // (register moves: midas_math.hpp)
    kmax = wave_umax64_dpp(kmax);
    ACK(31);
    if ((t & 63) == 0) { s_ext[t >> 6] = kmin; s_ext[16 + (t >> 6)] = kmax; }
    __syncthreads();
    ACK(32);
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        kmin = s_ext[w] < kmin ? s_ext[w] : kmin;
        kmax = s_ext[16 + w] > kmax ? s_ext[16 + w] : kmax;
    }
    ACK(33);
    // Enough copies of the smallest key?  (The pruned particles' zeros when particles are dropped, the ~80 particles on the
    // best-scoring entry when the best are duplicated: k is a per cent or two of the set.)  Then T is that key: no pass at all.
    {
        int eq = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < rows) eq += (j < mine_n && key[j] == kmin) ? 1 : 0;
        eq = lw_isum(eq);
        ACK(34);
        if ((t & 63) == 0 && eq) atomicAdd(&s_w[39], eq);
        __syncthreads();
    }
    const bool at_min = s_w[39] >= k;
    ACK(9);
    int top = (kmin == kmax || at_min) ? 0 : 64 - __builtin_clzll(kmin ^ kmax);  // bits [top, 64) are common to every key
#ifdef MIDAS_ANNEAL_CLOCKS
    long long ckl = wall_clock64();
#define ACKD(i) do { if (threadIdx.x == 0) { const long long nw = wall_clock64(); ctl_d[56 + (i)] += (double)(nw - ckl); ckl = nw; } } while (0)
#else
#define ACKD(i) do { } while (0)
#endif
    uint64_t prefix = top >= 64 ? 0ull : (kmin >> top);  // (top == 0: T = kmin, r = k)
    int krem = k;
#pragma unroll 1
    while (top > 0) {
        const int width = top < 11 ? top : 11, shift = top - width;
        s_h[t] = 0u; s_h[t + 1024] = 0u;
        __syncthreads();
        ACKD(10);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool match = top >= 64 || (key[j] >> top) == prefix;
            if (j < rows && j < mine_n && match) atomicAdd(&s_h[(uint32_t)(key[j] >> shift) & ((1u << width) - 1u)], 1u);
        }
        __syncthreads();
        ACKD(11);
        const int c0 = (int)s_h[2 * t], c1 = (int)s_h[2 * t + 1];
        int e0, e1, t0, t1;
        small_scan2(c0 + c1, 0, e0, e1, t0, t1, s_w);
        ACKD(12);
        if (krem > e0 && krem <= e0 + c0 + c1) {  // exactly one thread
            const bool first = krem <= e0 + c0;
            s_w[32] = 2 * t + (first ? 0 : 1);
            s_w[33] = krem - e0 - (first ? 0 : c0);
            s_w[34] = first ? c0 : c1;  // keys in the chosen bin
            s_w[35] = 0;                // (counter of the short finish below)
        }
        __syncthreads();
        prefix = (prefix << width) | (uint64_t)s_w[32];
        krem = s_w[33];
        const int in_bin = s_w[34];
        __syncthreads();
        ACKD(13);
        if (threadIdx.x == 0) ctl_d_pass(ctl_d);
        top = shift;
        // One value left?  Particles that share a nearest entry share its score and so their weight: a frame's 10^4 keys are
        // ~100 distinct values (and the pruned particles' zeros), the chosen bin usually holds ONE of them, many times - the
        // remaining digits would each be a pass of same-address atomics to learn nothing.  Somebody's key of the bin is the
        // candidate; if every key of the bin equals it, it is T.
        if (top > 0 && in_bin > 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < mine_n && (key[j] >> top) == prefix) s_T[0] = key[j];
            __syncthreads();
            const uint64_t cand = s_T[0];
            int eq = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < rows) eq += (j < mine_n && key[j] == cand) ? 1 : 0;
            eq = lw_isum(eq);
            if ((t & 63) == 0 && eq) atomicAdd(&s_w[35], eq);
            __syncthreads();
            const bool single = s_w[35] == in_bin;
            __syncthreads();
            if (t == 0) s_w[35] = 0;
            if (single) { prefix = cand; ACKD(14); break; }  // krem of them are taken, in index order
            __syncthreads();
        }
        // Short finish: distinct weights leave a handful of keys in the bin after one or two digits - further passes would each
        // cost a histogram, a scan and five barriers to tell one key from none.  With at most 64 keys left, one wave ranks
        // them: T = the key with #{< T} < krem <= #{<= T}.
        if (in_bin <= 64 && top > 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < mine_n && (key[j] >> top) == prefix) s_small[atomicAdd(&s_w[35], 1)] = key[j];
            __syncthreads();
            if (t < 64) {
                const uint64_t mk = t < in_bin ? s_small[t] : ~0ull;
                int lt = 0, le = 0;
                for (int m = 0; m < in_bin; ++m) {
                    const uint64_t o = s_small[m];
                    lt += o < mk ? 1 : 0;
                    le += o <= mk ? 1 : 0;
                }
                if (t < in_bin && lt < krem && krem <= le) {  // every lane that holds T writes the same two values
                    s_T[0] = mk;
                    s_w[33] = krem - lt;
                }
            }
            __syncthreads();
            prefix = s_T[0];
            krem = s_w[33];
            ACKD(14);
            break;
        }
    }
    const uint64_t T = prefix;
    const int r = krem;
    ACK(2);
    // ---- compaction in index order: per (row, wave) the counts of keys below / equal to T from ballots, one wave scans the 256
    // (row, wave) pairs in index order, a lane's place inside its wave's row is the ballot's bits below it
    {
        const int lane = t & 63, wv = t >> 6;
        const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
        int pl[16], pe[16];
        unsigned lbits = 0, ebits = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            pl[j] = 0; pe[j] = 0;
            if (j < rows) {
                const bool isl = j < mine_n && key[j] < T, ise = j < mine_n && key[j] == T;
                const uint64_t bl = __ballot(isl), be = __ballot(ise);
                pl[j] = __popcll(bl & below);
                pe[j] = __popcll(be & below);
                lbits |= isl ? (1u << j) : 0u;
                ebits |= ise ? (1u << j) : 0u;
                if (lane == 0) s_ce[j * 16 + wv] = make_int2(__popcll(bl), __popcll(be));
            } else if (lane == 0) {
                s_ce[j * 16 + wv] = make_int2(0, 0);
            }
        }
        __syncthreads();
        if (wv == 0) {
            const int2 c0 = s_ce[4 * lane], c1 = s_ce[4 * lane + 1], c2 = s_ce[4 * lane + 2], c3 = s_ce[4 * lane + 3];
            int sl = c0.x + c1.x + c2.x + c3.x, se = c0.y + c1.y + c2.y + c3.y;
            const int il = wave_iscan_dpp(sl), ie = wave_iscan_dpp(se);
            int el = il - sl, ee = ie - se;
            s_ce[4 * lane] = make_int2(el, ee);
            el += c0.x; ee += c0.y;
            s_ce[4 * lane + 1] = make_int2(el, ee);
            el += c1.x; ee += c1.y;
            s_ce[4 * lane + 2] = make_int2(el, ee);
            el += c2.x; ee += c2.y;
            s_ce[4 * lane + 3] = make_int2(el, ee);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < rows && j < mine_n) {
                const int i = 1024 * j + t;
                const int2 b = s_ce[j * 16 + wv];
                const int lb = b.x + pl[j], eb = b.y + pe[j];
                const bool isl = (lbits >> j) & 1u, ise = (ebits >> j) & 1u;
                const bool selected = isl || (ise && eb < r);
                const int sel_before = lb + (eb < r ? eb : r);
                if (mode == 1) {
                    if (!selected) src[i - sel_before] = i;
                } else if (selected) {
                    s_key[sel_before] = key[j];
                    s_idx[sel_before] = i;
                }
            }
        }
    }
    ACK(3);
    if (mode != 2) return;
    // ---- the k duplicates in topk's order: (key, index) ascending
    if (k <= 256) {
        // few duplicates (the usual case: the variance ratio moves by a per cent or two per frame): every pair counts the pairs
        // before it - one barrier pair instead of the network's log^2 stages of a 16-wave barrier each
        __syncthreads();
        uint64_t mk = 0;
        int32_t mi = 0;
        int rank = 0;
        if (t < k) {
            mk = s_key[t]; mi = s_idx[t];
            for (int m = 0; m < k; ++m) rank += pair_less(s_key[m], s_idx[m], mk, mi) ? 1 : 0;
        }
        ACK(4);
        if (t < k) src[n + rank] = mi;
        ACK(5);
        if (threadIdx.x == 0) { ctl_d_dbg(ctl_d, k); }
        return;
    }
    // larger sets: bitonic network over the next power of two
    int P = 2;
    while (P < k) P <<= 1;
    __syncthreads();
    for (int i = k + t; i < P; i += 1024) { s_key[i] = ~0ull; s_idx[i] = 0x7fffffff; }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = t; q < P / 2; q += 1024) {
                const int lo = 2 * q - (q & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t ka = s_key[lo], kb = s_key[hi];
                const int32_t ia = s_idx[lo], ib = s_idx[hi];
                const bool swap = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
                if (swap) { s_key[lo] = kb; s_key[hi] = ka; s_idx[lo] = ib; s_idx[hi] = ia; }
            }
            __syncthreads();
        }
    }
    ACK(4);
    for (int q = t; q < k; q += 1024) src[n + q] = s_idx[q];
    ACK(5);
    if (threadIdx.x == 0) { ctl_d_dbg(ctl_d, k); }
}


template <bool DECIDE>
__global__ __launch_bounds__(1024) void k_loop_anneal_small(int32_t* __restrict__ ctl_i, double* __restrict__ ctl_d,
                                                            const float* __restrict__ centers_all, const float* __restrict__ stds_all,
                                                            const int64_t* __restrict__ counts_all, float* __restrict__ centers_out,
                                                            float* __restrict__ stds_out, int32_t floor_n,
                                                            const double* __restrict__ w, int32_t* __restrict__ src,
                                                            const double* __restrict__ rot) {
#ifdef MIDAS_ANNEAL_CLOCKS
    const long long ck0 = wall_clock64();
#else
    const long long ck0 = 0;
#endif
    if (DECIDE && blockIdx.x == 1) { loop_rotations(ctl_i, counts_all, rot, centers_out); ACK(7); return; }
    __shared__ int s_w[40];
    const int t = threadIdx.x;
    if (t == 0) {
        int mode = ctl_i[LOOP_I_MODE], k = ctl_i[LOOP_I_K];  // !DECIDE: a plan made by the host (midas_anneal_select)
        if (DECIDE) loop_decide(ctl_i, ctl_d, centers_all, stds_all, counts_all, centers_out, stds_out, floor_n, mode, k);
        const int n0 = ctl_i[LOOP_I_N];
        if (n0 > LOOP_SMALL_MAX) { ctl_i[LOOP_I_ERR] |= 4; mode = 0; }  // the caller's bound was wrong: no annealing, flagged
        s_w[36] = mode; s_w[37] = k; s_w[38] = n0; s_w[39] = 0;
    }
    __syncthreads();
    ACK(0);
    __shared__ uint64_t s_key[LOOP_SMALL_PAIRS];
    __shared__ int32_t s_idx[LOOP_SMALL_PAIRS];
    uint64_t wkey[16];
    small_keys(w, s_w[38], wkey);
    anneal_small_select(s_w, wkey, src, ctl_i, ctl_d, ck0, s_key, s_idx);
#ifdef MIDAS_ANNEAL_CLOCKS
    if (threadIdx.x == 0) ctl_d[56 + 35] += 1.0;  // launches that got here
#endif
}

// ---- RESAMPLE -------------------------------------------------------------------------------------------------------
// block-local prefix of (e * valid)[src] (or x * valid when the weights are raw scores) in the spec order
__global__ __launch_bounds__(256) void k_loop_scan(const int32_t* __restrict__ ctl_i, const double* __restrict__ x,
                                                   const double* __restrict__ e, const uint8_t* __restrict__ valid,
                                                   const int32_t* __restrict__ src, double* __restrict__ lp,
                                                   double* __restrict__ btot, int32_t* __restrict__ bnan) {
    __shared__ double s_gtot[16];
    __shared__ double s_t[LDS_CHUNK_DOUBLES];  // (k_loop_xe: slot-per-lane in memory, chunk-per-thread for the sums)
    const int64_t n2 = ctl_i[LOOP_I_NSET];
    const int raw = ctl_i[LOOP_I_RAW];
    const int blk = blockIdx.x, t = threadIdx.x;
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK;
    if (bbase >= n2) return;
    double v[SCAN_CHUNK];
    int32_t sv[SCAN_CHUNK];
    bool nan = false;
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int64_t i = bbase + (int64_t)k * 256 + t;
        sv[k] = src[i < n2 ? i : n2 - 1];
    }
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const bool in = bbase + (int64_t)k * 256 + t < n2;
        const double num = raw ? x[sv[k]] : e[sv[k]];
        const double m = num * (valid[sv[k]] ? 1.0 : 0.0);
        nan |= in && m != m;
        s_t[lds_chunk_pos(k * 256 + t)] = in ? m : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) v[j] = s_t[17 * t + j];
    const double W = block_scan(v, v, s_gtot);
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) s_t[17 * t + j] = v[j];
    const int f = __syncthreads_or(nan ? 1 : 0);
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int64_t i = bbase + (int64_t)k * 256 + t;
        if (i < n2) lp[i] = s_t[lds_chunk_pos(k * 256 + t)];
    }
    if (t == 0) { btot[blk] = W; bnan[blk] = f; }
}

struct LoopResampleArgs {
    int32_t* ctl_i;
    double* ctl_d;
    const double* lp;
    const double* btot;
    const int32_t* bnan;
    const int32_t* src;
    const float* poses_prop;
    const double* w;
    const int32_t* nn_idx;
    const int32_t* labels;
    float* poses_out;
    double* weights_out;
    int32_t* hint_out;
    int32_t* labels_out;
    int32_t* ridx;
    int32_t mode;
    const double* u;
    float u32;
    uint64_t seed, step;
    double* log;
    const float* cluster_poses;
    const float* cluster_stds;
    int32_t* host_mirror;
    int32_t cap2;  // slots the prefix array holds (the launch's bound of the annealed set)
};

// n_set draws over cdf_i = (BP_b + lp_i) / total (last slot 1): lower bound (multinomial) / upper bound (systematic) by
// bisection on the exact values; the drawn particle's rows are fetched through src.  All-zero or NaN weights: the annealed
// set goes on unresampled (particle_filter.py:240-241).
__global__ __launch_bounds__(256) void k_loop_resample(LoopResampleArgs a) {
#ifdef MIDAS_ANNEAL_CLOCKS  // (phase clocks of workgroup 0, thread 0: tools/anneal_clocks.py)
    const long long ck0 = wall_clock64();
    double* ctl_d = a.ctl_d;
#define RCK(i) do { if (blockIdx.x == 0) ACK(i); } while (0)
#else
#define RCK(i) do { } while (0)
#endif
    __shared__ double s_bp[LAZY_MAX_BLOCKS + 1];
    __shared__ int s_flag;
    // Small sets (softmax weights: the prefix sums do not decrease): the prefix at the end of every 16-slot chunk is staged in LDS
    // by the whole workgroup (one batch of loads), a draw's search walks those and finishes inside ONE chunk - one cache line
    // of the prefix array.  The plain bisection is ~14 dependent trips to the L2 (7 of the kernel's 10.7 us at N ~ 10^4); any
    // search finds the same slot in a non-decreasing sequence (the values compared are the same exact quotients).
    constexpr int RS_CHUNKS = 2048;
    __shared__ double s_ce[RS_CHUNKS];
    const int t = threadIdx.x;
    // Chunk ends, block totals and NaN flags leave before the control block is looked at (it is a trip of its own): by position,
    // bounded by what the launch was sized for - what lies behind the live count is not used.
    // The table's stride is 16 slots while the launch's bound fits the table, else the next power of two that does (64 slots at
    // N = 100k: eleven probes of LDS + six of memory instead of seventeen dependent probes of memory with a division each -
    // 17 us of a 99 us frame).
    constexpr int RS_PRE = RS_CHUNKS / 256;
    int ssh = 4;
    while ((((int64_t)a.cap2 + ((int64_t)1 << ssh) - 1) >> ssh) > RS_CHUNKS) ++ssh;
    const int64_t stride = (int64_t)1 << ssh;
    const int nb_cap = (int)(((int64_t)a.cap2 + SCAN_BLOCK - 1) / SCAN_BLOCK);
    double ce[RS_PRE];
    const bool pre = a.cap2 > 0;
    if (pre) {
#pragma unroll
        for (int k = 0; k < RS_PRE; ++k) {
            const int64_t p = stride * (t + 256 * k) + stride - 1;
            ce[k] = a.lp[p < a.cap2 ? p : a.cap2 - 1];
        }
    }
    const int tb = t < nb_cap ? t : nb_cap - 1;  // (nb_cap <= LAZY_MAX_BLOCKS = 256: one block a thread)
    const double bt_t = a.btot[tb];
    const int bn_t = a.bnan[tb];
    const int64_t n2 = a.ctl_i[LOOP_I_NSET];
    const int64_t i = (int64_t)blockIdx.x * 256 + t;
    // The frame's bookkeeping (status, log row, counters, the host's mirror: ~2 us of one thread, loads and stores one after the
    // other) is done by the LAST workgroup of the launch when that one has no draws of its own - the launches are sized for a
    // third more than the set can hold, so it normally has none - instead of by the first behind its gathers: the kernel was as
    // long as workgroup 0.
    const bool last_idle = (int64_t)(gridDim.x - 1) * 256 >= n2 && gridDim.x > 1;
    const bool book = last_idle ? blockIdx.x == gridDim.x - 1 : blockIdx.x == 0;
    if ((int64_t)blockIdx.x * 256 >= n2 && !book) return;
    RCK(16);
    const int nb = (int)((n2 + SCAN_BLOCK - 1) / SCAN_BLOCK);
    const int64_t nch = (n2 + stride - 1) >> ssh;  // (<= RS_CHUNKS while n2 <= cap2)
    const bool two_level = a.ctl_i[LOOP_I_RAW] == 0 && nch <= RS_CHUNKS && n2 > 0;
    if (two_level) {  // (the chunk that straddles the live count ends at slot n2 - 1)
#pragma unroll
        for (int k = 0; k < RS_PRE; ++k) {
            const int c = t + 256 * k;
            if (c < (int)nch) s_ce[c] = stride * c + stride - 1 < n2 ? ce[k] : a.lp[n2 - 1];
        }
    }
    const int nbl = nb < nb_cap ? nb : nb_cap;  // (more alive than the launch was sized for: flagged elsewhere; stay inside the arrays)
    if (t < nbl) s_bp[t + 1] = bt_t;
    if (t == 0) s_flag = 0;
    __syncthreads();
    if (t < nbl && bn_t) s_flag = 1;
    if (t == 0) {
        double acc = 0.0;
        s_bp[0] = 0.0;
        for (int b = 0; b < nb; ++b) { const double w = s_bp[b + 1]; acc = acc + w; s_bp[b + 1] = acc; }
    }
    __syncthreads();
    const double total = s_bp[nb];
    const int status = (s_flag || total != total) ? 2 : (total == 0.0 ? 1 : 0);
    RCK(17);
    if (i < n2) {
        int64_t pick = i;
        if (!status) {
            const bool upper = a.mode == MIDAS_RESAMPLE_SYSTEMATIC;
            double uq;
            if (!upper) {
                uq = a.u ? a.u[i] : philox_uniform53((uint64_t)i, a.seed, a.step);
            } else {
                const float r = a.u32 >= 0.0f ? a.u32 : philox_uniform24(a.seed, a.step);
                const float off = r / (float)n2;
                uq = (double)i / (double)n2 + (double)off;
                uq = uq >= 1.0 ? uq - 1.0 : uq;
            }
            int64_t lo = 0, hi = n2;
            auto left_exact = [&](int64_t m) {
                const double c = m == n2 - 1 ? 1.0 : (s_bp[m >> 12] + a.lp[m]) / total;
                return upper ? c <= uq : c < uq;
            };
            if (two_level) {
                // the first chunk whose end is not left of the draw, then inside it - probed WITHOUT the division (total > 0 here:
                // bp + v against uq total; a float64 division is ~40 dependent instructions, fifteen probes were 3 of the kernel's
                // 10 us), then the exact predicate on the two neighbours of the boundary, walking where the two disagree: the
                // predicate on the quotients is monotone in the slot, so the result is the bisection's (resample_search.hpp does
                // the same for the large sets)
                const double tt = uq * total;
                auto left_fast = [&](int64_t m, double v_) {
                    if (m >= n2 - 1) return false;
                    const double c = s_bp[m >> 12] + v_;
                    return upper ? c <= tt : c < tt;
                };
                int cl = 0, ch = (int)nch;
                while (ch > cl) {
                    const int cm = cl + ((ch - cl) >> 1);
                    if (left_fast(stride * cm + stride - 1, s_ce[cm])) cl = cm + 1; else ch = cm;
                }
                lo = stride * cl;
                hi = lo + stride < n2 ? lo + stride : n2;
                if (cl >= (int)nch) lo = hi = n2;
                // (inside the chunk: four dependent probes of one cache line; its sixteen values fetched together - eight 16-byte
                // loads a lane, every lane a line of its own - made the kernel 1 us slower)
                while (hi > lo) {
                    const int64_t mid = lo + ((hi - lo) >> 1);
                    if (left_fast(mid, a.lp[mid])) lo = mid + 1; else hi = mid;
                }
                while (lo > 0 && !left_exact(lo - 1)) --lo;
                while (lo < n2 && left_exact(lo)) ++lo;
            } else {
                while (hi > lo) {
                    const int64_t mid = lo + ((hi - lo) >> 1);
                    if (left_exact(mid)) lo = mid + 1; else hi = mid;
                }
            }
            pick = lo < n2 ? lo : n2 - 1;
        }
        RCK(18);
        a.ridx[i] = (int32_t)pick;
        const int32_t s = a.src[pick];
        const float4* s4 = reinterpret_cast<const float4*>(a.poses_prop + (size_t)s * 16);
        float4* d4 = reinterpret_cast<float4*>(a.poses_out + (size_t)i * 16);
        const float4 r0 = s4[0], r1 = s4[1], r2 = s4[2], r3 = s4[3];
        d4[0] = r0; d4[1] = r1; d4[2] = r2; d4[3] = r3;
        a.weights_out[i] = a.w[s];
        a.hint_out[i] = a.nn_idx[s];
        a.labels_out[i] = a.labels[s];
        RCK(19);
    }
    if (book && t == 0) {
        const int32_t n = a.ctl_i[LOOP_I_N];
        a.ctl_i[LOOP_I_STATUS] = status;
        a.ctl_d[LOOP_D_TOTAL] = total;
        if (a.log) {
            double* L = a.log;
            L[0] = (double)a.ctl_i[LOOP_I_FRAME]; L[1] = (double)n; L[2] = (double)n2;
            L[3] = a.ctl_d[LOOP_D_RMSE_T]; L[4] = a.ctl_d[LOOP_D_RMSE_R];
            L[5] = (double)a.ctl_i[LOOP_I_KEPT]; L[6] = (double)a.ctl_i[LOOP_I_DRIFT]; L[7] = (double)status;
            L[8] = (double)a.ctl_i[LOOP_I_MODE]; L[9] = (double)a.ctl_i[LOOP_I_K];
            const int np = a.ctl_i[LOOP_I_NPRES];
            L[10] = (double)np; L[11] = a.ctl_d[LOOP_D_VAR]; L[12] = a.ctl_d[LOOP_D_S]; L[13] = (double)a.ctl_i[LOOP_I_RAW];
            // (with min_samples = n / 5 a frame has at most five clusters + the noise label; the row holds eight: bit 8 says
            // the centres of a frame with more present labels are truncated here - the engine's own arrays hold them all)
            // L[15] = THIS frame's condition bits: the word is cleared once the row holds it (it used to stay set, and every
            // later row repeated the first overflow)
            L[14] = (double)a.ctl_i[LOOP_I_NCL]; L[15] = (double)(a.ctl_i[LOOP_I_ERR] | (np > 8 ? 8 : 0));
            a.ctl_i[LOOP_I_ERR] = 0;
            for (int c = 0; c < 8 && c < np; ++c) {
                for (int k = 0; k < 16; ++k) L[16 + c * 19 + k] = (double)a.cluster_poses[c * 16 + k];
                for (int k = 0; k < 3; ++k) L[16 + c * 19 + 16 + k] = (double)a.cluster_stds[c * 3 + k];
            }
        }
        a.ctl_i[LOOP_I_FRAME] += 1;
        a.ctl_i[LOOP_I_N] = (int32_t)n2;  // the other workgroups read LOOP_I_NSET only
        if (a.host_mirror) {  // pinned host memory: the count first, then the frame number that says it is there
            __hip_atomic_store(&a.host_mirror[1], (int32_t)n2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&a.host_mirror[0], a.ctl_i[LOOP_I_FRAME], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#ifdef MIDAS_ANNEAL_CLOCKS
        ctl_d[56 + 22] += 1.0;  // launches
#endif
    }
    RCK(20);
}

// ---- host side ------------------------------------------------------------------------------------------------------
// largest float64 t2 with sqrt(t2) <= thr (see api.hip)
static double loop_squared_threshold(double thr) {
    if (!(thr >= 0.0)) return -1.0;
    if (std::isinf(thr)) return INFINITY;
    double t = thr * thr;
    while (std::sqrt(t) > thr) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) <= thr) t = std::nextafter(t, INFINITY);
    return t;
}

struct SelectScratch {
    uint32_t* hist;
    int32_t *state, *c_less, *c_eq, *sel_idx, *srt_idx;
    uint64_t *sel_key, *srt_key;
    size_t ksel;
};

static int select_scratch(midas_ctx* ctx, int64_t cap, SelectScratch& ss) {
    const unsigned nbcap = (unsigned)ceil_div(cap, SCAN_BLOCK);
    ss.ksel = (size_t)cap / 3 + 1;
    void* p;
    int rc;
#define SEL_SCRATCH(field, type, count)                                           \
    if ((rc = midas_scratch(ctx, (size_t)(count) * sizeof(type), &p))) return rc; \
    ss.field = (type*)p
    SEL_SCRATCH(hist, uint32_t, SEL_PASSES * SEL_BINS);
    SEL_SCRATCH(state, int32_t, 4 * (SEL_PASSES + 2));
    SEL_SCRATCH(c_less, int32_t, nbcap);
    SEL_SCRATCH(c_eq, int32_t, nbcap);
    SEL_SCRATCH(sel_key, uint64_t, ss.ksel);
    SEL_SCRATCH(sel_idx, int32_t, ss.ksel);
    SEL_SCRATCH(srt_key, uint64_t, ss.ksel);
    SEL_SCRATCH(srt_idx, int32_t, ss.ksel);
#undef SEL_SCRATCH
    return MIDAS_OK;
}

// ctl_i[N, MODE, K] and the cleared histograms / pass-0 state are in place: select, compact, order the duplicates
static int launch_select(midas_ctx* ctx, int64_t cap, const int32_t* ci, const double* w, int32_t* src, const SelectScratch& ss) {
    hipStream_t st = ctx->stream;
    const unsigned nbcap = (unsigned)ceil_div(cap, SCAN_BLOCK);
    hipLaunchKernelGGL(k_loop_select<0>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_select<1>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_select<2>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_select<3>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_select<4>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_select<5>, dim3(nbcap), dim3(256), 0, st, ci, w, ss.hist, ss.state);
    hipLaunchKernelGGL(k_loop_compact_count, dim3(nbcap), dim3(256), 0, st, ci, w, (const uint32_t*)ss.hist, ss.state, ss.c_less,
                       ss.c_eq);
    hipLaunchKernelGGL(k_loop_compact, dim3(nbcap), dim3(256), 0, st, ci, w, (const int32_t*)ss.state, (const int32_t*)ss.c_less,
                       (const int32_t*)ss.c_eq, src, ss.sel_key, ss.sel_idx);
    hipLaunchKernelGGL(k_loop_sort_chunks, dim3((unsigned)ceil_div((int64_t)ss.ksel, SORT_CHUNK)), dim3(1024), 0, st, ci,
                       (const uint64_t*)ss.sel_key, (const int32_t*)ss.sel_idx, ss.srt_key, ss.srt_idx);
    hipLaunchKernelGGL(k_loop_sort_rank, dim3((unsigned)ceil_div((int64_t)ss.ksel, 256)), dim3(256), 0, st, ci,
                       (const uint64_t*)ss.srt_key, (const int32_t*)ss.srt_idx, src);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

// a plan made by the host (the op-by-op annealing of the class surface): control block + cleared select state
__global__ __launch_bounds__(256) void k_anneal_plan(int32_t* __restrict__ ctl_i, uint32_t* __restrict__ hist,
                                                     int32_t* __restrict__ state, int32_t n, int32_t mode, int32_t k) {
    const int t = threadIdx.x;
    for (int i = t; i < SEL_PASSES * SEL_BINS; i += 256) hist[i] = 0u;
    for (int i = t; i < 4 * (SEL_PASSES + 2); i += 256) state[i] = (i == 2) ? k : 0;
    if (t < 32) ctl_i[t] = t == LOOP_I_N ? n : t == LOOP_I_MODE ? mode : t == LOOP_I_K ? k : 0;
}

__global__ __launch_bounds__(256) void k_loop_identity(int32_t n, int32_t* __restrict__ src) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) src[i] = i;
}

int launch_anneal_select(midas_ctx* ctx, int64_t N, const double* w, int32_t mode, int64_t k, int32_t ties, int32_t* src, int32_t* info) {
    void* ctl;
    int rc;
    if ((rc = midas_scratch(ctx, 32 * sizeof(int32_t), &ctl))) return rc;
    SelectScratch ss;
    if ((rc = select_scratch(ctx, N, ss))) return rc;
    hipLaunchKernelGGL(k_anneal_plan, dim3(1), dim3(256), 0, ctx->stream, (int32_t*)ctl, ss.hist, ss.state, (int32_t)N, mode, (int32_t)k);
    if (ties == MIDAS_TOPK_TIES_ATEN_CPU) return launch_topk_aten(ctx, N, (const int32_t*)ctl, w, src, info);
    const char* small = getenv("MIDAS_ANNEAL_SMALL");  // tests: the single-workgroup path on the same plan
    if (N <= LOOP_SMALL_MAX && small && atoi(small)) {
        hipLaunchKernelGGL(k_loop_identity, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, (int32_t)N, src);
        hipLaunchKernelGGL(k_loop_anneal_small<false>, dim3(1), dim3(1024), 0, ctx->stream, (int32_t*)ctl, (double*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const int64_t*)nullptr, (float*)nullptr, (float*)nullptr, 0, w, src,
                           (const double*)nullptr);
        LAUNCH_CHECK(ctx);
        return MIDAS_OK;
    }
    return launch_select(ctx, N, (const int32_t*)ctl, w, src, ss);
}

int launch_loop_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* t6, const midas_tree* t3,
                     const midas_loop_args& s, int32_t phases) {
    // the grids cover `cap` particles: the capacity, or the caller's upper bound of the live count
    const int64_t cap = (s.grid_n > 0 && s.grid_n < s.cap) ? s.grid_n : s.cap;
    const unsigned nbcap = (unsigned)ceil_div(cap, SCAN_BLOCK);
    int rc;
    hipStream_t st = ctx->stream;
    LoopWeightsArgs wa{};
    bool weights_merged = false;
    if (phases & MIDAS_LOOP_FRONT) {
        ParticleUpdateArgs pa;
        pa.N = cap;
        pa.n_live = s.ctl_i_dev + LOOP_I_N;
        pa.poses_in = s.poses_dev;
        pa.poses_prop = s.poses_prop_dev;
        pa.odom16 = s.odom16_dev;
        pa.tn = s.tn_dev;
        pa.rot = s.rot_dev;
        pa.std_t = s.std_t;
        pa.std_r = s.std_r;
        pa.seed = s.seed;
        pa.step = s.step;
        pa.hint_in = s.hint_dev;
        pa.nn_idx = s.nn_idx_dev;
        pa.scores = nullptr;  // deferred: k_loop_xe gathers the scores
        pa.valid = s.valid_dev;
        pa.t2 = loop_squared_threshold(s.prune_thr);
        pa.thr = s.prune_thr;
        pa.vlist = (t6->vlist && t6->vlist_mesh == t3) ? (const MeshRec*)t6->vlist : nullptr;
        pa.vscr = pa.vlist ? (const MeshScr*)t6->vscr : nullptr;
        pa.field = t3->field;
        pa.telemetry = (unsigned long long*)s.telemetry_dev;
        pa.gt16 = (s.gt16_dev && s.part_rmse_dev) ? s.gt16_dev : nullptr;
        pa.part_rmse = s.part_rmse_dev;
        if (s.score_stamps_dev && s.score_epoch) { pa.sp.stamps = s.score_stamps_dev; pa.sp.epoch = s.score_epoch; }
        bool fused = false;
        if (ctx->overlap)
            if ((rc = launch_frame_front(ctx, t6, t3, pa, cb, s.code_dev, s.scores_dev, &fused))) return rc;
        if (!fused) {
            if ((rc = launch_score(ctx, cb, 1, s.code_dev, s.scores_dev))) return rc;
            pa.sp.stamps = nullptr;  // scored densely just above
            if ((rc = launch_particle_update(ctx, t6, t3, pa))) return rc;
        }
        void* p;
        if ((rc = midas_scratch(ctx, (size_t)nbcap * (3 * sizeof(double) + 2 * sizeof(int32_t)), &p))) return rc;
        double* bsum = (double*)p;
        double *bmax = bsum + nbcap, *bmin = bmax + nbcap;
        int32_t* bkept = (int32_t*)(bmin + nbcap);
        int32_t* bnan = bkept + nbcap;
        hipLaunchKernelGGL(k_loop_xe, dim3(nbcap), dim3(256), 0, st, (const int32_t*)s.ctl_i_dev, (const double*)s.scores_dev,
                           (const int32_t*)s.nn_idx_dev, (const uint8_t*)s.valid_dev, s.softmax, s.unit_weights, s.x_dev, s.e_dev, bsum, bmax, bmin,
                           bkept, bnan, (int32_t)(cap < (1 << 30) ? cap : (1 << 30)), cb->K,
#ifdef MIDAS_ANNEAL_CLOCKS
                           s.ctl_d_dev
#else
                           (double*)nullptr
#endif
                           );
        wa.ctl_i = s.ctl_i_dev; wa.ctl_d = s.ctl_d_dev; wa.grid_n = (int32_t)(cap < (1 << 30) ? cap : (1 << 30)); wa.nbl = (int32_t)nbcap;
        wa.bsum = bsum; wa.bmax = bmax; wa.bmin = bmin; wa.bkept = bkept; wa.bnan = bnan;
        wa.x = (const double*)s.x_dev; wa.e = (const double*)s.e_dev; wa.valid = (const uint8_t*)s.valid_dev;
        wa.nn_idx = (const int32_t*)s.nn_idx_dev; wa.cb_poses = s.cb_poses_dev; wa.poses_prop = s.poses_prop_dev;
        wa.w_out = s.weights_dev; wa.src = s.src_dev; wa.part_rmse = (const double*)(pa.gt16 ? s.part_rmse_dev : nullptr);
        wa.softmax = s.softmax;
        // the weights at the head of the cluster-moment launch when that launch follows in this call with nothing in between
        // (DBSCAN reads the re-projected poses: frames that cluster keep the launch of their own)
        static const bool merge_env = !(getenv("MIDAS_LOOP_MERGE") && atoi(getenv("MIDAS_LOOP_MERGE")) == 0);
        weights_merged = merge_env && (phases & MIDAS_LOOP_ANNEAL) && !(phases & MIDAS_LOOP_DBSCAN);
        if (!weights_merged) hipLaunchKernelGGL(k_loop_weights, dim3(nbcap), dim3(256), 0, st, wa);
        LAUNCH_CHECK(ctx);
    }
    if (phases & MIDAS_LOOP_DBSCAN) {
        if ((rc = launch_dbscan(ctx, cap, s.ctl_i_dev + LOOP_I_N, s.poses_prop_dev, s.eps, -1, s.labels_dev,
                                s.ctl_i_dev + LOOP_I_NCL, s.ctl_i_dev + LOOP_I_ERR, LOOP_MAX_CLUSTERS - 1)))  // (the frame's cluster arrays hold that many)
            return rc;
    }
    if (phases & MIDAS_LOOP_ANNEAL) {
        void *part, *cen, *sd, *cnt, *rot;
        if ((rc = midas_scratch(ctx, LOOP_MAX_CLUSTERS * 10 * sizeof(double), &rot))) return rc;
        if ((rc = midas_scratch(ctx, (size_t)ceil_div(cap, 256) * LOOP_MAX_CLUSTERS * 36 * sizeof(double), &part))) return rc;
        if ((rc = midas_scratch(ctx, LOOP_MAX_CLUSTERS * 16 * sizeof(float), &cen))) return rc;
        if ((rc = midas_scratch(ctx, LOOP_MAX_CLUSTERS * 3 * sizeof(float), &sd))) return rc;
        if ((rc = midas_scratch(ctx, LOOP_MAX_CLUSTERS * sizeof(int64_t), &cnt))) return rc;
        SelectScratch ss;
        if ((rc = select_scratch(ctx, cap, ss))) return rc;
        if ((rc = launch_loop_cluster(ctx, cap, s.ctl_i_dev, s.poses_prop_dev, s.weights_dev, s.labels_dev, (double*)part, (float*)cen,
                                      (float*)sd, (int64_t*)cnt, (double*)rot, weights_merged ? &wa : nullptr)))
            return rc;
        if (s.anneal_frozen) {
            // live count == floor == the count annealing started from: the rule (particle_filter.py:421-446) cannot remove (needs
            // |n - floor| > 0) or duplicate (needs k + n <= init) - the decision's bookkeeping runs (cluster rows, variance), the ten
            // launches of the selection, which would each find mode 0 and leave, do not (45 us of a 118 us frame at N = 100k)
            hipLaunchKernelGGL(k_loop_decide, dim3(2), dim3(256), 0, st, s.ctl_i_dev, s.ctl_d_dev, (const float*)cen, (const float*)sd,
                               (const int64_t*)cnt, s.cluster_poses_dev, s.cluster_stds_dev, ss.hist, ss.state, s.floor, (const double*)rot, 1);
            LAUNCH_CHECK(ctx);
        } else if (s.topk_ties == MIDAS_TOPK_TIES_ATEN_CPU) {  // the reference's CPU tie choices (topk_aten.hip); the decision as always
            hipLaunchKernelGGL(k_loop_decide, dim3(2), dim3(256), 0, st, s.ctl_i_dev, s.ctl_d_dev, (const float*)cen, (const float*)sd,
                               (const int64_t*)cnt, s.cluster_poses_dev, s.cluster_stds_dev, ss.hist, ss.state, s.floor, (const double*)rot);
            if ((rc = launch_topk_aten(ctx, cap, s.ctl_i_dev, s.weights_dev, s.src_dev, nullptr))) return rc;
        } else if (s.anneal_small && cap <= LOOP_SMALL_MAX) {
            hipLaunchKernelGGL(k_loop_anneal_small<true>, dim3(2), dim3(1024), 0, st, s.ctl_i_dev, s.ctl_d_dev, (const float*)cen, (const float*)sd,
                               (const int64_t*)cnt, s.cluster_poses_dev, s.cluster_stds_dev, s.floor, (const double*)s.weights_dev, s.src_dev,
                               (const double*)rot);
            LAUNCH_CHECK(ctx);
        } else {
            hipLaunchKernelGGL(k_loop_decide, dim3(2), dim3(256), 0, st, s.ctl_i_dev, s.ctl_d_dev, (const float*)cen, (const float*)sd,
                               (const int64_t*)cnt, s.cluster_poses_dev, s.cluster_stds_dev, ss.hist, ss.state, s.floor, (const double*)rot);
            if ((rc = launch_select(ctx, cap, s.ctl_i_dev, s.weights_dev, s.src_dev, ss))) return rc;
        }
    }
    if (phases & MIDAS_LOOP_RESAMPLE) {
        // the annealed set may be a third larger than the particle set the bound was given for
        const int64_t cap2 = cap + cap / 3 + 1 < s.cap ? cap + cap / 3 + 1 : s.cap;
        const unsigned nb2 = (unsigned)ceil_div(cap2, SCAN_BLOCK);
        void *lp, *bt, *bn;
        if ((rc = midas_scratch(ctx, ((size_t)cap2 + SCAN_CHUNK) * sizeof(double), &lp))) return rc;  // (the resample reads whole chunks)
        if ((rc = midas_scratch(ctx, (size_t)nb2 * sizeof(double), &bt))) return rc;
        if ((rc = midas_scratch(ctx, (size_t)nb2 * sizeof(int32_t), &bn))) return rc;
        hipLaunchKernelGGL(k_loop_scan, dim3(nb2), dim3(256), 0, st, (const int32_t*)s.ctl_i_dev, (const double*)s.x_dev,
                           (const double*)s.e_dev, (const uint8_t*)s.valid_dev, (const int32_t*)s.src_dev, (double*)lp, (double*)bt,
                           (int32_t*)bn);
        LoopResampleArgs r;
        r.ctl_i = s.ctl_i_dev; r.ctl_d = s.ctl_d_dev;
        r.lp = (const double*)lp; r.btot = (const double*)bt; r.bnan = (const int32_t*)bn;
        r.src = s.src_dev; r.poses_prop = s.poses_prop_dev; r.w = s.weights_dev; r.nn_idx = s.nn_idx_dev; r.labels = s.labels_dev;
        r.poses_out = s.poses_dev; r.weights_out = s.weights_out_dev; r.hint_out = s.hint_dev; r.labels_out = s.labels_out_dev;
        r.ridx = s.ridx_dev; r.mode = s.resample_mode; r.u = s.u_dev; r.u32 = s.u32; r.seed = s.seed; r.step = s.step;
        r.log = s.log_dev; r.cluster_poses = s.cluster_poses_dev; r.cluster_stds = s.cluster_stds_dev;
        r.host_mirror = s.host_mirror;
        r.cap2 = (int32_t)(cap2 < (1 << 30) ? cap2 : (1 << 30));
        hipLaunchKernelGGL(k_loop_resample, dim3((unsigned)ceil_div(cap2, 256)), dim3(256), 0, st, r);
        LAUNCH_CHECK(ctx);
    }
    return MIDAS_OK;
}

MIDAS_WARM_TU(loop, k_loop_xe)

}  // namespace midas
