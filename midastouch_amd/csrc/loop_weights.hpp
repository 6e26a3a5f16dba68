// loop_weights.hpp - the weights of a loop frame from k_loop_xe's block results: S = blocks summed in order, the isclose guard,
// w = (e or x) / S * valid, every particle back onto its codebook pose when all of them were pruned (filter.py:176-179), and -
// first workgroup - the control block and the rmse (particle_filter.py:449-470).  Shared by k_loop_weights (loop.hip) and by
// k_loop_weights_moments (cluster.hip: the same work at the head of the cluster-moment launch, one launch less per frame).
// Every function is called by all 256 threads of the workgroup.
#pragma once
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

constexpr double LOOP_ISCLOSE_ATOL = 1e-8;  // torch.isclose default atol (particle_filter.py:460-463)

struct LoopWeightsArgs {
    int32_t* ctl_i;
    double* ctl_d;
    int32_t grid_n;        // the launches' bound of the live count
    int32_t nbl;           // 4096-slot blocks k_loop_xe was launched with (<= LAZY_MAX_BLOCKS)
    const double* bsum;    // its block results
    const double* bmax;
    const double* bmin;
    const int32_t* bkept;
    const int32_t* bnan;
    const double* x;
    const double* e;
    const uint8_t* valid;
    const int32_t* nn_idx;
    const float* cb_poses;
    float* poses_prop;
    double* w_out;
    int32_t* src;
    const double* part_rmse;  // nullable
    int32_t softmax;
};

// what a thread requests before the live count is looked at (the control block is another launch's output: these travel with it)
struct LoopWeightsPre {
    double sum, mx, mn;
    int kept, nan;
    double rm_p, rm_q;
};
MD LoopWeightsPre loop_weights_prefetch(const LoopWeightsArgs& a, bool first_wg) {
    const int t = threadIdx.x;
    const int tb = t < a.nbl ? t : a.nbl - 1;  // (one entry a thread: nbl <= LAZY_MAX_BLOCKS = the workgroup's threads)
    LoopWeightsPre p;
    p.sum = a.bsum[tb]; p.mx = a.bmax[tb]; p.mn = a.bmin[tb];
    p.kept = a.bkept[tb]; p.nan = a.bnan[tb];
    p.rm_p = 0.0; p.rm_q = 0.0;  // first workgroup: the first 256 waves' rmse partials (all of them for sets up to 16 384)
    if (first_wg && a.part_rmse) {
        const int nwl = (int)(((int64_t)a.grid_n + 63) / 64), kc = t < nwl ? t : nwl - 1;
        p.rm_p = a.part_rmse[2 * kc]; p.rm_q = a.part_rmse[2 * kc + 1];
    }
    return p;
}

struct LoopWeightsHead {
    int64_t n;
    double Sd, mx, mn;
    int kept, f;
    bool applied, drifted;
};
// s_sum: LAZY_MAX_BLOCKS doubles, s_red: 8 doubles, s_ired: 8 ints of LDS; one __syncthreads()
MD LoopWeightsHead loop_weights_head(const LoopWeightsArgs& a, const LoopWeightsPre& p, int64_t n, double* s_sum, double* s_red, int* s_ired) {
    const int t = threadIdx.x;
    int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    nb = nb < a.nbl ? nb : a.nbl;  // (more alive than the launches were sized for: flagged by the finalisation, the frame is undefined)
    double mx = -INFINITY, mn = INFINITY;
    int kept = 0, f = 0;
    bool anynan = false;
    if (t < nb) {
        s_sum[t] = p.sum;
        anynan |= p.mx != p.mx;
        mx = p.mx > mx ? p.mx : mx;
        mn = p.mn < mn ? p.mn : mn;
        kept += p.kept;
        f |= p.nan;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_xor(mx, o), c = __shfl_xor(mn, o);
        mx = u > mx ? u : mx;
        mn = c < mn ? c : mn;
    }
    kept = wave_isum_dpp(kept);
    f = __any(f != 0 || anynan) ? 1 : 0;
    if ((t & 63) == 0) { s_red[t >> 6] = mx; s_red[4 + (t >> 6)] = mn; s_ired[t >> 6] = kept; s_ired[4 + (t >> 6)] = f; }
    __syncthreads();
    mx = s_red[0]; mn = s_red[4]; kept = s_ired[0]; f = s_ired[4];
    for (int w = 1; w < 4; ++w) {
        mx = s_red[w] > mx ? s_red[w] : mx;
        mn = s_red[4 + w] < mn ? s_red[4 + w] : mn;
        kept += s_ired[w];
        f |= s_ired[4 + w];
    }
    if (f) { mx = NAN; mn = NAN; }
    double S = 0.0;
    for (int i = 0; i < nb; ++i) S = S + s_sum[i];
    LoopWeightsHead h;
    const bool close = __builtin_fabs(mx - mn) <= LOOP_ISCLOSE_ATOL;  // false on NaN
    h.n = n; h.mx = mx; h.mn = mn; h.kept = kept; h.f = f;
    h.applied = a.softmax != 0 && !close;
    h.Sd = h.applied ? S : 1.0;
    h.drifted = kept == 0 && n > 0;
    return h;
}

// one particle (i < n): its weight stored and returned, its place in the identity index list; the re-projection is the caller's
MD double loop_weight_store(const LoopWeightsArgs& a, const LoopWeightsHead& h, int64_t i, double e_i, double x_i, uint8_t valid_i) {
    const double num = h.applied ? e_i : x_i;
    const double w = num / h.Sd * (valid_i ? 1.0 : 0.0);
    a.w_out[i] = w;
    a.src[i] = (int32_t)i;  // until an ANNEAL phase says otherwise the annealed set is the particle set itself
    return w;
}

// first workgroup: rmse and the control block.  s_ab: 8 doubles of LDS; one __syncthreads()
MD void loop_weights_finalise(const LoopWeightsArgs& a, const LoopWeightsHead& h, const LoopWeightsPre& pre, double* s_ab) {
    const int t = threadIdx.x;
    const int64_t n = h.n;
    double p = 0.0, q = 0.0;
    if (a.part_rmse) {
        const int nw = (int)((n + 63) / 64);
        if (t < nw) { p += pre.rm_p; q += pre.rm_q; }
        for (int k = t + 256; k < nw; k += 256) { p += a.part_rmse[2 * k]; q += a.part_rmse[2 * k + 1]; }
        p = wave_sum_ordered(p);
        q = wave_sum_ordered(q);
        if ((t & 63) == 0) { s_ab[t >> 6] = p; s_ab[4 + (t >> 6)] = q; }
    }
    __syncthreads();
    if (t == 0) {
        if (a.part_rmse) {
            p = (s_ab[0] + s_ab[1]) + (s_ab[2] + s_ab[3]);
            q = (s_ab[4] + s_ab[5]) + (s_ab[6] + s_ab[7]);
            a.ctl_d[LOOP_D_RMSE_T] = __builtin_sqrt(p / (double)n);
            a.ctl_d[LOOP_D_RMSE_R] = __builtin_sqrt(q / (double)n);
        }
        a.ctl_d[LOOP_D_S] = h.Sd;
        a.ctl_d[LOOP_D_XMAX] = h.mx;
        a.ctl_d[LOOP_D_XMIN] = h.mn;
        a.ctl_i[LOOP_I_KEPT] = h.kept;
        a.ctl_i[LOOP_I_DRIFT] = h.drifted ? 1 : 0;
        a.ctl_i[LOOP_I_RAW] = h.applied ? 0 : 1;
        a.ctl_i[LOOP_I_NAN] = h.f;
        if (n > a.grid_n) a.ctl_i[LOOP_I_ERR] |= 4;  // the launches were sized for fewer particles than are alive
        a.ctl_i[LOOP_I_NSET] = (int32_t)n;
        a.ctl_i[LOOP_I_MODE] = 0;
        a.ctl_i[LOOP_I_K] = 0;
    }
}

}  // namespace midas
