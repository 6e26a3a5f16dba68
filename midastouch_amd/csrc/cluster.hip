// cluster.hip - K9: cluster centres of the labelled particle set.
//
// particle_filter.get_cluster_centers(method="quat_avg") (modules/particle_filter.py:153-206) with
// pose.xyz_quat_averaged (modules/pose.py:112-147): per cluster label the weights (float32) are flattened to 1 when
// isclose(max - min, 0); the rotation is Markley's quaternion mean = principal eigenvector of sum w q q^T / sum w over
// sign-fixed quaternions (qw >= 0), the translation the weighted mean, the spread sqrt(sum w (t - mean)^2 / sum w).
// The reference walks the clusters in a Python loop of ~25 small torch ops each; here one pass over the particles
// accumulates every cluster's moments (both the weighted and the flattened set, the choice needs the cluster's
// extrema) and one small kernel finishes each cluster (4x4 symmetric eigenproblem: the top eigenvector by repeated squaring in float64, cluster_rot.hpp).
//
// Summation order (deterministic): per particle wave a shuffle tree, per 256-thread block the four waves in order,
// then the blocks in order.
#include "midas_internal.hpp"
#include "midas_math.hpp"
#include "cluster_rot.hpp"
#include "loop_weights.hpp"

namespace midas {

// (the moments' layout - CL_MOM, M_* - and the closed forms made of them: cluster_rot.hpp)

// unit quaternion (x, y, z, w) of a rotation matrix, float64, branch on the largest diagonal term (Shepperd)
MD void quat_of(const float* P, double* q) {
    const double r00 = P[0], r01 = P[1], r02 = P[2], r10 = P[4], r11 = P[5], r12 = P[6], r20 = P[8], r21 = P[9], r22 = P[10];
    const double tr = r00 + r11 + r22;
    double x, y, z, w;
    if (tr > 0.0) {
        const double s = __builtin_sqrt(tr + 1.0) * 2.0;
        w = 0.25 * s; x = (r21 - r12) / s; y = (r02 - r20) / s; z = (r10 - r01) / s;
    } else if (r00 > r11 && r00 > r22) {
        const double s = __builtin_sqrt(1.0 + r00 - r11 - r22) * 2.0;
        w = (r21 - r12) / s; x = 0.25 * s; y = (r01 + r10) / s; z = (r02 + r20) / s;
    } else if (r11 > r22) {
        const double s = __builtin_sqrt(1.0 + r11 - r00 - r22) * 2.0;
        w = (r02 - r20) / s; x = (r01 + r10) / s; y = 0.25 * s; z = (r12 + r21) / s;
    } else {
        const double s = __builtin_sqrt(1.0 + r22 - r00 - r11) * 2.0;
        w = (r10 - r01) / s; x = (r02 + r20) / s; y = (r12 + r21) / s; z = 0.25 * s;
    }
    const double n = __builtin_sqrt(x * x + y * y + z * z + w * w);
    const double sg = w < 0.0 ? -1.0 : 1.0;  // antipodal fix (pose.py:126)
    q[0] = sg * x / n; q[1] = sg * y / n; q[2] = sg * z / n; q[3] = sg * w / n;
}

MD double cl_wsum(double v) { return wave_sum_ordered(v); }  // (the xor butterfly 32 .. 1 of the spec, by register moves: midas_math.hpp)
MD double cl_wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
MD double cl_wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// part[(c * nbs + block) * CL_MOM + m] (nbs = the launch's blocks: a cluster's partials lie together, whoever sums them can ask
// for them without knowing how many clusters there are); label_value(c) = the label cluster slot c stands for.  In two steps: what a particle
// contributes (pose rows, label, float32-rounded weight) and the accumulation, so that the loop step can compute the weight in
// the same launch (k_loop_weights_moments).
struct MomIn {
    float P[12];
    int64_t lab;
    double w;
};
template <typename LabelT>
MD MomIn moments_load(int64_t nc, const float* __restrict__ poses, const double* __restrict__ w64, const float* __restrict__ w32,
                      const LabelT* __restrict__ labels) {
    MomIn in;
    const float4* p4 = reinterpret_cast<const float4*>(poses + nc * 16);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float4 r = p4[i]; in.P[4 * i] = r.x; in.P[4 * i + 1] = r.y; in.P[4 * i + 2] = r.z; in.P[4 * i + 3] = r.w; }
    in.lab = (int64_t)labels[nc];
    // particles.weights.float() (:161): the reference averages with float32 weights
    in.w = w64 ? (double)(float)w64[nc] : w32 ? (double)w32[nc] : 0.0;  // (neither: the caller computes the weight itself)
    return in;
}

// skip: a cluster NONE of the workgroup's particles belongs to is not summed (34 wave sums and two barriers each: with six
// clusters alive and one or two of them in a workgroup, 7.5 of the loop launch's 11 us) - its partials are written directly.
// They are not simply +0.0: the sums are of 0 x (a product of quaternion components, a translation) over every lane, which is -0.0
// where EVERY lane's factor is negative - the signs are taken once per workgroup (13 of them) - and NaN where some lane's pose
// is not finite (then nothing is skipped).  Same bits as the sums (tests/test_gpu_knobs.py: MIDAS_MOMENTS_SKIP=0).
template <typename LV>
MD void moments_accumulate(bool live, const MomIn& in, int C, LV label_value, double* __restrict__ part, double (*s_w)[CL_MOM],
                           bool skip = true) {
    __shared__ unsigned long long s_mem[4];
    __shared__ unsigned s_sgn[4];
    const size_t nbs = gridDim.x;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const float* P = in.P;
    const int64_t lab = in.lab;
    const double w = in.w;
    double q[4];
    quat_of(P, q);
    const double tx = P[3], ty = P[7], tz = P[11];
    unsigned long long wgmask = ~0ull;
    unsigned sgn = 0;
    if (skip && C <= 64) {
        unsigned long long mm = 0;
        for (int c = 0; c < C; ++c) mm |= __any(live && lab == label_value(c)) ? (1ull << c) : 0ull;
        unsigned sg = 0;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                const double qq = q[i] * q[j];
                sg |= __all(__builtin_signbit(qq) ? 1 : 0) ? (1u << k) : 0u;
                ++k;
            }
        sg |= __all(__builtin_signbit(tx) ? 1 : 0) ? (1u << 10) : 0u;
        sg |= __all(__builtin_signbit(ty) ? 1 : 0) ? (1u << 11) : 0u;
        sg |= __all(__builtin_signbit(tz) ? 1 : 0) ? (1u << 12) : 0u;
        const bool fin = __builtin_isfinite(q[0]) && __builtin_isfinite(q[1]) && __builtin_isfinite(q[2]) && __builtin_isfinite(q[3]) &&
                         __builtin_isfinite(tx) && __builtin_isfinite(ty) && __builtin_isfinite(tz);
        sg |= __all(fin ? 1 : 0) ? (1u << 31) : 0u;
        if (lane == 0) { s_mem[wv] = mm; s_sgn[wv] = sg; }
        __syncthreads();
        sgn = s_sgn[0] & s_sgn[1] & s_sgn[2] & s_sgn[3];
        if ((sgn >> 31) & 1u) wgmask = s_mem[0] | s_mem[1] | s_mem[2] | s_mem[3];  // (something not finite here: 0 x inf is part of the sums)
    }
    double v[CL_MOM];
    for (int c = 0; c < C; ++c) {
        if (!((wgmask >> (c & 63)) & 1ull)) {  // (workgroup-uniform)
            if (t < CL_MOM) {
                double r = 0.0;
                if (t == M_WMAX) r = -INFINITY;
                else if (t == M_WMIN) r = INFINITY;
                else {
                    int bit = -1;
                    if (t >= M_QQW && t < M_QQW + 10) bit = t - M_QQW;
                    else if (t >= M_QQ1 && t < M_QQ1 + 10) bit = t - M_QQ1;
                    else if (t >= M_TW && t < M_TW + 3) bit = 10 + t - M_TW;
                    else if (t >= M_T1 && t < M_T1 + 3) bit = 10 + t - M_T1;
                    if (bit >= 0 && ((sgn >> bit) & 1u)) r = -0.0;
                }
                part[((size_t)c * nbs + blockIdx.x) * CL_MOM + t] = r;
            }
            continue;
        }
        const bool mine = live && lab == label_value(c);
        const double a = mine ? w : 0.0, b = mine ? 1.0 : 0.0;
        v[M_SW] = a; v[M_CNT] = b; v[M_WMAX] = 0.0; v[M_WMIN] = 0.0;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                const double qq = q[i] * q[j];
                v[M_QQW + k] = a * qq; v[M_QQ1 + k] = b * qq;
                ++k;
            }
        v[M_TW] = a * tx; v[M_TW + 1] = a * ty; v[M_TW + 2] = a * tz;
        v[M_T1] = b * tx; v[M_T1 + 1] = b * ty; v[M_T1 + 2] = b * tz;
        v[M_TTW] = a * tx * tx; v[M_TTW + 1] = a * ty * ty; v[M_TTW + 2] = a * tz * tz;
        v[M_TT1] = b * tx * tx; v[M_TT1 + 1] = b * ty * ty; v[M_TT1 + 2] = b * tz * tz;
        const double mx = cl_wmax(mine ? w : -INFINITY), mn = cl_wmin(mine ? w : INFINITY);
#pragma unroll
        for (int m = 0; m < CL_MOM; ++m)
            if (m != M_WMAX && m != M_WMIN) v[m] = cl_wsum(v[m]);
        __syncthreads();  // the previous cluster's LDS values have been read
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < CL_MOM; ++m) s_w[wv][m] = v[m];
            s_w[wv][M_WMAX] = mx; s_w[wv][M_WMIN] = mn;
        }
        __syncthreads();
        if (t < CL_MOM) {
            double r;
            if (t == M_WMAX) { r = s_w[0][t]; for (int i = 1; i < 4; ++i) r = s_w[i][t] > r ? s_w[i][t] : r; }
            else if (t == M_WMIN) { r = s_w[0][t]; for (int i = 1; i < 4; ++i) r = s_w[i][t] < r ? s_w[i][t] : r; }
            else r = ((s_w[0][t] + s_w[1][t]) + s_w[2][t]) + s_w[3][t];
            part[((size_t)c * nbs + blockIdx.x) * CL_MOM + t] = r;
        }
    }
}

template <typename LabelT, typename LV>
MD void cluster_moments_body(int64_t N, const float* __restrict__ poses, const double* __restrict__ w64,
                             const float* __restrict__ w32, const LabelT* __restrict__ labels, int C, LV label_value,
                             double* __restrict__ part, double (*s_w)[CL_MOM], bool skip) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = n < N;
    const MomIn in = moments_load(live ? n : N - 1, poses, w64, w32, labels);
    moments_accumulate(live, in, C, label_value, part, s_w, skip);
}

__global__ __launch_bounds__(256) void k_cluster_moments(int64_t N, const float* __restrict__ poses, const double* __restrict__ w64,
                                                         const float* __restrict__ w32, const int64_t* __restrict__ labels, int C,
                                                         const int64_t* __restrict__ label_values, double* __restrict__ part, bool skip) {
    __shared__ double s_w[4][CL_MOM];
    cluster_moments_body(N, poses, w64, w32, labels, C, [&](int c) { return label_values[c]; }, part, s_w, skip);
}

// loop engine: labels are DBSCAN's int32 values in [-1, ncl), cluster slot c stands for label c - 1; the particle count
// and ncl come from the control block (midas_loop_step)
__global__ __launch_bounds__(256) void k_loop_cluster_moments(const int32_t* __restrict__ ctl_i, const float* __restrict__ poses,
                                                              const double* __restrict__ w64, const int32_t* __restrict__ labels,
                                                              double* __restrict__ part, bool skip) {
    __shared__ double s_w[4][CL_MOM];
    const int64_t n = ctl_i[LOOP_I_N];
    int C = ctl_i[LOOP_I_NCL] + 1;
    C = C > LOOP_MAX_CLUSTERS ? LOOP_MAX_CLUSTERS : C;
    if ((int64_t)blockIdx.x * 256 >= n) return;
    cluster_moments_body(n, poses, w64, (const float*)nullptr, labels, C, [](int c) { return (int64_t)(c - 1); }, part, s_w, skip);
}

// The same with the frame's weights computed at its head (loop_weights.hpp: k_loop_weights' arithmetic, one particle a thread):
// S and the guard from k_loop_xe's block results by every workgroup for itself, the particle's weight stored and used at once,
// the first workgroup finalises the control block.  One launch less per frame (~5 us of a 90 us frame at N ~ 10^4), and the
// moments need not read the weights back.  Everything a thread reads is requested before the live count is looked at.
#ifdef MIDAS_ANNEAL_CLOCKS  // phase clocks of workgroup 0, thread 0 (tools/anneal_clocks.py; slots 24 .. 29 of the profiling block)
#define WCK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl_d[56 + (i)] += (double)(wall_clock64() - wck0); } while (0)
#else
#define WCK(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_loop_weights_moments(LoopWeightsArgs a, const int32_t* __restrict__ labels,
                                                              double* __restrict__ part, bool skip) {
#ifdef MIDAS_ANNEAL_CLOCKS
    const long long wck0 = wall_clock64();
#endif
    __shared__ double s_w[4][CL_MOM];
    __shared__ double s_sum[LAZY_MAX_BLOCKS];
    __shared__ double s_red[8], s_ab[8];
    __shared__ int s_ired[8];
    const int t = threadIdx.x;
    const int64_t idx = (int64_t)blockIdx.x * 256 + t, ic = idx < a.grid_n ? idx : a.grid_n - 1;
    const LoopWeightsPre pre = loop_weights_prefetch(a, blockIdx.x == 0);
    const double e_i = a.e[ic], x_i = a.x[ic];
    const uint8_t v_i = a.valid[ic];
    const int32_t nn_i = a.nn_idx[ic];
    MomIn in = moments_load(ic, a.poses_prop, (const double*)nullptr, (const float*)nullptr, labels);  // (w: below)
    const int64_t n = a.ctl_i[LOOP_I_N];
    int C = a.ctl_i[LOOP_I_NCL] + 1;
    C = C > LOOP_MAX_CLUSTERS ? LOOP_MAX_CLUSTERS : C;
    if ((int64_t)blockIdx.x * 256 >= n && blockIdx.x != 0) return;
    WCK(24);
    const LoopWeightsHead h = loop_weights_head(a, pre, n, s_sum, s_red, s_ired);
    WCK(25);
    const bool live = idx < n;
    double w = 0.0;
    if (live) {
        w = loop_weight_store(a, h, idx, e_i, x_i, v_i);
        if (h.drifted) {  // every particle back onto its codebook pose; the moments are taken of the re-projected set
            const float4* s4 = reinterpret_cast<const float4*>(a.cb_poses + (size_t)nn_i * 16);
            float4* d4 = reinterpret_cast<float4*>(a.poses_prop + (size_t)idx * 16);
            const float4 r0 = s4[0], r1 = s4[1], r2 = s4[2], r3 = s4[3];
            d4[0] = r0; d4[1] = r1; d4[2] = r2; d4[3] = r3;
            in.P[0] = r0.x; in.P[1] = r0.y; in.P[2] = r0.z; in.P[3] = r0.w;
            in.P[4] = r1.x; in.P[5] = r1.y; in.P[6] = r1.z; in.P[7] = r1.w;
            in.P[8] = r2.x; in.P[9] = r2.y; in.P[10] = r2.z; in.P[11] = r2.w;
        }
    }
    in.w = (double)(float)w;  // particles.weights.float() (:161)
    WCK(26);
    if ((int64_t)blockIdx.x * 256 < n) moments_accumulate(live, in, C, [](int c) { return (int64_t)(c - 1); }, part, s_w, skip);
    WCK(27);
    if (blockIdx.x == 0) loop_weights_finalise(a, h, pre, s_ab);
    WCK(28);
#ifdef MIDAS_ANNEAL_CLOCKS
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl_d[56 + 29] += 1.0;
#endif
}

// one 64-thread workgroup per cluster: blocks summed in order, then the closed forms
// rot_out (loop step): the normalised moment matrix goes there (10 doubles per cluster) and the rotation entries of the
// centre are left to whoever solves it (cluster_rotation_write, beside the annealing); nullptr: solved here
MD void cluster_finish_body(int nblocks, size_t nbs, int c, const double* __restrict__ part, float* __restrict__ centers,
                            float* __restrict__ stds, int64_t* __restrict__ counts, double* s_m, double* __restrict__ rot_out = nullptr) {
    const int t = threadIdx.x;
    // blocks in order (the sums' order is part of the arithmetic).  The partials of CB blocks are staged in LDS by all 64 threads
    // (independent loads in flight together), then moment t walks them - at N = 100 k (391 blocks) the direct form was
    // thirteen dependent trips by 36 threads: 66 us, the longest kernel of the loop frame
    constexpr int CB = 128;
    __shared__ double s_part[CB * CL_MOM];
    double r = 0.0;
    // (eighteen loads in flight per thread before the first LDS store: left as one load and one store per iteration the
    // compiler waits for each load in turn - 72 dependent trips, slower than the form this replaces)
    constexpr int LB = 18;
    const int NT = (int)blockDim.x;  // 64 (midas_cluster_centers) or 256 (loop step: a chunk is one batch of loads)
    const double* __restrict__ pc = part + (size_t)c * nbs * CL_MOM;  // the cluster's partials lie together, block after block
    // 256 threads: a chunk of CB blocks is ONE batch of loads, and the next chunk's batch is in flight while this one's chain is
    // walked (at N = 100k - 391 blocks, four chunks - the chunks' trips came one after the other: 22 us)
    const bool one_batch = NT * LB >= CB * CL_MOM;
    double e_mx = -INFINITY, e_mn = INFINITY, e_first_mx = 0.0, e_first_mn = 0.0;  // (second wave: the extrema, see below)
    double xn[LB];
    auto issue = [&](int b0n) {
        const int nbn = nblocks - b0n < CB ? nblocks - b0n : CB;
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int i = t + NT * j, ic = i < nbn * CL_MOM ? i : nbn * CL_MOM - 1;
            xn[j] = pc[(size_t)b0n * CL_MOM + ic];
        }
    };
    if (one_batch && nblocks > 0) issue(0);
    for (int b0 = 0; b0 < nblocks; b0 += CB) {
        const int nb = nblocks - b0 < CB ? nblocks - b0 : CB;
        if (one_batch) {
#pragma unroll
            for (int j = 0; j < LB; ++j)
                if (t + NT * j < nb * CL_MOM) s_part[t + NT * j] = xn[j];
        } else {
            for (int i0 = t; i0 < nb * CL_MOM; i0 += NT * LB) {
                double x[LB];
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    const int i = i0 + NT * j, ic = i < nb * CL_MOM ? i : nb * CL_MOM - 1;
                    x[j] = pc[(size_t)b0 * CL_MOM + ic];
                }
#pragma unroll
                for (int j = 0; j < LB; ++j)
                    if (i0 + NT * j < nb * CL_MOM) s_part[i0 + NT * j] = x[j];
            }
        }
        __syncthreads();
        if (one_batch && b0 + CB < nblocks) issue(b0 + CB);
        // sixteen LDS reads in flight, then the chain in block order.  The sums' chain is ONE dependent addition per block: the
        // two extrema (selects in the chain) are walked by lanes of their own where the workgroup has a second wave; whole batches
        // first, the remainder one by one (no guards in the chain).  With a read, a wait and three branches per block the walk was
        // 60 of the kernel's 65 us at 391 blocks.
        const bool two_waves = blockDim.x >= 128;
        const bool sums = t < CL_MOM && !(two_waves && (t == M_WMAX || t == M_WMIN));
        const bool ext = two_waves ? (t == 64 || t == 65) : (t == M_WMAX || t == M_WMIN);
        const int m = ext && two_waves ? (t == 64 ? M_WMAX : M_WMIN) : t;
        if (sums && !(ext && !two_waves)) {
            int b1 = 0;
            if (b0 == 0) { r = s_part[m]; b1 = 1; }
            for (; b1 + 16 <= nb; b1 += 16) {
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = s_part[(b1 + j) * CL_MOM + m];
#pragma unroll
                for (int j = 0; j < 16; ++j) r = r + x[j];
            }
            for (; b1 < nb; ++b1) r = r + s_part[b1 * CL_MOM + m];
        }
        if (two_waves) {
            // The extrema need no order: `x > r ? x : r` walked from the first block's value takes the largest value that is not NaN
            // (NaN never wins) unless the FIRST is NaN (then nothing replaces it) - the second wave's lanes take every 64th block
            // each and meet at the end; walked block after block by one lane the two selects a block were the longest chain of
            // the kernel at 391 blocks.
            if (t >= 64 && t < 128) {
                const int l = t - 64;
                if (b0 == 0) { e_first_mx = s_part[M_WMAX]; e_first_mn = s_part[M_WMIN]; }
                for (int b1 = l; b1 < nb; b1 += 64) {
                    const double x = s_part[b1 * CL_MOM + M_WMAX], y = s_part[b1 * CL_MOM + M_WMIN];
                    e_mx = x > e_mx ? x : e_mx;
                    e_mn = y < e_mn ? y : e_mn;
                }
            }
        } else if (ext) {
            const bool is_max = m == M_WMAX;
            int b1 = 0;
            if (b0 == 0) { r = s_part[m]; b1 = 1; }
            for (; b1 + 16 <= nb; b1 += 16) {
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = s_part[(b1 + j) * CL_MOM + m];
#pragma unroll
                for (int j = 0; j < 16; ++j) r = is_max ? (x[j] > r ? x[j] : r) : (x[j] < r ? x[j] : r);
            }
            for (; b1 < nb; ++b1) {
                const double x = s_part[b1 * CL_MOM + m];
                r = is_max ? (x > r ? x : r) : (x < r ? x : r);
            }
        }
        __syncthreads();
    }
    {
        const bool two_waves = blockDim.x >= 128;
        if (two_waves && t >= 64 && t < 128) {
            const double mx = wave_max_dpp(e_mx), mn = wave_min_dpp(e_mn);
            if (t == 64) {
                s_m[M_WMAX] = nblocks > 0 ? (e_first_mx != e_first_mx ? e_first_mx : mx) : 0.0;
                s_m[M_WMIN] = nblocks > 0 ? (e_first_mn != e_first_mn ? e_first_mn : mn) : 0.0;
            }
        } else if (t < CL_MOM && !(two_waves && (t == M_WMAX || t == M_WMIN))) s_m[t] = r;
    }
    __syncthreads();
    if (t != 0) return;
    cluster_close(s_m, centers + (size_t)c * 16, stds + (size_t)c * 3, counts ? counts + c : nullptr,
                  rot_out ? rot_out + (size_t)c * 10 : nullptr);
}

__global__ __launch_bounds__(64) void k_cluster_finish(int nblocks, int C, const double* __restrict__ part,
                                                       float* __restrict__ centers, float* __restrict__ stds,
                                                       int64_t* __restrict__ counts) {
    __shared__ double s_m[CL_MOM];
    cluster_finish_body(nblocks, (size_t)nblocks, blockIdx.x, part, centers, stds, counts, s_m);
}

// loop engine: cluster slot c = blockIdx.x of the LOOP_MAX_CLUSTERS launched; rows of label c - 1
__global__ __launch_bounds__(256) void k_loop_cluster_finish(const int32_t* __restrict__ ctl_i, const double* __restrict__ part,
                                                            float* __restrict__ centers, float* __restrict__ stds,
                                                            int64_t* __restrict__ counts, double* __restrict__ rot, int32_t nbs) {
    __shared__ double s_m[CL_MOM];
    const int64_t n = ctl_i[LOOP_I_N];
    int C = ctl_i[LOOP_I_NCL] + 1;
    C = C > LOOP_MAX_CLUSTERS ? LOOP_MAX_CLUSTERS : C;
    if ((int)blockIdx.x >= C) return;
    cluster_finish_body((int)((n + 255) / 256), (size_t)nbs, blockIdx.x, part, centers, stds, counts, s_m, rot);
}

static bool moments_skip() {  // MIDAS_MOMENTS_SKIP=0: every cluster summed by every workgroup (the parity test's other side)
    static const bool on = !(getenv("MIDAS_MOMENTS_SKIP") && atoi(getenv("MIDAS_MOMENTS_SKIP")) == 0);
    return on;
}

// rot: LOOP_MAX_CLUSTERS x 10 doubles - the moment matrices whose eigenproblem the annealing kernel's second workgroup solves
// (the decision only needs the translation spreads: the eigenvector runs beside the selection)
int launch_loop_cluster(midas_ctx* ctx, int64_t cap, const int32_t* ctl_i, const float* poses, const double* w64,
                        const int32_t* labels, double* part, float* centers, float* stds, int64_t* counts, double* rot,
                        const LoopWeightsArgs* weights) {
    if (weights)
        hipLaunchKernelGGL(k_loop_weights_moments, dim3((unsigned)ceil_div(cap, 256)), dim3(256), 0, ctx->stream, *weights, labels, part,
                           moments_skip());
    else
        hipLaunchKernelGGL(k_loop_cluster_moments, dim3((unsigned)ceil_div(cap, 256)), dim3(256), 0, ctx->stream, ctl_i, poses, w64,
                           labels, part, moments_skip());
    hipLaunchKernelGGL(k_loop_cluster_finish, dim3(LOOP_MAX_CLUSTERS), dim3(256), 0, ctx->stream, ctl_i, (const double*)part, centers,
                       stds, counts, rot, (int32_t)ceil_div(cap, 256));
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_cluster_centers(midas_ctx* ctx, int64_t N, const float* poses, const double* w64, const float* w32,
                           const int64_t* labels, int32_t C, const int64_t* label_values, float* centers, float* stds,
                           int64_t* counts) {
    const int nb = (int)ceil_div(N, 256);
    void* part;
    int rc = midas_scratch(ctx, (size_t)nb * C * CL_MOM * sizeof(double), &part);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cluster_moments, dim3((unsigned)nb), dim3(256), 0, ctx->stream, N, poses, w64, w32, labels, (int)C,
                       label_values, (double*)part, moments_skip());
    hipLaunchKernelGGL(k_cluster_finish, dim3((unsigned)C), dim3(64), 0, ctx->stream, nb, (int)C, (const double*)part, centers,
                       stds, counts);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(cluster, k_cluster_moments)

}  // namespace midas
