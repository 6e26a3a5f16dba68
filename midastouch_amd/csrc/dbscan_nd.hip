// dbscan_nd.hip - exact DBSCAN of N points in 2 .. 6 dimensions, all pairs (the "logmap" method).
//
// modules/particle_filter.py:218-223: cluster_particles(method="logmap") hands the 6-d SE(3) logarithms of the particles to
// sklearn's DBSCAN (eps, min_samples = N / 5; sklearn 1.7: KDTree on float64 copies, neighbour <=> sum_k (x_k - y_k)^2 <= eps^2
// with the squares added in coordinate order).  The grid of dbscan.hip (cells of side eps / sqrt 3 whose points are all mutual
// neighbours) does not carry over to six dimensions - a rotation vector's range is 600 eps at the reference's eps = 1e-2 - and the
// method is not on the filter's loop (filter/filter.py:183 clusters the translations), so this is the plain form: every pass
// compares all pairs, a 256-point tile of the columns in LDS per step (every thread reads the SAME column point: a broadcast),
// float64 differences, squares and sums in coordinate order without contraction - sklearn's predicate.
//   count   core_i = #{j : |x_i - x_j|^2 <= eps^2} >= min_samples (the point itself counts, as in sklearn)
//   spread  comp_i = min(comp_j : j core, within eps of the core point i), from comp_i = i, with pointer jumping, until nothing
//           changes: comp_i = the smallest index of i's density-connected set of core points = the core point sklearn's scan
//           (dbscan_inner, index order) starts that cluster from - so clusters numbered by ascending root are sklearn's numbers
//   label   a core point: its cluster; any other point: the SMALLEST cluster number among the core points within eps (sklearn
//           expands the clusters in number order and a labelled border point is never relabelled), none: noise (-1)
// 10^10 pairs per pass at N = 100 k: a few milliseconds each; sklearn needs minutes on the host.
#include "midas_internal.hpp"

namespace midas {
namespace {

constexpr int DBN_T = 256;

template <int DIM>
struct DbnTile {
    double p[DIM][DBN_T];
    int32_t v[DBN_T];  // per column: count -> unused; spread -> comp (or -1: not core); label -> cluster (or -1)
};

template <int DIM>
__device__ __forceinline__ void dbn_load_tile(DbnTile<DIM>& s, const double* __restrict__ pts, const int32_t* __restrict__ val, int64_t N, int64_t j0) {
    const int t = threadIdx.x;
    const int64_t j = j0 + t;
    if (j < N) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) s.p[k][t] = pts[j * DIM + k];
        s.v[t] = val ? val[j] : 0;
    } else {
#pragma unroll
        for (int k = 0; k < DIM; ++k) s.p[k][t] = NAN;  // never within eps of anything
        s.v[t] = -1;
    }
}

template <int DIM>
__device__ __forceinline__ bool dbn_near(const double* q, const DbnTile<DIM>& s, int j, double r2) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double d = q[k] - s.p[k][j];
        acc = acc + d * d;  // (compiled without contraction: a product and a sum, like the host's loop)
    }
    return acc <= r2;
}

// MODE 0: neighbour counts -> core flags.  MODE 1: one spread step over the core points.  MODE 2: labels.
template <int DIM, int MODE>
__global__ __launch_bounds__(DBN_T) void k_dbn_pass(int64_t N, const double* __restrict__ pts, double r2, int64_t min_samples,
                                                   const int32_t* __restrict__ col_val, int32_t* __restrict__ out, int32_t* __restrict__ changed) {
    __shared__ DbnTile<DIM> s;
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * DBN_T + t;
    const bool live = i < N;
    double q[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) q[k] = live ? pts[i * DIM + k] : NAN;
    const int32_t own = (MODE != 0 && live) ? col_val[i] : -1;
    // MODE 1: rows that are not core points have nothing to do; MODE 2: core points keep their cluster
    const bool active = MODE == 0 ? live : MODE == 1 ? (live && own >= 0) : (live && own < 0);
    int64_t cnt = 0;
    int32_t best = MODE == 1 ? own : 0x7fffffff;
    const bool any_active = __syncthreads_or(active ? 1 : 0) != 0;  // (uniform) a workgroup without active rows skips the columns
    for (int64_t j0 = 0; any_active && j0 < N; j0 += DBN_T) {
        __syncthreads();
        dbn_load_tile<DIM>(s, pts, MODE == 0 ? nullptr : col_val, N, j0);
        __syncthreads();
        if (MODE == 0) {
#pragma unroll 4
            for (int j = 0; j < DBN_T; ++j) cnt += dbn_near<DIM>(q, s, j, r2) ? 1 : 0;
        } else {
            for (int j = 0; j < DBN_T; ++j) {
                const int32_t v = s.v[j];
                if (v < 0) continue;  // (uniform: every thread looks at the same column) not a core point
                if (active && v < best && dbn_near<DIM>(q, s, j, r2)) best = v;
            }
        }
    }
    if (!live) return;
    if (MODE == 0) {
        out[i] = cnt >= min_samples ? (int32_t)i : -1;  // comp_i = i for a core point
    } else if (MODE == 1) {
        if (own >= 0) {
            out[i] = best;
            if (best != own) atomicOr(changed, 1);
        } else {
            out[i] = -1;
        }
    } else {
        out[i] = own >= 0 ? own : (best == 0x7fffffff ? -1 : best);
    }
}

// comp_i <- comp_{comp_i} (roots point at themselves; values only ever decrease towards the root, so in place is safe)
__global__ __launch_bounds__(256) void k_dbn_jump(int64_t N, int32_t* __restrict__ comp, int32_t* __restrict__ changed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t c = comp[i];
    if (c < 0) return;
    const int32_t r = comp[c];
    if (r >= 0 && r < c) { comp[i] = r; atomicOr(changed, 1); }
}

// roots (comp_i == i) numbered in index order: one workgroup walks the array (N / 1024 steps; the numbers are a prefix count)
__global__ __launch_bounds__(1024) void k_dbn_number(int64_t N, const int32_t* __restrict__ comp, int32_t* __restrict__ rootnum, int32_t* __restrict__ ncl) {
    __shared__ int s_w[16];
    __shared__ int s_run;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) s_run = 0;
    __syncthreads();
    for (int64_t b = 0; b < N; b += 1024) {
        const int64_t i = b + t;
        const bool root = i < N && comp[i] == (int32_t)i;
        const unsigned long long m = __ballot(root);
        if (lane == 0) s_w[w] = __popcll(m);
        __syncthreads();
        int before = s_run;
        for (int k = 0; k < w; ++k) before += s_w[k];
        if (root) rootnum[i] = before + __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (t == 0) { int tot = 0; for (int k = 0; k < 16; ++k) tot += s_w[k]; s_run += tot; }
        __syncthreads();
    }
    if (t == 0) ncl[0] = s_run;
}

// cluster of every core point (its root's number), -1 elsewhere
__global__ __launch_bounds__(256) void k_dbn_core_label(int64_t N, const int32_t* __restrict__ comp, const int32_t* __restrict__ rootnum, int32_t* __restrict__ lab) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t c = comp[i];
    lab[i] = c >= 0 ? rootnum[c] : -1;
}

template <int DIM>
int dbscan_nd(midas_ctx* ctx, int64_t N, const double* pts, double eps, int64_t min_samples, int32_t* labels, int32_t* info) {
    hipStream_t st = ctx->stream;
    void* p;
    int rc;
    if ((rc = midas_scratch(ctx, (size_t)N * sizeof(int32_t), &p))) return rc;
    int32_t* comp_a = (int32_t*)p;
    if ((rc = midas_scratch(ctx, (size_t)N * sizeof(int32_t), &p))) return rc;
    int32_t* comp_b = (int32_t*)p;
    if ((rc = midas_scratch(ctx, (size_t)N * sizeof(int32_t), &p))) return rc;
    int32_t* rootnum = (int32_t*)p;
    if ((rc = midas_scratch(ctx, 64, &p))) return rc;
    int32_t* changed = (int32_t*)p;
    const unsigned g = (unsigned)ceil_div(N, DBN_T), g256 = (unsigned)ceil_div(N, 256);
    const double r2 = eps * eps;
    hipLaunchKernelGGL((k_dbn_pass<DIM, 0>), dim3(g), dim3(DBN_T), 0, st, N, pts, r2, min_samples, (const int32_t*)nullptr, comp_a, changed);
    int32_t iters = 0, flag = 1;
    const int32_t max_iters = 1 << 16;
    while (flag && iters < max_iters) {
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(changed, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL((k_dbn_pass<DIM, 1>), dim3(g), dim3(DBN_T), 0, st, N, pts, r2, min_samples, (const int32_t*)comp_a, comp_b, changed);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_dbn_jump, dim3(g256), dim3(256), 0, st, N, comp_b, changed);
        MIDAS_HIP_CHECK(ctx, hipGetLastError());
        MIDAS_HIP_CHECK(ctx, hipMemcpyAsync(&flag, changed, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(st));  // (not on the filter's loop: one look per spread step)
        int32_t* tmp = comp_a; comp_a = comp_b; comp_b = tmp;
        ++iters;
    }
    hipLaunchKernelGGL(k_dbn_number, dim3(1), dim3(1024), 0, st, N, (const int32_t*)comp_a, rootnum, info);
    hipLaunchKernelGGL(k_dbn_core_label, dim3(g256), dim3(256), 0, st, N, (const int32_t*)comp_a, (const int32_t*)rootnum, comp_b);
    hipLaunchKernelGGL((k_dbn_pass<DIM, 2>), dim3(g), dim3(DBN_T), 0, st, N, pts, r2, min_samples, (const int32_t*)comp_b, labels, changed);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    const int32_t it = flag ? -1 : iters;  // -1: the spread did not settle within the bound (labels are not final)
    MIDAS_HIP_CHECK(ctx, hipMemcpyAsync(info + 1, &it, sizeof(int32_t), hipMemcpyHostToDevice, st));
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(st));  // `it` lives on this frame
    return MIDAS_OK;
}

}  // namespace

int launch_dbscan_points(midas_ctx* ctx, int64_t N, int32_t dim, const double* pts, double eps, int64_t min_samples, int32_t* labels, int32_t* info) {
    if (min_samples < 0) min_samples = N / 5;
    switch (dim) {
        case 2: return dbscan_nd<2>(ctx, N, pts, eps, min_samples, labels, info);
        case 3: return dbscan_nd<3>(ctx, N, pts, eps, min_samples, labels, info);
        case 4: return dbscan_nd<4>(ctx, N, pts, eps, min_samples, labels, info);
        case 5: return dbscan_nd<5>(ctx, N, pts, eps, min_samples, labels, info);
        case 6: return dbscan_nd<6>(ctx, N, pts, eps, min_samples, labels, info);
        default: return midas_set_error(ctx, MIDAS_ERR_INVALID, "dim", "midas_dbscan_points takes 2 .. 6 dimensions");
    }
}

MIDAS_WARM_TU(dbscan_nd, k_dbn_jump)

}  // namespace midas
