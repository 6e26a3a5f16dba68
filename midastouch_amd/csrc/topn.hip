// topn.hip - per-row top-n of a score matrix and the best pose error among them.
//
// eval/single_touch_test.py:35-73 (`top_n_error`): C = pairwise cosine similarity of the codebook's embeddings,
// diagonal set to 0, per row the n (= 25) best-scoring entries (np.argpartition) and the smallest pose distance among
// them.  The similarity rows come from the scoring kernels (midas_score: float64 GEMV per query, or midas_score_batch:
// all queries of a tile in one pass over the codebook on the matrix cores); this kernel does the selection and the
// distance in ONE pass over each row, without materialising the K x K matrix.
//
// One 256-thread workgroup per query row.  Candidates above a running threshold are appended to an LDS buffer; when the
// buffer fills it is sorted (bitonic, value descending, index ascending on ties - np.argpartition leaves ties
// unspecified) and cut to the n best, whose last value becomes the threshold.  After the first cut ~n/1024 of a tile
// passes, so a row of 50 k scores costs about three sorts.
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

constexpr int TN_CAP = 2048;   // candidate slots (power of two: bitonic network)
constexpr int TN_TILE = 1024;  // scores examined between two capacity checks (4 per thread)

MD bool tn_before(double va, int ia, double vb, int ib) { return va > vb || (va == vb && ia < ib); }

// sort the TN_CAP slots: value descending, index ascending; unused slots hold (-inf, INT_MAX)
MD void tn_sort(double* s_v, int* s_i) {
    const int t = threadIdx.x;
    for (int k = 2; k <= TN_CAP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int e = t; e < TN_CAP; e += 256) {
                const int p = e ^ j;
                if (p > e) {
                    const bool up = (e & k) == 0;  // this pair sorts "best first"
                    const double va = s_v[e], vb = s_v[p];
                    const int ia = s_i[e], ib = s_i[p];
                    const bool swap = up ? tn_before(vb, ib, va, ia) : tn_before(va, ia, vb, ib);
                    if (swap) { s_v[e] = vb; s_v[p] = va; s_i[e] = ib; s_i[p] = ia; }
                }
            }
            __syncthreads();
        }
}

// DOTS: the row is a float32 panel row of raw dot products (selfsim.hip; stride ld), turned into cosines here with the
// float64 row norms - (double)dot / (|E_self| |E_j|), the division midas_score_batch performs in its epilogue
template <bool DOTS>
__global__ __launch_bounds__(256) void k_topn_pose_error(int64_t K, const void* __restrict__ scores_, int64_t ld, const double* __restrict__ norms,
                                                         int64_t row0, int n, const double* __restrict__ feat, int d,
                                                         double* __restrict__ err_out, int32_t* __restrict__ idx_out) {
    __shared__ double s_v[TN_CAP];
    __shared__ int s_i[TN_CAP];
    __shared__ int s_cnt;
    __shared__ double s_thr;
    __shared__ double s_red[4];
    const int t = threadIdx.x;
    const int64_t row = blockIdx.x, self = row0 + row;
    const double* __restrict__ x = reinterpret_cast<const double*>(scores_) + row * ld;
    const float* __restrict__ xf = reinterpret_cast<const float*>(scores_) + row * ld;
    const double nself = DOTS ? norms[self] : 1.0;
    for (int e = t; e < TN_CAP; e += 256) { s_v[e] = -INFINITY; s_i[e] = 0x7fffffff; }
    if (t == 0) { s_cnt = 0; s_thr = -INFINITY; }
    __syncthreads();
    bool cut_once = false;
    for (int64_t base = 0; base < K; base += TN_TILE) {
        const double thr = s_thr;
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = base + k * 256 + t, jc = j < K ? j : K - 1;
            v[k] = DOTS ? (double)xf[jc] / (nself * norms[jc]) : x[jc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = base + k * 256 + t;
            double val = j == self ? 0.0 : v[k];  // np.fill_diagonal(C, 0) (:64)
            // before the first cut everything is a candidate (NaN scores are never selected)
            if (j < K && (cut_once ? val >= thr : val == val)) {
                const int slot = atomicAdd(&s_cnt, 1);
                s_v[slot] = val;
                s_i[slot] = (int)j;
            }
        }
        __syncthreads();
        const int cnt_now = s_cnt;
        __syncthreads();  // everybody has read the count before anybody appends again
        if (cnt_now > TN_CAP - TN_TILE || base + TN_TILE >= K) {  // the next tile might not fit / the row is done: cut to n
            tn_sort(s_v, s_i);
            const int keep = n < cnt_now ? n : cnt_now;
            for (int e = keep + t; e < TN_CAP; e += 256) { s_v[e] = -INFINITY; s_i[e] = 0x7fffffff; }
            __syncthreads();
            if (t == 0) { s_cnt = keep; s_thr = keep == n ? s_v[n - 1] : -INFINITY; }
            cut_once = keep == n;
            __syncthreads();
        }
    }
    // s_v / s_i[0 .. cnt): the best entries, best first.  Ties with the threshold value were kept as ">=" candidates and
    // resolved by the sort (smaller index first).
    const int cnt = s_cnt;
    double best = INFINITY;
    if (t < cnt) {
        const int j = s_i[t];
        double acc = 0.0;
        for (int c = 0; c < d; ++c) {
            const double df = feat[(int64_t)j * d + c] - feat[self * d + c];
            acc = acc + df * df;
        }
        best = __builtin_sqrt(acc);
        if (idx_out) idx_out[row * n + t] = j;
    } else if (t < n && idx_out) {
        idx_out[row * n + t] = -1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(best, o); best = u < best ? u : best; }
    if ((t & 63) == 0) s_red[t >> 6] = best;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < 4; ++i) best = s_red[i] < best ? s_red[i] : best;
        err_out[row] = best;
    }
}

int launch_topn_pose_error(midas_ctx* ctx, int32_t B, int64_t K, const double* scores, int64_t row0, int32_t n,
                           const double* feat, int32_t d, double* err_out, int32_t* idx_out) {
    hipLaunchKernelGGL(k_topn_pose_error<false>, dim3((unsigned)B), dim3(256), 0, ctx->stream, K, (const void*)scores, K, (const double*)nullptr, row0, (int)n,
                       feat, (int)d, err_out, idx_out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_topn_pose_error_dots(midas_ctx* ctx, int32_t B, int64_t K, const float* panel, int64_t ld, const double* norms, int64_t row0, int32_t n,
                                const double* feat, int32_t d, double* err_out, int32_t* idx_out) {
    hipLaunchKernelGGL(k_topn_pose_error<true>, dim3((unsigned)B), dim3(256), 0, ctx->stream, K, (const void*)panel, ld, norms, row0, (int)n, feat,
                       (int)d, err_out, idx_out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

}  // namespace midas
