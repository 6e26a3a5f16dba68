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
template <int NT>
MD void tn_sort(double* s_v, int* s_i) {
    const int t = threadIdx.x;
    for (int k = 2; k <= TN_CAP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int e = t; e < TN_CAP; e += NT) {
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

struct TnShared {
    double v[TN_CAP];
    int i[TN_CAP];
    int cnt;
    double thr;
    double red[32];
};

// Best pose error among the cnt (<= n) selected entries sh.i[0 .. cnt) (best first) and the index row.
template <int NT>
MD void tn_finish(TnShared& sh, int cnt, int64_t row, int64_t self, int n, const double* __restrict__ feat, int d,
                  double* __restrict__ err_out, int32_t* __restrict__ idx_out) {
    const int t = threadIdx.x;
    double best = INFINITY;
    if (t < cnt) {
        const int j = sh.i[t];
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
    if ((t & 63) == 0) sh.red[t >> 6] = best;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < NT / 64; ++i) best = sh.red[i] < best ? sh.red[i] : best;
        err_out[row] = best;
    }
}

// One pass over a row of any length by a workgroup of NT threads.  DOTS: the row is a float32 panel row of raw dot products
// (selfsim.hip; stride ld), turned into cosines here with the float64 row norms - (double)dot / (|E_self| |E_j|), the division
// midas_score_batch performs in its epilogue
template <bool DOTS, int NT>
MD void tn_row_stream(TnShared& sh, int64_t K, const void* __restrict__ scores_, int64_t ld, const double* __restrict__ norms, int64_t row0,
                      int n, const double* __restrict__ feat, int d, double* __restrict__ err_out, int32_t* __restrict__ idx_out) {
    constexpr int VPT = TN_TILE / NT;
    static_assert(VPT >= 1 && VPT * NT == TN_TILE, "a tile is whole rounds of the workgroup");
    const int t = threadIdx.x;
    const int64_t row = blockIdx.x, self = row0 + row;
    const double* __restrict__ x = reinterpret_cast<const double*>(scores_) + row * ld;
    const float* __restrict__ xf = reinterpret_cast<const float*>(scores_) + row * ld;
    const double nself = DOTS ? norms[self] : 1.0;
    for (int e = t; e < TN_CAP; e += NT) { sh.v[e] = -INFINITY; sh.i[e] = 0x7fffffff; }
    if (t == 0) { sh.cnt = 0; sh.thr = -INFINITY; }
    __syncthreads();
    bool cut_once = false;
    for (int64_t base = 0; base < K; base += TN_TILE) {
        const double thr = sh.thr;
        double v[VPT];
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int64_t j = base + k * NT + t, jc = j < K ? j : K - 1;
            v[k] = DOTS ? (double)xf[jc] / (nself * norms[jc]) : x[jc];
        }
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int64_t j = base + k * NT + t;
            double val = j == self ? 0.0 : v[k];  // np.fill_diagonal(C, 0) (:64)
            // before the first cut everything is a candidate (NaN scores are never selected)
            if (j < K && (cut_once ? val >= thr : val == val)) {
                const int slot = atomicAdd(&sh.cnt, 1);
                sh.v[slot] = val;
                sh.i[slot] = (int)j;
            }
        }
        __syncthreads();
        const int cnt_now = sh.cnt;
        __syncthreads();  // everybody has read the count before anybody appends again
        if (cnt_now > TN_CAP - TN_TILE || base + TN_TILE >= K) {  // the next tile might not fit / the row is done: cut to n
            tn_sort<NT>(sh.v, sh.i);
            const int keep = n < cnt_now ? n : cnt_now;
            for (int e = keep + t; e < TN_CAP; e += NT) { sh.v[e] = -INFINITY; sh.i[e] = 0x7fffffff; }
            __syncthreads();
            if (t == 0) { sh.cnt = keep; sh.thr = keep == n ? sh.v[n - 1] : -INFINITY; }
            cut_once = keep == n;
            __syncthreads();
        }
    }
    // sh.v / sh.i[0 .. cnt): the best entries, best first.  Ties with the threshold value were kept as ">=" candidates and
    // resolved by the sort (smaller index first).
    tn_finish<NT>(sh, sh.cnt, row, self, n, feat, d, err_out, idx_out);
}

template <bool DOTS>
__global__ __launch_bounds__(256) void k_topn_pose_error(int64_t K, const void* __restrict__ scores_, int64_t ld, const double* __restrict__ norms,
                                                         int64_t row0, int n, const double* __restrict__ feat, int d,
                                                         double* __restrict__ err_out, int32_t* __restrict__ idx_out) {
    __shared__ TnShared sh;
    tn_row_stream<DOTS, 256>(sh, K, scores_, ld, norms, row0, n, feat, d, err_out, idx_out);
}

// ---- register-resident selection of a panel row (the GEMM path of top_n_error) -------------------------------------------------
// The streaming kernel above walks a row tile by tile (a barrier pair per 1024 scores, a bitonic sort of 2048 slots per cut):
// 1.1 ms per panel of 4096 rows x 50 k float32 dots = 0.74 TB/s, as long as the GEMM that wrote the panel.  Here a workgroup of
// 1024 threads holds the WHOLE row's screen values in registers (NV4 float4 per thread, the loads in flight together), screens in float32 and
// decides in float64:
//   1. s_j = dot_j * rinv_j (rinv = float32 of 1 / |E_j|; |E_self| > 0 scales a row uniformly and is left out), diagonal 0,
//      NaN -> -inf;
//   2. the maxima of the 32 half-waves: their n-th largest T has >= n scores at or above it, and about 32 ln(32 / (32 - n))
//      scores above it in expectation (49 at n = 25) - one rank count in one wave, no sort;
//   3. every score >= T - 1e-6 |T| is a candidate (the float32 screen is within 1.2e-7 relative of the float64 value: anything
//      that can be among the n best in float64 passes), appended to LDS with one atomic per wave and round;
//   4. candidates get the float64 value the streaming kernel computes, (double)dot / (|E_self| |E_j|), and their rank among the
//      candidates by (value descending, index ascending) - ranks below n are the answer, identical to the streaming kernel's.
// More candidates than slots (masses of equal scores) or n > TF_MAXN: the row goes through tn_row_stream<true, 1024> instead.
constexpr int TF_NT = 1024, TF_CAP = 1024, TF_MAXN = 28;
static_assert(TF_CAP <= TN_CAP, "the candidate arrays are the streaming buffers");

template <int NV4>
__global__ __launch_bounds__(TF_NT) void k_topn_dots_rows(int64_t K, const float* __restrict__ panel, int64_t ld, const double* __restrict__ norms,
                                                          const float* __restrict__ rinv, int64_t row0, int n,
                                                          const double* __restrict__ feat, int d, double* __restrict__ err_out,
                                                          int32_t* __restrict__ idx_out) {
    __shared__ TnShared sh;
    __shared__ float s_gmax[32];
    __shared__ float s_cut;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t row = blockIdx.x, self = row0 + row;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(panel + row * ld);
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rinv);
    // the float32 screen of column jj
    const int Ki = (int)K, selfi = (int)self;  // K <= 65 536 here: 32-bit column numbers
    auto scr = [&](float dot, float r, int jj) {
        float s = dot * r;
        s = jj == selfi ? 0.0f : s;
        return (jj < Ki && s == s) ? s : -INFINITY;
    };
    float4 sv[NV4];  // the whole row's screen values: thread t holds columns 4 (k 1024 + t) .. + 3 (ld is a multiple of 128: in bounds)
    float mx = -INFINITY;
    int tt = t;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int j4 = k * TF_NT + tt, j = 4 * j4;
        const bool in = j < Ki;
        const float4 dv = in ? x4[j4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ri = in ? r4[j4] : make_float4(0.f, 0.f, 0.f, 0.f);
        sv[k] = make_float4(scr(dv.x, ri.x, j), scr(dv.y, ri.y, j + 1), scr(dv.z, ri.z, j + 2), scr(dv.w, ri.w, j + 3));
        mx = fmaxf(mx, fmaxf(fmaxf(sv[k].x, sv[k].y), fmaxf(sv[k].z, sv[k].w)));
        // loads in flight four columns-of-four at a time: all 2 NV4 of them above the arithmetic would need the registers twice
        if (k % 4 == 3) asm volatile("" : "+v"(tt) : "v"(sv[k].x));  // the next four's addresses wait for this value
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((lane & 31) == 0) s_gmax[t >> 5] = mx;
    if (t == 0) sh.cnt = 0;
    __syncthreads();
    if (wave == 0) {  // the n-th largest of the 32 half-wave maxima (rank by value, then by group number)
        const float g = s_gmax[lane & 31];
        int rank = 0;
        for (int m = 0; m < 32; ++m) {
            const float gm = __shfl(g, m);
            rank += (gm > g || (gm == g && m < (lane & 31))) ? 1 : 0;
        }
        if (lane < 32 && rank == n - 1) s_cut = g > -INFINITY ? g - 1e-6f * fabsf(g) : -INFINITY;
    }
    __syncthreads();
    const float cut = s_cut;
    const double nself = norms[self];
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int j = 4 * (k * TF_NT + t);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s = c == 0 ? sv[k].x : c == 1 ? sv[k].y : c == 2 ? sv[k].z : sv[k].w;
            const bool cand = s >= cut && s > -INFINITY;
            const unsigned long long b = __ballot(cand);
            if (b) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh.cnt, __popcll(b));
                base = __shfl(base, 0);
                const int slot = base + __popcll(b & ((1ull << lane) - 1ull));
                if (cand && slot < TF_CAP) sh.i[slot] = j + c;  // candidate columns; sh.v: their float64 values below
            }
        }
    }
    __syncthreads();
    const int cnt = sh.cnt;
    if (cnt > TF_CAP || !(nself > 0.0)) {  // uniform: masses of equal scores (the streaming form has no capacity to exceed), or a row
                                           // whose own norm is not positive (its cosines are NaN / signed infinities: no common scale)
        __syncthreads();
        tn_row_stream<true, TF_NT>(sh, K, panel, ld, norms, row0, n, feat, d, err_out, idx_out);
        return;
    }
    int myj = 0x7fffffff;
    double myv = -INFINITY;
    if (t < cnt) {
        myj = sh.i[t];
        myv = myj == self ? 0.0 : (double)panel[row * ld + myj] / (nself * norms[myj]);  // the dot again: one scattered load per candidate
        sh.v[t] = myv;
    }
    __syncthreads();
    int rank = 0;
    if (t < cnt)
        for (int m = 0; m < cnt; ++m) rank += tn_before(sh.v[m], sh.i[m], myv, myj) ? 1 : 0;
    __syncthreads();  // every rank is counted before the slots are reordered
    const int keep = cnt < n ? cnt : n;
    if (t < cnt && rank < n) { sh.i[rank] = myj; }
    __syncthreads();
    tn_finish<TF_NT>(sh, keep, row, self, n, feat, d, err_out, idx_out);
}

__global__ __launch_bounds__(256) void k_topn_rinv(int64_t K, int64_t ld, const double* __restrict__ norms, float* __restrict__ rinv) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < ld) rinv[j] = j < K ? (float)(1.0 / norms[j]) : 0.0f;
}

int launch_topn_pose_error(midas_ctx* ctx, int32_t B, int64_t K, const double* scores, int64_t row0, int32_t n,
                           const double* feat, int32_t d, double* err_out, int32_t* idx_out) {
    hipLaunchKernelGGL(k_topn_pose_error<false>, dim3((unsigned)B), dim3(256), 0, ctx->stream, K, (const void*)scores, K, (const double*)nullptr, row0, (int)n,
                       feat, (int)d, err_out, idx_out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_topn_rinv(midas_ctx* ctx, int64_t K, int64_t ld, const double* norms, float* rinv) {
    hipLaunchKernelGGL(k_topn_rinv, dim3((unsigned)ceil_div(ld, 256)), dim3(256), 0, ctx->stream, K, ld, norms, rinv);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// rinv (ld floats, launch_topn_rinv) selects the register-resident kernel where the row fits (K <= 65 536, n <= 28); NULL or a
// larger problem: the streaming kernel
int launch_topn_pose_error_dots(midas_ctx* ctx, int32_t B, int64_t K, const float* panel, int64_t ld, const double* norms, const float* rinv,
                                int64_t row0, int32_t n, const double* feat, int32_t d, double* err_out, int32_t* idx_out) {
    const bool fast = rinv && n <= TF_MAXN && K <= 16 * 4096 && ld % 4 == 0 && (uintptr_t)panel % 16 == 0 && (uintptr_t)rinv % 16 == 0;
#define MIDAS_TF(NV4)                                                                                                              \
    hipLaunchKernelGGL(k_topn_dots_rows<NV4>, dim3((unsigned)B), dim3(TF_NT), 0, ctx->stream, K, panel, ld, norms, rinv, row0, (int)n, \
                       feat, (int)d, err_out, idx_out)
    if (!fast)
        hipLaunchKernelGGL(k_topn_pose_error<true>, dim3((unsigned)B), dim3(256), 0, ctx->stream, K, (const void*)panel, ld, norms, row0, (int)n,
                           feat, (int)d, err_out, idx_out);
    else if (K <= 4 * 4096) MIDAS_TF(4);
    else if (K <= 8 * 4096) MIDAS_TF(8);
    else if (K <= 13 * 4096) MIDAS_TF(13);
    else MIDAS_TF(16);
#undef MIDAS_TF
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(topn, k_topn_rinv)

}  // namespace midas
