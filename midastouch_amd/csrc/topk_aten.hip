// topk_aten.hip - particle_filter.annealing's `torch.topk` with the choices ATen's CPU kernel makes inside a tie
// (modules/particle_filter.py:433-441; MIDAS_TOPK_TIES_ATEN_CPU).
//
// Which members of a tie survive - and in which order the duplicates are appended - is not a property of the weights but of
// the algorithm the reference's torch build runs: ATen `topk_impl_loop` (TopKImpl.h) fills a queue of (value, index) pairs and
// calls libstdc++'s std::partial_sort when k * 64 <= n, else std::nth_element (+ std::sort of the first k - 1 when sorted).
// Ties are the normal case here (particles that share a codebook entry share a weight, pruned particles all weigh 0), so
// the seeded / host-draw mode that replays the reference's particle set has to walk the same algorithm.  The control flow of
// those algorithms is sequential by definition; what one wave can do in parallel without changing a single move is done so:
//
//   * unguarded Hoare partition (nth_element, sort): the scan from the left stops at the positions whose value is not
//     below the pivot, the scan from the right at those not above it, and every swap happens behind both pointers - so the
//     m-th swap exchanges the m-th "left stopper" with the m-th "right stopper" of the ORIGINAL segment while the former
//     lies left of the latter.  Both lists come from ballots over 64-element tiles, the swaps are independent.
//   * heap select (partial_sort): an element enters the heap only if it beats the top; a tile of 64 candidates is tested
//     against the current top with one ballot, the winners are sifted in one at a time (in LDS while k <= 3968).
//   * the leaves of sort (<= 16 elements, then one final insertion sort = a stable sort of every leaf, since all of a leaf
//     is <= all of the next): a rank sort across lanes.
//   * the depth-limit fallbacks (heap select / heap sort on a segment) run as written; adversarial inputs reach them
//     (tests/test_gpu_topk_aten.py builds such inputs with the oracle's adversary).
//
// Spec: oracle/aten_topk.c, pinned against torch.topk (tests/test_aten_topk.py).  One wave per call; 0.3 - 5 ms at
// N = 100k: this is the mode that follows the reference move for move, not the fast one (loop.hip's radix select, ties by index).
#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

#define LAUNCH_CHECK(ctx) MIDAS_HIP_CHECK(ctx, hipGetLastError())

struct alignas(16) TkPair { double v; int32_t i; int32_t pad; };

constexpr int TK_HEAP_LDS = 3968;   // heap of the partial-sort path kept in LDS up to this k (62 KB)
constexpr int TK_POS_LDS = 7936;    // stopper lists of a partition kept in LDS up to this segment length (2 x 31 KB)
constexpr int TK_STACK = 96;        // pending segments of sort (depth limit 2 lg n <= 62)

template <bool LARGEST>
MD bool tk_comp(double x, double y) {
    return LARGEST ? ((x != x && y == y) || x > y) : ((x == x && y != y) || x < y);
}

MD int tk_lane() { return (int)__lane_id(); }
MD int tk_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
MD void tk_sync() { __syncthreads(); }  // one-wave workgroup: orders this wave's memory operations at workgroup scope
MD unsigned long long tk_lt_mask() { return (1ull << tk_lane()) - 1ull; }

// ---- bits/stl_heap.h, executed by every lane on the same (uniform) values ----------------------------------------------
template <bool LARGEST>
MD void tk_push_heap(TkPair* h, int hole, int top, TkPair value) {
    int parent = (hole - 1) / 2;
    while (hole > top && tk_comp<LARGEST>(h[parent].v, value.v)) {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = value;
}

template <bool LARGEST>
MD void tk_adjust_heap(TkPair* h, int hole, int len, TkPair value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (tk_comp<LARGEST>(h[child].v, h[child - 1].v)) child--;
        h[hole] = h[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    tk_push_heap<LARGEST>(h, hole, top, value);
}

template <bool LARGEST>
MD void tk_make_heap(TkPair* h, int len) {
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        const TkPair v = h[parent];
        tk_adjust_heap<LARGEST>(h, parent, len, v);
        if (parent == 0) return;
    }
}

template <bool LARGEST>
MD void tk_sort_heap(TkPair* h, int len) {
    while (len > 1) {
        --len;
        const TkPair v = h[len];
        h[len] = h[0];
        tk_adjust_heap<LARGEST>(h, 0, len, v);
    }
}

// __heap_select(h, h + len, rest_end): the heap is h[0, len) (LDS or in place), the candidates q[from, to)
template <bool LARGEST>
MD void tk_heap_select(TkPair* h, int len, TkPair* q, int from, int to) {
    tk_make_heap<LARGEST>(h, len);
    tk_sync();
    const int lane = tk_lane();
    for (int base0 = from; base0 < to; base0 += 256) {
        TkPair cs[4];  // four tiles of 64 candidates requested together (one round trip instead of four); looked at in order
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = base0 + 64 * u + lane;
            cs[u] = q[x < to ? x : to - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = base0 + 64 * u + lane;
            const bool in = x < to;
            const TkPair c = cs[u];
            double top = h[0].v;
            unsigned long long m = __ballot(in && tk_comp<LARGEST>(c.v, top));
            while (m) {
                const int b = __builtin_ctzll(m);
                TkPair val;  // __pop_heap(first, middle, i): value = *i; *i = *first; adjust(first, 0, len, value)
                val.v = __shfl(c.v, b);
                val.i = __shfl(c.i, b);
                val.pad = 0;
                if (lane == b) q[x] = h[0];
                tk_adjust_heap<LARGEST>(h, 0, len, val);
                top = h[0].v;
                m = __ballot(in && lane > b && tk_comp<LARGEST>(c.v, top));
            }
        }
    }
    tk_sync();
}

// ---- bits/stl_algo.h -----------------------------------------------------------------------------------------------------
MD void tk_swap(TkPair* q, int a, int b) {  // (uniform)
    const TkPair t = q[a];
    q[a] = q[b];
    q[b] = t;
}

template <bool LARGEST>
MD void tk_median_to_first(TkPair* q, int result, int a, int b, int c) {
    const double va = q[a].v, vb = q[b].v, vc = q[c].v;
    int pick;
    if (tk_comp<LARGEST>(va, vb)) {
        if (tk_comp<LARGEST>(vb, vc)) pick = b;
        else if (tk_comp<LARGEST>(va, vc)) pick = c;
        else pick = a;
    } else if (tk_comp<LARGEST>(va, vc)) pick = a;
    else if (tk_comp<LARGEST>(vb, vc)) pick = c;
    else pick = b;
    tk_swap(q, result, pick);
}

// __unguarded_partition_pivot(first, last): returns the cut.  lpos / rpos: room for last - first positions each.
template <bool LARGEST>
MD int tk_partition_pivot(TkPair* q, int first, int last, int* lpos, int* rpos) {
    const int lane = tk_lane();
    tk_median_to_first<LARGEST>(q, first, first + 1, first + (last - first) / 2, last - 1);
    tk_sync();
    const double p = q[first].v;
    // stoppers of the scan from the left, ascending, over [first + 1, last)
    int cl = 0;
    for (int base = first + 1; base < last; base += 256) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = base + 64 * u + lane;
            v[u] = q[x < last ? x : last - 1].v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = base + 64 * u + lane;
            const bool f = x < last && !tk_comp<LARGEST>(v[u], p);
            const unsigned long long m = __ballot(f);
            if (f) lpos[cl + __popcll(m & tk_lt_mask())] = x;
            cl += __popcll(m);
        }
    }
    // stoppers of the scan from the right, descending, over [first, last) (the pivot itself ends it)
    int cr = 0;
    for (int top = last; top > first; top -= 256) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = top - 1 - 64 * u - lane;
            v[u] = q[x >= first ? x : first].v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = top - 1 - 64 * u - lane;
            const bool f = x >= first && !tk_comp<LARGEST>(p, v[u]);
            const unsigned long long m = __ballot(f);
            if (f) rpos[cr + __popcll(m & tk_lt_mask())] = x;
            cr += __popcll(m);
        }
    }
    tk_sync();
    // swaps: while the m-th left stopper lies left of the m-th right stopper
    const int both = cl < cr ? cl : cr;
    int s = 0;
    for (int base = 0; base < both; base += 64) {
        const int r = base + lane;
        const bool ok = r < both && lpos[r] < rpos[r];
        const unsigned long long m = __ballot(ok);
        s += __popcll(m);
        if (ok) {
            const int a = lpos[r], b = rpos[r];
            const TkPair ta = q[a], tb = q[b];
            q[a] = tb;
            q[b] = ta;
        }
        if (m != ~0ull) break;
    }
    int cut = s > 0 ? rpos[s - 1] : last;
    if (s < cl) { const int l = lpos[s]; cut = l < cut ? l : cut; }
    tk_sync();
    return tk_uni(cut);
}

// a stable sort of q[first, last), last - first <= 64 (insertion sort's result): rank across lanes
template <bool LARGEST>
MD void tk_stable_small(TkPair* q, int first, int last) {
    const int m = last - first, lane = tk_lane();
    if (m < 2) return;
    const TkPair mine = q[first + (lane < m ? lane : 0)];
    int rank = 0;
    for (int j = 0; j < m; ++j) {
        const double vj = __shfl(mine.v, j);
        rank += (tk_comp<LARGEST>(vj, mine.v) || (!tk_comp<LARGEST>(mine.v, vj) && j < lane)) ? 1 : 0;
    }
    tk_sync();
    if (lane < m) q[first + rank] = mine;
    tk_sync();
}

struct TkShared {
    union {
        TkPair heap[TK_HEAP_LDS];
        struct { int l[TK_POS_LDS]; int r[TK_POS_LDS]; } pos;
    };
};

// __introselect(first, nth, last, depth_limit)
template <bool LARGEST>
MD void tk_introselect(TkPair* q, int first, int nth, int last, int depth, int* lpos_g, int* rpos_g, TkShared& sh, int* fallbacks) {
    while (last - first > 3) {
        if (depth == 0) {
            tk_heap_select<LARGEST>(q + first, nth + 1 - first, q, nth + 1, last);
            tk_swap(q, first, nth);
            tk_sync();
            if (fallbacks) *fallbacks += 1;
            return;
        }
        --depth;
        const bool lds = last - first <= TK_POS_LDS;
        const int cut = tk_partition_pivot<LARGEST>(q, first, last, lds ? sh.pos.l : lpos_g, lds ? sh.pos.r : rpos_g);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    tk_stable_small<LARGEST>(q, first, last);
}

// std::sort(q + first, q + last): __introsort_loop with an explicit stack of pending right halves (they are independent,
// any order gives the same array), leaves finished on the spot
template <bool LARGEST>
MD void tk_sort(TkPair* q, int first0, int last0, int* lpos_g, int* rpos_g, TkShared& sh, int* stack, int* fallbacks) {
    if (last0 - first0 < 2) return;
    int depth0 = 0;
    for (int n = last0 - first0; n > 1; n >>= 1) ++depth0;
    int sp = 0;
    int first = first0, last = last0, depth = 2 * depth0;
    for (;;) {
        while (last - first > 16) {
            if (depth == 0) {
                tk_make_heap<LARGEST>(q + first, last - first);
                tk_sort_heap<LARGEST>(q + first, last - first);
                tk_sync();
                if (fallbacks) *fallbacks += 1;
                first = last;  // (nothing left of this segment)
                break;
            }
            --depth;
            const bool lds = last - first <= TK_POS_LDS;
            const int cut = tk_partition_pivot<LARGEST>(q, first, last, lds ? sh.pos.l : lpos_g, lds ? sh.pos.r : rpos_g);
            stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth;  // __introsort_loop(cut, last, depth_limit)
            ++sp;
            last = cut;
        }
        tk_stable_small<LARGEST>(q, first, last);
        if (sp == 0) break;
        --sp;
        tk_sync();
        first = tk_uni(stack[3 * sp]); last = tk_uni(stack[3 * sp + 1]); depth = tk_uni(stack[3 * sp + 2]);
    }
}

template <bool LARGEST>
MD void tk_topk(TkPair* q, int n, int k, bool sorted, int* lpos_g, int* rpos_g, TkShared& sh, int* stack, int* fallbacks) {
    const int lane = tk_lane();
    if ((long long)k * 64 <= (long long)n) {  // std::partial_sort(queue, queue + k, queue + n)
        const bool lds = k <= TK_HEAP_LDS;
        TkPair* h = lds ? sh.heap : q;
        if (lds) {
            for (int j = lane; j < k; j += 64) sh.heap[j] = q[j];
            tk_sync();
        }
        tk_heap_select<LARGEST>(h, k, q, k, n);
        if (sorted) tk_sort_heap<LARGEST>(h, k);
        tk_sync();
        if (lds)
            for (int j = lane; j < k; j += 64) q[j] = sh.heap[j];
    } else {  // std::nth_element(queue, queue + k - 1, queue + n) [+ std::sort(queue, queue + k - 1)]
        int depth = 0;
        for (int m = n; m > 1; m >>= 1) ++depth;
        tk_introselect<LARGEST>(q, 0, k - 1, n, 2 * depth, lpos_g, rpos_g, sh, fallbacks);
        if (sorted) tk_sort<LARGEST>(q, 0, k - 1, lpos_g, rpos_g, sh, stack, fallbacks);
    }
    tk_sync();
}

// queue[j] = (w[j], j); the removal marks cleared
__global__ __launch_bounds__(256) void k_topk_init(const int32_t* __restrict__ ctl_i, const double* __restrict__ w, TkPair* __restrict__ q,
                                                   uint8_t* __restrict__ mark) {
    const int n = ctl_i[LOOP_I_N], mode = ctl_i[LOOP_I_MODE];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (!mode || j >= n) return;
    TkPair p;
    p.v = w[j]; p.i = j; p.pad = 0;
    q[j] = p;
    mark[j] = 0;
}

// One wave: the first k of the queue become torch.topk's output (mode 1: the k smallest, order irrelevant - they are removed;
// mode 2: the k largest, sorted).  info[0] += depth-limit fallbacks taken (tests).
__global__ __launch_bounds__(64) void k_topk_select(const int32_t* __restrict__ ctl_i, TkPair* __restrict__ q, int* __restrict__ lpos,
                                                    int* __restrict__ rpos, int* __restrict__ info) {
    __shared__ TkShared sh;
    __shared__ int stack[3 * TK_STACK];
    const int n = ctl_i[LOOP_I_N], mode = ctl_i[LOOP_I_MODE], k = ctl_i[LOOP_I_K];
    if (!mode || k <= 0 || k > n) return;
    int fb = 0;
    if (mode == 2) tk_topk<true>(q, n, k, true, lpos, rpos, sh, stack, &fb);
    else tk_topk<false>(q, n, k, false, lpos, rpos, sh, stack, &fb);
    if (info && tk_lane() == 0) info[0] += fb;
}

// mode 1: mark the k removed particles; mode 2: src[n + j] = the j-th best, src[0, n) = identity
__global__ __launch_bounds__(256) void k_topk_emit(const int32_t* __restrict__ ctl_i, const TkPair* __restrict__ q, uint8_t* __restrict__ mark,
                                                   int32_t* __restrict__ src) {
    const int n = ctl_i[LOOP_I_N], mode = ctl_i[LOOP_I_MODE], k = ctl_i[LOOP_I_K];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    if (mode != 1) src[j] = j;
    if (j >= k) return;
    if (mode == 1) mark[q[j].i] = 1;
    else if (mode == 2) src[n + j] = q[j].i;
}

// mode 1: the survivors in their order.  Per 4096-slot block the number of marks, then every block adds up the blocks
// before it (at most 256) and writes its survivors.
__global__ __launch_bounds__(256) void k_topk_count(const int32_t* __restrict__ ctl_i, const uint8_t* __restrict__ mark, int32_t* __restrict__ cnt) {
    __shared__ int s_w[4];
    const int n = ctl_i[LOOP_I_N];
    if (ctl_i[LOOP_I_MODE] != 1) return;
    const int64_t bbase = (int64_t)blockIdx.x * SCAN_BLOCK;
    const int t = threadIdx.x;
    int c = 0;
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = bbase + (int64_t)j * 256 + t;
        c += (i < n && mark[i]) ? 1 : 0;
    }
    c = wave_isum_dpp(c);
    if ((t & 63) == 0) s_w[t >> 6] = c;
    __syncthreads();
    if (t == 0) cnt[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void k_topk_compact(const int32_t* __restrict__ ctl_i, const uint8_t* __restrict__ mark,
                                                      const int32_t* __restrict__ cnt, int32_t* __restrict__ src) {
    __shared__ int s_w[4];
    __shared__ int s_before;
    const int n = ctl_i[LOOP_I_N];
    if (ctl_i[LOOP_I_MODE] != 1) return;
    const int blk = blockIdx.x, t = threadIdx.x;
    const int64_t bbase = (int64_t)blk * SCAN_BLOCK;
    if (bbase >= n) return;
    int before = 0;
    for (int i = t; i < blk; i += 256) before += cnt[i];
    before = wave_isum_dpp(before);
    if ((t & 63) == 0) s_w[t >> 6] = before;
    __syncthreads();
    if (t == 0) s_before = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
    int removed = s_before;
    // thread t owns the 16 consecutive slots bbase + 16 t ..
    const int64_t base = bbase + (int64_t)t * SCAN_CHUNK;
    unsigned bits = 0;
    for (int j = 0; j < SCAN_CHUNK; ++j) bits |= (base + j < n && mark[base + j]) ? (1u << j) : 0u;
    const int mine = __popc(bits);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((t & 63) >= o) incl += v;
    }
    __syncthreads();
    if ((t & 63) == 63) s_w[t >> 6] = incl;
    __syncthreads();
    removed += incl - mine;
    for (int wv = 0; wv < (t >> 6); ++wv) removed += s_w[wv];
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        if (i >= n) break;
        if ((bits >> j) & 1u) ++removed;
        else src[i - removed] = (int32_t)i;
    }
}

// ctl_i[N, MODE, K] are in place (k_loop_decide / k_anneal_plan): src = the annealed set under ATen's CPU rule
int launch_topk_aten(midas_ctx* ctx, int64_t cap, const int32_t* ci, const double* w, int32_t* src, int32_t* info) {
    void* p;
    int rc;
    if ((rc = midas_scratch(ctx, (size_t)cap * sizeof(TkPair), &p))) return rc;
    TkPair* q = (TkPair*)p;
    if ((rc = midas_scratch(ctx, (size_t)cap * sizeof(int), &p))) return rc;
    int* lpos = (int*)p;
    if ((rc = midas_scratch(ctx, (size_t)cap * sizeof(int), &p))) return rc;
    int* rpos = (int*)p;
    if ((rc = midas_scratch(ctx, (size_t)cap, &p))) return rc;
    uint8_t* mark = (uint8_t*)p;
    const unsigned nb = (unsigned)ceil_div(cap, SCAN_BLOCK);
    if ((rc = midas_scratch(ctx, (size_t)nb * sizeof(int32_t), &p))) return rc;
    int32_t* cnt = (int32_t*)p;
    hipStream_t st = ctx->stream;
    const unsigned g = (unsigned)ceil_div(cap, 256);
    hipLaunchKernelGGL(k_topk_init, dim3(g), dim3(256), 0, st, ci, w, q, mark);
    hipLaunchKernelGGL(k_topk_select, dim3(1), dim3(64), 0, st, ci, q, lpos, rpos, (int*)info);
    hipLaunchKernelGGL(k_topk_emit, dim3(g), dim3(256), 0, st, ci, (const TkPair*)q, mark, src);
    hipLaunchKernelGGL(k_topk_count, dim3(nb), dim3(256), 0, st, ci, (const uint8_t*)mark, cnt);
    hipLaunchKernelGGL(k_topk_compact, dim3(nb), dim3(256), 0, st, ci, (const uint8_t*)mark, (const int32_t*)cnt, src);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

MIDAS_WARM_TU(topk_aten, k_topk_select)

}  // namespace midas
