/* aten_topk.c - TEST INFRASTRUCTURE (CPU oracle; only tests/, smoke() and bench.py's cpu_baseline leg may use it).
 *
 * Restates which elements, in which order, `torch.topk` returns on the CPU for a one-dimensional float64 tensor - the call
 * particle_filter.annealing makes (/root/reference/midastouch/modules/particle_filter.py:433-441).  Inside a tie the answer
 * is not a property of the values but of the algorithm, so the algorithm is restated:
 *
 *   ATen (third-party dependency of the reference, absent from /root/reference; the build installed here is torch 2.10):
 *   aten/src/ATen/native/cpu/SortingKernel.cpp `topk_kernel` -> aten/src/ATen/native/TopKImpl.h `topk_impl_loop`:
 *     queue[j] = (value_j, j);  use_partial_sort = k * 64 <= n;
 *     partial sort : std::partial_sort(queue, queue + k, queue + n, cmp)
 *     otherwise    : std::nth_element(queue, queue + k - 1, queue + n, cmp); if (sorted) std::sort(queue, queue + k - 1, cmp)
 *     cmp largest  : (isnan(x) && !isnan(y)) || x > y         cmp smallest : (!isnan(x) && isnan(y)) || x < y
 *     output j = queue[j], j < k
 *   libstdc++ (GCC; bits/stl_algo.h, bits/stl_heap.h - the published algorithms, unchanged for two decades):
 *     partial_sort = __heap_select + __sort_heap;  nth_element = __introselect (depth limit 2 lg n, median of three to the
 *     front, unguarded Hoare partition, insertion sort below 4 elements, __heap_select when the limit is spent);
 *     sort = __introsort_loop (threshold 16, same partition, heap sort when the limit is spent) + __final_insertion_sort.
 *
 * PINNED by tests/test_aten_topk.py against torch.topk itself (torch is an installed library on every box): random
 * tie-heavy inputs over both branches, both directions, NaN, signed zeros, and inputs built by an adversary against the
 * median-of-three partition so that both depth-limit fallbacks run.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define API __attribute__((visibility("default")))

typedef struct { double v; int64_t i; } pair_t;

typedef int (*cmp_fn)(const pair_t*, const pair_t*);
static int cmp_largest(const pair_t* x, const pair_t* y) { return (isnan(x->v) && !isnan(y->v)) || (x->v > y->v); }
static int cmp_smallest(const pair_t* x, const pair_t* y) { return (!isnan(x->v) && isnan(y->v)) || (x->v < y->v); }

/* ---- bits/stl_heap.h ---------------------------------------------------------------------------------------------- */
static void push_heap_(pair_t* first, int64_t hole, int64_t top, pair_t value, cmp_fn comp) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && comp(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void adjust_heap_(pair_t* first, int64_t hole, int64_t len, pair_t value, cmp_fn comp) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(first + child, first + (child - 1))) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, comp);
}

static void pop_heap_(pair_t* first, pair_t* last, pair_t* result, cmp_fn comp) {
    const pair_t value = *result;
    *result = *first;
    adjust_heap_(first, 0, last - first, value, comp);
}

static void make_heap_(pair_t* first, pair_t* last, cmp_fn comp) {
    const int64_t len = last - first;
    if (len < 2) return;
    int64_t parent = (len - 2) / 2;
    for (;;) {
        adjust_heap_(first, parent, len, first[parent], comp);
        if (parent == 0) return;
        parent--;
    }
}

static void heap_select_(pair_t* first, pair_t* middle, pair_t* last, cmp_fn comp) {
    make_heap_(first, middle, comp);
    for (pair_t* i = middle; i < last; ++i)
        if (comp(i, first)) pop_heap_(first, middle, i, comp);
}

static void sort_heap_(pair_t* first, pair_t* last, cmp_fn comp) {
    while (last - first > 1) {
        --last;
        pop_heap_(first, last, last, comp);
    }
}

/* ---- bits/stl_algo.h ---------------------------------------------------------------------------------------------- */
static void swap_(pair_t* a, pair_t* b) { const pair_t t = *a; *a = *b; *b = t; }

static void move_median_to_first_(pair_t* result, pair_t* a, pair_t* b, pair_t* c, cmp_fn comp) {
    if (comp(a, b)) {
        if (comp(b, c)) swap_(result, b);
        else if (comp(a, c)) swap_(result, c);
        else swap_(result, a);
    } else if (comp(a, c)) swap_(result, a);
    else if (comp(b, c)) swap_(result, c);
    else swap_(result, b);
}

static pair_t* unguarded_partition_(pair_t* first, pair_t* last, pair_t* pivot, cmp_fn comp) {
    for (;;) {
        while (comp(first, pivot)) ++first;
        --last;
        while (comp(pivot, last)) --last;
        if (!(first < last)) return first;
        swap_(first, last);
        ++first;
    }
}

static pair_t* partition_pivot_(pair_t* first, pair_t* last, cmp_fn comp) {
    pair_t* mid = first + (last - first) / 2;
    move_median_to_first_(first, first + 1, mid, last - 1, comp);
    return unguarded_partition_(first + 1, last, first, comp);
}

static void unguarded_linear_insert_(pair_t* last, cmp_fn comp) {
    const pair_t val = *last;
    pair_t* next = last - 1;
    while (comp(&val, next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

static void insertion_sort_(pair_t* first, pair_t* last, cmp_fn comp) {
    if (first == last) return;
    for (pair_t* i = first + 1; i != last; ++i) {
        if (comp(i, first)) {
            const pair_t val = *i;
            for (pair_t* j = i; j != first; --j) *j = *(j - 1);
            *first = val;
        } else
            unguarded_linear_insert_(i, comp);
    }
}

static int lg_(int64_t n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

static int64_t g_fallbacks;  /* depth-limit fallbacks taken by the last call (tests check that the adversarial inputs reach them) */

static void introselect_(pair_t* first, pair_t* nth, pair_t* last, int64_t depth_limit, cmp_fn comp) {
    while (last - first > 3) {
        if (depth_limit == 0) {
            ++g_fallbacks;
            heap_select_(first, nth + 1, last, comp);
            swap_(first, nth);
            return;
        }
        --depth_limit;
        pair_t* cut = partition_pivot_(first, last, comp);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort_(first, last, comp);
}

static void introsort_loop_(pair_t* first, pair_t* last, int64_t depth_limit, cmp_fn comp) {
    while (last - first > 16) {
        if (depth_limit == 0) {
            ++g_fallbacks;
            heap_select_(first, last, last, comp);
            sort_heap_(first, last, comp);
            return;
        }
        --depth_limit;
        pair_t* cut = partition_pivot_(first, last, comp);
        introsort_loop_(cut, last, depth_limit, comp);
        last = cut;
    }
}

static void sort_(pair_t* first, pair_t* last, cmp_fn comp) {
    if (first == last) return;
    introsort_loop_(first, last, lg_(last - first) * 2, comp);
    if (last - first > 16) {
        insertion_sort_(first, first + 16, comp);
        for (pair_t* i = first + 16; i != last; ++i) unguarded_linear_insert_(i, comp);
    } else
        insertion_sort_(first, last, comp);
}

/* torch.topk(values, k, largest, sorted).indices for a 1-d float64 tensor on the CPU.  Returns 0, or -1 when k is out of
 * range / memory is short.  *fallbacks (optional) = depth-limit fallbacks the call went through. */
API int mo_aten_topk(const double* values, int64_t n, int64_t k, int largest, int sorted, int64_t* idx_out, double* val_out,
                     int64_t* fallbacks) {
    if (k < 0 || k > n) return -1;
    if (fallbacks) *fallbacks = 0;
    if (k == 0) return 0;
    pair_t* q = (pair_t*)malloc((size_t)n * sizeof(pair_t));
    if (!q) return -1;
    for (int64_t j = 0; j < n; ++j) { q[j].v = values[j]; q[j].i = j; }
    const cmp_fn comp = largest ? cmp_largest : cmp_smallest;
    g_fallbacks = 0;
    if (k * 64 <= n) {
        heap_select_(q, q + k, q + n, comp);
        sort_heap_(q, q + k, comp);
    } else {
        /* nth_element(first, nth, last) returns at once when nth == last: cannot happen, nth = k - 1 < n */
        introselect_(q, q + k - 1, q + n, lg_(n) * 2, comp);
        if (sorted) sort_(q, q + k - 1, comp);
    }
    for (int64_t j = 0; j < k; ++j) {
        idx_out[j] = q[j].i;
        if (val_out) val_out[j] = q[j].v;
    }
    if (fallbacks) *fallbacks = g_fallbacks;
    free(q);
    return 0;
}

/* An input of length n on which the median-of-three partition above degenerates (McIlroy's adversary, "A Killer Adversary
 * for Quicksort", 1999, played against nth_element / sort of THIS file): values are decided lazily while the algorithm
 * compares them, so that every pivot turns out to be among the smallest.  The values written are a permutation of 0..n-1
 * (as doubles); `for_sort` plays against sort_() over the whole array, otherwise against introselect_ at position nth. */
static double* adv_val;
static int64_t adv_nsolid, adv_candidate, adv_gas;
static int adv_cmp(const pair_t* x, const pair_t* y) {
    const int64_t a = x->i, b = y->i;
    if (adv_val[a] == (double)adv_gas && adv_val[b] == (double)adv_gas) {
        if (a == adv_candidate) adv_val[a] = (double)adv_nsolid++;
        else adv_val[b] = (double)adv_nsolid++;
    }
    if (adv_val[a] == (double)adv_gas) adv_candidate = a;
    else if (adv_val[b] == (double)adv_gas) adv_candidate = b;
    return adv_val[a] < adv_val[b];
}

API int mo_aten_topk_killer(int64_t n, int64_t nth, int for_sort, double* values_out) {
    pair_t* q = (pair_t*)malloc((size_t)n * sizeof(pair_t));
    adv_val = (double*)malloc((size_t)n * sizeof(double));
    if (!q || !adv_val) { free(q); free(adv_val); return -1; }
    adv_gas = n - 1;
    adv_nsolid = 0;
    adv_candidate = 0;
    for (int64_t j = 0; j < n; ++j) { adv_val[j] = (double)adv_gas; q[j].v = 0.0; q[j].i = j; }
    if (for_sort) sort_(q, q + n, adv_cmp);
    else introselect_(q, q + nth, q + n, lg_(n) * 2, adv_cmp);
    /* what stayed gas gets the remaining values in index order */
    for (int64_t j = 0; j < n; ++j) values_out[j] = adv_val[j] == (double)adv_gas ? (double)adv_nsolid++ : adv_val[j];
    free(q);
    free(adv_val);
    return 0;
}
