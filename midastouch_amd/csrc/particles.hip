// particles.hip - per-particle kernels: SE(3) propagate (K2), 6-d pose feature (K3), exact nearest
// neighbour on static KD-trees (K4), the fused particle update of the step, rmse and pose checks.
//
// One lane owns one particle.  The NN search is latency-bound pointer chasing over a tree that lives
// in L2 (K=50k: 128 KB of nodes + 1.6 MB of points), so workgroups are single waves (64 threads) to
// spread the N/64 waves evenly over the 256 CUs, and the traversal is stack-free (a bit-trail of
// pending far children in one register) so it needs no LDS and no scratch.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <vector>

#include "midas_internal.hpp"
#include "midas_math.hpp"
#include "score_body.hpp"
#include "peer_row.hpp"
#include "resample_search.hpp"

namespace midas {

// =================================================================================================
// spatial index: host build (balanced median splits, three binary splits per 8-ary level)
// =================================================================================================
template <class KD>
struct HostTree {
    std::vector<typename KD::Box> boxes;
    std::vector<typename KD::Point> pts;
    std::vector<int32_t> inv_perm;
    std::vector<Nbr6> nbrs;      // dim 6 only
    std::vector<float> rho_out;  // dim 6 only
    std::vector<int32_t> twin;   // dim 6 only
    int levels = 0;              // 8-ary levels
};

static inline int64_t level_offset(int l) { return (((int64_t)1 << (3 * l)) - 1) / 7; }

// binary node `b` (1-based heap) at binary depth `depth`; every third depth is an 8-ary node
template <class KD>
static void build_rec(HostTree<KD>& t, const typename KD::T* P, std::vector<int32_t>& perm, uint64_t b, int64_t lo,
                      int64_t hi, int depth) {
    using T = typename KD::T;
    constexpr int DIM = KD::DIM;
    typename KD::Box bx;
    for (int d = 0; d < DIM; ++d) { bx.lo[d] = INFINITY; bx.hi[d] = -INFINITY; }
    for (int64_t i = lo; i < hi; ++i)
        for (int d = 0; d < DIM; ++d) {
            T v = P[(int64_t)perm[i] * DIM + d];
            bx.lo[d] = v < bx.lo[d] ? v : bx.lo[d];
            bx.hi[d] = v > bx.hi[d] ? v : bx.hi[d];
        }
    if (depth % 3 == 0) {
        const int l = depth / 3;
        const int64_t local = (int64_t)b - ((int64_t)1 << depth);
        t.boxes[level_offset(l) + local] = bx;
        if (l == t.levels) {
            std::sort(perm.begin() + lo, perm.begin() + hi);
            for (int64_t i = lo; i < hi; ++i) {
                typename KD::Point p;
                for (int d = 0; d < DIM; ++d) p.c[d] = P[(int64_t)perm[i] * DIM + d];
                p.idx = perm[i];
                if constexpr (DIM == 6) p.pad = 0;
                const int64_t slot = local * LEAF_CAP + (i - lo);
                t.pts[slot] = p;
                t.inv_perm[perm[i]] = (int32_t)slot;
            }
            return;
        }
    }
    int best_dim = 0;
    T best_spread = -1;
    for (int d = 0; d < DIM; ++d)
        if (bx.hi[d] - bx.lo[d] > best_spread) { best_spread = bx.hi[d] - bx.lo[d]; best_dim = d; }
    const int64_t mid = lo + (hi - lo + 1) / 2;
    auto cmp = [&](int32_t a, int32_t c) {
        T va = P[(int64_t)a * DIM + best_dim], vc = P[(int64_t)c * DIM + best_dim];
        return va < vc || (va == vc && a < c);
    };
    if (mid < hi) std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, cmp);
    build_rec(t, P, perm, 2 * b, lo, mid, depth + 1);
    build_rec(t, P, perm, 2 * b + 1, mid, hi, depth + 1);
}

template <class KD>
static HostTree<KD> build_tree(const typename KD::T* P, int64_t K) {
    HostTree<KD> t;
    int levels = 0;
    while ((((int64_t)LEAF_CAP) << (3 * levels)) < K) ++levels;
    t.levels = levels;
    const int64_t nleaves = (int64_t)1 << (3 * levels);
    t.boxes.resize(level_offset(levels + 1) + 1);  // +1: the 64-byte unified fetch reads 16 bytes past a box
    typename KD::Point pad;
    for (int d = 0; d < KD::DIM; ++d) pad.c[d] = INFINITY;
    pad.idx = 0x7fffffff;
    if constexpr (KD::DIM == 6) pad.pad = 0;
    t.pts.assign(nleaves * LEAF_CAP, pad);
    t.inv_perm.resize(K);
    std::vector<int32_t> perm(K);
    std::iota(perm.begin(), perm.end(), 0);
    build_rec(t, P, perm, 1u, 0, K, 0);
    return t;
}

// ---- neighbour graph of the codebook features (hint fast path) ---------------------------------
// For every entry k: its NBR_M nearest other entries sorted by (distance, index), float64 distances,
// rho rounded DOWN to float32 so that it never over-states a true distance.
namespace {
struct HeapItem { double d; int64_t idx; };
inline bool heap_less(const HeapItem& a, const HeapItem& b) { return a.d < b.d || (a.d == b.d && a.idx < b.idx); }

template <class KD>
double host_box_d2(const double* q, const typename KD::Box& b) {
    double d = 0.0;
    for (int j = 0; j < KD::DIM; ++j) {
        double a = (double)b.lo[j] - q[j], c = q[j] - (double)b.hi[j];
        double m = a > c ? a : c;
        if (m > 0) d += m * m;
    }
    return d;
}

// k nearest points of q (float64 distances), excluding original index `self` (-1: none)
template <class KD>
void knn_rec(const HostTree<KD>& t, const double* q, int64_t self, int64_t node, int level, std::vector<HeapItem>& heap,
             size_t k) {
    if (level == t.levels) {
        const typename KD::Point* lp = t.pts.data() + (size_t)(node - level_offset(level)) * LEAF_CAP;
        for (int j = 0; j < LEAF_CAP; ++j) {
            if (lp[j].idx == 0x7fffffff || (int64_t)lp[j].idx == self) continue;
            double d = 0.0;
            for (int a = 0; a < KD::DIM; ++a) { double x = q[a] - (double)lp[j].c[a]; d += x * x; }
            HeapItem it{d, (int64_t)lp[j].idx};
            if (heap.size() < k) {
                heap.push_back(it);
                std::push_heap(heap.begin(), heap.end(), heap_less);
            } else if (heap_less(it, heap.front())) {
                std::pop_heap(heap.begin(), heap.end(), heap_less);
                heap.back() = it;
                std::push_heap(heap.begin(), heap.end(), heap_less);
            }
        }
        return;
    }
    std::pair<double, int> order[8];
    for (int j = 0; j < 8; ++j) order[j] = {host_box_d2<KD>(q, t.boxes[8 * node + 1 + j]), j};
    std::sort(order, order + 8);
    for (int j = 0; j < 8; ++j)
        if (heap.size() < k || order[j].first <= heap.front().d)
            knn_rec<KD>(t, q, self, 8 * node + 1 + order[j].second, level + 1, heap, k);
}

template <class F>
void parallel_for(int64_t n, F&& work) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; ++i) th.emplace_back(work, n * i / nt, n * (i + 1) / nt);
    for (auto& x : th) x.join();
}

inline float round_down_f32(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, 0.0f);
    return f;
}
}  // namespace

static void build_neighbour_graph(HostTree<Kd6>& t, const float* P, int64_t K) {
    t.nbrs.resize((size_t)K * NBR_REC);
    t.rho_out.resize(K);
    t.twin.resize(K);
    parallel_for(K, [&](int64_t k0, int64_t k1) {
        std::vector<HeapItem> heap;
        for (int64_t k = k0; k < k1; ++k) {
            heap.clear();
            double q[6];
            for (int a = 0; a < 6; ++a) q[a] = (double)P[k * 6 + a];
            knn_rec<Kd6>(t, q, k, 0, 0, heap, (size_t)NBR_M + 1);
            std::sort(heap.begin(), heap.end(), heap_less);
            {   // record 0 = the entry itself (rho 0): the scan needs no other lookup
                Nbr6 r;
                for (int a = 0; a < 6; ++a) r.c[a] = P[k * 6 + a];
                r.idx = (int32_t)k;
                r.rho = 0.0f;
                t.nbrs[(size_t)k * NBR_REC] = r;
            }
            for (int s2 = 0; s2 < NBR_M; ++s2) {
                Nbr6 r;
                if ((size_t)s2 < heap.size()) {
                    const int64_t j = heap[s2].idx;
                    for (int a = 0; a < 6; ++a) r.c[a] = P[j * 6 + a];
                    r.idx = (int32_t)j;
                    r.rho = round_down_f32(std::sqrt(heap[s2].d));
                } else {
                    for (int a = 0; a < 6; ++a) r.c[a] = INFINITY;
                    r.idx = 0x7fffffff;
                    r.rho = INFINITY;
                }
                t.nbrs[(size_t)k * NBR_REC + 1 + s2] = r;
            }
            t.rho_out[k] = heap.size() > (size_t)NBR_M ? round_down_f32(std::sqrt(heap[NBR_M].d)) : INFINITY;
            // twin across the rotation-angle-pi cut: the feature's rotation part is w = 0.01 log(R); a pose whose
            // angle crosses pi reappears at w - 2 pi 0.01 w/|w|.  The entry nearest to that image is the right
            // second hint for a particle whose own feature has just flipped.
            const double wn = std::sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5]);
            t.twin[k] = -1;
            if (wn > 0.01 * (M_PI - 0.6)) {
                double qf[6] = {q[0], q[1], q[2], 0, 0, 0};
                const double sc = (wn - 0.01 * 2.0 * M_PI) / wn;
                for (int a = 3; a < 6; ++a) qf[a] = q[a] * sc;
                heap.clear();
                knn_rec<Kd6>(t, qf, -1, 0, 0, heap, 1);
                if (!heap.empty() && heap[0].idx != k) t.twin[k] = (int32_t)heap[0].idx;
            }
        }
    });
}

// ---- mesh-vertex lists anchored at the codebook entries (prune fast path) --------------------------
// For entry k: the MESH_M mesh vertices nearest to its translation t_k, sorted by rho = |v - t_k|
// (rounded down), and rho_out = distance of the next vertex.  A particle whose NN entry is k can only be
// within thr of a vertex v if rho_v <= thr + |t_q - t_k|, so scanning the list in order decides the prune
// exactly unless the list runs out first.
int attach_mesh_impl(midas_ctx* ctx, midas_tree* t6, const midas_tree* t3, const float* cb_poses_dev) {
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (!index_build_on_host()) return build_vertex_lists_device(ctx, t6, t3, cb_poses_dev);
    const HostTree<Kd3>* mesh = reinterpret_cast<const HostTree<Kd3>*>(t3->host);
    if (!mesh) return midas_set_error(ctx, MIDAS_ERR_INVALID, "attach_mesh", "mesh tree has no host copy");
    const int64_t K = t6->K;
    std::vector<float> poses((size_t)K * 16);
    MIDAS_HIP_CHECK(ctx, hipMemcpy(poses.data(), cb_poses_dev, poses.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<MeshRec> recs((size_t)K * MESH_REC);
    parallel_for(K, [&](int64_t k0, int64_t k1) {
        std::vector<HeapItem> heap;
        for (int64_t k = k0; k < k1; ++k) {
            heap.clear();
            const double q[3] = {(double)poses[k * 16 + 3], (double)poses[k * 16 + 7], (double)poses[k * 16 + 11]};
            knn_rec<Kd3>(*mesh, q, -1, 0, 0, heap, (size_t)MESH_M + 1);
            std::sort(heap.begin(), heap.end(), heap_less);
            for (int s2 = 0; s2 < MESH_M; ++s2) {
                MeshRec r;
                if ((size_t)s2 < heap.size()) {
                    const typename Kd3::Point& p = mesh->pts[mesh->inv_perm[heap[s2].idx]];
                    r.c[0] = p.c[0]; r.c[1] = p.c[1]; r.c[2] = p.c[2];
                    r.rho = round_down_f32(std::sqrt(heap[s2].d));
                } else {
                    r.c[0] = r.c[1] = r.c[2] = INFINITY;
                    r.rho = INFINITY;
                }
                r.pad = 0;
                recs[(size_t)k * MESH_REC + 1 + s2] = r;
            }
            MeshRec hd;
            hd.c[0] = q[0]; hd.c[1] = q[1]; hd.c[2] = q[2];
            hd.rho = heap.size() > (size_t)MESH_M ? round_down_f32(std::sqrt(heap[MESH_M].d)) : INFINITY;
            hd.pad = 0;
            recs[(size_t)k * MESH_REC] = hd;
        }
    });
    if (t6->vlist) { (void)hipFree(t6->vlist); t6->vlist = nullptr; }
    MIDAS_HIP_CHECK(ctx, hipMalloc(&t6->vlist, recs.size() * sizeof(MeshRec)));
    MIDAS_HIP_CHECK(ctx, hipMemcpy(t6->vlist, recs.data(), recs.size() * sizeof(MeshRec), hipMemcpyHostToDevice));
    t6->vlist_mesh = t3;
    return build_vertex_screen(ctx, t6);
}

template <class KD>
static int upload_tree(midas_ctx* ctx, const HostTree<KD>& h, int64_t K, midas_tree* out) {
    auto up = [&](const void* src, size_t bytes, void** dst) -> int {
        MIDAS_HIP_CHECK(ctx, hipMalloc(dst, bytes ? bytes : 16));
        if (bytes) MIDAS_HIP_CHECK(ctx, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return MIDAS_OK;
    };
    int rc;
    if ((rc = up(h.boxes.data(), h.boxes.size() * sizeof(typename KD::Box), &out->boxes))) return rc;
    if ((rc = up(h.pts.data(), h.pts.size() * sizeof(typename KD::Point), &out->pts))) return rc;
    if ((rc = up(h.inv_perm.data(), h.inv_perm.size() * sizeof(int32_t), (void**)&out->inv_perm))) return rc;
    if (!h.nbrs.empty()) {
        if ((rc = up(h.nbrs.data(), h.nbrs.size() * sizeof(Nbr6), &out->nbrs))) return rc;
        if ((rc = up(h.rho_out.data(), h.rho_out.size() * sizeof(float), (void**)&out->rho_out))) return rc;
        if ((rc = up(h.twin.data(), h.twin.size() * sizeof(int32_t), (void**)&out->twin))) return rc;
    }
    out->levels = h.levels;
    out->K = K;
    return MIDAS_OK;
}

// The distance field of a dim-3 tree's vertices (MeshField): the bounding box grown by FIELD_EXPAND, cubic cells sized so that the
// grid has at most FIELD_MAX_CELLS of them, every cell's value by the exact search (k_field_build).  MIDAS_MESH_FIELD=0: none.
constexpr double FIELD_EXPAND = 0.0025;            // m: decides "outside the grid = pruned" for thresholds below it (the reference's is 0.002)
constexpr double FIELD_MIN_CELL = 2.5e-5;          // m: cells no finer than this (shell half-width < 0.022 mm)
constexpr int64_t FIELD_MAX_CELLS = (int64_t)1 << 26;  // 256 MB of float32 (c4's mug: 0.23 mm cells; with 2^22 cells of 0.59 mm the undecided shell held 1700 particles a frame)
static int build_mesh_field(midas_ctx* ctx, midas_tree* t, const double* pts, int64_t K);

int tree_build_impl(midas_ctx* ctx, int32_t dim, int64_t K, const void* points_dev, midas_tree* out) {
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (dim == 6) {
        std::vector<float> host((size_t)K * 6);
        MIDAS_HIP_CHECK(ctx, hipMemcpy(host.data(), points_dev, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        HostTree<Kd6> h = build_tree<Kd6>(host.data(), K);
        if (index_build_on_host()) {
            build_neighbour_graph(h, host.data(), K);
            return upload_tree<Kd6>(ctx, h, K, out);
        }
        // the box tree (K log K) on the host, the neighbour graph (K x K) on the device (index_build.hip)
        int rc = upload_tree<Kd6>(ctx, h, K, out);
        if (rc) return rc;
        return build_neighbour_graph_device(ctx, out);
    }
    std::vector<double> host((size_t)K * 3);
    MIDAS_HIP_CHECK(ctx, hipMemcpy(host.data(), points_dev, host.size() * sizeof(double), hipMemcpyDeviceToHost));
    HostTree<Kd3>* h = new HostTree<Kd3>(build_tree<Kd3>(host.data(), K));
    out->host = h;  // kept for attach_mesh (host k-NN over the mesh vertices)
    int rc = upload_tree<Kd3>(ctx, *h, K, out);
    if (rc) return rc;
    return build_mesh_field(ctx, out, host.data(), K);
}

void tree_free_host(midas_tree* t) {
    if (t->host) {
        if (t->dim == 3) delete reinterpret_cast<HostTree<Kd3>*>(t->host);
        t->host = nullptr;
    }
}

template <class KD>
static TreeView<KD> view_of(const midas_tree* t) {
    TreeView<KD> v;
    v.boxes = (const typename KD::Box*)t->boxes;
    v.pts = (const typename KD::Point*)t->pts;
    v.inv_perm = t->inv_perm;
    v.nbrs = (const Nbr6*)t->nbrs;
    v.rho_out = t->rho_out;
    v.twin = t->twin;
    v.levels = t->levels;
    v.K = t->K;
    return v;
}

// =================================================================================================
// KD-tree: device traversal
// =================================================================================================
MD float dist2(const float* q, const Point6& p) {
    float d0 = q[0] - p.c[0], d1 = q[1] - p.c[1], d2 = q[2] - p.c[2];
    float d3 = q[3] - p.c[3], d4 = q[4] - p.c[4], d5 = q[5] - p.c[5];
    float d = d0 * d0;
    d = fmaf_(d1, d1, d);
    d = fmaf_(d2, d2, d);
    d = fmaf_(d3, d3, d);
    d = fmaf_(d4, d4, d);
    d = fmaf_(d5, d5, d);
    return d;
}
MD double dist2(const double* q, const Point3& p) {
    double d0 = q[0] - p.c[0], d1 = q[1] - p.c[1], d2 = q[2] - p.c[2];
    double d = d0 * d0;
    d = fma_(d1, d1, d);
    d = fma_(d2, d2, d);
    return d;
}

// Lower bound of dist2(q, p) over every p inside the box, IN THE COMPUTED ARITHMETIC: each per-axis
// offset is <= |q_j - p_j| after rounding (subtraction and max are monotone) and the fma chain has the
// same shape as dist2, so monotonicity of rounding carries the bound through.
MD float box_dist2(const float* q, const Box6& b) {
    float t[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float a = b.lo[j] - q[j], c = q[j] - b.hi[j];
        float m = a > c ? a : c;
        t[j] = m > 0.0f ? m : 0.0f;
    }
    float d = t[0] * t[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) d = fmaf_(t[j], t[j], d);
    return d;
}
MD double box_dist2(const double* q, const Box3& b) {
    double t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double a = b.lo[j] - q[j], c = q[j] - b.hi[j];
        double m = a > c ? a : c;
        t[j] = m > 0.0 ? m : 0.0;
    }
    double d = t[0] * t[0];
    d = fma_(t[1], t[1], d);
    d = fma_(t[2], t[2], d);
    return d;
}

MD int64_t level_offset_dev(int l) { return (((int64_t)1 << (3 * l)) - 1) / 7; }

// ---- octet-cooperative exact search ------------------------------------------------------------
// Eight lanes (an octet) serve ONE query: at a node each lane tests one child box, at a leaf two point
// slots.  The child distances of every level on the current path stay in an LDS column (cd[level][lane]),
// so backtracking touches no memory; each loop iteration issues one round of global loads (a 48-byte box
// or two 32-byte points per lane).  A wave therefore advances eight queries at a time and finishes a query
// in ~(levels + a few) rounds instead of the ~13-level descents of a binary tree walked per lane.
//
// All state below is octet-uniform except the lane's own child distance.  `cand` packs, per level, the
// 8-bit set of children still worth visiting.  Children are visited in order of box distance (3 low
// mantissa bits replaced by the child number: that only orders the visits, pruning uses exact values).
MD uint32_t octet_bits(bool pred, int octet) { return (uint32_t)((__ballot(pred) >> (8 * octet)) & 0xffull); }

// Cross-lane moves inside an octet as DPP modifiers (no LDS round trip): lane^1, lane^2 (quad permutes)
// and lane <-> 7-lane (row_half_mirror); applied in that order they form an 8-lane all-reduce butterfly.
template <int CTRL>
MD uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141;

MD uint32_t octet_min(uint32_t v) {
    uint32_t t = dpp_u32<DPP_XOR1>(v); v = t < v ? t : v;
    t = dpp_u32<DPP_XOR2>(v); v = t < v ? t : v;
    t = dpp_u32<DPP_HALF_MIRROR>(v); v = t < v ? t : v;
    return v;
}

MD uint32_t order_key(float d, int j) { return (__float_as_uint(d) & ~7u) | (uint32_t)j; }
MD uint32_t order_key(double d, int j) { return (__float_as_uint(__double2float_rd(d)) & ~7u) | (uint32_t)j; }

template <int CTRL>
MD void best_step(float& d, int64_t& i) {
    const float od = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(d)));
    const int oi = (int)dpp_u32<CTRL>((uint32_t)(int)i);
    if (od < d || (od == d && (int64_t)oi < i)) { d = od; i = oi; }
}
template <int CTRL>
MD void best_step(double& d, int64_t& i) {  // mesh search: only the distance matters
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    const uint64_t ob = ((uint64_t)dpp_u32<CTRL>((uint32_t)(b >> 32)) << 32) | dpp_u32<CTRL>((uint32_t)b);
    const double od = __longlong_as_double((long long)ob);
    if (od < d) d = od;
    (void)i;
}
// butterfly minimum over the octet; for the 6-d tree with the tie rule (smaller index wins)
template <typename T>
MD void octet_best(T& d, int64_t& i) {
    best_step<DPP_XOR1>(d, i);
    best_step<DPP_XOR2>(d, i);
    best_step<DPP_HALF_MIRROR>(d, i);
}

// One octet, one query.  `run` is octet-uniform; inactive octets fall through.  On entry (best, bi) is a
// valid candidate or (+inf, 0); on exit the exact minimum of the spec distance, ties to the smallest index.
// EXISTS: stop at the first point with d <= best (the entry bound); returns whether one was found.
template <class KD, bool EXISTS, bool STATS = false>
MD bool octet_search(const TreeView<KD>& tv, const typename KD::T* q, typename KD::T& best, int64_t& bi, bool run,
                     typename KD::T* cd, int* n_leaves = nullptr, int* n_nodes = nullptr) {
    using T = typename KD::T;
    static_assert(sizeof(typename KD::Box) == 48 && sizeof(typename KD::Point) == 32, "64-byte unified fetch");
    const int lane = threadIdx.x & 63, octet = lane >> 3, j = lane & 7;
    const int L = tv.levels;
    const int64_t leaf0 = level_offset_dev(L);
    int l = 0;
    int64_t n = 0;
    uint64_t cand = 0;
    bool found = false;
    bool enter = run && L > 0;   // fetch + test the children of n (level l)
    bool leaf = run && L == 0;   // fetch + scan leaf n
    while (__any(run)) {
        // one fetch per iteration whatever the octet is doing: this lane's child box (48 B) or its two
        // point slots (2 x 32 B) - 64 bytes from one base address, so a single wait covers both cases
        const uint4* src = reinterpret_cast<const uint4*>(tv.boxes);
        if (enter) src = reinterpret_cast<const uint4*>(tv.boxes + (8 * n + 1 + j));
        if (leaf) src = reinterpret_cast<const uint4*>(tv.pts + (size_t)(n - leaf0) * LEAF_CAP + 2 * j);
        uint4 r[4];
        if (run && (enter || leaf)) { r[0] = src[0]; r[1] = src[1]; r[2] = src[2]; r[3] = src[3]; }
        if (run) {
            if (enter) {
                typename KD::Box bx;
                __builtin_memcpy(&bx, r, sizeof(bx));
                const T d = box_dist2(q, bx);
                cd[l * 64] = d;
                if (STATS) ++*n_nodes;
                const uint32_t m = octet_bits(d <= best, octet);
                cand = (cand & ~(0xffull << (8 * l))) | ((uint64_t)m << (8 * l));
                enter = false;
            } else if (leaf) {
                typename KD::Point p0, p1;
                __builtin_memcpy(&p0, r, sizeof(p0));
                __builtin_memcpy(&p1, r + 2, sizeof(p1));
                if (STATS) ++*n_leaves;
                T d0 = dist2(q, p0), d1 = dist2(q, p1);
                int64_t i0 = p0.idx, i1 = p1.idx;
                if (d1 < d0 || (d1 == d0 && i1 < i0)) { d0 = d1; i0 = i1; }
                if (!(d0 == d0)) { d0 = INFINITY; i0 = 0x7fffffff; }  // NaN never wins
                octet_best(d0, i0);
                if (EXISTS) {
                    if (d0 <= best) { best = d0; bi = i0; found = true; run = false; }
                } else if (d0 < best || (d0 == best && i0 < bi)) {
                    best = d0;
                    bi = i0;
                }
                leaf = false;
                if (L == 0) run = false;
                else n = (n - 1) >> 3;  // back to the parent; l already points at it
            }
        }
        if (run) {
            // next child at level l: the nearest still-alive candidate
            const T d = cd[l * 64];
            const bool alive = ((cand >> (8 * l + j)) & 1ull) && (d <= best);
            const uint32_t m = octet_bits(alive, octet);
            if (m == 0) {
                if (l == 0) run = false;
                else { --l; n = (n - 1) >> 3; }
            } else {
                const uint32_t jm = octet_min(alive ? order_key(d, j) : 0xffffffffu) & 7u;
                cand = (cand & ~(0xffull << (8 * l))) | ((uint64_t)(m & ~(1u << jm)) << (8 * l));
                n = 8 * n + 1 + jm;
                if (l + 1 == L) leaf = true;
                else { ++l; enter = true; }
            }
        }
    }
    return found;
}

// Wave-level driver: lanes with `need` set get their query served by an octet, eight queries per round.
template <class KD, bool EXISTS, bool STATS = false>
MD bool wave_search(const TreeView<KD>& tv, const typename KD::T* q, typename KD::T& best, int64_t& bi, bool need,
                    typename KD::T* cd_base, int* n_leaves = nullptr, int* n_nodes = nullptr) {
    using T = typename KD::T;
    constexpr int DIM = KD::DIM;
    const int lane = threadIdx.x & 63, octet = lane >> 3;
    uint64_t todo = __ballot(need);
    bool found = false;
    while (todo) {
        // owner of this octet = the octet-th set bit of todo
        uint64_t t = todo;
        int owner = -1;
        for (int k = 0; k <= octet && t; ++k) {
            owner = k == octet ? (int)__builtin_ctzll(t) : -1;
            t &= t - 1;
        }
        const bool active = owner >= 0;
        const int src = active ? owner : lane;
        T qq[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) qq[d] = __shfl(q[d], src);
        T b = __shfl(best, src);
        int64_t i = (int64_t)__shfl((long long)bi, src);
        int nl = 0, nn = 0;
        const bool f = octet_search<KD, EXISTS, STATS>(tv, qq, b, i, active, cd_base + lane, &nl, &nn);
        // hand the result back to the owner lanes
        const int rank = (int)__builtin_popcountll(todo & ((1ull << lane) - 1ull));
        const bool served = ((todo >> lane) & 1ull) && rank < 8;
        const int from = 8 * (rank < 8 ? rank : 0);
        const T rb = __shfl(b, from);
        const int64_t ri = (int64_t)__shfl((long long)i, from);
        const int rf = __shfl((int)f, from);
        const int rl = __shfl(nl, from), rn = __shfl(nn, from);
        if (served) {
            best = rb;
            bi = ri;
            found = rf != 0;
            if (STATS) { *n_leaves += rl; *n_nodes += rn; }
        }
        for (int k = 0; k < 8 && todo; ++k) todo &= todo - 1;
    }
    return found;
}

// Hint fast path.  h = a codebook entry near q (the NN of the particle's ancestor).  The candidates
// {h} U N(h) are scanned in order of rho = |F_h - F_s|; every entry not yet scanned is at least
// rho - |q - F_h| away from q (triangle inequality), so once that exceeds the best distance found the
// search is certified complete and returns the exact answer of the full search.  The comparison
// carries a 3e-5 relative margin on squared distances, two orders above float32 rounding of the
// six-term sums, so "certified" also holds for the COMPUTED distances and their tie rule.
// Returns true when certified; otherwise (best, bi) is a valid bound for the tree search.
#ifndef MIDAS_NN_BATCH
#define MIDAS_NN_BATCH 8
#endif
#ifndef MIDAS_MESH_BATCH
#define MIDAS_MESH_BATCH 8
#endif
#ifndef MIDAS_NN_SOLO
#define MIDAS_NN_SOLO 32
#endif
#ifndef MIDAS_MESH_SOLO
#define MIDAS_MESH_SOLO 16
#endif
constexpr int NN_BATCH = MIDAS_NN_BATCH, MESH_BATCH = MIDAS_MESH_BATCH;  // records per round trip of the per-lane scans
constexpr int NN_SOLO = MIDAS_NN_SOLO;      // records a lane scans by itself before the wave takes over its list
constexpr int MESH_SOLO = MIDAS_MESH_SOLO;
static_assert(NN_SOLO % NN_BATCH == 0 && NBR_M % NN_BATCH == 0 && MESH_SOLO % MESH_BATCH == 0 && MESH_M % MESH_BATCH == 0,
              "scan batches must tile the solo prefixes and the lists");

// ---- half-record screening ------------------------------------------------------------------------------------------
// A record is two 16-byte pieces: lo = c[0..3], hi = {c[4], c[5], idx, rho}.  dist2 adds the six squared differences
// in order, every step an fma onto the sum so far, so the sum after four terms P4 is a LOWER bound of the finished
// distance in the computed arithmetic (adding a non-negative term and rounding never lowers a sum).  A record whose
// P4 already exceeds the best distance can neither win nor tie: its second piece is not fetched at all.  On the
// bench workloads 1 - 6 of 32 records pass the screen (the lists are sorted by 6-d distance from the entry, and most
// of a neighbour's offset is in the translation), so a scan issues about half the loads - and the particle kernels
// are bound by the number of scattered 16-byte loads a CU's vector cache can look up, not by bytes.
// The certificate needs rho only once per batch: of the batch's last record (the largest; everything behind is farther).
#ifndef MIDAS_SCREEN
#define MIDAS_SCREEN 1
#endif
MD float part4(const float* q, const float4& lo) {
    const float d0 = q[0] - lo.x, d1 = q[1] - lo.y, d2 = q[2] - lo.z, d3 = q[3] - lo.w;
    float d = d0 * d0;
    d = fmaf_(d1, d1, d);
    d = fmaf_(d2, d2, d);
    d = fmaf_(d3, d3, d);
    return d;
}
MD float full_from(const float* q, float p4, const float4& hi) {  // == dist2(q, record), bit for bit
    const float d4 = q[4] - hi.x, d5 = q[5] - hi.y;
    float d = fmaf_(d4, d4, p4);
    d = fmaf_(d5, d5, d);
    return d;
}
// P[j] for a per-lane j as a chain of selects on registers (the empty asm keeps the compiler from turning the chain
// back into an indexed array, which it would put in scratch memory)
template <int B>
MD float pick(const float* P, int j) {
    float v = P[0];
#pragma unroll
    for (int k = 1; k < B; ++k) {
        v = j == k ? P[k] : v;
        asm volatile("" : "+v"(v));
    }
    return v;
}
template <int B>
MD int pick(const int* P, int j) {
    int v = P[0];
#pragma unroll
    for (int k = 1; k < B; ++k) {
        v = j == k ? P[k] : v;
        asm volatile("" : "+v"(v));
    }
    return v;
}

// Scans records [0, NN_SOLO) of entry h's list (record 0 = the entry itself, fetched whole with the first batch so that
// r = |q - F_h| costs no round trip of its own).  Per batch: the first pieces of its records and the second piece of
// its last one in one round trip; then the second pieces of up to two records that pass the screen in another (the
// lines are in the vector cache by then); a lane with more takes them one at a time (rare).
template <bool FIRST>
MD void nn6_hint_batch(const float4* __restrict__ nb4, int s0, const float* q, int32_t h, float& best, int64_t& bi, float& r,
                       float& rslack, int& scanned, bool& certified) {
    float4 lo[NN_BATCH];
#pragma unroll
    for (int j = 0; j < NN_BATCH; ++j) lo[j] = nb4[2 * (s0 + j)];
    const float4 hl = nb4[2 * (s0 + NN_BATCH - 1) + 1];
    float P[NN_BATCH];
#pragma unroll
    for (int j = 0; j < NN_BATCH; ++j) P[j] = part4(q, lo[j]);
    if (FIRST) {  // the entry itself: the starting candidate (a NaN distance stays, as in a serial scan)
        const float4 h0 = nb4[1];
        best = full_from(q, P[0], h0);
        bi = h;
        r = __builtin_sqrtf(best);
        rslack = -8e-7f * r;
    }
    unsigned mask = 0;
#pragma unroll
    for (int j = FIRST ? 1 : 0; j < NN_BATCH; ++j) mask |= (P[j] <= best ? 1u : 0u) << j;
    scanned += __popc(mask);
    const unsigned m1 = mask & ~(1u << (NN_BATCH - 1)), m2 = m1 & (m1 - 1u);
    // two second pieces, unconditionally (a lane without a candidate re-reads a piece it holds: conditional loads would
    // be waited for one at a time)
    const int j1 = m1 ? __builtin_ctz(m1) : NN_BATCH - 1, j2 = m2 ? __builtin_ctz(m2) : NN_BATCH - 1;
    const float4 ha = nb4[2 * (s0 + j1) + 1];
    const float4 hb = nb4[2 * (s0 + j2) + 1];
    // candidate updates as selects (the short-circuit form compiled to exec-mask regions: slower, removed in round 6)
    int b32 = (int)bi;  // list indices are int32
    auto take = [&](float p4, const float4& hi, bool on) {  // no short circuits: selects instead of exec-mask regions
        const float d = full_from(q, p4, hi);
        const int32_t id = __float_as_int(hi.z);
        const bool better = on & ((d < best) | ((d == best) & (id < b32)));
        best = better ? d : best;
        b32 = better ? id : b32;
    };
    take(pick<NN_BATCH>(P, j1), ha, m1 != 0);
    take(pick<NN_BATCH>(P, j2), hb, m2 != 0);
    take(P[NN_BATCH - 1], hl, (mask >> (NN_BATCH - 1)) != 0);
    unsigned rest = m2 & (m2 - 1u);
    while (rest) {  // more than two candidates among the batch's first records
        const int j = __builtin_ctz(rest);
        rest &= rest - 1u;
        take(pick<NN_BATCH>(P, j), nb4[2 * (s0 + j) + 1], true);
    }
    bi = b32;
    // every record behind this batch is at least this far (lower bound of |q - F| with slack for the rounding of r, rho)
    const float g = fmaf_(hl.w - r, 0.9999996f, rslack);
    certified = g > 0.0f && g * g * 0.99997f > best;
}

// Pivot switch across the rotation-angle-pi cut.  The feature's rotation part is 0.01 log(R): a particle whose rotation angle
// passes pi reappears 2 pi 0.01 = 63 mm-equivalents away from its ancestor's nearest entry, and no list of that entry can
// certify anything for it - the lane would walk all NBR_M records (eight cooperative passes of cold fetches) before the
// twin entry gets its turn.  With uniformly distributed yaws about 0.5 % of the particles of a spread cloud cross the cut in a
// frame, i.e. every second wave has such a lane and ends 30 us after the others (phase clocks of the diffuse regime,
// profiles/r03_diffuse_*).  So: a lane that finds itself farther than FLIP_R from the hinted entry (nothing near an entry is:
// codebook spacings are millimetres) continues from the entry's TWIN - the entry nearest to the hinted one's image across the
// cut - whose index travels with the first batch.  Any pivot is a correct pivot (the certificate is relative to the list
// scanned, the continuation and the tree search stay behind it), so this changes which records are read, never the answer.
constexpr float FLIP_R = 0.02f;
MD bool nn6_hint_scan_screened(const TreeView<Kd6>& tv, const float* q, int32_t& h, float& best, int64_t& bi, int* n_scanned,
                               float* r_out = nullptr) {
    const float4* __restrict__ nb4 = reinterpret_cast<const float4*>(tv.nbrs + (size_t)h * NBR_REC);
    float r = 0.f, rslack = 0.f;
    int scanned = 0;
    bool certified = false;
    const int32_t tw = tv.twin[h];  // with the first batch: behind it, it would be a round trip of its own
    nn6_hint_batch<true>(nb4, 0, q, h, best, bi, r, rslack, scanned, certified);
    if (!certified && r > FLIP_R && tw >= 0) {
        h = tw;
        nb4 = reinterpret_cast<const float4*>(tv.nbrs + (size_t)h * NBR_REC);
        scanned = 0;
        nn6_hint_batch<true>(nb4, 0, q, h, best, bi, r, rslack, scanned, certified);
    }
#pragma unroll 1
    for (int s0 = NN_BATCH; s0 < NN_SOLO && !certified; s0 += NN_BATCH)
        nn6_hint_batch<false>(nb4, s0, q, h, best, bi, r, rslack, scanned, certified);
    if (r_out) *r_out = r;
    if (n_scanned) *n_scanned = scanned;
    return certified;
}

// The unscreened form (whole records, one round trip per batch): what the batch step and the largest particle sets run -
// there the waves are many and short of registers, and a batch in two round trips costs more than the loads it saves
// (c5: 353 -> 376 us per batch frame with the screen, c2's front 32.2 -> 30.8 us).
// Scans records [0, NN_SOLO) of entry h's list (record 0 = the entry itself); the first batch is fetched
// together with the entry so that r = |q - F_h| costs no round trip of its own.
// (measured and dropped: a greedy hop to a closer entry's list - no effect at c2, hints are rarely stale; none at c5 either (round 6:
// 278 us per batch frame with and without): there the nearest entry is about as far as the hinted one - the feature's rotation part
// spreads the entries over five dimensions - so no pivot shortens the proof, 47 of a wave's 64 lanes go on to nn6_coop either way)
MD bool nn6_hint_scan(const TreeView<Kd6>& tv, const float* q, int32_t& h, float& best, int64_t& bi, int* n_scanned,
                      float* r_out = nullptr) {
    const Nbr6* nb = tv.nbrs + (size_t)h * NBR_REC;
    float r = 0.f, rslack = 0.f;
    int scanned = 0;
    int b32 = h;  // record 0 overwrites the incoming candidate; list indices are int32
    bool certified = false;
    int32_t tw = tv.twin[h];  // pivot switch across the angle-pi cut (see nn6_hint_scan_screened); once
#pragma unroll 1
    for (int s0 = 0; s0 < NN_SOLO && !certified; s0 += NN_BATCH) {
        if (s0 == NN_BATCH && r > FLIP_R && tw >= 0) {  // far from the hinted entry: its twin's list from the start
            h = tw;
            tw = -1;
            b32 = h;
            nb = tv.nbrs + (size_t)h * NBR_REC;
            s0 = 0;
            scanned = 0;
        }
        Nbr6 e[NN_BATCH];
#pragma unroll
        for (int j = 0; j < NN_BATCH; ++j) e[j] = nb[s0 + j];
#pragma unroll
        for (int j = 0; j < NN_BATCH; ++j) {
            Point6 p;
#pragma unroll
            for (int a = 0; a < 6; ++a) p.c[a] = e[j].c[a];
            const float d = dist2(q, p);
            if (s0 == 0 && j == 0) {  // the entry itself: the starting candidate (a NaN distance stays, as in a serial scan)
                best = d;
                r = __builtin_sqrtf(best);
                rslack = -8e-7f * r;
            } else {
                // Serial semantics without branches (the compiler turned the nested conditions into exec-mask regions, ~25
                // scalar / mask instructions per record): a record counts while no earlier one has certified;
                // lower bound of |q - F| for this and every later record, with slack for the rounding of r and rho
                const float g = fmaf_(e[j].rho - r, 0.9999996f, rslack);
                certified |= (g > 0.0f) & (g * g * 0.99997f > best);
                const bool better = !certified & ((d < best) | ((d == best) & (e[j].idx < b32)));
                best = better ? d : best;
                b32 = better ? e[j].idx : b32;
                scanned += certified ? 0 : 1;
            }
        }
    }
    bi = b32;
    if (r_out) *r_out = r;
    if (n_scanned) *n_scanned = scanned;
    return certified;
}


// ---- wave-cooperative continuation of the list scans ------------------------------------------------
// Most lanes certify inside their first batch of records; the few that do not used to walk the rest of
// their list alone (up to 32 dependent round trips) while 60 lanes idled.  Here the whole wave serves
// them one at a time: 64 records per round trip, one per lane, reduced with the exact (distance, index)
// tie rule.  Any evaluated candidate bounds the answer from above, so after a chunk the proof is the same
// triangle-inequality test on the chunk's LAST record (largest rho): everything beyond it is farther.
MD float rl_f32(float v, int lane) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane)); }
MD int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// minimum over the wave of (d, idx) with ties to the smaller idx; result uniform
MD void wave_best(float& d, int& i) {
#define MIDAS_STEP(CTRL)                                                              \
    {                                                                                 \
        const float od = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(d)));          \
        const int oi = (int)dpp_u32<CTRL>((uint32_t)i);                                \
        const bool ob = (od < d) | ((od == d) & (oi < i));                            \
        d = ob ? od : d;                                                               \
        i = ob ? oi : i;                                                               \
    }
    MIDAS_STEP(DPP_XOR1) MIDAS_STEP(DPP_XOR2) MIDAS_STEP(DPP_HALF_MIRROR) MIDAS_STEP(0x140 /* row_mirror */)
#undef MIDAS_STEP
    float bd = rl_f32(d, 0);
    int bi = rl_i32(i, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float od = rl_f32(d, r);
        const int oi = rl_i32(i, r);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    d = bd;
    i = bi;
}

// one full cooperative scan of entry h's list for the query of lane `owner`, records [first, NBR_M];
// returns certified; (bb, bi) in/out uniform
MD bool coop_scan_list(const TreeView<Kd6>& tv, const float* qq, int h, int first, float rr, float& bb, int& bi) {
    const int lane = threadIdx.x & 63;
    const Nbr6* nb = tv.nbrs + (size_t)h * NBR_REC;
    const float rslack = -8e-7f * rr;
    for (int c0 = first; c0 <= NBR_M; c0 += 64) {
        const int s = c0 + lane;
        float d = INFINITY, rho = INFINITY;
        int id = 0x7fffffff;
        if (s <= NBR_M) {
            const Nbr6 e = nb[s];
            Point6 p;
#pragma unroll
            for (int a = 0; a < 6; ++a) p.c[a] = e.c[a];
            d = dist2(qq, p);
            if (!(d == d)) d = INFINITY;
            id = e.idx;
            rho = e.rho;
        }
        wave_best(d, id);
        if (d < bb || (d == bb && id < bi)) { bb = d; bi = id; }
        const int last = (c0 + 63 <= NBR_M ? c0 + 63 : NBR_M) - c0;
        const float g = fmaf_(rl_f32(rho, last) - rr, 0.9999996f, rslack);
        if (g > 0.0f && g * g * 0.99997f > bb) return true;
    }
    const float g = fmaf_(tv.rho_out[h] - rr, 0.9999996f, rslack);
    return g > 0.0f && g * g * 0.99997f > bb;
}

// serve the lanes in `need`: continue their hint scan after the solo records, then try the twin entry.
// minimum over a 16-lane row of (d, idx), ties to the smaller idx; every lane of the row gets the result
MD void row_best(float& d, int& i) {
#define MIDAS_STEP(CTRL)                                                              \
    {                                                                                 \
        const float od = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(d)));          \
        const int oi = (int)dpp_u32<CTRL>((uint32_t)i);                                \
        const bool ob = (od < d) | ((od == d) & (oi < i));                            \
        d = ob ? od : d;                                                               \
        i = ob ? oi : i;                                                               \
    }
    MIDAS_STEP(DPP_XOR1) MIDAS_STEP(DPP_XOR2) MIDAS_STEP(DPP_HALF_MIRROR) MIDAS_STEP(0x140 /* row_mirror */)
#undef MIDAS_STEP
}

// COOP_G owners at a time, one per group of 64 / COOP_G lanes: the group walks the next 64 records of its owner's list in
// 64 / L steps of L (all the loads of a lane in flight together) and reduces inside the group with DPP - the owners are
// evaluated by the same instructions, where the whole-wave form spent them once per owner.  The certificate is the
// one of the 64-record chunk (its last record's rho against the final best); an owner it does not settle comes back
// in the next pass with its next 64 records, until its list is exhausted (then: the list's outer radius, the twin).
// A wave of c2 has ~10 open owners (up to ~20): with four per pass (16-lane rows) that was three to five dependent
// round trips, with eight it is two or three.
// (Looking at the stamps only after the prune was measured twice - round 3: front 34 -> 45 us, round 5 with the prediction
// list: 29.7k -> 29.2k steps/s - the particle waves run in lock step, so with the look deferred nearly every wave still
// finds the old stamps and exchanges.  The claim is looked at where it is issued.)
#ifndef MIDAS_SCORE_ROUNDS
#define MIDAS_SCORE_ROUNDS 2  // quads of codebook rows a scoring wave of the fused front streams (dense scoring)
#endif
#ifndef MIDAS_CLAIM_HASH
#define MIDAS_CLAIM_HASH 1  // leaders of the row claims through an LDS hash table (score_body.hpp claim_rows_issue); 0: ballot rounds
#endif
#ifndef MIDAS_COOP_G
#define MIDAS_COOP_G 8
#endif
// records an owner gets per pass (64: eight steps of eight lanes; 32 halves the records fetched past the certificate
// on codebooks where a typical list needs 40 - 60 of them)
#ifndef MIDAS_COOP_CHUNK
#define MIDAS_COOP_CHUNK 64
#endif
// (measured and dropped: piece-contiguous fetches of the group, DESIGN.md notebook "MIDAS_COOP_PIECES")
constexpr int COOP_G = MIDAS_COOP_G, COOP_L = 64 / COOP_G, COOP_CHUNK = MIDAS_COOP_CHUNK, COOP_STEPS = COOP_CHUNK / COOP_L;
static_assert(COOP_G == 4 || COOP_G == 8 || COOP_G == 16, "owners per pass");
static_assert(COOP_STEPS >= 1 && COOP_STEPS * COOP_L == COOP_CHUNK, "a chunk is whole steps of the group");
// minimum over a group of COOP_L lanes of (d, idx), ties to the smaller idx; every lane of the group gets the result
MD void group_best(float& d, int& i) {
#define MIDAS_STEP(CTRL)                                                              \
    {                                                                                 \
        const float od = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(d)));          \
        const int oi = (int)dpp_u32<CTRL>((uint32_t)i);                                \
        const bool ob = (od < d) | ((od == d) & (oi < i));                            \
        d = ob ? od : d;                                                               \
        i = ob ? oi : i;                                                               \
    }
    MIDAS_STEP(DPP_XOR1) MIDAS_STEP(DPP_XOR2)
    if (COOP_L >= 8) MIDAS_STEP(DPP_HALF_MIRROR)
    if (COOP_L >= 16) MIDAS_STEP(0x140 /* row_mirror */)
#undef MIDAS_STEP
}

template <bool SCREEN = false>
MD void nn6_coop(const TreeView<Kd6>& tv, const float* q, int32_t hint, float r_lane, float& best, int64_t& bi, bool need,
                 bool& done) {
    const int lane = threadIdx.x & 63, grp = lane / COOP_L, j = lane % COOP_L;
    int nrec = NN_SOLO;  // next record of this lane's list (owners only)
    // pass after pass: every open owner gets its next 64 records, COOP_G owners per instruction stream
    for (;;) {
        const bool open_lane = need && !done && nrec <= NBR_M;
        unsigned long long todo = __ballot(open_lane);
        if (!todo) break;
        const int my_rank = (int)__builtin_popcountll(todo & ((1ull << lane) - 1ull));  // rank among this pass's owners
        int served = 0;
        while (todo != 0) {
            int mine = -1;
#pragma unroll
            for (int k = 0; k < COOP_G; ++k) {  // (the owners' lane numbers through LDS instead of these scalar steps: no change, dropped)
                const int o = todo ? (int)__builtin_ctzll(todo) : -1;
                todo &= todo - 1;  // 0 & anything stays 0
                mine = grp == k ? o : mine;
            }
            const int src = mine >= 0 ? mine : lane;
            float qq[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) qq[d] = __shfl(q[d], src);
            const float rr = __shfl(r_lane, src);
            float bb = __shfl(best, src);
            int b_i = __shfl((int)bi, src);
            // (shuffles stay unconditional: a lane outside the branch could not serve as a source)
            const int hh_s = __shfl(hint, src), first_s = __shfl(nrec, src);
            const int hh = mine >= 0 ? hh_s : 0;
            const int first = mine >= 0 ? first_s : 0;  // records first .. first+COOP_CHUNK-1, clamped to the list
            float d = INFINITY, rho_last = 0.f;
            int id = 0x7fffffff;
            if (SCREEN) {
            // half-record screening (see part4): first pieces of the lane's records and the second piece of its last one,
            // then the second pieces of the records whose partial distance does not exceed the owner's best
            const float4* __restrict__ nb4 = reinterpret_cast<const float4*>(tv.nbrs + (size_t)hh * NBR_REC);
            float4 lo[COOP_STEPS];
            int sc[COOP_STEPS];
#pragma unroll
            for (int m = 0; m < COOP_STEPS; ++m) {
                const int s = first + COOP_L * m + j;
                sc[m] = s <= NBR_M ? s : NBR_M;
                lo[m] = nb4[2 * sc[m]];
            }
            const float4 hl = nb4[2 * sc[COOP_STEPS - 1] + 1];
            float P[COOP_STEPS];
            unsigned mask = 0;
#pragma unroll
            for (int m = 0; m < COOP_STEPS; ++m) {
                P[m] = part4(qq, lo[m]);
                const bool in = first + COOP_L * m + j <= NBR_M;
                mask |= (in && P[m] <= bb ? 1u : 0u) << m;
            }
            const unsigned m1 = mask & ~(1u << (COOP_STEPS - 1)), m2 = m1 & (m1 - 1u);
            const int k1 = m1 ? __builtin_ctz(m1) : COOP_STEPS - 1, k2 = m2 ? __builtin_ctz(m2) : COOP_STEPS - 1;
            const int s1 = pick<COOP_STEPS>(sc, k1), s2 = pick<COOP_STEPS>(sc, k2);
            const float4 ha = nb4[2 * s1 + 1];
            const float4 hb = nb4[2 * s2 + 1];
            auto take = [&](float p4, const float4& hi, bool on) {
                const float dm = full_from(qq, p4, hi);
                const int im = __float_as_int(hi.z);
                const bool better = on & ((dm < d) | ((dm == d) & (im < id)));  // NaN never wins; selects, no branches
                d = better ? dm : d;
                id = better ? im : id;
            };
            take(pick<COOP_STEPS>(P, k1), ha, m1 != 0);
            take(pick<COOP_STEPS>(P, k2), hb, m2 != 0);
            take(P[COOP_STEPS - 1], hl, (mask >> (COOP_STEPS - 1)) != 0);
            unsigned rest = m2 & (m2 - 1u);
            while (rest) {
                const int k = __builtin_ctz(rest);
                rest &= rest - 1u;
                take(pick<COOP_STEPS>(P, k), nb4[2 * pick<COOP_STEPS>(sc, k) + 1], true);
            }
            rho_last = hl.w;  // of this lane's last record: the group's last lane holds the chunk's last (when the chunk is whole)
            } else {
            const Nbr6* nb = tv.nbrs + (size_t)hh * NBR_REC;
            Nbr6 e[COOP_STEPS];
#pragma unroll
            for (int m = 0; m < COOP_STEPS; ++m) {
                const int s = first + COOP_L * m + j;
                e[m] = nb[s <= NBR_M ? s : NBR_M];
            }
#pragma unroll
            for (int m = 0; m < COOP_STEPS; ++m) {
                Point6 p;
#pragma unroll
                for (int a = 0; a < 6; ++a) p.c[a] = e[m].c[a];
                const float dm = dist2(qq, p);
                const bool in = first + COOP_L * m + j <= NBR_M;
                const bool better = in & ((dm < d) | ((dm == d) & (e[m].idx < id)));  // NaN never wins; no short circuits: no branches
                d = better ? dm : d;
                id = better ? e[m].idx : id;
                rho_last = in ? e[m].rho : rho_last;
            }
            }
            group_best(d, id);
            {
                const bool gb = (d < bb) | ((d == bb) & (id < b_i));
                bb = gb ? d : bb;
                b_i = gb ? id : b_i;
            }
            // largest rho scanned = the last valid record of the chunk (clamped loads repeat the list's last record);
            // once the list is exhausted the bound is the distance of the first entry NOT in it
            rho_last = __shfl(rho_last, lane | (COOP_L - 1));
            const bool at_end = first + COOP_CHUNK - 1 >= NBR_M;
            const float bound = at_end ? tv.rho_out[hh] : rho_last;
            const float gg = fmaf_(bound - rr, 0.9999996f, -8e-7f * rr);
            const bool cert = gg > 0.0f && gg * gg * 0.99997f > bb;
            // hand the groups' results to the owners: the owner with rank r among this pass sits in group r - served
            const int from = COOP_L * ((my_rank - served) & (COOP_G - 1));
            const float rb = __shfl(bb, from);
            const int ri = __shfl(b_i, from);
            const int rc = __shfl((int)cert, from);
            if (open_lane && my_rank >= served && my_rank < served + COOP_G) { best = rb; bi = ri; done = rc != 0; nrec += COOP_CHUNK; }
            served += COOP_G;
        }
    }
    // lists exhausted without a certificate: second chance from the entry across the angle-pi cut, whole wave (rare)
    unsigned long long open = __ballot(need && !done);
    while (open) {
        const int o = (int)__builtin_ctzll(open);
        open &= open - 1;
        const int h1 = rl_i32(hint, o);
        const int tw = tv.twin[h1];
        if (tw < 0) continue;
        float q1[6];
#pragma unroll
        for (int dd = 0; dd < 6; ++dd) q1[dd] = rl_f32(q[dd], o);
        float b1 = rl_f32(best, o);
        int i1 = rl_i32((int)bi, o);
        const Nbr6 ts = tv.nbrs[(size_t)tw * NBR_REC];  // record 0 = the twin itself
        Point6 pt;
#pragma unroll
        for (int a = 0; a < 6; ++a) pt.c[a] = ts.c[a];
        const float r2 = __builtin_sqrtf(dist2(q1, pt));
        const bool c1 = coop_scan_list(tv, q1, tw, 0, r2, b1, i1);
        if (lane == o) { best = b1; bi = i1; done = c1; }
    }
}

// prune: continue the vertex-list scan of the lanes in `need` (mv < 0 after their solo records); owners in
// groups of COOP_G as above, `lim` (the "cannot be within thr" radius) comes from the owner lane
MD double rl_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)rl_i32((int)(unsigned)b, lane), hi = (unsigned)rl_i32((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

MD double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__shfl((int)(unsigned)b, src), hi = (unsigned)__shfl((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Four owners at a time, one per 16-lane row, records MESH_SOLO+1 .. MESH_SOLO+64 of each owner's vertex list in four
// steps of 16 (loads in flight together).  Serial semantics inside a row: records in order, the first event decides
// ("provably too far" before "hit" on the same record).  An owner the 64 records do not settle continues with the
// whole wave, 64 records per step.
MD void mesh_coop(const MeshRec* __restrict__ vlist, int32_t h, const double* tq, double t2, double lim_lane, bool need, int& mv) {
    const int lane = threadIdx.x & 63, row = lane >> 4, j = lane & 15;
    unsigned long long todo = __ballot(need);
    const int my_rank = (int)__builtin_popcountll(todo & ((1ull << lane) - 1ull));
    int served = 0;
    while (todo) {
        int owner[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            owner[k] = todo ? (int)__builtin_ctzll(todo) : -1;
            todo &= todo - 1;
        }
        const int mine = row == 0 ? owner[0] : row == 1 ? owner[1] : row == 2 ? owner[2] : owner[3];
        const int src = mine >= 0 ? mine : lane;
        double q3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) q3[d] = shfl_f64(tq[d], src);
        const double lim = shfl_f64(lim_lane, src);
        const int hh = __shfl(h, src);
        const MeshRec* vl = vlist + (size_t)(mine >= 0 ? hh : 0) * MESH_REC + (1 + MESH_SOLO) + j;
        MeshRec e[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) e[m] = vl[16 * m];
        int res = -1;  // row-uniform
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            Point3 p;
            p.c[0] = e[m].c[0]; p.c[1] = e[m].c[1]; p.c[2] = e[m].c[2];
            const bool hit = dist2(q3, p) <= t2;
            const bool stop = (double)e[m].rho * (1.0 - 1e-7) > lim;
            const unsigned hits = (unsigned)(__ballot(hit) >> (16 * row)) & 0xffffu;
            const unsigned stops = (unsigned)(__ballot(stop) >> (16 * row)) & 0xffffu;
            const int fh = hits ? __builtin_ctz(hits) : 16, fs = stops ? __builtin_ctz(stops) : 16;
            if (res < 0) {
                if (fh < 16 && fh < fs) res = 1;
                else if (fs < 16) res = 0;
            }
        }
        const int from = 16 * ((my_rank - served) & 3);
        const int rres = __shfl(res, from);
        const bool in_group = need && my_rank >= served && my_rank < served + 4;
        if (in_group) mv = rres;
        served += 4;
        // owners the 64 records did not settle (rare): the rest of the list with the whole wave, then the list's outer radius
        unsigned long long open = __ballot(in_group && mv < 0);
        while (open) {
            const int o = (int)__builtin_ctzll(open);
            open &= open - 1;
            double q1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) q1[d] = rl_f64(tq[d], o);
            const double lim1 = rl_f64(lim_lane, o);
            const MeshRec* v1 = vlist + (size_t)rl_i32(h, o) * MESH_REC;
            int r1 = -1;
            for (int c0 = 1 + MESH_SOLO + 64; c0 <= MESH_M && r1 < 0; c0 += 64) {
                const int s = c0 + lane;
                bool hit = false;
                // (lanes past the end of the list decide nothing: with +inf here they "stopped" the scan, and a list that was
                // merely EXHAUSTED - a dense mesh, the particle 2 mm off its entry - read as "provably too far" instead of going
                // to the tree search; two particles in 640 000 at c5, found by the exhaustive check of round 6)
                float rho = -INFINITY;
                if (s <= MESH_M) {
                    const MeshRec r = v1[s];
                    Point3 p;
                    p.c[0] = r.c[0]; p.c[1] = r.c[1]; p.c[2] = r.c[2];
                    rho = r.rho;
                    hit = dist2(q1, p) <= t2;
                }
                const unsigned long long hits = __ballot(hit);
                const unsigned long long stops = __ballot((double)rho * (1.0 - 1e-7) > lim1);
                const int fh = hits ? (int)__builtin_ctzll(hits) : 64, fs = stops ? (int)__builtin_ctzll(stops) : 64;
                if (fh < 64 && fh < fs) r1 = 1;
                else if (fs < 64) r1 = 0;
            }
            if (r1 < 0) r1 = ((double)v1[0].rho * (1.0 - 1e-7) > lim1) ? 0 : -1;
            if (lane == o) mv = r1;
        }
    }
}


// Prune fast path: decide "some mesh vertex within thr of tq" from the vertex list of the particle's NN
// entry h.  Returns 1 (valid: an actual vertex passes the exact test d2 <= t2), 0 (invalid: every vertex not
// yet scanned is provably farther than thr, triangle inequality with slack far above float64 rounding) or
// -1 (list exhausted: the caller runs the tree search).
// PRE: header and first batch (records 0 .. MESH_BATCH) were fetched by the caller ahead of time (registers: a compile-time
// choice - a pointer that may or may not refer to them would put them in scratch memory)
template <bool PRE = false>
MD int mesh_list_check(const MeshRec* __restrict__ vlist, int32_t h, const double* tq, double t2, double thr,
                       int max_records = MESH_M, double* lim_out = nullptr, const MeshRec* pre = nullptr) {
    const MeshRec* vl = vlist + (size_t)h * MESH_REC;
    const MeshRec hd = PRE ? pre[0] : vl[0];
    Point3 ph;
    ph.c[0] = hd.c[0]; ph.c[1] = hd.c[1]; ph.c[2] = hd.c[2];
    const double delta = __builtin_sqrt(dist2(tq, ph)) * (1.0 + 1e-12);
    const double lim = thr * (1.0 + 1e-9) + delta + 1e-12;  // a vertex with rho*(1-1e-7) > lim cannot be within thr of tq
    if (lim_out) *lim_out = lim;
    // a batch is evaluated branch-free (all its loads in one round trip), then resolved in record order:
    // per record "provably too far" is tested before "hit", so the first event decides
    for (int s0 = 1; s0 <= max_records; s0 += MESH_BATCH) {
        MeshRec e[MESH_BATCH];
        if (PRE && s0 == 1) {
#pragma unroll
            for (int j = 0; j < MESH_BATCH; ++j) e[j] = pre[1 + j];
        } else {
#pragma unroll
            for (int j = 0; j < MESH_BATCH; ++j) e[j] = vl[s0 + j];
        }
        unsigned hits = 0, stops = 0;
#pragma unroll
        for (int j = 0; j < MESH_BATCH; ++j) {
            Point3 p;
            p.c[0] = e[j].c[0]; p.c[1] = e[j].c[1]; p.c[2] = e[j].c[2];
            stops |= ((double)e[j].rho * (1.0 - 1e-7) > lim ? 1u : 0u) << j;
            hits |= (dist2(tq, p) <= t2 ? 1u : 0u) << j;
        }
        if (hits | stops) {
            const int fh = hits ? __builtin_ctz(hits) : 32, fs = stops ? __builtin_ctz(stops) : 32;
            return fh < fs ? 1 : 0;
        }
    }
    if (max_records < MESH_M) return -1;
    return ((double)hd.rho * (1.0 - 1e-7) > lim) ? 0 : -1;
}

// The same decision from the float32 screening copy of the list (half the bytes per record - the particle kernels are bound
// by the bytes their scattered loads move through the vector cache, tools/probes/ta_probe.hip - and float32 instead of
// float64 arithmetic).  tq is a float32 value already (a pose entry) and so is the header; a vertex v was rounded to
// nearest, |v_f - v| <= 2^-24 |v| per coordinate, and |v| <= |tq| + d, so the true distance d and the one between the
// float32 points d~ satisfy |d - d~| <= E + 1.1e-7 d~ with E = 2.5e-7 (|tq_x| + |tq_y| + |tq_z|); the computed squared
// distance is within 4e-7 (relative) of d~^2.  Hence, with 4e-6 of relative slack on the squares:
//   d2f <= (thr - E)^2 (1 - 4e-6)  =>  d <= thr  (a sure hit: the exact test d2 <= t2 holds - t2 is thr^2 to 1e-16),
//   d2f >= (thr + E)^2 (1 + 4e-6)  =>  d >  thr  (a sure miss),
// and anything between (about one record in 10^5; also NaN) is AMBIGUOUS: the lane returns -2 and the caller decides it
// with mesh_list_check on the float64 records.  "Provably too far" uses a bound that is never below the exact path's
// (a later stop is still a correct stop): rho > (thr + |tq - header| (1 + 1e-6)) (1 + 1e-6).  Events in record order,
// stop before hit on the same record, as in mesh_list_check; 1 / 0 / -1 mean the same.
MD float dist2f3(const float* q, const MeshScr& p) {
    const float d0 = q[0] - p.c[0], d1 = q[1] - p.c[1], d2 = q[2] - p.c[2];
    float d = d0 * d0;
    d = fmaf_(d1, d1, d);
    d = fmaf_(d2, d2, d);
    return d;
}
template <bool PRE = false>
MD int mesh_screen_check(const MeshScr* __restrict__ vscr, int32_t h, const float* tqf, double thr, int max_records,
                         double* lim_out, const MeshScr* pre = nullptr) {
    const MeshScr* vs = vscr + (size_t)h * MESH_REC;
    const MeshScr hd = PRE ? pre[0] : vs[0];
    const float thr_up = __double2float_ru(thr), thr_dn = __double2float_rd(thr);
    const float E = 2.5e-7f * (__builtin_fabsf(tqf[0]) + __builtin_fabsf(tqf[1]) + __builtin_fabsf(tqf[2]));
    const float lo = thr_dn - E, hi = thr_up + E;
    const float t2lo = lo > 0.0f ? lo * lo * (1.0f - 4e-6f) : -1.0f;  // no sure hits when the threshold is within E
    const float t2hi = hi * hi * (1.0f + 4e-6f);
    const float delta_up = __builtin_sqrtf(dist2f3(tqf, hd)) * (1.0f + 1e-6f);
    const float limf = (thr_up + delta_up) * (1.0f + 1e-6f) + 1e-30f;
    if (lim_out) *lim_out = (double)limf;
    for (int s0 = 1; s0 <= max_records; s0 += MESH_BATCH) {
        MeshScr e[MESH_BATCH];
        if (PRE && s0 == 1) {
#pragma unroll
            for (int j = 0; j < MESH_BATCH; ++j) e[j] = pre[1 + j];
        } else {
#pragma unroll
            for (int j = 0; j < MESH_BATCH; ++j) e[j] = vs[s0 + j];
        }
        unsigned hits = 0, stops = 0, amb = 0;
#pragma unroll
        for (int j = 0; j < MESH_BATCH; ++j) {
            const float d = dist2f3(tqf, e[j]);
            const bool hit = d <= t2lo, miss = d >= t2hi;
            stops |= (e[j].rho > limf ? 1u : 0u) << j;
            hits |= (hit ? 1u : 0u) << j;
            amb |= ((hit | miss) ? 0u : 1u) << j;
        }
        if (hits | stops | amb) {
            const int fh = hits ? __builtin_ctz(hits) : 32, fs = stops ? __builtin_ctz(stops) : 32, fa = amb ? __builtin_ctz(amb) : 32;
            if (fs <= fh && fs <= fa) return 0;
            return fh < fa ? 1 : -2;
        }
    }
    if (max_records < MESH_M) return -1;
    return hd.rho > limf ? 0 : -1;
}

// Wave-level NN: per-lane hint scan, then the octets serve the lanes it could not certify.
// Must be called by every lane of the wave (`live` = this lane holds a query).
template <bool STATS = false, bool SCREEN = false>
MD bool nn6_wave(const TreeView<Kd6>& tv, const float* q, bool live, int32_t hint, int32_t& idx, float& d2, float* cd,
                 int* n_leaves = nullptr, int* n_nodes = nullptr, int* n_scanned = nullptr, long long* t_solo = nullptr) {
    float best = INFINITY;
    int64_t bi = 0;
    bool done = !live;
    const bool hinted = live && hint >= 0 && (int64_t)hint < tv.K;
    float r_lane = 0.f;
    if (hinted)  // records 0 .. NN_SOLO-1, per lane; `hint` comes back as the pivot whose list was scanned
        done = SCREEN ? nn6_hint_scan_screened(tv, q, hint, best, bi, n_scanned, &r_lane)
                      : nn6_hint_scan(tv, q, hint, best, bi, n_scanned, &r_lane);
    if (t_solo) *t_solo = clock64();
    nn6_coop<SCREEN>(tv, q, hint, r_lane, best, bi, hinted && !done, done);  // the rest, whole wave per lane
    wave_search<Kd6, false, STATS>(tv, q, best, bi, !done, cd, n_leaves, n_nodes);
    idx = (int32_t)bi;
    d2 = best;
    return !done;  // this lane needed the tree search
}

// =================================================================================================
// standalone kernels
// =================================================================================================
MD void load_pose(const float* p, float* P) {
    const float4* v = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 r = v[i];
        P[i * 4 + 0] = r.x; P[i * 4 + 1] = r.y; P[i * 4 + 2] = r.z; P[i * 4 + 3] = r.w;
    }
}
MD void store_pose(float* p, const float* P) {
    float4* v = reinterpret_cast<float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = make_float4(P[i * 4 + 0], P[i * 4 + 1], P[i * 4 + 2], P[i * 4 + 3]);
}

// the part of the motion model that does not depend on the particle's pose: NO = O @ Tn(noise of slot n) - in two halves (the draws;
// the noise transform and the product), so that a caller may put a round trip of its own under each
MD void noise_draws(int64_t n, int64_t n_global, const float* tn_arr, const float* rot_arr, float std_t, float std_r, uint64_t seed,
                    uint64_t step, float* tn, float* rot) {
    if (tn_arr) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { tn[j] = tn_arr[n * 3 + j]; rot[j] = rot_arr[n * 3 + j]; }
        // (the host draws are consumed inside this branch: values still "in flight" at the join make the compiler wait
        // for every outstanding load there, including ones the caller issued to travel during the arithmetic below)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(tn[j]), "+v"(rot[j]));
    } else {
        float z[6];
        philox_normals6((uint64_t)n_global, seed, step, z);
#pragma unroll
        for (int j = 0; j < 3; ++j) { tn[j] = z[j] * std_t; rot[j] = z[3 + j] * std_r; }
    }
}
MD void noise_apply(const float* O, const float* tn, const float* rot, float* NO) {
    float Tn[16];
    noise_transform(tn, rot, Tn);
    mat4_mul(O, Tn, NO);
}
MD void noise_odom(int64_t n, int64_t n_global, const float* O, const float* tn_arr, const float* rot_arr, float std_t,
                   float std_r, uint64_t seed, uint64_t step, float* NO) {
    float tn[3], rot[3];
    noise_draws(n, n_global, tn_arr, rot_arr, std_t, std_r, seed, step, tn, rot);
    noise_apply(O, tn, rot, NO);
}

MD void propagate_one(int64_t n, int64_t n_global, const float* P, const float* O, const float* tn_arr,
                      const float* rot_arr, float std_t, float std_r, uint64_t seed, uint64_t step, float* out) {
    float NO[16];
    noise_odom(n, n_global, O, tn_arr, rot_arr, std_t, std_r, seed, step, NO);
    mat4_mul(P, NO, out);
}

__global__ __launch_bounds__(64) void k_propagate(int64_t N, const float* __restrict__ in, float* __restrict__ out,
                                                  const float* __restrict__ odom, const float* __restrict__ tn,
                                                  const float* __restrict__ rot, float std_t, float std_r,
                                                  uint64_t seed, uint64_t step) {
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float P[16], O[16], R[16];
    load_pose(in + n * 16, P);
#pragma unroll
    for (int i = 0; i < 16; ++i) O[i] = odom[i];
    propagate_one(n, n, P, O, tn, rot, std_t, std_r, seed, step, R);
    store_pose(out + n * 16, R);
}

__global__ __launch_bounds__(64) void k_feature(int64_t N, const float* __restrict__ poses, float wt, float wr,
                                                float* __restrict__ feat) {
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float P[16], f[6];
    load_pose(poses + n * 16, P);
    se3_feature(P, wt, wr, f);
#pragma unroll
    for (int j = 0; j < 6; ++j) feat[n * 6 + j] = f[j];
}

__global__ __launch_bounds__(64) void k_nn6(TreeView<Kd6> tv, int64_t N, const float* __restrict__ feat,
                                            const int32_t* __restrict__ hint, int32_t* __restrict__ idx,
                                            float* __restrict__ d2out) {
    __shared__ float s_cd[KD_MAX_LEVELS * 64];
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int j = 0; j < 6; ++j) q[j] = feat[n * 6 + j];
    }
    int32_t bi;
    float bd;
    nn6_wave(tv, q, live, (live && hint) ? hint[n] : -1, bi, bd, s_cd);
    if (live) {
        idx[n] = bi;
        if (d2out) d2out[n] = bd;
    }
}

// diagnostic twin: per query, leaves / nodes visited by the octet search (0 / -(1 + records scanned) when the
// hint scan certified the answer)
__global__ __launch_bounds__(64) void k_nn6_stats(TreeView<Kd6> tv, int64_t N, const float* __restrict__ feat,
                                                  const int32_t* __restrict__ hint, int32_t* __restrict__ leaves,
                                                  int32_t* __restrict__ nodes) {
    __shared__ float s_cd[KD_MAX_LEVELS * 64];
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int j = 0; j < 6; ++j) q[j] = feat[n * 6 + j];
    }
    int32_t bi;
    float bd;
    int nl = 0, nn = 0, ns = -1;
    nn6_wave<true>(tv, q, live, (live && hint) ? hint[n] : -1, bi, bd, s_cd, &nl, &nn, &ns);
    if (live) {
        leaves[n] = nl;
        nodes[n] = (nl == 0 && nn == 0 && ns >= 0) ? -(ns + 1) : nn;
    }
}

// ---- the mesh's distance field (MeshField, midas_internal.hpp) ---------------------------------------------------------------
// The centre of cell (ix, iy, iz) as ONE float32 expression, used by the builder and by the look-up alike (the stored distance
// belongs to exactly this point).
MD float field_centre(const MeshField& f, int axis, int i) { return fmaf_((float)i + 0.5f, f.h, f.lo[axis]); }

// A look-up in two halves: field_fetch requests the cell's value (early: the translation is known once the motion model is done,
// the answer travels under the nearest-neighbour search), field_decide turns it into 1 (some vertex within thr, certain),
// 0 (none, certain) or -1 (the shell around the threshold, NaN, no field: the exact path decides).
struct FieldProbe { float v = 0.f, rho = 0.f; int state = 2; };  // state 0: inside the grid, 1: outside it, 2: unknown
MD FieldProbe field_fetch(const MeshField& f, const float* tq, bool live) {
    FieldProbe pr;
    if (!f.d || !live) return pr;
    const float gx = (tq[0] - f.lo[0]) * f.inv_h, gy = (tq[1] - f.lo[1]) * f.inv_h, gz = (tq[2] - f.lo[2]) * f.inv_h;
    if (!(gx == gx && gy == gy && gz == gz)) return pr;  // NaN: unknown
    if (!(gx >= 0.f && gy >= 0.f && gz >= 0.f && gx < (float)f.n[0] && gy < (float)f.n[1] && gz < (float)f.n[2])) { pr.state = 1; return pr; }
    const int ix = (int)gx, iy = (int)gy, iz = (int)gz;  // (a value that rounding put into the neighbouring cell is served by that cell: rho says how far its centre is)
    const float dx = tq[0] - field_centre(f, 0, ix), dy = tq[1] - field_centre(f, 1, iy), dz = tq[2] - field_centre(f, 2, iz);
    pr.rho = __builtin_sqrtf(fmaf_(dz, dz, fmaf_(dy, dy, dx * dx)));
    pr.v = f.d[((int64_t)iz * f.n[1] + iy) * f.n[0] + ix];
    pr.state = 0;
    return pr;
}
MD int field_decide(const MeshField& f, const FieldProbe& pr, double thr) {
    if (pr.state == 2 || !(thr >= 0.0)) return -1;
    const float thr_up = __double2float_ru(thr), thr_dn = __double2float_rd(thr);
    if (pr.state == 1) return f.expand > thr_up * 1.00001f ? 0 : -1;  // outside the grown bounding box: farther than `expand` from every vertex
    // true distance d, stored v = float(d_centre) (nearest: 6e-8 relative), rho computed to 4e-7 relative on float32 coordinates
    // that are exact: |d - v| <= rho + 1e-6 (v + rho), and the exact path's comparison is d <= thr up to 1e-16
    const float slack = 2e-6f * (pr.v + pr.rho + thr_up) + 1e-30f;
    if (pr.v + pr.rho + slack <= thr_dn) return 1;
    if (pr.v - pr.rho - slack >= thr_up) return 0;
    return -1;
}

// builder: exact distance (float64 search of the 3-d tree, as k_nn3) from every cell centre of a slab of the grid
__global__ __launch_bounds__(64) void k_field_build(TreeView<Kd3> tv, MeshField f, int64_t c0, int64_t ncells, float* __restrict__ out) {
    __shared__ double s_cd[KD_MAX_LEVELS * 64];
    const int64_t n = c0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = n < ncells;
    double q[3] = {0.0, 0.0, 0.0};
    if (live) {
        const int ix = (int)(n % f.n[0]), iy = (int)((n / f.n[0]) % f.n[1]), iz = (int)(n / ((int64_t)f.n[0] * f.n[1]));
        q[0] = (double)field_centre(f, 0, ix); q[1] = (double)field_centre(f, 1, iy); q[2] = (double)field_centre(f, 2, iz);
    }
    double best = INFINITY;
    int64_t bi = 0;
    wave_search<Kd3, false>(tv, q, best, bi, live, s_cd);
    if (live) out[n] = (float)__builtin_sqrt(best);
}

__global__ __launch_bounds__(64) void k_nn3(TreeView<Kd3> tv, int64_t N, const float* __restrict__ poses,
                                            double* __restrict__ dist) {
    __shared__ double s_cd[KD_MAX_LEVELS * 64];
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    double q[3] = {0.0, 0.0, 0.0};
    if (live) { q[0] = (double)poses[n * 16 + 3]; q[1] = (double)poses[n * 16 + 7]; q[2] = (double)poses[n * 16 + 11]; }
    double best = INFINITY;
    int64_t bi = 0;
    wave_search<Kd3, false>(tv, q, best, bi, live, s_cd);
    if (live) dist[n] = __builtin_sqrt(best);
}

// check_quats (modules/particle_filter.py:347-357): flag poses whose rotation yields a NaN or
// zero-norm quaternion.  theseus' to_quaternion is derived from trace / off-diagonal terms; a pose
// is flagged when any rotation entry is non-finite or the quaternion's squared norm
// (1 + tr)/4 + |axis|^2-style reconstruction collapses to 0 - in practice only NaN/Inf poses.
__global__ __launch_bounds__(256) void k_check_poses(int64_t N, const float* __restrict__ poses,
                                                     uint8_t* __restrict__ flag, int32_t* __restrict__ count) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (n < N) {
        const float* P = poses + n * 16;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += P[i * 4 + j] * P[i * 4 + j];
        bad = !(acc > 0.0f) || !(acc < INFINITY);  // NaN, Inf or all-zero rotation
        flag[n] = bad ? 1 : 0;
    }
    unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (int32_t)__popcll(m));
}

// rmse partials: per-wave (sum e_t^2, sum ang^2) in float64
MD void rmse_terms(const float* P, const float* G, double& et2, double& ang2) {
    float dx = G[3] - P[3], dy = G[7] - P[7], dz = G[11] - P[11];
    float e2 = fmaf_(dz, dz, fmaf_(dy, dy, dx * dx));
    float tr = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = G[i * 4] * P[i * 4];
        acc = fmaf_(G[i * 4 + 1], P[i * 4 + 1], acc);
        acc = fmaf_(G[i * 4 + 2], P[i * 4 + 2], acc);
        tr += acc;
    }
    float ang = acosf((tr - 1.0f) * 0.5f) * 57.2957795130823209f;
    if (ang != ang) ang = 0.0f;
    if (ang > 180.0f) ang -= 360.0f;
    if (ang < -180.0f) ang += 360.0f;
    et2 = (double)e2;
    ang2 = (double)ang * (double)ang;
}

MD double wave_sum(double v) { return wave_sum_ordered(v); }  // (the xor butterfly 32 .. 1 of the spec, by register moves: midas_math.hpp)
MD double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
MD double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

__global__ __launch_bounds__(64) void k_rmse_part(int64_t N, const float* __restrict__ poses,
                                                  const float* __restrict__ gt, double* __restrict__ part) {
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (n < N) {
        float P[16], G[16];
        load_pose(poses + n * 16, P);
#pragma unroll
        for (int i = 0; i < 16; ++i) G[i] = gt[i];
        rmse_terms(P, G, a, b);
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}

__global__ __launch_bounds__(256) void k_rmse_final(int64_t N, int nb, const double* __restrict__ part,
                                                    double* __restrict__ out2) {
    __shared__ double sa[4], sb[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (sa[0] + sa[1]) + (sa[2] + sa[3]);
        b = (sb[0] + sb[1]) + (sb[2] + sb[3]);
        out2[0] = __builtin_sqrt(a / (double)N);
        out2[1] = __builtin_sqrt(b / (double)N);
    }
}


// ---- resample of the previous frame as a prologue of the particle update (LazyResample) -------------------------
// Workgroup part: guard, sequential block prefix, exact cdf at the block ends into LDS (every thread of the
// 256-thread workgroup takes part).  rs_lds: [0, nb) block prefix | [256, 256+nb) block ends | 512: total, 513: S,
// 514: apply | [516, 516+nb) block sums of e | [LAZY_WG_W, +nb) block totals (the guide tables' bin width).  Same arithmetic as
// k_tail_b / k_tail_b2.
constexpr int LAZY_WG_W = 3 * LAZY_MAX_BLOCKS + 8, LAZY_WG_LDS = 4 * LAZY_MAX_BLOCKS + 8;
constexpr double LAZY_ISCLOSE_ATOL = 1e-8;
MD void lazy_tables(const LazyResample& rs, double* rs_lds) {
    __shared__ double s_ex[12];
    const int t = threadIdx.x;
    const int b = t < rs.nb ? t : rs.nb - 1;  // nb <= 256: one block per thread, clamped loads
    const double bt = rs.btot[b], btr = rs.btot_raw[b], bs = rs.bsum_e[b], bx = rs.bmax[b], bn = rs.bmin[b];
    const int32_t status = rs.status_prev[0];  // with the records: read in lazy_source it was a round trip of its own
    const bool in = t < rs.nb;
    double mx = in ? bx : -INFINITY, mn = in ? bn : INFINITY;
    const bool nan = in && ((bx != bx) || (bn != bn));
    mx = wave_max_dpp(mx);  // (DPP moves: midas_math.hpp)
    mn = wave_min_dpp(mn);
    const bool wn = __any(nan);
    if ((t & 63) == 0) { s_ex[t >> 6] = mx; s_ex[4 + (t >> 6)] = mn; s_ex[8 + (t >> 6)] = wn ? 1.0 : 0.0; }
    __syncthreads();
    mx = s_ex[0]; mn = s_ex[4];
    double f = s_ex[8];
    for (int w = 1; w < 4; ++w) { mx = s_ex[w] > mx ? s_ex[w] : mx; mn = s_ex[4 + w] < mn ? s_ex[4 + w] : mn; f += s_ex[8 + w]; }
    if (f != 0.0) { mx = NAN; mn = NAN; }
    const bool apply = rs.softmax && !(__builtin_fabs(mx - mn) <= LAZY_ISCLOSE_ATOL);
    double* s_bp = rs_lds;
    double* s_w = rs_lds + 256;
    double* s_se = rs_lds + 516;
    if (in) { s_w[t] = apply ? bt : btr; s_se[t] = bs; }
    __syncthreads();
    if (t == 0) {
        double acc = 0.0, S = 0.0;
        for (int i = 0; i < rs.nb; ++i) { s_bp[i] = acc; acc = acc + s_w[i]; S = S + s_se[i]; }
        rs_lds[512] = acc;
        rs_lds[513] = apply ? S : 1.0;
        rs_lds[514] = apply ? 1.0 : 0.0;
        rs_lds[515] = status != 0 ? 1.0 : 0.0;
    }
    __syncthreads();
    const double total = rs_lds[512];
    // exact cdf at the last slot of every block: (BP_b + W_b) / total - the block total IS the block-local prefix at
    // the block's last slot (same additions in the same order); the last block ends at N-1, forced to 1
    const double wb = in ? s_w[t] : 0.0;
    __syncthreads();
    if (in) rs_lds[LAZY_WG_W + t] = wb;
    if (in) s_w[t] = (t == rs.nb - 1) ? 1.0 : (s_bp[t] + wb) / total;
    __syncthreads();
}

// The same tables built by ONE wave for itself (nb <= 64: N <= 262144), split in two so that the pose-independent half of
// the motion model runs between the loads and their use: no workgroup barrier, no serial LDS loop - the sequential block
// prefix is a left fold over lane values read with v_readlane (the same additions in the same order as lazy_tables).
// Layout of the wave's block (LAZY_WAVE_LDS doubles): [0, 64) block prefix | [64, 128) block ends | 128 total | 130 apply |
// 131 status of the previous frame | [132, 196) block totals (the guide table's bin width, GUIDE_BINS).
constexpr int LAZY_WAVE_LD = 64, LAZY_WAVE_LDS = 3 * LAZY_WAVE_LD + 4;
struct LazyRecords { double bt, btr, bx, bn; int32_t status; };
MD LazyRecords lazy_records_load(const LazyResample& rs) {
    const int lane = threadIdx.x & 63;
    const int b = lane < rs.nb ? lane : rs.nb - 1;
    LazyRecords r;
    r.bt = rs.btot[b]; r.btr = rs.btot_raw[b]; r.bx = rs.bmax[b]; r.bn = rs.bmin[b];
    r.status = rs.status_prev[0];
    return r;
}
MD void lazy_tables_wave(const LazyResample& rs, const LazyRecords& r, double* rs_lds) {
    const int lane = threadIdx.x & 63;
    const bool in = lane < rs.nb;
    double mx = in ? r.bx : -INFINITY, mn = in ? r.bn : INFINITY;
    const bool nan = in && ((r.bx != r.bx) || (r.bn != r.bn));
    mx = wave_max_dpp(mx);  // (DPP moves: midas_math.hpp)
    mn = wave_min_dpp(mn);
    if (__any(nan)) { mx = NAN; mn = NAN; }
    const bool apply = rs.softmax && !(__builtin_fabs(mx - mn) <= LAZY_ISCLOSE_ATOL);
    const double w = in ? (apply ? r.bt : r.btr) : 0.0;
    // The sequential prefix of the block totals: the totals go through LDS (every lane reads the same eight values a round -
    // broadcast reads, all requested before the first addition) and every lane runs the same chain of additions, keeping the
    // value it passes at its own block.  Blocks past nb hold +0.0: adding them changes nothing (the sum never is -0.0).
    // (v_readlane with the block number in a scalar register cost two hazards and a branch per block: 1.4 us of a wave's life)
    double* s_wb = rs_lds + 2 * LAZY_WAVE_LD + 4;
    s_wb[lane] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double acc = 0.0, bp = 0.0;
    for (int i0 = 0; i0 < rs.nb; i0 += 8) {
        double wv[8];
        const double2* p2 = reinterpret_cast<const double2*>(s_wb + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double2 t = p2[j]; wv[2 * j] = t.x; wv[2 * j + 1] = t.y; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bp = lane == i0 + j ? acc : bp;
            acc = acc + wv[j];
        }
    }
    const double total = acc;
    if (in) {
        rs_lds[lane] = bp;
        rs_lds[LAZY_WAVE_LD + lane] = (lane == rs.nb - 1) ? 1.0 : (bp + w) / total;
    }
    if (lane == 0) {
        rs_lds[2 * LAZY_WAVE_LD] = total;
        rs_lds[2 * LAZY_WAVE_LD + 2] = apply ? 1.0 : 0.0;
        rs_lds[2 * LAZY_WAVE_LD + 3] = r.status != 0 ? 1.0 : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane part: the source particle of slot n (what k_tail_b2 writes to ridx[n]).
// ld = stride of the table block: 256 (lazy_tables, one block per workgroup) or LAZY_WAVE_LD (lazy_tables_wave)
// gend_lds / lp_lds (both or gend_lds alone): the caller's LDS copies of the chunk-end / per-slot tables
// mid: arithmetic of the caller's that does not depend on the search, run once while the guide entries travel (NoMid: none; a lane
// that leaves the search before that point has not run it - the caller looks at its own flag)
template <typename GT = const double*, typename LT = const double*, typename MID = NoMid>
MD int64_t lazy_source(const LazyResample& rs, const double* rs_lds, int64_t n, int64_t N, int ld = 256, GT gend_lds = nullptr,
                       LT lp_lds = nullptr, MID mid = MID()) {
    const double* s_bp = rs_lds;
    const double* s_end = rs_lds + ld;
    const double total = rs_lds[2 * ld];
    const bool apply = rs_lds[2 * ld + 2] != 0.0;
    const bool bad_total = !(total == total) || total == 0.0;
    if (rs_lds[2 * ld + 3] != 0.0 || bad_total) return n;  // unusable weights: the resampler keeps the particles
    const double* __restrict__ lp = apply ? rs.lp : rs.lp_raw;
    const double* __restrict__ gend = apply ? rs.gend : rs.gend_raw;
    double tq;
    bool upper;
    if (rs.mode == MIDAS_RESAMPLE_MULTINOMIAL) {
        tq = rs.u ? rs.u[n] : philox_uniform53((uint64_t)(n + rs.key_base), rs.seed, rs.step);
        upper = false;
    } else {
        const float r = rs.u32 >= 0.0f ? rs.u32 : philox_uniform24(rs.seed + (uint64_t)rs.traj, rs.step);
        const float off = r / (float)N;
        tq = (double)n / (double)N + (double)off;
        tq = tq >= 1.0 ? tq - 1.0 : tq;
        upper = true;
    }
    auto left_exact = [&](double c) { return upper ? (c <= tq) : (c < tq); };
    // block: first b whose exact end value is not left of the draw
    int lo = 0, hi = rs.nb;
    while (hi > lo) {
        const int mid = lo + ((hi - lo) >> 1);
        if (left_exact(s_end[mid])) lo = mid + 1; else hi = mid;
    }
    if (lo >= rs.nb) return N - 1;
    if constexpr (__is_same(LT, lds_cdp)) {
        return search_in_block_t<lds_cdp, GT>(lp_lds, gend, apply ? rs.ggend : rs.ggend_raw, lo, N, N - 1, s_bp[lo], total, tq, upper, gend_lds);
    } else {
        // guide table of the block: the unit from one entry pair (the block totals sit behind the tables, per wave or per workgroup)
        const guide_t* guide = apply ? rs.guide : rs.guide_raw;
        return search_in_block_t<const double*, GT, MID>(lp, gend, apply ? rs.ggend : rs.ggend_raw, lo, N, N - 1, s_bp[lo], total, tq, upper, gend_lds,
                                                         guide, guide ? rs_lds[(ld == LAZY_WAVE_LD ? 2 * LAZY_WAVE_LD + 4 : LAZY_WG_W) + lo] : 0.0, mid);
    }
}

// =================================================================================================
// fused particle update of the step
// =================================================================================================
// One wave = 64 consecutive particles of trajectory `traj`; `wave` counts the waves of that trajectory,
// `nwaves` = waves per trajectory (strides of the per-wave partial arrays), s_cd = this wave's LDS columns.
// WT (with rs_lds): the wave builds the resample tables itself (lazy_tables_wave; rs_lds = its own LAZY_WAVE_LDS doubles)
// SCREEN: half-record screening in the list scans (see part4)
// PREF: the vertex list's header and first batch are requested before the sparse-scoring claim (72 more registers: only
// where two waves per SIMD are all the launch needs, N <= 131072 in one-wave workgroups)
// STATS (profiling instantiations only, chosen by MIDAS_ABLATE != 0): per-wave phase clocks, scan statistics and the ablation
// switches; the production instantiations read no clock and test no switch
// PRES: the presorted form (pre_order / pre_src, batch kernels only) is compiled in; elsewhere the arguments are ignored
template <bool WT = false, bool SCREEN = false, bool PREF = false, bool STATS = false, bool PRES = false>
MD void particle_update_wave(const TreeView<Kd6>& t6, const TreeView<Kd3>& t3, ParticleUpdateArgs a, int64_t wave,
                             int nwaves, int traj, double* s_cd, double* rs_lds = nullptr) {
    const int lane = threadIdx.x & 63;
    if (a.n_live) {  // variable particle count: the grid covers the capacity, the waves past the live set leave
        const int64_t nl = *a.n_live;
        a.N = nl < a.N ? nl : a.N;
        if (wave * 64 >= a.N && wave != 0) return;
    }
    if (traj) {  // batch of trajectories: every per-trajectory array is (B, ...) contiguous
        const int64_t b = traj, o = b * a.N;
        a.poses_in += o * 16; a.poses_prop += o * 16; a.odom16 += b * 16;
        if (a.tn) { a.tn += o * 3; a.rot += o * 3; }
        if (a.hint_in) a.hint_in += o;
        a.nn_idx += o; a.valid += o;
        if (a.scores) { a.scores += b * a.score_stride; a.x += o; a.e += o; a.part_max += b * nwaves; a.part_min += b * nwaves; }
        if (a.gt16) { a.gt16 += b * 16; a.part_rmse += 2 * b * nwaves; }
        if (a.status_reset) a.status_reset += 2 * b;
        if (a.sp.stamps) {  // sparse scoring per trajectory: its own stamps, tactile code and score row
            a.sp.stamps += b * a.score_stride; a.sp.scores += b * a.score_stride; a.sp.code += b * (int64_t)(a.sp.nj * 64);
        }
        a.slot_base += o;
        if (a.rs.enabled) {  // pipelined batch: per-trajectory table blocks, previous-frame arrays and draws
            const int64_t ts = b * a.rs.tstride;
            a.rs.e += ts; a.rs.x_raw += ts; a.rs.lp += ts; a.rs.lp_raw += ts; a.rs.gend += ts; a.rs.gend_raw += ts;
            a.rs.ggend += ts; a.rs.ggend_raw += ts; a.rs.bsum_e += ts; a.rs.btot += ts; a.rs.btot_raw += ts; a.rs.bmax += ts; a.rs.bmin += ts;
            a.rs.poses_prev += o * 16; a.rs.nn_prev += o; a.rs.status_prev += 2 * b;
            if (a.rs.ridx_out) a.rs.ridx_out += o;
            if (a.rs.u) a.rs.u += o;
            a.rs.key_base = o;
            a.rs.traj = traj;
        }
        if (a.pre_order) { a.pre_order += o; a.pre_src += o; if (a.pre_rmse_terms) a.pre_rmse_terms += 2 * o; }
    }
    // presorted (wave-uniform): lane `rank` of the launch works on slot order[rank] - slots that start from the same codebook
    // entry sit side by side, so the list records a wave's lanes ask for are mostly the SAME addresses (one look-up, one line)
    const bool presorted = PRES && a.pre_order != nullptr;
    const int64_t rank = wave * 64 + lane;
    const bool live = rank < a.N;
    int64_t n = rank;
    int32_t src_pre = 0;
    if (presorted) {
        n = a.pre_order[live ? rank : 0];
        src_pre = a.pre_src[live ? rank : 0];
    }
    if (wave == 0 && lane == 0) {
        if (a.status_reset) { a.status_reset[0] = 0; a.status_reset[1] = 0; }
        if (a.flags_reset) { a.flags_reset[0] = 0.0; a.flags_reset[1] = 0.0; }
        if (a.sp.next_count) *a.sp.next_count = 0;  // this frame's tail appends the next frame's prediction list
    }
    unsigned long long st_nn = 0, st_mesh = 0, st_scan = 0;
    int st_rows = 0;
    const bool dense_scores = scores_dense(a.sp);  // requested here, looked at after the nearest-neighbour search
    const int ablate = STATS ? a.ablate : 0;
    long long tc[10];  // phase clocks, reported with MIDAS_ABLATE=4
    long long tp[4] = {0, 0, 0, 0};  // ... and inside the first phase (MIDAS_ABLATE=4 + 128: reported in place of phases 4 .. 7)
#define MIDAS_TICK(i) do { if (STATS) tc[i] = clock64(); } while (0)
#define MIDAS_PTICK(i) do { if (STATS) tp[i] = clock64(); } while (0)
    MIDAS_TICK(0);
    const long long wall0 = STATS ? wall_clock64() : 0;  // 100 MHz
    double x = 0.0, et2 = 0.0, ang2 = 0.0;
    float R[16], f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) R[i] = 0.f;
    // source of the particle: its own slot, or - resample of the previous frame folded in - slot src of the previous
    // frame's propagated poses
    int64_t src = n;
    float NO[16];
    if (WT) {
        // the block records travel while the pose-independent half of the motion model (draws, noise transform,
        // O @ Tn: half of the propagate's arithmetic) is computed; the tables are then built from registers
        // (memory operations come back in order: the odometry is requested BEFORE the records, or waiting for it would
        // be waiting for them)
        float O[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = a.odom16[i];
        __builtin_amdgcn_sched_barrier(0);
        LazyRecords rec;
        if (!presorted) rec = lazy_records_load(a.rs);
        __builtin_amdgcn_sched_barrier(0);
        // (the draws under the records' round trip; the noise transform and O @ Tn under the guide entries' - see lazy_source)
        float tnv[3] = {0.f, 0.f, 0.f}, rotv[3] = {0.f, 0.f, 0.f};
        bool no_done = false;
        if (live) noise_draws(n, n + a.slot_base, a.tn, a.rot, a.std_t, a.std_r, a.seed, a.step, tnv, rotv);
        if (!presorted) lazy_tables_wave(a.rs, rec, rs_lds);
        MIDAS_PTICK(0);  // records there, tables built (draws done under their trip)
        auto mid = [&]() { noise_apply(O, tnv, rotv, NO); no_done = true; };
        if (rs_lds && live && !presorted && !(ablate & 8)) {
            src = lazy_source(a.rs, rs_lds, n, a.N, LAZY_WAVE_LD, (const double*)nullptr, (const double*)nullptr, mid);
            if (a.rs.ridx_out) a.rs.ridx_out[n] = (int32_t)src;
        }
        if (live && !no_done) mid();
        MIDAS_PTICK(1);  // source slot known (guide entries, prefix piece)
    }
    if (rs_lds && live) {
        if (presorted) {
            src = src_pre;  // (ridx_out was written by the presort)
        } else if (!WT) {
            // ablate 8 (profiling): no search, own slot
            src = (ablate & 8) ? n : lazy_source(a.rs, rs_lds, n, a.N, 256);
            if (a.rs.ridx_out) a.rs.ridx_out[n] = (int32_t)src;
        }
    }
    const float* pose_src = rs_lds ? a.rs.poses_prev : a.poses_in;
    // the sharded frame with the unpack folded in (midas_shard_run): the particle of slot n is row n of this rank's inbox, stored
    // there by the owner of its source; the rows are complete - the route kernel in front of this launch ended with every
    // rank's completion flag in
    const bool from_inbox = !WT && a.inbox.rows != nullptr;
    PeerRow row;
    if (from_inbox && live) row = peer_row_load(a.inbox.rows, n);
    // the hint travels with the pose (behind the store of the propagated pose it would be a round trip of its own)
    const int32_t hint = !live ? -1 : from_inbox ? (int32_t)(row.head[1] & 0xFFFFFFFFull) : rs_lds ? a.rs.nn_prev[src] : a.hint_in ? a.hint_in[n] : -1;
    if (live) {
        float P[16];
        if (from_inbox) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                P[2 * k] = __int_as_float((int)(row.pose[k] & 0xFFFFFFFFull));
                P[2 * k + 1] = __int_as_float((int)(row.pose[k] >> 32));
            }
        } else {
            load_pose(pose_src + src * 16, P);
        }
        if (WT) {
            mat4_mul(P, NO, R);
            MIDAS_PTICK(2);  // source row there, propagated
        } else {
            float O[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) O[i] = a.odom16[i];
            propagate_one(n, n + a.slot_base, P, O, a.tn, a.rot, a.std_t, a.std_r, a.seed, a.step, R);
        }
        store_pose(a.poses_prop + n * 16, R);
        se3_feature(R, 0.99f, 0.01f, f);
    }
    // the prune's distance-field cell is requested now (it needs the translation only): the answer arrives under the search
    const float tq_f[3] = {R[3], R[7], R[11]};
    const FieldProbe probe = field_fetch(a.field, tq_f, live && !(ablate & 2));
    MIDAS_TICK(1);
    // nearest codebook entry
    int32_t bi = 0;
    float bd;
    if (ablate & 1) {  // profiling only: trust the hint
        bi = hint < 0 ? 0 : hint;
    } else {
        int nscan = 0;
        const bool fb = nn6_wave<false, SCREEN>(t6, f, live, hint, bi, bd, reinterpret_cast<float*>(s_cd), nullptr, nullptr, STATS ? &nscan : nullptr,
                                                STATS ? &tc[2] : nullptr);
        MIDAS_TICK(3);
        if (a.telemetry && (ablate & 4)) {  // MIDAS_ABLATE=4: scan statistics (profiling only), flushed at the end
            st_nn = __ballot(live && nscan >= NN_SOLO - 1);
            st_scan = (unsigned long long)wave_sum((double)nscan);
        }
        if (a.telemetry) {
            const unsigned long long m = __ballot(fb);
            if (lane == 0 && m) atomicAdd(&a.telemetry[0], (unsigned long long)__popcll(m));
        }
    }
    // sparse scoring: the first particle of the frame on an entry has it scored - the exchanges leave here, the answers are
    // looked at after the prune
    // (small-set regime, registers to spare: the vertex list's header and first batch are requested before the claim, whose
    // look at the stamps is a round trip of its own)
    // (of the float32 screening copy; without one the float64 list is read after the claim)
    constexpr bool PRE = PREF;
    // prune, first word: the distance field (1 valid, 0 invalid: certain; -1: the vertex lists / the tree decide)
    int mv = live ? field_decide(a.field, probe, a.thr) : -1;
    const bool lists_needed = __ballot(live && mv < 0) != 0;  // (wave-uniform: a wave whose particles are all decided skips the lists)
    MeshScr pre[PRE ? 1 + MESH_BATCH : 1];
    if (PRE && a.vscr != nullptr && lists_needed) {
        const MeshScr* vs = a.vscr + (size_t)(live ? bi : 0) * MESH_REC;
#pragma unroll
        for (int j = 0; j < (PRE ? 1 + MESH_BATCH : 1); ++j) pre[j] = vs[j];
    }
    RowClaim claim{false, 0u};
    if (a.sp.stamps && !(ablate & 16)) {  // ablate 16 (profiling): nobody scores
        claim = claim_rows_issue(a.sp, live, bi, MIDAS_CLAIM_HASH ? reinterpret_cast<int*>(s_cd) : nullptr);
        st_rows = score_claimed_rows_nj(a.sp, claim, bi, dense_scores);
    }
    MIDAS_TICK(9);
    // prune: valid <=> some mesh vertex within sqrt(t2) of the particle
    double q3[3] = {(double)R[3], (double)R[7], (double)R[11]};
    double best = a.t2;
    int64_t vi = 0;
    if (ablate & 2) mv = 1;
    else if (a.vlist && lists_needed) {
        double lim_lane = 0.0;
        const bool open = live && mv < 0;  // the lanes the field left undecided
        if (a.vscr) {  // first records, per lane: float32 screening copy, the float64 records only for what it cannot decide
            const float tqf[3] = {R[3], R[7], R[11]};
            if (open) mv = mesh_screen_check<PRE>(a.vscr, bi, tqf, a.thr, MESH_SOLO, &lim_lane, pre);
            if (__ballot(mv == -2)) {
                if (mv == -2) mv = mesh_list_check<false>(a.vlist, bi, q3, a.t2, a.thr, MESH_SOLO, &lim_lane);
            }
        } else if (open) {
            mv = mesh_list_check<false>(a.vlist, bi, q3, a.t2, a.thr, MESH_SOLO, &lim_lane);
        }
        MIDAS_TICK(4);
        if (a.telemetry && (ablate & 4)) st_mesh = __ballot(live && mv < 0);
        mesh_coop(a.vlist, bi, q3, a.t2, lim_lane, live && mv < 0, mv);             // the rest, whole wave per lane
    }
    MIDAS_TICK(5);
    bool ok = wave_search<Kd3, true>(t3, q3, best, vi, live && mv < 0, s_cd);
    if (a.telemetry) {
        const unsigned long long m = __ballot(live && mv < 0);
        if (lane == 0 && m) atomicAdd(&a.telemetry[1], (unsigned long long)__popcll(m));
    }
    if (mv >= 0) ok = mv == 1;
    if (a.telemetry && st_rows && lane == 0) atomicAdd(&a.telemetry[2], (unsigned long long)st_rows);  // rows scored by particle waves
    MIDAS_TICK(6);
    if (live) {
        a.nn_idx[n] = bi;
        if (a.scores) {  // nullptr: the scoring runs concurrently, the tail gathers the scores
            x = a.scores[bi];
            a.x[n] = x;
            a.e[n] = exp_spec(x - 1.0);
        }
        a.valid[n] = ok ? 1 : 0;
        if (a.gt16) {
            float G[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) G[i] = a.gt16[i];
            rmse_terms(R, G, et2, ang2);
        }
    }
    MIDAS_TICK(7);
    // per-wave extrema of x over live lanes
    const double NEG = -INFINITY, POS = INFINITY;
    if (a.scores) {
        double mx = wave_max(live ? x : NEG), mn = wave_min(live ? x : POS);
        if (lane == 0) { a.part_max[wave] = mx; a.part_min[wave] = mn; }
    }
    if (a.gt16) {
        if (presorted && a.pre_rmse_terms) {
            // a presorted wave holds other slots than 64 wave .. 64 wave + 63: its sum would be a different (and, the order inside
            // a group being what the LDS atomics made it, run-dependent) grouping of the same terms.  The terms go out by slot.
            if (live) reinterpret_cast<double2*>(a.pre_rmse_terms)[n] = make_double2(et2, ang2);
        } else {
            et2 = wave_sum(et2);
            ang2 = wave_sum(ang2);
            if (lane == 0) { a.part_rmse[2 * wave] = et2; a.part_rmse[2 * wave + 1] = ang2; }
        }
    }
    if (STATS && a.telemetry && (ablate & 4) && lane == 0) {
        // MIDAS_ABLATE=4: per-wave scan statistics and phase clocks, plain stores into the wave's own 16 slots
        // behind the 16 cumulative counters (the caller sized the buffer 16 + 16 * waves)
        tc[8] = clock64();
        unsigned long long* w = a.telemetry + 16 + 16 * ((size_t)traj * nwaves + wave);
        w[0] += (unsigned long long)st_rows;                 // codebook rows this wave scored (sparse scoring)
        w[1] = (unsigned long long)wall0;                    // start of the wave (100 MHz wall clock, not cumulative)
        w[2] += (unsigned long long)__popcll(st_nn); w[4] += st_nn ? 1 : 0;
        w[3] += (unsigned long long)__popcll(st_mesh); w[5] += st_mesh ? 1 : 0;
        w[6] += st_scan;
        w[7] += (unsigned long long)(wall_clock64() - wall0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long long d = i == 3 ? tc[9] - tc[3] : i == 4 ? tc[5] - tc[9] : tc[i + 1] - tc[i];  // [3] claim + scoring, [4] prune lists
            // + 128: the first phase in four pieces instead of phases 4 .. 7: records + tables, source slot, source row + product,
            // store + feature + field probe
            if ((ablate & 128) && i >= 4) d = !tp[0] ? 0 : i == 4 ? tp[0] - tc[0] : i == 5 ? tp[1] - tp[0] : i == 6 ? tp[2] - tp[1] : tc[1] - tp[2];  // (frames without a folded resample: nothing)
            w[8 + i] += (unsigned long long)d;
        }
    }
#undef MIDAS_TICK
#undef MIDAS_PTICK
}

template <bool STATS>
__global__ __launch_bounds__(64) void k_particle_update(TreeView<Kd6> t6, TreeView<Kd3> t3, ParticleUpdateArgs a) {
    __shared__ double s_cd[KD_MAX_LEVELS * 64];  // child-distance columns, reused by both searches
    particle_update_wave<false, false, false, STATS>(t6, t3, a, blockIdx.x, gridDim.x, blockIdx.y, s_cd);
}

// Front kernel of the fused single-trajectory step: the particle update (latency-bound: dependent scattered
// fetches, ~1.5 waves per SIMD) and the codebook scoring (HBM-bound stream) have no dependency on each other -
// the scores are only gathered in the tail - so they share ONE launch: workgroups [0, n_pu) run four
// particle waves each and start first, workgroups [n_pu, ...) stream sixteen codebook rows each behind them
// and fill the memory pipes the particle waves leave idle.
// LAZY: the resample of the previous frame runs as a prologue of the particle waves (midas_lazy_step).
// LAZY 2: the same with the tables built per wave (nb <= 64), which also frees the workgroup size: FW = waves per
// workgroup.  With FW = 1 the 1563 particle waves of c2 spread 6 - 7 per CU; workgroups of four land 4 or 8 on a CU.
// SCR = false (batch of trajectories, grid.y): whole-record list scans - the screen costs the batch step more than it saves
#ifndef MIDAS_FRONT_OCC
#define MIDAS_FRONT_OCC 1  // waves per SIMD the single-trajectory forms are compiled for (1 = no register cap: 234 registers, two waves)
#endif
#ifndef MIDAS_FRONT4_OCC
#define MIDAS_FRONT4_OCC 4  // ... and the four-wave workgroups with workgroup-level tables = sets beyond 131 072 particles: several rounds of waves, four a SIMD (c3, N = 1 M: 190 -> 181 us; 3: no gain).  NOT the four-wave form of the dense front at smaller N (per-wave tables): one round of waves, the cap cost it 4 us of 29
#endif
#ifndef MIDAS_BATCH_OCC
#define MIDAS_BATCH_OCC 1  // waves per SIMD the batch form (SCR = false) is compiled for (1 = no register cap)
#endif
#ifdef MIDAS_DEBUG_CLOCKS  // wall-clock span of the front's particle waves [0, 1] and of its scoring waves [2, 3] (tools/tg_clocks.py)
__device__ long long g_ff_clk[16384];  // per frame parity and workgroup: start, end
#define FF_T0 const long long ff_t0_ = wall_clock64()
#define FF_END do { if (threadIdx.x == 0 && blockIdx.x < 4096) { long long* c_ = g_ff_clk + (a.step & 1) * 8192; c_[2 * blockIdx.x] = ff_t0_; c_[2 * blockIdx.x + 1] = wall_clock64(); } } while (0)
#else
#define FF_T0 do { } while (0)
#define FF_END do { } while (0)
#endif
template <typename T, int NJ, int LAZY, int FW, bool SCR = true, bool PREF = false, bool STATS = false>
__global__ __launch_bounds__(64 * FW, (!SCR && FW == 1) ? MIDAS_BATCH_OCC : (FW == 4 && LAZY == 1) ? MIDAS_FRONT4_OCC : MIDAS_FRONT_OCC) void k_frame_front(TreeView<Kd6> t6, TreeView<Kd3> t3, ParticleUpdateArgs a,
                                                         int n_pu, int nwaves, const T* __restrict__ emb,
                                                         const double* __restrict__ norms, const double* __restrict__ code,
                                                         double* __restrict__ scores, int64_t K) {
    static_assert(LAZY != 1 || FW == 4, "the workgroup-level tables take 256 threads");
    FF_T0;
    __shared__ double s_cd[FW][KD_MAX_LEVELS * 64];
    __shared__ alignas(16) double s_rs[LAZY == 1 ? LAZY_WG_LDS : LAZY == 2 ? FW * LAZY_WAVE_LDS : 8];
    const int w = threadIdx.x >> 6;
    // (a batch's trajectories bound to XCDs - XCD c serving the trajectories c mod 8 so that its L2 sees an eighth of the batch's
    // lists - was measured and dropped: 371 against 319 us per c5 batch frame)
    const unsigned bx = blockIdx.x, by = blockIdx.y;
    if ((int)bx < n_pu) {
        if (LAZY == 1) lazy_tables(a.rs, s_rs);
        const int64_t wave = (int64_t)bx * FW + w;
        if (wave < nwaves) {
            // one-wave workgroups = the small-set regime (see launch_frame_front): screened scans
            constexpr bool SCREEN = FW == 1 && MIDAS_SCREEN && SCR;
            const int traj = (int)by;
            if (LAZY == 2) particle_update_wave<true, SCREEN, PREF, STATS, !SCR && !STATS>(t6, t3, a, wave, nwaves, traj, s_cd[w], s_rs + w * LAZY_WAVE_LDS);
            else particle_update_wave<false, SCREEN, PREF, STATS>(t6, t3, a, wave, nwaves, traj, s_cd[w], LAZY ? s_rs : nullptr);
        }
    } else if (a.sp.list) {  // prediction list: the rows the previous frame used, four per wave-instruction
        if ((int)bx == n_pu && threadIdx.x == 0 && a.telemetry) {  // rows scored off the list (cumulative, for the bench's byte count)
            const int c = *a.sp.list_count;
            if (a.sp.dense_thr > 0 && c > a.sp.dense_thr) atomicAdd(&a.telemetry[3], (unsigned long long)a.sp.K);  // all of them
            else if (c > 0) atomicAdd(&a.telemetry[3], (unsigned long long)(c < a.sp.list_cap ? c : a.sp.list_cap));
        }
        if (!(STATS && (a.ablate & 64)))  // ablate 64 (profiling): the list is not scored - what its stream costs the particle waves
            score_list_wave<NJ>(a.sp, (int)(bx - n_pu) * FW + w, ((int)gridDim.x - n_pu) * FW);
    } else {
        // all K rows (the dense K1 beside the particle waves): MIDAS_SCORE_ROUNDS consecutive quads of rows a wave, requested
        // together (score_wave_multi, score_body.hpp)
        const int64_t w0 = ((int64_t)(bx - n_pu) * FW + w) * MIDAS_SCORE_ROUNDS;
        if (w0 * 4 < K) score_wave_multi<T, NJ, MIDAS_SCORE_ROUNDS>(emb, norms, code, scores, K, w0);
    }
    FF_END;
}

// =================================================================================================
// two-kernel form of the front: (A) resample prologue + propagate + feature, beside the codebook scoring;
// (B) nearest neighbour + prune with FOUR lanes per particle
// =================================================================================================
// At N = 100k the particle waves of the single front kernel are 1.5 per SIMD and every one of them walks its whole chain
// of dependent fetches alone.  The chain's second half - list scans - parallelises over records:
// a quad of lanes fetches the 32 solo records of the neighbour list (then the 16 of the vertex list) in ONE round trip
// instead of four (two).  That needs four times the waves, which do not fit beside the 127-register front and the
// scoring stream; as a kernel of its own (no propagate state, no scoring) they do.  The hand-over is 32 bytes per
// particle (6-d feature + hint); the results are the ones of the single-kernel form bit for bit (exact NN with the same
// tie rule, the same "first event in record order" of the prune list).
struct alignas(16) PuFeat { float f[6]; int32_t hint; int32_t pad; };
static_assert(sizeof(PuFeat) == 32, "two 16-byte pieces per particle");

// part A of a particle wave: what particle_update_wave does before the nearest-neighbour search, plus its rmse epilogue
MD void particle_front_wave(ParticleUpdateArgs a, int64_t wave, const double* rs_lds, PuFeat* __restrict__ feat) {
    const int lane = threadIdx.x & 63;
    if (a.n_live) {
        const int64_t nl = *a.n_live;
        a.N = nl < a.N ? nl : a.N;
        if (wave * 64 >= a.N && wave != 0) return;
    }
    const int64_t n = wave * 64 + lane;
    const bool live = n < a.N;
    if (wave == 0 && lane == 0) {
        if (a.status_reset) { a.status_reset[0] = 0; a.status_reset[1] = 0; }
        if (a.flags_reset) { a.flags_reset[0] = 0.0; a.flags_reset[1] = 0.0; }
        if (a.sp.next_count) *a.sp.next_count = 0;  // this frame's tail appends the next frame's prediction list
    }
    double et2 = 0.0, ang2 = 0.0;
    int64_t src = n;
    if (rs_lds && live) {
        src = lazy_source(a.rs, rs_lds, n, a.N);
        if (a.rs.ridx_out) a.rs.ridx_out[n] = (int32_t)src;
    }
    const float* pose_src = rs_lds ? a.rs.poses_prev : a.poses_in;
    if (live) {
        float P[16], O[16], R[16], f[6];
        load_pose(pose_src + src * 16, P);
        const int32_t hint = rs_lds ? a.rs.nn_prev[src] : a.hint_in ? a.hint_in[n] : -1;
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = a.odom16[i];
        propagate_one(n, n + a.slot_base, P, O, a.tn, a.rot, a.std_t, a.std_r, a.seed, a.step, R);
        store_pose(a.poses_prop + n * 16, R);
        se3_feature(R, 0.99f, 0.01f, f);
        float4* o = reinterpret_cast<float4*>(feat + n);
        o[0] = make_float4(f[0], f[1], f[2], f[3]);
        o[1] = make_float4(f[4], f[5], __int_as_float(hint), 0.f);
        if (a.gt16) {
            float G[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) G[i] = a.gt16[i];
            rmse_terms(R, G, et2, ang2);
        }
    }
    if (a.gt16) {
        et2 = wave_sum(et2);
        ang2 = wave_sum(ang2);
        if (lane == 0) { a.part_rmse[2 * wave] = et2; a.part_rmse[2 * wave + 1] = ang2; }
    }
}

template <typename T, int NJ, bool LAZY>
__global__ __launch_bounds__(256) void k_frame_front_a(ParticleUpdateArgs a, int n_pu, int nwaves, PuFeat* __restrict__ feat,
                                                       const T* __restrict__ emb, const double* __restrict__ norms,
                                                       const double* __restrict__ code, double* __restrict__ scores, int64_t K) {
    __shared__ double s_rs[LAZY ? LAZY_WG_LDS : 8];
    const int w = threadIdx.x >> 6;
    if ((int)blockIdx.x < n_pu) {
        if (LAZY) lazy_tables(a.rs, s_rs);
        const int64_t wave = (int64_t)blockIdx.x * 4 + w;
        if (wave < nwaves) particle_front_wave(a, wave, LAZY ? s_rs : nullptr, feat);
    } else {
        score_wave<T, NJ, 0>(emb, norms, code, scores, K, (int64_t)(blockIdx.x - n_pu) * 4 + w);
    }
}

template <int CTRL>
MD float dpp_f32(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }

// part B: a wave = 64/LPP particles x LPP lanes (4 or 2); lane g of a group takes records g, g + LPP, ... of its
// particle's lists: with four lanes the 32 solo records of the neighbour list are one round trip of eight records per
// lane, with two lanes two; header + 16 records of the vertex list are one round trip either way.
static_assert(NN_SOLO == 32 && MESH_SOLO == 16, "the group scans fetch 32 / 16 records");
#ifndef MIDAS_NNP_OCC
#define MIDAS_NNP_OCC 2  // the two-kernel form only serves small sets now (<= 10 240 particles, the loop step): registers over occupancy
#endif
template <int LPP>
MD void particle_nn_prune_wg(const TreeView<Kd6>& t6, const TreeView<Kd3>& t3, ParticleUpdateArgs a, const PuFeat* __restrict__ feat) {
    static_assert(LPP == 4 || LPP == 2, "lanes per particle");
    // quad_perm selectors inside a group of LPP lanes: broadcast of its first / last lane
    constexpr int BC_FIRST = LPP == 4 ? 0x00 : 0xA0, BC_LAST = LPP == 4 ? 0xFF : 0xF5;
    constexpr int PPW = 64 / LPP;          // particles per wave
    constexpr int NN_PASSES = 32 / (8 * LPP);  // round trips of eight records per lane
    constexpr int MESH_PER_LANE = 16 / LPP;
    __shared__ double s_cd[4][KD_MAX_LEVELS * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane & (LPP - 1);
    if (a.n_live) {
        const int64_t nl = *a.n_live;
        a.N = nl < a.N ? nl : a.N;
        if ((int64_t)blockIdx.x * 4 * PPW >= a.N) return;  // whole workgroup past the live set
    }
    const int64_t p = ((int64_t)blockIdx.x * 4 + w) * PPW + lane / LPP;
    const bool live = p < a.N, owner = g == 0;
    const int64_t pc = live ? p : (a.N > 0 ? a.N - 1 : 0);
    const float4* fp = reinterpret_cast<const float4*>(feat + pc);
    const float4 f0 = fp[0], f1 = fp[1];
    const float q[6] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y};
    int32_t hint = live ? __float_as_int(f1.z) : -1;
    // the prune's distance-field cell (MeshField) is requested now, under the search: the translation is in the propagated pose
    const float* pr = a.poses_prop + pc * 16;
    const float tq_f[3] = {pr[3], pr[7], pr[11]};
    const FieldProbe probe = field_fetch(a.field, tq_f, live);
    // ---- nearest codebook entry: the solo records of the hinted entry's list ----
    float best = INFINITY, r_lane = 0.f;
    int64_t bi = 0;
    bool done = !live;
    const bool hinted = live && hint >= 0 && (int64_t)hint < t6.K;
    int32_t piv = hinted ? hint : 0;  // the entry whose list is scanned: the hint, or its twin across the angle-pi cut
    int32_t tw = t6.twin[piv];        // (see nn6_hint_scan_screened: a particle far from the hinted entry has crossed the cut)
    const Nbr6* nb = t6.nbrs + (size_t)piv * NBR_REC;
    int pass = 0;                     // group-uniform: the group's next batch of 8 LPP records
#pragma unroll 1
    while (__any(hinted && !done && pass < NN_PASSES)) {
        const int pc_ = pass < NN_PASSES ? pass : NN_PASSES - 1;  // finished groups re-read their last batch (unused)
        Nbr6 e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = nb[pc_ * 8 * LPP + LPP * j + g];
        float d0 = 0.f, ld = INFINITY;
        int li = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Point6 pt;
#pragma unroll
            for (int c = 0; c < 6; ++c) pt.c[c] = e[j].c[c];
            const float d = dist2(q, pt);
            if (j == 0) d0 = d;  // record 0 (first lane, first pass) is the entry itself: the starting candidate, below
            const bool cand = !(pass == 0 && j == 0 && g == 0);
            if (cand && (d < ld || (d == ld && e[j].idx < li))) { ld = d; li = e[j].idx; }  // NaN never wins
        }
#define MIDAS_GSTEP(CTRL)                                                          \
        {                                                                          \
            const float od = dpp_f32<CTRL>(ld);                                    \
            const int oi = (int)dpp_u32<CTRL>((uint32_t)li);                        \
            if (od < ld || (od == ld && oi < li)) { ld = od; li = oi; }            \
        }
        MIDAS_GSTEP(DPP_XOR1)
        if (LPP == 4) MIDAS_GSTEP(DPP_XOR2)
#undef MIDAS_GSTEP
        d0 = dpp_f32<BC_FIRST>(d0);
        const float rho_last = dpp_f32<BC_LAST>(e[7].rho);  // the largest rho fetched so far
        bool flip = false;
        if (hinted && !done && pass < NN_PASSES) {
            if (pass == 0) {
                best = d0;  // a NaN distance stays, as in the serial scan
                bi = piv;
                r_lane = __builtin_sqrtf(d0);
            }
            if (ld < best || (ld == best && (int64_t)li < bi)) { best = ld; bi = li; }
            // every record behind the last one fetched is at least this far (lower bound with slack for the rounding
            // of r and rho, as in nn6_hint_scan): nothing unseen can beat or tie the best
            const float gg = fmaf_(rho_last - r_lane, 0.9999996f, -8e-7f * r_lane);
            done = gg > 0.0f && gg * gg * 0.99997f > best;
            flip = pass == 0 && !done && r_lane > FLIP_R && tw >= 0;
        }
        if (flip) {  // the twin's list from its start (once: tw = -1)
            piv = tw;
            tw = -1;
            nb = t6.nbrs + (size_t)piv * NBR_REC;
        } else {
            ++pass;
        }
    }
    hint = hinted ? piv : hint;
    nn6_coop(t6, q, hint, r_lane, best, bi, owner && hinted && !done, done);  // the rest of the list, owners = first lanes
    const bool fb = owner && live && !done;
    wave_search<Kd6, false>(t6, q, best, bi, fb, reinterpret_cast<float*>(s_cd[w]));
    if (a.telemetry) {
        const unsigned long long m = __ballot(fb);
        if (lane == 0 && m) atomicAdd(&a.telemetry[0], (unsigned long long)__popcll(m));
    }
    const int32_t nn = (int32_t)dpp_u32<BC_FIRST>((uint32_t)(int32_t)bi);
    // ---- prune: header + records 1 .. 16 of the entry's vertex list in one round trip, requested BEFORE the row claim of the
    // sparse scoring so that the claim's look at the stamps (a round trip of its own) runs beside it ----
    const double q3[3] = {(double)tq_f[0], (double)tq_f[1], (double)tq_f[2]};
    int mv = live ? field_decide(a.field, probe, a.thr) : -1;  // 1 valid, 0 invalid (the distance field: certain), -1 undecided
    const bool lists_needed = __ballot(live && mv < 0) != 0;    // (wave-uniform)
    double lim = 0.0;
    MeshRec hd, e[MESH_PER_LANE];
    if (a.vlist && lists_needed) {
        const MeshRec* vl = a.vlist + (size_t)(live ? nn : 0) * MESH_REC;
        hd = vl[0];
#pragma unroll
        for (int j = 0; j < MESH_PER_LANE; ++j) e[j] = vl[1 + LPP * j + g];
    }
    RowClaim claim{false, 0u};
    if (a.sp.stamps) {
        claim = claim_rows_issue(a.sp, owner && live, nn, MIDAS_CLAIM_HASH ? reinterpret_cast<int*>(s_cd[w]) : nullptr);
        const int nr = score_claimed_rows_nj(a.sp, claim, nn);
        if (a.telemetry && nr && lane == 0) atomicAdd(&a.telemetry[2], (unsigned long long)nr);
    }
    if (a.vlist && lists_needed) {
        Point3 ph;
        ph.c[0] = hd.c[0]; ph.c[1] = hd.c[1]; ph.c[2] = hd.c[2];
        const double delta = __builtin_sqrt(dist2(q3, ph)) * (1.0 + 1e-12);
        lim = a.thr * (1.0 + 1e-9) + delta + 1e-12;  // as mesh_list_check
        unsigned hits = 0, stops = 0;
#pragma unroll
        for (int j = 0; j < MESH_PER_LANE; ++j) {
            Point3 pt;
            pt.c[0] = e[j].c[0]; pt.c[1] = e[j].c[1]; pt.c[2] = e[j].c[2];
            const int pos = LPP * j + g;  // record 1 + pos
            stops |= ((double)e[j].rho * (1.0 - 1e-7) > lim ? 1u : 0u) << pos;
            hits |= (dist2(q3, pt) <= a.t2 ? 1u : 0u) << pos;
        }
        hits |= dpp_u32<DPP_XOR1>(hits); stops |= dpp_u32<DPP_XOR1>(stops);
        if (LPP == 4) { hits |= dpp_u32<DPP_XOR2>(hits); stops |= dpp_u32<DPP_XOR2>(stops); }
        if (live && mv < 0 && (hits | stops)) {  // the first event in record order decides, "provably too far" before "hit"
            const int fh = hits ? __builtin_ctz(hits) : 32, fs = stops ? __builtin_ctz(stops) : 32;
            mv = fh < fs ? 1 : 0;
        }
        mesh_coop(a.vlist, nn, q3, a.t2, lim, owner && live && mv < 0, mv);
    }
    double bestd = a.t2;
    int64_t vi = 0;
    const bool fb3 = owner && live && mv < 0;
    bool ok = wave_search<Kd3, true>(t3, q3, bestd, vi, fb3, s_cd[w]);
    if (a.telemetry) {
        const unsigned long long m = __ballot(fb3);
        if (lane == 0 && m) atomicAdd(&a.telemetry[1], (unsigned long long)__popcll(m));
    }
    if (mv >= 0) ok = mv == 1;
    if (owner && live) {
        a.nn_idx[p] = nn;
        a.valid[p] = ok ? 1 : 0;
    }
}

template <int LPP>
__global__ __launch_bounds__(256, MIDAS_NNP_OCC) void k_particle_nn_prune(TreeView<Kd6> t6, TreeView<Kd3> t3, ParticleUpdateArgs a,
                                                                          const PuFeat* __restrict__ feat) {
    particle_nn_prune_wg<LPP>(t6, t3, a, feat);
}

// Both parts in ONE launch for the loop step's small sets (no folded resample, sparse scoring: no streaming workgroups): a
// workgroup's first wave is part A for the 64 particles the workgroup's four waves then search with four lanes each.  Part A is
// a chain of round trips that one wave per 64 particles carries as well as four waves per 256 did; what goes is a launch
// boundary (~4 us of a frame of 85) and the first touch of the hand-over records by another launch.
__global__ __launch_bounds__(256, MIDAS_NNP_OCC) void k_front_small(TreeView<Kd6> t6, TreeView<Kd3> t3, ParticleUpdateArgs a, int nwaves,
                                                                    PuFeat* __restrict__ feat) {
    if (threadIdx.x < 64 && (int)blockIdx.x < nwaves) particle_front_wave(a, (int64_t)blockIdx.x, nullptr, feat);
    __syncthreads();  // (drains the first wave's stores of the hand-over records: the other waves read them through the L2)
    particle_nn_prune_wg<4>(t6, t3, a, feat);
}

// =================================================================================================
// launchers
// =================================================================================================
int launch_se3_feature(midas_ctx* ctx, int64_t N, const float* poses, float w, float* feat6) {
    if (N == 0) return MIDAS_OK;
    // (1.0 - w) * t is a python-float times a float32 tensor in the reference: the scalar is rounded to float32
    const float wt = (float)(1.0 - (double)w), wr = w;
    hipLaunchKernelGGL(k_feature, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx->stream, N, poses, wt, wr, feat6);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_propagate(midas_ctx* ctx, int64_t N, const float* in, float* out, const float* odom, const float* tn,
                     const float* rot, float std_t, float std_r, uint64_t seed, uint64_t step) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_propagate, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx->stream, N, in, out, odom, tn,
                       rot, std_t, std_r, seed, step);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_check_poses(midas_ctx* ctx, int64_t N, const float* poses, uint8_t* flag, int32_t* count) {
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, sizeof(int32_t), ctx->stream));
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_check_poses, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, poses, flag, count);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_nn6(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, const int32_t* hint, int32_t* idx,
               float* d2) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_nn6, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx->stream, view_of<Kd6>(t), N, feat6, hint,
                       idx, d2);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// ---- exact k nearest codebook entries (tactile_tree.py:43-58 with n_neighbors > 1) --------------------------------------
// One wave per query, brute force over the tree's leaf slots (empty slots carry +inf coordinates): every lane keeps the k
// best of the slots it visits in its own LDS column, sorted by (distance, index); the wave then merges the 64 columns,
// taking the smallest head k times.  Distances are the spec's dist2 chain, ties go to the smaller index, so column 0 of
// the result is what midas_nn6 returns.  Not on the filter's path (it uses nn = 1): a query costs one pass over the codebook.
__global__ __launch_bounds__(64) void k_knn6(TreeView<Kd6> tv, int64_t N, const float* __restrict__ feat, int k,
                                            int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    extern __shared__ unsigned char s_knn[];
    float* s_d = reinterpret_cast<float*>(s_knn);
    int* s_i = reinterpret_cast<int*>(s_d + (size_t)k * 64);
    const int lane = threadIdx.x & 63;
    const int64_t n = blockIdx.x;
    float q[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) q[a] = feat[n * 6 + a];
    for (int s = 0; s < k; ++s) { s_d[s * 64 + lane] = INFINITY; s_i[s * 64 + lane] = 0x7fffffff; }
    const int64_t nslots = ((int64_t)LEAF_CAP) << (3 * tv.levels);
    float worst_d = INFINITY;
    int worst_i = 0x7fffffff;
    for (int64_t slot = lane; slot < nslots; slot += 64) {
        const Point6 p = tv.pts[slot];
        const float d = dist2(q, p);
        const int id = p.idx;
        if (d < worst_d || (d == worst_d && id < worst_i)) {  // NaN never enters
            int pos = k - 1;
            while (pos > 0) {
                const float pd = s_d[(pos - 1) * 64 + lane];
                const int pi = s_i[(pos - 1) * 64 + lane];
                if (pd < d || (pd == d && pi < id)) break;
                s_d[pos * 64 + lane] = pd;
                s_i[pos * 64 + lane] = pi;
                --pos;
            }
            s_d[pos * 64 + lane] = d;
            s_i[pos * 64 + lane] = id;
            worst_d = s_d[(k - 1) * 64 + lane];
            worst_i = s_i[(k - 1) * 64 + lane];
        }
    }
    int ptr = 0;
    for (int r = 0; r < k; ++r) {
        const float hd = ptr < k ? s_d[ptr * 64 + lane] : INFINITY;
        const int hi = ptr < k ? s_i[ptr * 64 + lane] : 0x7fffffff;
        float bd = hd;
        int bi = hi;
        wave_best(bd, bi);
        if (lane == 0) {
            idx_out[n * k + r] = bi;
            if (d2_out) d2_out[n * k + r] = bd;
        }
        if (hi == bi && bi != 0x7fffffff) ++ptr;  // an entry sits in exactly one column
    }
}

int launch_knn6(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, int32_t k, int32_t* idx, float* d2) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_knn6, dim3((unsigned)N), dim3(64), (size_t)k * 64 * 8, ctx->stream, view_of<Kd6>(t), N, feat6, (int)k,
                       idx, d2);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_nn6_stats(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, const int32_t* hint,
                     int32_t* leaves, int32_t* nodes) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_nn6_stats, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx->stream, view_of<Kd6>(t), N, feat6,
                       hint, leaves, nodes);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

static int build_mesh_field(midas_ctx* ctx, midas_tree* t, const double* pts, int64_t K) {
    const char* env = getenv("MIDAS_MESH_FIELD");
    if ((env && env[0] == '0') || K <= 0) return MIDAS_OK;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t k = 0; k < K; ++k)
        for (int d = 0; d < 3; ++d) {
            const double v = pts[3 * k + d];
            if (!(v == v) || std::isinf(v)) return MIDAS_OK;  // non-finite vertices: no field (the exact paths deal with them as before)
            lo[d] = v < lo[d] ? v : lo[d];
            hi[d] = v > hi[d] ? v : hi[d];
        }
    double ext[3], vol = 1.0;
    for (int d = 0; d < 3; ++d) { ext[d] = (hi[d] - lo[d]) + 2.0 * FIELD_EXPAND * 1.01; vol *= ext[d]; }
    double h = std::cbrt(vol / (double)FIELD_MAX_CELLS);
    if (h < FIELD_MIN_CELL) h = FIELD_MIN_CELL;  // (a small mesh does not need the whole budget: the shell is thin enough)
    MeshField f;
    for (int iter = 0; iter < 8; ++iter) {  // (rounding the counts up can exceed the budget: grow the cells a little)
        int64_t cells = 1;
        for (int d = 0; d < 3; ++d) { f.n[d] = (int32_t)std::ceil(ext[d] / h) + 1; cells *= f.n[d]; }
        if (cells <= FIELD_MAX_CELLS) break;
        h *= 1.03;
    }
    f.h = (float)h;
    f.inv_h = 1.0f / f.h;
    // the grid's corner: at or below lo - 1.01 expand as a float32 (a point "outside" must really be beyond the grown box)
    for (int d = 0; d < 3; ++d) f.lo[d] = std::nextafter((float)(lo[d] - FIELD_EXPAND * 1.01), -INFINITY);
    // ... and the far faces: n cells of f.h must reach hi + expand (the counts were taken with the double h: check with the float)
    for (int d = 0; d < 3; ++d)
        while ((double)f.lo[d] + (double)f.n[d] * (double)f.h < hi[d] + FIELD_EXPAND * 1.005) ++f.n[d];
    f.expand = (float)FIELD_EXPAND;
    const int64_t cells = (int64_t)f.n[0] * f.n[1] * f.n[2];
    float* dev = nullptr;
    if (hipMalloc((void**)&dev, (size_t)cells * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return MIDAS_OK; }  // no memory: no field
    f.d = dev;
    const TreeView<Kd3> tv = view_of<Kd3>(t);
    const int64_t SLAB = (int64_t)1 << 24;  // cells per launch
    for (int64_t c0 = 0; c0 < cells; c0 += SLAB) {
        const int64_t n = cells - c0 < SLAB ? cells - c0 : SLAB;
        hipLaunchKernelGGL(k_field_build, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, ctx->stream, tv, f, c0, cells, dev);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(dev);
        return midas_set_error(ctx, MIDAS_ERR_HIP, "k_field_build", "building the mesh's distance field failed");
    }
    t->field = f;
    return MIDAS_OK;
}

int launch_nn3(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* poses, double* dist) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_nn3, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx->stream, view_of<Kd3>(t), N, poses, dist);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_rmse(midas_ctx* ctx, int64_t N, const float* poses, const float* gt16, double* out2) {
    MIDAS_REQUIRE(ctx, N > 0);
    const int nb = (int)ceil_div(N, 64);
    void* part;
    int rc = midas_scratch(ctx, (size_t)nb * 2 * sizeof(double), &part);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rmse_part, dim3(nb), dim3(64), 0, ctx->stream, N, poses, gt16, (double*)part);
    hipLaunchKernelGGL(k_rmse_final, dim3(1), dim3(256), 0, ctx->stream, N, nb, (const double*)part, out2);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int particle_update_blocks(int64_t N) { return (int)ceil_div(N, 64); }

int launch_particle_update(midas_ctx* ctx, const midas_tree* t6, const midas_tree* t3, const ParticleUpdateArgs& a_in) {
    if (a_in.N == 0) return MIDAS_OK;
    ParticleUpdateArgs a = a_in;
    // MIDAS_ABLATE (profiling only, results become wrong): bit 0 skips the NN search, bit 1 the mesh prune
    static const int ablate = getenv("MIDAS_ABLATE") ? atoi(getenv("MIDAS_ABLATE")) : 0;
    a.ablate = ablate;
    const dim3 grid((unsigned)particle_update_blocks(a.N), (unsigned)(a.batch > 1 ? a.batch : 1));
    if (ablate) hipLaunchKernelGGL(k_particle_update<true>, grid, dim3(64), 0, ctx->stream, view_of<Kd6>(t6), view_of<Kd3>(t3), a);
    else hipLaunchKernelGGL(k_particle_update<false>, grid, dim3(64), 0, ctx->stream, view_of<Kd6>(t6), view_of<Kd3>(t3), a);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// fused front: returns MIDAS_ERR_UNSUPPORTED-like 1 when the codebook layout has no fused instantiation
// =================================================================================================
// presort: the folded resample's sources, and an execution order that groups the slots by their hint
// =================================================================================================
// A particle wave's list scans start from the hinted entry's neighbour and vertex lists.  In slot order a wave's 64 particles start
// from ~50 different entries (c5, frames 20 - 70: the set sits on 200 - 600 entries per trajectory, the multinomial draws scatter
// them over the slots), so every list record a wave touches is a look-up and a line of its own; in an order that keeps equal hints
// together it is 3 - 5 entries per wave: the lanes ask for the SAME addresses.  The hint of slot n is nn_prev[src(n)] - known once
// the resample search is done - so the search moves out of the front into k_presort_search (one lane per slot, the front's own
// functions: same sources), and k_presort_group builds the order per chunk of 16384 slots in one workgroup: an LDS hash table of
// the chunk's hints (count per hint, first come first served), an exclusive scan of the counts, a scatter.  The order inside a
// group is whatever the LDS atomics made it; nothing depends on it (a particle's arithmetic does not know its lane).
constexpr int PS_CHUNK = 4096, PS_THREADS = 1024, PS_PER = PS_CHUNK / PS_THREADS, PS_TAB = 4096;  // (chunks of 16384 slots in one workgroup
// a trajectory: 18 us of serialised LDS atomics on 64 of the 256 CUs)
constexpr int PS_GEND_MAX = 2048;  // chunk ends the search kernel stages in LDS (N <= 32768); beyond, the two line fetches

MD void presort_offset_traj(ParticleUpdateArgs& a, int traj) {
    if (!traj) return;
    const int64_t b = traj, o = b * a.N, ts = b * a.rs.tstride;
    a.rs.e += ts; a.rs.x_raw += ts; a.rs.lp += ts; a.rs.lp_raw += ts; a.rs.gend += ts; a.rs.gend_raw += ts;
    a.rs.ggend += ts; a.rs.ggend_raw += ts; a.rs.bsum_e += ts; a.rs.btot += ts; a.rs.btot_raw += ts; a.rs.bmax += ts; a.rs.bmin += ts;
    a.rs.poses_prev += o * 16; a.rs.nn_prev += o; a.rs.status_prev += 2 * b;
    if (a.rs.ridx_out) a.rs.ridx_out += o;
    if (a.rs.u) a.rs.u += o;
    a.rs.key_base = o;
    a.rs.traj = traj;
}

// slot n -> src[n] (= lazy_source, what the front computes for itself otherwise), hint[n] = nn_prev[src].  Four waves a workgroup:
// every wave builds the block tables for itself (lazy_tables_wave), then the four copy the trajectory's chunk-end table (the
// softmax or the raw variant, as the guard decided) into LDS - N = 10 000: 5 KB - and a lane finds its chunk there; the scattered
// fetches of a search drop from 25 sixteen-byte pieces to 9 (this kernel is bound by the vector cache's look-up rate: 35 -> 15 us).
__global__ __launch_bounds__(256) void k_presort_search(ParticleUpdateArgs a, int32_t* __restrict__ src_out, int32_t* __restrict__ hint_out) {
    __shared__ alignas(16) double s_rs[4][LAZY_WAVE_LDS];
    __shared__ double s_gend[PS_GEND_MAX];
    const int traj = (int)blockIdx.y, t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int64_t o = (int64_t)traj * a.N;
    presort_offset_traj(a, traj);
    const LazyRecords rec = lazy_records_load(a.rs);
    lazy_tables_wave(a.rs, rec, s_rs[w]);
    const bool staged = a.rs.ng <= PS_GEND_MAX;
    if (staged) {
        const bool apply = s_rs[w][2 * LAZY_WAVE_LD + 2] != 0.0;  // (every wave computes the same guard)
        const double* __restrict__ g = apply ? a.rs.gend : a.rs.gend_raw;
        for (int i = t; i < a.rs.ng; i += 256) s_gend[i] = g[i];
        __syncthreads();
    }
    const int64_t n = (int64_t)blockIdx.x * 256 + t;
    if (n >= a.N) return;
    const int64_t src = staged ? lazy_source<lds_cdp>(a.rs, s_rs[w], n, a.N, LAZY_WAVE_LD, (lds_cdp)s_gend)
                               : lazy_source(a.rs, s_rs[w], n, a.N, LAZY_WAVE_LD);
    if (a.rs.ridx_out) a.rs.ridx_out[n] = (int32_t)src;
    src_out[o + n] = (int32_t)src;
    hint_out[o + n] = a.rs.nn_prev[src];
    (void)lane;
}

// chunk c of trajectory b: order[o + base + pos] = slot, srcr[o + base + pos] = its source, equal hints adjacent
// deal > 0: the grouped sequence is dealt to the chunk's waves in runs of `deal` slots (wave w takes runs w, w + W, ...): a wave
// then holds 64 / deal entries' particles instead of one entry's - the particles of a hard entry (whose cooperative
// continuations serve one owner per pass) spread over many waves again, while `deal` lanes still ask for the same records.
__global__ __launch_bounds__(PS_THREADS) void k_presort_group(int64_t N, const int32_t* __restrict__ src, const int32_t* __restrict__ hint,
                                                              int32_t* __restrict__ order, int32_t* __restrict__ srcr, int deal) {
    __shared__ int s_key[PS_TAB];
    __shared__ int s_cnt[PS_TAB];
    __shared__ int s_w[PS_THREADS / 64];
    __shared__ int s_fail;
    const int t = threadIdx.x;
    const int64_t o = (int64_t)blockIdx.y * N, base = (int64_t)blockIdx.x * PS_CHUNK;
    const int64_t end = base + PS_CHUNK < N ? base + PS_CHUNK : N;
    for (int i = t; i < PS_TAB; i += PS_THREADS) { s_key[i] = -2; s_cnt[i] = 0; }
    if (t == 0) s_fail = 0;
    __syncthreads();
    int slot[PS_PER], rk[PS_PER], sv[PS_PER];
    int32_t hv[PS_PER];
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {  // the chunk's hints and sources: coalesced, all in flight together
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        const int64_t nc = n < end ? n : end - 1;
        hv[j] = hint[o + nc];
        sv[j] = src[o + nc];
    }
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        slot[j] = -1; rk[j] = 0;
        if (n < end) {
            const int key = hv[j] < 0 ? -1 : hv[j];
            unsigned h = ((unsigned)key * 2654435761u) >> 20;  // 12 bits
            for (int probe = 0; probe < 64; ++probe) {
                const int old = atomicCAS(&s_key[h], -2, key);
                if (old == -2 || old == key) { slot[j] = (int)h; rk[j] = atomicAdd(&s_cnt[h], 1); break; }
                h = (h + 1) & (PS_TAB - 1);
            }
            if (slot[j] < 0) s_fail = 1;  // more distinct hints than the table takes: slot order for this chunk
        }
    }
    __syncthreads();
    // exclusive scan of the PS_TAB counts: eight per thread
    constexpr int E = PS_TAB / PS_THREADS;
    int v[E], mine = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) { v[k] = s_cnt[t * E + k]; mine += v[k]; }
    const int incl = wave_iscan_dpp(mine);  // (DPP row shifts and broadcasts: midas_math.hpp)
    if ((t & 63) == 63) s_w[t >> 6] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) run += s_w[w];
    const bool fail = s_fail != 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; ++k) { s_cnt[t * E + k] = run; run += v[k]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        if (n < end) {
            int64_t pos = fail ? n - base : (int64_t)(s_cnt[slot[j]] + rk[j]);
            const int64_t W = (end - base) >> 6;  // whole waves of the chunk; the ragged rest keeps its place
            if (deal > 0 && !fail && pos < (W << 6)) {
                const int64_t q = pos / deal, within = pos - q * deal;
                pos = ((q % W) << 6) + (q / W) * deal + within;
            }
            order[o + base + pos] = (int32_t)n;
            srcr[o + base + pos] = sv[j];
        }
    }
}

// Both steps in ONE kernel for trajectories of up to PS_LP_MAX particles (c5: 10 000): a workgroup stages the trajectory's whole
// per-slot prefix table (80 KB) and its chunk-end table in LDS - coalesced - and every level of its slots' searches reads LDS;
// the only scattered fetch left is the hint nn_prev[src].  (The two-kernel form is bound by the vector cache's look-up rate on
// the searches' line fetches: 26 + 11 us at c5; this one is a launch less and ~12 us.)  Same sources, same grouping.
constexpr int PS_LP_MAX = 10240;
struct PresortLds {  // dynamic LDS of k_presort_fused
    double lp[PS_LP_MAX];
    double gend[PS_LP_MAX / SCAN_CHUNK];
    double rs[PS_THREADS / 64][LAZY_WAVE_LDS];
    int key[PS_TAB];
    int cnt[PS_TAB];
    int w[PS_THREADS / 64];
    int fail;
};
__global__ __launch_bounds__(PS_THREADS) void k_presort_fused(ParticleUpdateArgs a, int32_t* __restrict__ order, int32_t* __restrict__ srcr, int deal, int chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_raw[];
    PresortLds& L = *reinterpret_cast<PresortLds*>(ps_raw);
    const int traj = (int)blockIdx.y, t = threadIdx.x, wv = t >> 6;
    // chunk (<= PS_CHUNK, whole waves): slots per workgroup - the launcher cuts a trajectory into as many chunks as fill the CUs
    const int64_t N = a.N, o = (int64_t)traj * N, base = (int64_t)blockIdx.x * chunk;
    const int64_t end = base + chunk < N ? base + chunk : N;
    presort_offset_traj(a, traj);
    const LazyRecords rec = lazy_records_load(a.rs);
    lazy_tables_wave(a.rs, rec, L.rs[wv]);
    {
        const bool apply = L.rs[wv][2 * LAZY_WAVE_LD + 2] != 0.0;  // (every wave computes the same guard)
        const double2* __restrict__ lsrc = reinterpret_cast<const double2*>(apply ? a.rs.lp : a.rs.lp_raw);  // padded to 16 values (tables_of)
        const int n2 = (int)((N + 1) >> 1);
        for (int i = t; i < n2; i += PS_THREADS) reinterpret_cast<double2*>(L.lp)[i] = lsrc[i];
        const double* __restrict__ g = apply ? a.rs.gend : a.rs.gend_raw;
        for (int i = t; i < a.rs.ng; i += PS_THREADS) L.gend[i] = g[i];
        for (int i = t; i < PS_TAB; i += PS_THREADS) { L.key[i] = -2; L.cnt[i] = 0; }
        if (t == 0) L.fail = 0;
    }
    __syncthreads();
    int slot[PS_PER], rk[PS_PER], sv[PS_PER];
    int32_t hv[PS_PER];
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        sv[j] = 0; hv[j] = -1;
        if (n < end) {
            const int64_t src = lazy_source<lds_cdp, lds_cdp>(a.rs, L.rs[wv], n, N, LAZY_WAVE_LD, (lds_cdp)L.gend, (lds_cdp)L.lp);
            if (a.rs.ridx_out) a.rs.ridx_out[n] = (int32_t)src;
            sv[j] = (int)src;
            hv[j] = a.rs.nn_prev[src];
        }
    }
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        slot[j] = -1; rk[j] = 0;
        if (n < end) {
            const int key = hv[j] < 0 ? -1 : hv[j];
            unsigned h = ((unsigned)key * 2654435761u) >> 20;  // 12 bits
            for (int probe = 0; probe < 64; ++probe) {
                const int old = atomicCAS(&L.key[h], -2, key);
                if (old == -2 || old == key) { slot[j] = (int)h; rk[j] = atomicAdd(&L.cnt[h], 1); break; }
                h = (h + 1) & (PS_TAB - 1);
            }
            if (slot[j] < 0) L.fail = 1;
        }
    }
    __syncthreads();
    constexpr int E = PS_TAB / PS_THREADS;
    int v[E], mine = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) { v[k] = L.cnt[t * E + k]; mine += v[k]; }
    const int incl = wave_iscan_dpp(mine);  // (DPP row shifts and broadcasts: midas_math.hpp)
    if ((t & 63) == 63) L.w[t >> 6] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) run += L.w[w];
    const bool fail = L.fail != 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; ++k) { L.cnt[t * E + k] = run; run += v[k]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) {
        const int64_t n = base + (int64_t)j * PS_THREADS + t;
        if (n < end) {
            int64_t pos = fail ? n - base : (int64_t)(L.cnt[slot[j]] + rk[j]);
            const int64_t W = (end - base) >> 6;
            if (deal > 0 && !fail && pos < (W << 6)) {
                const int64_t q = pos / deal, within = pos - q * deal;
                pos = ((q % W) << 6) + (q / W) * deal + within;
            }
            order[o + base + pos] = (int32_t)n;
            srcr[o + base + pos] = sv[j];
        }
    }
}

// per-wave rmse sums in SLOT order from the presorted front's per-slot terms: exactly what an unsorted wave leaves in part_rmse
__global__ __launch_bounds__(64) void k_rmse_parts(int64_t N, int nwaves, const double* __restrict__ terms, double* __restrict__ part_rmse) {
    const int64_t b = blockIdx.y, n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const double2 v = n < N ? reinterpret_cast<const double2*>(terms + 2 * b * N)[n] : make_double2(0.0, 0.0);
    const double p = wave_sum(v.x), q = wave_sum(v.y);
    if (threadIdx.x == 0) { part_rmse[2 * (b * nwaves + blockIdx.x)] = p; part_rmse[2 * (b * nwaves + blockIdx.x) + 1] = q; }
}

// the two launches in front of a frame front with folded resample and per-wave tables; fills a.pre_order / a.pre_src
static int launch_presort(midas_ctx* ctx, ParticleUpdateArgs& a) {
    void *p_src = nullptr, *p_hint = nullptr, *p_order, *p_srcr;
    const size_t bytes = (size_t)a.batch * (size_t)a.N * sizeof(int32_t);
    int rc;
    if ((rc = midas_scratch(ctx, bytes, &p_order))) return rc;
    if ((rc = midas_scratch(ctx, bytes, &p_srcr))) return rc;
    static const int run_env = getenv("MIDAS_PRESORT_RUN") ? atoi(getenv("MIDAS_PRESORT_RUN")) : 8;
    const int run = (run_env == 1 || run_env == 2 || run_env == 4 || run_env == 8 || run_env == 16 || run_env == 32) ? run_env : 0;  // divisors of 64; else none
    static const bool fused_env = !(getenv("MIDAS_PRESORT_FUSED") && getenv("MIDAS_PRESORT_FUSED")[0] == '0');
    if (fused_env && a.N <= PS_LP_MAX) {
        // per device: the dynamic-LDS limit of the kernel and the CU count (a second GPU's context must not inherit the first's)
        constexpr int MAXDEV = 64;
        static bool attr_set[MAXDEV] = {};
        static int ncu_dev[MAXDEV] = {};
        const int di = ctx->device >= 0 && ctx->device < MAXDEV ? ctx->device : 0;
        if (!attr_set[di] || ctx->device != di) {
            MIDAS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_presort_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PresortLds)));
            attr_set[di] = true;
        }
        // One workgroup per CU is all the kernel's LDS allows, and a workgroup's life is a chain of round trips whatever its share: as many
        // chunks per trajectory as fill the chip (c5: 64 trajectories x 4 chunks of 2560 slots on 256 CUs instead of 3 of 4096 -
        // 285 / 270 -> 277 / 266 us per batch frame; 5 or 8 chunks - a second round of workgroups - lose: 290 / 285), whole waves each
        if (!ncu_dev[di] || ctx->device != di) {
            hipDeviceProp_t prop;
            ncu_dev[di] = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        const int ncu = ncu_dev[di];
        static const int chunk_env = getenv("MIDAS_PRESORT_CHUNK") ? atoi(getenv("MIDAS_PRESORT_CHUNK")) : 0;
        int64_t nch = ceil_div(a.N, PS_CHUNK);
        if (ncu / a.batch > nch) nch = ncu / a.batch;
        int64_t chunk = ceil_div(ceil_div(a.N, nch), 64) * 64;
        if (chunk < 1024) chunk = 1024;  // (a chunk groups its own slots only: small ones share few list records)
        if (chunk_env >= 64 && chunk_env <= PS_CHUNK && chunk_env % 64 == 0) chunk = chunk_env;
        hipLaunchKernelGGL(k_presort_fused, dim3((unsigned)ceil_div(a.N, chunk), (unsigned)a.batch), dim3(PS_THREADS), sizeof(PresortLds), ctx->stream,
                           a, (int32_t*)p_order, (int32_t*)p_srcr, run, (int)chunk);
    } else {
        if ((rc = midas_scratch(ctx, bytes, &p_src))) return rc;  // (the two-kernel form hands sources and hints over through memory)
        if ((rc = midas_scratch(ctx, bytes, &p_hint))) return rc;
        hipLaunchKernelGGL(k_presort_search, dim3((unsigned)ceil_div(a.N, 256), (unsigned)a.batch), dim3(256), 0, ctx->stream, a, (int32_t*)p_src, (int32_t*)p_hint);
        hipLaunchKernelGGL(k_presort_group, dim3((unsigned)ceil_div(a.N, PS_CHUNK), (unsigned)a.batch), dim3(PS_THREADS), 0, ctx->stream, a.N,
                           (const int32_t*)p_src, (const int32_t*)p_hint, (int32_t*)p_order, (int32_t*)p_srcr, run);
    }
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    a.pre_order = (const int32_t*)p_order;
    a.pre_src = (const int32_t*)p_srcr;
    if (a.gt16) {
        void* p_terms;
        if ((rc = midas_scratch(ctx, (size_t)a.batch * (size_t)a.N * 2 * sizeof(double), &p_terms))) return rc;
        a.pre_rmse_terms = (double*)p_terms;
    }
    return MIDAS_OK;
}

int launch_frame_front(midas_ctx* ctx, const midas_tree* t6, const midas_tree* t3, const ParticleUpdateArgs& a_in,
                       const midas_codebook* cb, const double* code, double* scores, bool* launched) {
    *launched = false;
    if (a_in.N == 0 || cb->dtype != MIDAS_F32) return MIDAS_OK;
    // a batch of trajectories (grid.y) only in the pipelined form with per-wave tables and sparse scoring (midas_lazy_step_batch)
    if (a_in.batch > 1 && !(a_in.rs.enabled && a_in.rs.nb <= LAZY_WAVE_LD && a_in.sp.stamps)) return MIDAS_OK;
    if ((uintptr_t)cb->emb % 16 != 0 || (uintptr_t)code % 16 != 0) return MIDAS_OK;
    if (cb->D != 512 && cb->D != 256 && cb->D != 128 && cb->D != 1024) return MIDAS_OK;
    ParticleUpdateArgs a = a_in;
    static const int ablate = getenv("MIDAS_ABLATE") ? atoi(getenv("MIDAS_ABLATE")) : 0;
    a.ablate = ablate;
    a.scores = nullptr;  // deferred: the tail gathers the scores
    const int nwaves = particle_update_blocks(a.N), n_pu = (nwaves + 3) / 4;
    if (a.sp.stamps) {  // sparse scoring: the particle waves score the rows they need, no streaming workgroups
        a.sp.emb = (const float*)cb->emb; a.sp.norms = cb->norms; a.sp.code = code; a.sp.scores = scores; a.sp.nj = cb->D / 64;
    }
    // prediction list: scored by streaming workgroups of the single-kernel form only; elsewhere the tags are not honoured
    // (a row stamped pred_tag is then simply stale and gets claimed: same scores)
    static const int list_wgs_env = getenv("MIDAS_LIST_WAVES") ? atoi(getenv("MIDAS_LIST_WAVES")) : 1024;
    const bool use_list = a.sp.stamps && a.sp.list && a.batch <= 1 && list_wgs_env > 0;
    // MIDAS_DENSE_ROWS=<rows>: a frame whose prediction list holds more rows scores the whole codebook with the streaming waves
    // and its particle waves claim nothing (decided on the device, per frame).  Off by default: measured on the frames after a
    // wide start (20k -> 400 distinct rows over the driver's window) it changes nothing (19.2 - 19.5k steps/s either way, thresholds
    // 1500 / 3125 / 6000, 1024 - 4096 streaming waves) - with the prediction lists the claims are no longer what those frames wait for.
    const char* dense_env = getenv("MIDAS_DENSE_ROWS");
    a.sp.K = cb->K;
    a.sp.dense_thr = use_list && dense_env ? atoi(dense_env) : 0;
    const unsigned grid = (unsigned)(n_pu + (a.sp.stamps ? 0 : ceil_div(cb->K, 16)));
    const float* emb = (const float*)cb->emb;
    // Two-kernel form (group-parallel list scans, see k_particle_nn_prune) for small particle sets (round 1's rule was "while
    // its N/16 waves fit the chip at once": 65536 particles).  Measured at K = 50k, D = 512 the pipelined frame
    // gains 6 - 15 % for N = 4k .. 40k; when the particle set is materialised every frame the extra launch boundary only pays
    // off for the smallest sets; at N = 100k the four-lane form (two rounds of waves) loses 4 %, the two-lane form (one
    // round, two trips) 8 %, at N = 1M 7 %: there the single kernel stays.
    // MIDAS_SPLIT_FRONT = 0 never, 2 always with 4 lanes per particle, 3 always with 2.
    static const int split_env = getenv("MIDAS_SPLIT_FRONT") ? atoi(getenv("MIDAS_SPLIT_FRONT")) : 1;
    // a live count in device memory (loop engine): the two-kernel form while the caller's bound of the count (a.N here) is small -
    // the set shrinks within a few frames of annealing; a set held at 100k takes the single kernel (42 -> ~22 us at N = 100k)
    // (re-measured after the single kernel's second pass - per-wave tables, one-wave workgroups, screened scans: pipelined,
    // split / single at N = 6k 29.6k / 27.5k steps/s, 8k 29.0k / 27.9k, 12k 28.3k / 28.5k, 20k 25.8k / 27.1k, 65k 21.0k / 22.9k:
    // the two-kernel form now pays up to ~10 000 particles instead of 65 536; the loop step (live count) keeps it: 126 / 138 us)
    // (round 4, with the guide tables in the folded search: the single kernel wins from N = 1000 up - pipelined, split / single at
    // N = 1k 27.4 / 26.4 us a step, 3k 28.8 / 26.8, 8k 28.7 / 26.7, 12k 27.0 / 27.1, 20k 27.5 / 27.5 - so the pipelined form splits no more)
    const bool split_front = a.batch <= 1 && !a.inbox.rows && (split_env >= 2 || (split_env == 1 && ((a.rs.enabled && a.N <= 512) || (!a.rs.enabled && a.N <= 2048) || (a.n_live && a.N <= 16384))));
    const int lpp = split_env == 3 ? 2 : 4;
    if (split_front && !(a.ablate & 7)) {
        a.sp.pred_tag = 0; a.sp.list = nullptr;  // (next_count stays: the tail appends whatever form the front had)
        void* feat;
        int rc = midas_scratch(ctx, (size_t)a.N * sizeof(PuFeat), &feat);
        if (rc) return rc;
        static const bool small_env = !(getenv("MIDAS_FRONT_SMALL") && atoi(getenv("MIDAS_FRONT_SMALL")) == 0);
        if (small_env && !a.rs.enabled && a.sp.stamps && lpp == 4) {  // (no streaming workgroups beside part A: grid == n_pu)
            hipLaunchKernelGGL(k_front_small, dim3((unsigned)ceil_div(a.N, 64)), dim3(256), 0, ctx->stream, view_of<Kd6>(t6), view_of<Kd3>(t3),
                               a, nwaves, (PuFeat*)feat);
            MIDAS_HIP_CHECK(ctx, hipGetLastError());
            *launched = true;
            return MIDAS_OK;
        }
#define MIDAS_FRONT_A(NJ)                                                                                            \
    if (a.rs.enabled)                                                                                                \
        hipLaunchKernelGGL((k_frame_front_a<float, NJ, true>), dim3(grid), dim3(256), 0, ctx->stream, a, n_pu, nwaves,  \
                           (PuFeat*)feat, emb, cb->norms, code, scores, cb->K);                                      \
    else                                                                                                             \
        hipLaunchKernelGGL((k_frame_front_a<float, NJ, false>), dim3(grid), dim3(256), 0, ctx->stream, a, n_pu, nwaves, \
                           (PuFeat*)feat, emb, cb->norms, code, scores, cb->K)
        switch (cb->D) {
            case 512: MIDAS_FRONT_A(8); break;
            case 256: MIDAS_FRONT_A(4); break;
            case 128: MIDAS_FRONT_A(2); break;
            default: MIDAS_FRONT_A(16); break;
        }
#undef MIDAS_FRONT_A
        if (lpp == 4)
            hipLaunchKernelGGL(k_particle_nn_prune<4>, dim3((unsigned)ceil_div(a.N, 64)), dim3(256), 0, ctx->stream,
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, (const PuFeat*)feat);
        else
            hipLaunchKernelGGL(k_particle_nn_prune<2>, dim3((unsigned)ceil_div(a.N, 128)), dim3(256), 0, ctx->stream,
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, (const PuFeat*)feat);
        MIDAS_HIP_CHECK(ctx, hipGetLastError());
        *launched = true;
        return MIDAS_OK;
    }
    // the single kernel: workgroup-level tables (more than 64 summation blocks) or per-wave tables, then FW waves per
    // workgroup (MIDAS_FRONT_WAVES = 1 | 4; default 1 with sparse scoring, 4 beside the streaming workgroups)
    static const int fw_env = getenv("MIDAS_FRONT_WAVES") ? atoi(getenv("MIDAS_FRONT_WAVES")) : 0;
    static const int wt_env = getenv("MIDAS_WAVE_TABLES") ? atoi(getenv("MIDAS_WAVE_TABLES")) : 1;
    const bool wave_tables = a.rs.enabled && a.rs.nb <= LAZY_WAVE_LD && wt_env != 0;
    const int fw = a.batch > 1 ? 1 : (a.rs.enabled && !wave_tables) ? 4 : fw_env == 1 || fw_env == 4 ? fw_env : (a.sp.stamps ? 1 : 4);
    const int n_pu_fw = (nwaves + fw - 1) / fw;
    if (!use_list) { a.sp.pred_tag = 0; a.sp.list = nullptr; }
    // presort (see k_presort_search): the batch step (grid.y trajectories), on by default, MIDAS_PRESORT=0 switches it off.
    // Measured at c5 (profiles/r04_c5_presort.txt): grouped and dealt to the waves in runs of 8 (MIDAS_PRESORT_RUN), 298 / 287 us per
    // batch frame against 313 / 303 without - the front itself drops from ~290 to ~237 us, the two launches in front of it cost
    // 53 us (the search they moved out of the front included).  Grouped WITHOUT the deal (a wave = one entry's particles) it loses:
    // a hard entry's particles then share waves, their cooperative continuations (one owner's list per pass) queue up inside a
    // wave instead of spreading over the launch, and the front swings between 180 and 480 us (362 / 331 us).  Compiled into the
    // batch kernels only (SCR = false): tried in the single-trajectory front too, the two launches cost c2 more than the front's
    // whole list phase (14.5k against 23.7k steps/s) and the untaken branches 2 us.
    static const int presort_env = getenv("MIDAS_PRESORT") ? atoi(getenv("MIDAS_PRESORT")) : 1;
    if (wave_tables && fw == 1 && a.batch > 1 && !a.inbox.rows && !a.ablate && !a.n_live && presort_env != 0) {
        const int rc = launch_presort(ctx, a);
        if (rc) return rc;
    }
    const unsigned grid_fw = (unsigned)(n_pu_fw + (a.sp.stamps ? (use_list ? (list_wgs_env + fw - 1) / fw : 0) : ceil_div(cb->K, 4 * fw * MIDAS_SCORE_ROUNDS)));
    // profiling instantiations (MIDAS_ABLATE != 0; D = 512, one-wave workgroups): phase clocks, scan statistics, ablation switches
    if (a.ablate && cb->D == 512 && fw == 1) {
        bool done = true;
        if (a.batch > 1)
            hipLaunchKernelGGL((k_frame_front<float, 8, 2, 1, false, false, true>), dim3(grid_fw, (unsigned)a.batch), dim3(64), 0, ctx->stream,
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K);
        else if (!a.rs.enabled)
            hipLaunchKernelGGL((k_frame_front<float, 8, 0, 1, true, false, true>), dim3(grid_fw), dim3(64), 0, ctx->stream,
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K);
        else if (wave_tables && a.N <= 131072)
            hipLaunchKernelGGL((k_frame_front<float, 8, 2, 1, true, true, true>), dim3(grid_fw), dim3(64), 0, ctx->stream,
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K);
        else done = false;
        if (done) {
            MIDAS_HIP_CHECK(ctx, hipGetLastError());
            *launched = true;
            return MIDAS_OK;
        }
    }
    // the vertex-list prefetch (PREF) also in the form without folded resample: eager engine 58.8 -> 55.3 us per frame at c2; not
    // where the particles come from a shard's inbox (57.4 - 58.0 us either way: the row's registers are in use there).
    // MIDAS_PREF_PLAIN=0: off
    static const bool pref_env = !(getenv("MIDAS_PREF_PLAIN") && getenv("MIDAS_PREF_PLAIN")[0] == '0');
    const bool pref_plain = pref_env && !a.inbox.rows;
#define MIDAS_FRONT_L(NJ, LZ, FW)                                                                                     \
    hipLaunchKernelGGL((k_frame_front<float, NJ, LZ, FW>), dim3(grid_fw), dim3(64 * FW), 0, ctx->stream,               \
                       view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K)
#define MIDAS_FRONT(NJ)                                                                                              \
    if (a.batch > 1)                                                                                                  \
        hipLaunchKernelGGL((k_frame_front<float, NJ, 2, 1, false>), dim3(grid_fw, (unsigned)a.batch), dim3(64), 0, ctx->stream, \
                           view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K);  \
    else if (!a.rs.enabled) {                                                                                         \
        if (fw == 1 && a.N <= 131072 && pref_plain)                                                                   \
            hipLaunchKernelGGL((k_frame_front<float, NJ, 0, 1, true, true>), dim3(grid_fw), dim3(64), 0, ctx->stream,   \
                               view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K); \
        else if (fw == 1) MIDAS_FRONT_L(NJ, 0, 1); else MIDAS_FRONT_L(NJ, 0, 4); }                                    \
    else if (!wave_tables) MIDAS_FRONT_L(NJ, 1, 4);                                                                   \
    else if (fw == 1 && a.N <= 131072)                                                                                \
        hipLaunchKernelGGL((k_frame_front<float, NJ, 2, 1, true, true>), dim3(grid_fw), dim3(64), 0, ctx->stream,     \
                           view_of<Kd6>(t6), view_of<Kd3>(t3), a, n_pu_fw, nwaves, emb, cb->norms, code, scores, cb->K); \
    else if (fw == 1) MIDAS_FRONT_L(NJ, 2, 1);                                                                        \
    else MIDAS_FRONT_L(NJ, 2, 4)
    switch (cb->D) {
        case 512: MIDAS_FRONT(8); break;
        case 256: MIDAS_FRONT(4); break;
        case 128: MIDAS_FRONT(2); break;
        default: MIDAS_FRONT(16); break;
    }
#undef MIDAS_FRONT
#undef MIDAS_FRONT_L
    if (a.pre_rmse_terms)  // presorted launch with rmse: the per-wave sums the tail reads, formed in slot order
        hipLaunchKernelGGL(k_rmse_parts, dim3((unsigned)nwaves, (unsigned)a.batch), dim3(64), 0, ctx->stream, a.N, nwaves,
                           (const double*)a.pre_rmse_terms, a.part_rmse);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    *launched = true;
    return MIDAS_OK;
}

// per-wave partials of the particle update -> two extrema and (optionally) two rmse sums
__global__ __launch_bounds__(256) void k_reduce_partials(int np, const double* __restrict__ pmax,
                                                         const double* __restrict__ pmin, const double* __restrict__ prm,
                                                         double* __restrict__ extrema2, double* __restrict__ rmse_sums2) {
    __shared__ double s0[4], s1[4], s2[4], s3[4];
    double a = -INFINITY, b = INFINITY, p = 0.0, q = 0.0;
    bool nan = false;
    for (int i = threadIdx.x; i < np; i += 256) {
        if (pmax) {
            double u = pmax[i], v = pmin[i];
            nan |= (u != u) || (v != v);
            a = u > a ? u : a;
            b = v < b ? v : b;
        }
        if (prm) { p += prm[2 * i]; q += prm[2 * i + 1]; }
    }
    a = wave_max(a);
    b = wave_min(b);
    p = wave_sum(p);
    q = wave_sum(q);
    const bool wnan = __any(nan);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s0[w] = wnan ? NAN : a; s1[w] = wnan ? NAN : b; s2[w] = p; s3[w] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool bad = false;
        for (int i = 0; i < 4; ++i) {
            bad |= s0[i] != s0[i];
            a = s0[i] > a ? s0[i] : a;
            b = s1[i] < b ? s1[i] : b;
        }
        if (extrema2) {
            extrema2[0] = bad ? NAN : a;
            extrema2[1] = bad ? NAN : b;
        }
        if (rmse_sums2) {
            rmse_sums2[0] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
            rmse_sums2[1] = (s3[0] + s3[1]) + (s3[2] + s3[3]);
        }
    }
}

int launch_reduce_partials(midas_ctx* ctx, int np, const double* pmax, const double* pmin, const double* prm,
                           double* extrema2, double* rmse_sums2) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, ctx->stream, np, pmax, pmin, prm, extrema2, rmse_sums2);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

#ifdef MIDAS_DEBUG_CLOCKS
int debug_ff_clocks(long long* io8192, int reset) {
    if (reset) {
        static long long zero[16384];
        return hipMemcpyToSymbol(HIP_SYMBOL(g_ff_clk), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(io8192, HIP_SYMBOL(g_ff_clk), 16384 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
MIDAS_WARM_TU(particles, k_propagate)

}  // namespace midas
