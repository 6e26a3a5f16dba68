// dbscan.hip - particle_filter.cluster_particles(method="euclidean") on the device (modules/particle_filter.py:208-217):
// sklearn.cluster.DBSCAN(eps, min_samples = N / 5).fit(translations).labels_.
//
// The reference runs it on the host every 50th frame; at N = 100 000 that is seconds (neighbour lists of a spread
// cloud) to minutes (a converged one: N^2 / 2 list entries), i.e. the one step that would keep the loop from ever running
// at sensor rate.  min_samples is huge (N / 5), which shapes the algorithm:
//   * uniform grid of side h = 0.577 eps (cube diagonal < eps): all points of a cell are mutual neighbours, candidates
//     live in the 5 x 5 x 5 cells around a point (two cells away is already > eps apart in that axis);
//   * core test: cell population first (no distance at all); then, from the TIGHT bounding box of every cell's points, a
//     lower bound (cells whose box lies wholly within eps of the point count in full) and an upper bound (cells whose box
//     is wholly beyond eps count nothing) over the 125 cells; only a point neither bound decides gets exact tests, and
//     those only against the cells its ball cuts, with an early exit at min_samples.  (Round 2 had the 125-cell population
//     as the only bound: a converged cloud of 100k particles spread over a few cells of 12k - below N / 5 each - sent 30k
//     points into 20k exact tests each, 34 ms a call.)
//   * components: the core points of a cell form a clique -> one representative per cell; a core point only needs ONE
//     partner within eps per neighbouring cell to join that cell's component (lock-free union-find, larger root under
//     smaller, so a root is the smallest particle index of its cluster);
//   * numbering as sklearn's scan does: clusters in the order of their first core point; a border point takes the
//     smallest number among the clusters whose core points reach it.
//   * extent: a cloud of up to 128 cells per axis (0.74 m at eps = 1e-2) indexes its cells directly; a larger one (a wide
//     init_filter start on a big object, a caller's small eps) keeps them in a hash table of the occupied cells (64-bit key =
//     the three cell coordinates, open addressing, 2^21 slots) - same algorithm, the 125 neighbour cells looked up by key.  No
//     extent limit short of 2^21 cells per axis; round 3 fell back to the host's sklearn there;
//   * any number of clusters: up to 62 are ranked in LDS, more by a prefix sum over the root flags (midas_dbscan; the loop
//     step keeps its 62-cluster arrays - min_samples = N / 5 allows about five).
// Predicate (sklearn's KD-tree on float64 copies): ((dx*dx) + dy*dy) + dz*dz <= eps*eps, accumulated in that order.
// Pinned by fixture G9 (labels written by the reference's own cluster_particles) through the oracle's O(N^2) restatement.
#include <cmath>

#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

#define DB_LAUNCH_CHECK(ctx) MIDAS_HIP_CHECK(ctx, hipGetLastError())

constexpr int DB_MAXDIM = 128;                     // cells per axis
constexpr int DB_MAXCELLS = DB_MAXDIM * DB_MAXDIM * DB_MAXDIM;
constexpr int DB_MAXROOTS = LOOP_MAX_CLUSTERS - 1;  // labels -1 .. 62: what the LDS ranking (and the loop step's cluster arrays) hold
constexpr int DB_HASH_BITS = 21;                    // cell coordinates of the hashed form: 0 .. 2^21 - 1 per axis
constexpr unsigned long long DB_EMPTY = ~0ull;
static_assert(DB_MAXCELLS == (1 << DB_HASH_BITS), "the hash table has as many slots as the dense grid has cells");
constexpr double DB_CELL = 0.577;                   // cell side / eps, below 1 / sqrt(3)

struct DbGrid {         // written by k_db_setup
    double ox, oy, oz;  // origin
    double h;           // cell side
    int32_t dx, dy, dz; // cells per axis
    int32_t ncells;
    int32_t n;          // points
    int32_t ms;         // min_samples
    int32_t nroots;
    int32_t err;
    int32_t nwork;      // points whose core test needs distances
    int32_t hashed;     // 1: the cells are the slots of the hash table (cloud wider than DB_MAXDIM cells in some axis)
    int32_t ncore_cells;  // cells that hold a core point (k_db_clique lists them for k_db_link)
};

struct DbArgs {
    int64_t N;              // capacity / host count
    const int32_t* n_dev;   // nullable: live count
    const float* poses;     // x 16
    double eps, r2;
    int64_t min_samples;    // < 0 -> n / 5
    DbGrid* grid;
    float* part;            // [blocks x 6] bounds partials
    int32_t* cell_count;    // [DB_MAXCELLS] population, then cursor
    int32_t* cell_start;    // [DB_MAXCELLS + 1]
    int32_t* cell_rep;      // [DB_MAXCELLS] smallest particle index among the cell's core points (INT_MAX: none)
    int32_t* cell_num;      // [DB_MAXCELLS] cluster number of the cell's core points (-1: none)
    uint32_t* cell_box;     // [6][DB_MAXCELLS] tight bounds of the cell's points (lo x y z, hi x y z) as order-preserving keys
    int32_t* cid;           // [N] cell of particle i
    int32_t* s_orig;        // [N] sorted position -> particle
    float4* s_pt;           // [N] sorted position -> (x, y, z, cell id as int bits)
    uint8_t* s_core;        // [N] by sorted position
    int32_t* parent;        // [N] by particle (core points only)
    int32_t* roots;         // [DB_MAXROOTS + 1]
    int32_t* work;          // [N] sorted positions whose core test needs distances
    int32_t* core_cells;    // [N] the cells that hold a core point (at most one per point)
    unsigned long long* hkeys;  // [DB_MAXCELLS] hashed form: key of the cell in this slot (DB_EMPTY: free)
    int32_t* rank;          // [N + 1] more than DB_MAXROOTS clusters: root flag by particle, then its exclusive prefix sum
    int32_t max_clusters;   // 0: any number; otherwise the clusters beyond that many stay unnumbered (err |= 2)
    int32_t* labels;        // [N] out
    int32_t* ncl_out;       // out: number of clusters
    int32_t* err_out;       // nullable: |= 2 cluster limit, |= 32 non-finite coordinates / more than 2^21 cells per axis, |= 64 wide cloud of more than 2^20 points
};

__device__ __forceinline__ int64_t db_n(const DbArgs& a) {
    if (!a.n_dev) return a.N;
    const int64_t n = *a.n_dev;
    return n < a.N ? n : a.N;
}

__global__ __launch_bounds__(256) void k_db_bounds(DbArgs a) {
    __shared__ float s[6][4];
    const int64_t n = db_n(a);
    const int t = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < n; i += (int64_t)gridDim.x * 256) {
        const float* P = a.poses + i * 16;
        const float c[3] = {P[3], P[7], P[11]};
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = c[d] < lo[d] ? c[d] : lo[d]; hi[d] = c[d] > hi[d] ? c[d] : hi[d]; }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float x = __shfl_xor(lo[d], o), y = __shfl_xor(hi[d], o);
            lo[d] = x < lo[d] ? x : lo[d];
            hi[d] = y > hi[d] ? y : hi[d];
        }
    if ((t & 63) == 0)
        for (int d = 0; d < 3; ++d) { s[d][t >> 6] = lo[d]; s[3 + d][t >> 6] = hi[d]; }
    __syncthreads();
    if (t < 6) {
        float v = s[t][0];
        for (int w = 1; w < 4; ++w) v = t < 3 ? (s[t][w] < v ? s[t][w] : v) : (s[t][w] > v ? s[t][w] : v);
        a.part[blockIdx.x * 6 + t] = v;
    }
}

__global__ __launch_bounds__(64) void k_db_setup(DbArgs a, int nblocks) {
    if (threadIdx.x != 0) return;
    const int64_t n = db_n(a);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < nblocks; ++b)
        for (int d = 0; d < 3; ++d) {
            const float x = a.part[b * 6 + d], y = a.part[b * 6 + 3 + d];
            lo[d] = x < lo[d] ? x : lo[d];
            hi[d] = y > hi[d] ? y : hi[d];
        }
    DbGrid g;
    g.h = a.eps * DB_CELL;
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
    int dims[3];
    int err = 0;
    int hashed = 0;
    for (int d = 0; d < 3; ++d) {
        const double ext = (double)hi[d] - (double)lo[d];
        double c = n > 0 ? floor(ext / g.h) + 1.0 : 1.0;
        if (!(c >= 1.0)) c = 1.0;                      // NaN extents
        if (c > (double)DB_MAXDIM) hashed = 1;
        if (c > (double)(1 << DB_HASH_BITS)) { c = (double)(1 << DB_HASH_BITS); err |= 1; }  // (6 km at eps = 1e-2, or infinite coordinates)
        dims[d] = (int)c;
    }
    if (hashed && n > DB_MAXCELLS / 2) { hashed = 0; err |= 2; }  // the table is sized for a load of one half (err 2: a wide cloud of more than 2^20 points)
    if (!hashed)
        for (int d = 0; d < 3; ++d)
            if (dims[d] > DB_MAXDIM) dims[d] = DB_MAXDIM;  // (only with err set: clamped as before)
    g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
    g.ncells = hashed ? DB_MAXCELLS : g.dx * g.dy * g.dz;
    g.hashed = hashed;
    g.n = (int32_t)n;
    g.ms = a.min_samples < 0 ? (int32_t)(n / 5) : (int32_t)a.min_samples;
    g.nroots = 0;
    g.err = err;
    g.nwork = 0;
    g.ncore_cells = 0;
    *a.grid = g;
}

// float32 <-> uint32 keys whose unsigned order is the floats' order (atomicMin / atomicMax on bounds)
__device__ __forceinline__ uint32_t db_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float db_unkey(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

// Squared distance bounds between point p and the tight box of cell c2, IN THE PREDICATE'S ARITHMETIC (float64 on the float32
// coordinates, ((dx dx) + dy dy) + dz dz): for every point q of the cell |p_a - q_a| lies between the per-axis gap and the
// per-axis far distance, and rounding is monotone through the differences, squares and sums - so maxd2 <= r2 means EVERY
// point of the cell passes the exact test and mind2 > r2 means none does.
__device__ __forceinline__ void db_box_bounds(const DbArgs& a, int c2, const float4& p, double& mind2, double& maxd2) {
    double lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = (double)db_unkey(a.cell_box[(size_t)d * DB_MAXCELLS + c2]);
        hi[d] = (double)db_unkey(a.cell_box[(size_t)(3 + d) * DB_MAXCELLS + c2]);
    }
    const double pc[3] = {(double)p.x, (double)p.y, (double)p.z};
    double gmin[3], gmax[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double u = pc[d] - lo[d], v = hi[d] - pc[d];  // both >= 0 inside the box
        gmax[d] = u > v ? u : v;
        const double below = lo[d] - pc[d], above = pc[d] - hi[d];
        const double g = below > above ? below : above;
        gmin[d] = g > 0.0 ? g : 0.0;
    }
    mind2 = gmin[0] * gmin[0]; mind2 += gmin[1] * gmin[1]; mind2 += gmin[2] * gmin[2];
    maxd2 = gmax[0] * gmax[0]; maxd2 += gmax[1] * gmax[1]; maxd2 += gmax[2] * gmax[2];
}

__device__ __forceinline__ void db_cell_coords(const DbGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    // (clamped as doubles: a coordinate at infinity must not reach the int conversion)
    const double hi_x = (double)(g.dx - 1), hi_y = (double)(g.dy - 1), hi_z = (double)(g.dz - 1);
    double fx = floor(((double)x - g.ox) / g.h), fy = floor(((double)y - g.oy) / g.h), fz = floor(((double)z - g.oz) / g.h);
    fx = !(fx >= 0.0) ? 0.0 : fx > hi_x ? hi_x : fx;
    fy = !(fy >= 0.0) ? 0.0 : fy > hi_y ? hi_y : fy;
    fz = !(fz >= 0.0) ? 0.0 : fz > hi_z ? hi_z : fz;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
}
__device__ __forceinline__ unsigned long long db_key64(int cx, int cy, int cz) {
    return (unsigned long long)cx | ((unsigned long long)cy << DB_HASH_BITS) | ((unsigned long long)cz << (2 * DB_HASH_BITS));
}
__device__ __forceinline__ unsigned db_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k & (unsigned)(DB_MAXCELLS - 1);
}
// slot of the cell with this key, inserted if absent (k_db_count only)
__device__ __forceinline__ int db_hash_insert(unsigned long long* hkeys, unsigned long long key) {
    unsigned s = db_hash(key);
    while (true) {
        const unsigned long long cur = __hip_atomic_load(&hkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (int)s;
        if (cur == DB_EMPTY) {
            const unsigned long long old = atomicCAS(&hkeys[s], DB_EMPTY, key);
            if (old == DB_EMPTY || old == key) return (int)s;
        }
        s = (s + 1) & (unsigned)(DB_MAXCELLS - 1);
    }
}
// slot of the cell with this key or -1 (after k_db_count: the table is read-only)
__device__ __forceinline__ int db_hash_find(const unsigned long long* hkeys, unsigned long long key) {
    unsigned s = db_hash(key);
    while (true) {
        const unsigned long long cur = hkeys[s];
        if (cur == key) return (int)s;
        if (cur == DB_EMPTY) return -1;
        s = (s + 1) & (unsigned)(DB_MAXCELLS - 1);
    }
}

// ---- one atomic per distinct cell of a wave ----------------------------------------------------------------------------
// A gathered cloud sits in a handful of cells: per-lane atomics on one counter serialise (8000 particles in three cells:
// 63 us for a counting pass).  The lanes of a wave that share a cell are served by one atomic of their leader; after
// DB_AGG_KEYS distinct cells (a spread cloud: no contention there) the remaining lanes go one by one.
constexpr int DB_AGG_KEYS = 6;
// returns this lane's value of the counter before its own increment, as a per-lane atomicAdd(.., 1) would (any order)
__device__ __forceinline__ int db_cell_fetch_inc(int* counters, int c, bool active) {
    const int lane = threadIdx.x & 63;
    int result = 0;
    unsigned long long todo = __ballot(active);
    for (int it = 0; it < DB_AGG_KEYS && todo; ++it) {
        const int l = __builtin_ctzll(todo);
        const int cl = __shfl(c, l);
        const unsigned long long same = __ballot(active && c == cl) & todo;
        int base = 0;
        if (lane == l) base = atomicAdd(&counters[cl], (int)__popcll(same));
        base = __shfl(base, l);
        if ((same >> lane) & 1ull) result = base + (int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) result = atomicAdd(&counters[c], 1);
    return result;
}
// atomicMin(&arr[c], val) for the active lanes
__device__ __forceinline__ void db_cell_min(int32_t* arr, int c, int32_t val, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    for (int it = 0; it < DB_AGG_KEYS && todo; ++it) {
        const int l = __builtin_ctzll(todo);
        const int cl = __shfl(c, l);
        const unsigned long long same = __ballot(active && c == cl) & todo;
        int32_t m = ((same >> lane) & 1ull) ? val : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int32_t x = __shfl_xor(m, o); m = x < m ? x : m; }
        if (lane == l) atomicMin(&arr[cl], m);
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicMin(&arr[c], val);
}

// the hashed form's table starts empty (a no-op for a cloud the dense grid holds)
__global__ __launch_bounds__(256) void k_db_hclear(DbArgs a) {
    if (!a.grid->hashed) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < DB_MAXCELLS; i += (int64_t)gridDim.x * 256) a.hkeys[i] = DB_EMPTY;
}

__global__ __launch_bounds__(256) void k_db_count(DbArgs a) {
    const DbGrid g = *a.grid;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < g.n;
    int c = 0;
    if (in) {
        const float* P = a.poses + i * 16;
        int cx, cy, cz;
        db_cell_coords(g, P[3], P[7], P[11], cx, cy, cz);
        c = g.hashed ? db_hash_insert(a.hkeys, db_key64(cx, cy, cz)) : (cz * g.dy + cy) * g.dx + cx;
        a.cid[i] = c;
    }
    (void)db_cell_fetch_inc(a.cell_count, c, in);
}

// exclusive scan of the cell populations (one workgroup); the counters become the scatter cursors
__global__ __launch_bounds__(1024) void k_db_scan(DbArgs a) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const DbGrid g = *a.grid;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < g.ncells; base += 1024 * 8) {
        int v[8], mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = base + t * 8 + j;
            v[j] = c < g.ncells ? a.cell_count[c] : 0;
            mine += v[j];
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(incl, o);
            if ((t & 63) >= o) incl += x;
        }
        if ((t & 63) == 63) s_w[t >> 6] = incl;
        __syncthreads();
        int before = s_carry + incl - mine;
        for (int w = 0; w < (t >> 6); ++w) before += s_w[w];
        int run = before;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = base + t * 8 + j;
            if (c < g.ncells) {
                a.cell_start[c] = run; a.cell_count[c] = run; a.cell_rep[c] = 0x7fffffff; a.cell_num[c] = -1;
#pragma unroll
                for (int d = 0; d < 3; ++d) { a.cell_box[(size_t)d * DB_MAXCELLS + c] = 0xFFFFFFFFu; a.cell_box[(size_t)(3 + d) * DB_MAXCELLS + c] = 0u; }
            }
            run += v[j];
        }
        __syncthreads();
        if (t == 1023) s_carry = run;
        __syncthreads();
    }
    if (t == 0) a.cell_start[g.ncells] = s_carry;
}

__global__ __launch_bounds__(256) void k_db_scatter(DbArgs a) {
    const DbGrid g = *a.grid;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < g.n;
    const int c = in ? a.cid[i] : 0;
    const int p = db_cell_fetch_inc(a.cell_count, c, in);
    float x = 0.f, y = 0.f, z = 0.f;
    if (in) {
        const float* P = a.poses + i * 16;
        x = P[3]; y = P[7]; z = P[11];
        a.s_orig[p] = (int32_t)i;
        a.s_pt[p] = make_float4(x, y, z, __int_as_float(c));
    }
    // tight bounds of the cell's points: the lanes of a wave that share a cell reduce first (one atomic per cell and bound)
    const uint32_t k[3] = {db_key(x), db_key(y), db_key(z)};
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(in);
    for (int it = 0; it < DB_AGG_KEYS && todo; ++it) {
        const int l = __builtin_ctzll(todo);
        const int cl = __shfl(c, l);
        const unsigned long long same = __ballot(in && c == cl) & todo;
        const bool mine = (same >> lane) & 1ull;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            uint32_t mn = mine ? k[d] : 0xFFFFFFFFu, mx = mine ? k[d] : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t u = (uint32_t)__shfl_xor((int)mn, o), v = (uint32_t)__shfl_xor((int)mx, o);
                mn = u < mn ? u : mn;
                mx = v > mx ? v : mx;
            }
            if (lane == l) {
                atomicMin(&a.cell_box[(size_t)d * DB_MAXCELLS + cl], mn);
                atomicMax(&a.cell_box[(size_t)(3 + d) * DB_MAXCELLS + cl], mx);
            }
        }
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            atomicMin(&a.cell_box[(size_t)d * DB_MAXCELLS + c], k[d]);
            atomicMax(&a.cell_box[(size_t)(3 + d) * DB_MAXCELLS + c], k[d]);
        }
    }
}

__device__ __forceinline__ bool db_within(const float4& p, const float4& q, double r2) {
    const double dx = (double)p.x - (double)q.x, dy = (double)p.y - (double)q.y, dz = (double)p.z - (double)q.z;
    double d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d <= r2;
}

// walks the 5 x 5 x 5 cells around cell c (own cell included when `own`); f(cell) returns true to stop
template <typename F>
__device__ __forceinline__ void db_for_cells(const DbGrid& g, const unsigned long long* __restrict__ hkeys, int c, bool own, F f) {
    int cx, cy, cz;
    if (g.hashed) {
        const unsigned long long k = hkeys[c];
        const unsigned m = (1u << DB_HASH_BITS) - 1u;
        cx = (int)((unsigned)k & m); cy = (int)((unsigned)(k >> DB_HASH_BITS) & m); cz = (int)((unsigned)(k >> (2 * DB_HASH_BITS)) & m);
    } else {
        cx = c % g.dx; cy = (c / g.dx) % g.dy; cz = c / (g.dx * g.dy);
    }
    for (int z = cz - 2 < 0 ? 0 : cz - 2; z <= (cz + 2 >= g.dz ? g.dz - 1 : cz + 2); ++z)
        for (int y = cy - 2 < 0 ? 0 : cy - 2; y <= (cy + 2 >= g.dy ? g.dy - 1 : cy + 2); ++y)
            for (int x = cx - 2 < 0 ? 0 : cx - 2; x <= (cx + 2 >= g.dx ? g.dx - 1 : cx + 2); ++x) {
                int c2;
                if (g.hashed) {
                    if (x == cx && y == cy && z == cz) c2 = c;
                    else if ((c2 = db_hash_find(hkeys, db_key64(x, y, z))) < 0) continue;  // nobody lives there
                } else {
                    c2 = (z * g.dy + y) * g.dx + x;
                }
                if (c2 == c && !own) continue;
                if (f(c2)) return;
            }
}

// The 5 x 5 x 5 cells around a cell, ONE PER LANE (two turns of a wave: lanes 0 .. 63 take neighbours 0 .. 63, then 64 .. 124): the
// kernels that give a point a whole wave used to walk the 125 cells one after the other with every lane doing the same box
// test - 125 dependent look-ups a point.  nbr = 25 (z + 2) + 5 (y + 2) + (x + 2) offset index; returns the cell or -1 (outside the
// grid, nobody lives there, or the centre cell itself unless `own`).
__device__ __forceinline__ void db_cell_xyz(const DbGrid& g, const unsigned long long* __restrict__ hkeys, int c, int& cx, int& cy, int& cz) {
    if (g.hashed) {
        const unsigned long long k = hkeys[c];
        const unsigned m = (1u << DB_HASH_BITS) - 1u;
        cx = (int)((unsigned)k & m); cy = (int)((unsigned)(k >> DB_HASH_BITS) & m); cz = (int)((unsigned)(k >> (2 * DB_HASH_BITS)) & m);
    } else {
        cx = c % g.dx; cy = (c / g.dx) % g.dy; cz = c / (g.dx * g.dy);
    }
}
__device__ __forceinline__ int db_neighbour_cell(const DbGrid& g, const unsigned long long* __restrict__ hkeys, int c, int cx, int cy, int cz,
                                                 int nbr, bool own) {
    if (nbr >= 125) return -1;
    const int x = cx + nbr % 5 - 2, y = cy + (nbr / 5) % 5 - 2, z = cz + nbr / 25 - 2;
    if (x < 0 || y < 0 || z < 0 || x >= g.dx || y >= g.dy || z >= g.dz) return -1;
    if (nbr == 62) return own ? c : -1;  // the centre
    if (g.hashed) return db_hash_find(hkeys, db_key64(x, y, z));
    return (z * g.dy + y) * g.dx + x;
}

// core <=> at least min_samples points within eps (itself included).  Pass 1, one thread per point, decides what needs no
// distance: the own cell alone reaches min_samples (all of it is within eps), or the 125 cells together cannot.  The rest
// goes to a worklist.
__global__ __launch_bounds__(256) void k_db_core(DbArgs a) {
    DbGrid* gp = a.grid;
    const DbGrid g = *gp;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = p < g.n;
    int c = 0, state = 0;  // 1 core, 0 not core, 2 undecided
    int32_t orig = 0;
    if (in) {
        const float4 me = a.s_pt[p];
        c = __float_as_int(me.w);
        const int cnt = a.cell_start[c + 1] - a.cell_start[c];
        state = cnt >= g.ms ? 1 : 0;
        if (!state) {
            int lower = cnt, upper = cnt;  // the own cell: all of it within eps
            db_for_cells(g, a.hkeys, c, false, [&](int c2) {
                const int pop = a.cell_start[c2 + 1] - a.cell_start[c2];
                if (pop == 0) return false;
                double mind2, maxd2;
                db_box_bounds(a, c2, me, mind2, maxd2);
                if (maxd2 <= a.r2) { lower += pop; upper += pop; }
                else if (mind2 <= a.r2) upper += pop;
                return lower >= g.ms;
            });
            state = lower >= g.ms ? 1 : (upper >= g.ms ? 2 : 0);
        }
        orig = a.s_orig[p];
        a.parent[orig] = orig;
        if (state == 2) a.work[atomicAdd(&gp->nwork, 1)] = (int32_t)p;
        else a.s_core[p] = (uint8_t)state;
    }
    db_cell_min(a.cell_rep, c, orig, in && state == 1);
}

// Pass 2, one wave per undecided point.  The 124 cells around it are classified one per lane (whole box within eps: counts in
// full; box beyond eps: nothing; box cut by the ball: exact tests) and the whole cells added up by the wave; then only the cut
// cells are scanned, 64 candidates at a time, and the count stops as soon as min_samples is reached OR can no longer be reached
// (round 5 walked the 125 cells one by one, every lane the same box test, and counted a point that could not become core to
// the end: 1 - 2 ms of the DBSCAN frame at N = 100k).
__global__ __launch_bounds__(256) void k_db_core_count(DbArgs a) {
    const DbGrid g = *a.grid;
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * 256) >> 6);
    for (int wi = wave; wi < g.nwork; wi += nwaves) {
        const int32_t p = a.work[wi];
        const float4 me = a.s_pt[p];
        const int c = __float_as_int(me.w);
        int cx, cy, cz;
        db_cell_xyz(g, a.hkeys, c, cx, cy, cz);
        int cnt = a.cell_start[c + 1] - a.cell_start[c];  // the own cell: all of it within eps
        int pot = 0;                                        // points of the cut cells not yet looked at
        int c2v[2], popv[2];
        unsigned long long cut[2];
#pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            const int c2 = db_neighbour_cell(g, a.hkeys, c, cx, cy, cz, lane + 64 * turn, false);
            int pop = 0, cls = 0;  // cls 1: whole cell within eps, 2: cut
            if (c2 >= 0) {
                pop = a.cell_start[c2 + 1] - a.cell_start[c2];
                if (pop > 0) {
                    double mind2, maxd2;
                    db_box_bounds(a, c2, me, mind2, maxd2);
                    cls = maxd2 <= a.r2 ? 1 : (mind2 <= a.r2 ? 2 : 0);
                }
            }
            c2v[turn] = c2; popv[turn] = pop;
            cnt += wave_isum_dpp(cls == 1 ? pop : 0);
            pot += wave_isum_dpp(cls == 2 ? pop : 0);
            cut[turn] = __ballot(cls == 2);
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        pot = __builtin_amdgcn_readfirstlane(pot);
#pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            unsigned long long m = cut[turn];
            while (m && cnt < g.ms && cnt + pot >= g.ms) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1;
                const int c2 = __shfl(c2v[turn], l), pop = __shfl(popv[turn], l);
                const int b0 = a.cell_start[c2], e = b0 + pop;
                pot -= pop;
                for (int q0 = b0; q0 < e && cnt < g.ms; q0 += 256) {  // four tiles of candidates requested together (one round trip)
                    float4 cand[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int q = q0 + 64 * u + lane; cand[u] = a.s_pt[q < e ? q : e - 1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = q0 + 64 * u + lane;
                        cnt += __popcll(__ballot(q < e && db_within(me, cand[u], a.r2)));
                    }
                }
            }
        }
        if (lane == 0) {
            const bool core = cnt >= g.ms;
            a.s_core[p] = core ? 1 : 0;
            if (core) atomicMin(&a.cell_rep[c], a.s_orig[p]);
        }
    }
}

__device__ __forceinline__ int32_t db_find(int32_t* parent, int32_t i) {
    int32_t r = i;
    while (true) {
        const int32_t pr = __atomic_load_n(&parent[r], __ATOMIC_RELAXED);
        if (pr == r) break;
        r = pr;
    }
    return r;
}

// lock-free union: the larger root is hooked under the smaller one, so a root is its component's smallest index
__device__ __forceinline__ void db_union(int32_t* parent, int32_t x, int32_t y) {
    while (true) {
        x = db_find(parent, x);
        y = db_find(parent, y);
        if (x == y) return;
        if (x > y) { const int32_t t = x; x = y; y = t; }
        if (atomicCAS(&parent[y], y, x) == y) return;
    }
}

// the core points of a cell are a clique: hang each under the cell's representative
__global__ __launch_bounds__(256) void k_db_clique(DbArgs a) {
    DbGrid* gp = a.grid;
    const DbGrid g = *gp;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n || !a.s_core[p]) return;
    const int c = __float_as_int(a.s_pt[p].w);
    const int32_t orig = a.s_orig[p], rep = a.cell_rep[c];
    if (orig != rep) a.parent[orig] = rep;
    else a.core_cells[atomicAdd(&gp->ncore_cells, 1)] = c;  // the representative lists its cell (once per cell)
}

// Two cells' components join when ONE core point of the one is within eps of ONE core point of the other.  One WAVE per cell
// that holds core points, against the cells around it with a higher index (a pair is looked at once): the pair's components
// are compared once, by the wave - not by every core point of the cell (round 5: one thread per core point, 125 cells each,
// two walks of the union-find per cell: at N = 100 k in a converged cloud 25 M atomic loads of the same few roots, 5.2 ms of
// the 7.3 ms DBSCAN frame); then the tight boxes of the two cells (wholly beyond eps: nothing; wholly within: joined);
// then the cell's core points 64 at a time against the other cell's box, and only a point the box does not decide against
// that cell's core points, 64 at a time, until the first hit.
__device__ __forceinline__ void db_cell_box(const DbArgs& a, int c, double* lo, double* hi) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = (double)db_unkey(a.cell_box[(size_t)d * DB_MAXCELLS + c]);
        hi[d] = (double)db_unkey(a.cell_box[(size_t)(3 + d) * DB_MAXCELLS + c]);
    }
}
__global__ __launch_bounds__(256) void k_db_link(DbArgs a) {
    const DbGrid g = *a.grid;
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * 256) >> 6);
    for (int ci = wave; ci < g.ncore_cells; ci += nwaves) {
        const int c = a.core_cells[ci];
        const int32_t rep = a.cell_rep[c];
        const int b1 = a.cell_start[c], e1 = a.cell_start[c + 1];
        double lo1[3], hi1[3];
        db_cell_box(a, c, lo1, hi1);
        db_for_cells(g, a.hkeys, c, false, [&](int c2) {  // (wave-uniform: every lane walks the same cells)
            if (c2 <= c) return false;                    // the pair belongs to the cell with the lower index
            const int32_t rep2 = a.cell_rep[c2];
            if (rep2 == 0x7fffffff) return false;         // no core point there
            if (db_find(a.parent, rep2) == db_find(a.parent, rep)) return false;
            // the two tight boxes, in the predicate's arithmetic (see db_box_bounds: rounding is monotone through it)
            double lo2[3], hi2[3], mind2 = 0.0, maxd2 = 0.0;
            db_cell_box(a, c2, lo2, hi2);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double ga = lo2[d] - hi1[d], gb = lo1[d] - hi2[d];
                double gm = ga > gb ? ga : gb;
                gm = gm > 0.0 ? gm : 0.0;
                const double fa = hi2[d] - lo1[d], fb = hi1[d] - lo2[d];
                const double fm = fa > fb ? fa : fb;
                if (d == 0) { mind2 = gm * gm; maxd2 = fm * fm; } else { mind2 += gm * gm; maxd2 += fm * fm; }
            }
            if (mind2 > a.r2) return false;
            bool linked = maxd2 <= a.r2;  // every pair is within eps: so are two core points
            const int b2 = a.cell_start[c2], e2 = a.cell_start[c2 + 1];
            for (int p0 = b1; p0 < e1 && !linked; p0 += 64) {
                const int p = p0 + lane, pc = p < e1 ? p : e1 - 1;
                const bool active = p < e1 && a.s_core[pc];
                const float4 me = a.s_pt[pc];
                double mn, mx;
                db_box_bounds(a, c2, me, mn, mx);
                // (the other cell's box holds its non-core points too: "all of it within eps" still reaches its core points)
                if (__any(active && mx <= a.r2)) { linked = true; break; }
                unsigned long long open = __ballot(active && mn <= a.r2);
                while (open && !linked) {
                    const int l = (int)__builtin_ctzll(open);
                    open &= open - 1;
                    float4 mq;
                    mq.x = __shfl(me.x, l); mq.y = __shfl(me.y, l); mq.z = __shfl(me.z, l); mq.w = 0.f;
                    for (int q0 = b2; q0 < e2; q0 += 64) {
                        const int q = q0 + lane, qc = q < e2 ? q : e2 - 1;
                        if (__any(q < e2 && a.s_core[qc] && db_within(mq, a.s_pt[qc], a.r2))) { linked = true; break; }
                    }
                }
            }
            if (linked && lane == 0) db_union(a.parent, rep, rep2);
            return false;
        });
    }
}

// flatten; the roots (a cluster's first core point in index order) are collected (the first DB_MAXROOTS in a list for the LDS
// ranking; all of them as flags by particle index for the general ranking)
__global__ __launch_bounds__(256) void k_db_roots(DbArgs a) {
    DbGrid* gp = a.grid;
    const DbGrid g = *gp;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n) return;
    const int32_t orig = a.s_orig[p];
    int flag = 0;
    if (a.s_core[p]) {
        const int32_t r = db_find(a.parent, orig);
        a.parent[orig] = r;  // the unions are complete: values only ever move towards the root, a plain store is safe
        if (r == orig) {
            const int slot = atomicAdd(&gp->nroots, 1);
            if (slot < DB_MAXROOTS) a.roots[slot] = orig;
            flag = 1;
        }
    }
    a.rank[orig] = flag;
}

// more clusters than the LDS ranking holds: rank[i] := number of roots among the particles before i (one workgroup; a no-op
// otherwise)
__global__ __launch_bounds__(1024) void k_db_rank(DbArgs a) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const DbGrid g = *a.grid;
    if (g.nroots <= DB_MAXROOTS || (a.max_clusters > 0 && a.max_clusters <= DB_MAXROOTS)) return;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < g.n; base += 1024 * 8) {
        int v[8], mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + t * 8 + j;
            v[j] = i < g.n ? a.rank[i] : 0;
            mine += v[j];
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(incl, o);
            if ((t & 63) >= o) incl += x;
        }
        if ((t & 63) == 63) s_w[t >> 6] = incl;
        __syncthreads();
        int run = s_carry + incl - mine;
        for (int w = 0; w < (t >> 6); ++w) run += s_w[w];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + t * 8 + j;
            if (i < g.n) a.rank[i] = run;
            run += v[j];
        }
        __syncthreads();
        if (t == 1023) s_carry = run;
        __syncthreads();
    }
}

// cluster number of a root = its rank among the roots; per cell the number of its core points' cluster
__global__ __launch_bounds__(256) void k_db_number(DbArgs a) {
    __shared__ int32_t s_roots[DB_MAXROOTS + 1];
    DbGrid* gp = a.grid;
    const DbGrid g = *gp;
    int nr = g.nroots;
    // general = ranked through the prefix sum; otherwise the first DB_MAXROOTS roots are ranked here and the clusters beyond
    // that limit (the loop step's arrays) stay unnumbered
    const bool general = nr > DB_MAXROOTS && !(a.max_clusters > 0 && a.max_clusters <= DB_MAXROOTS);
    const bool over = !general && nr > DB_MAXROOTS;
    nr = over ? DB_MAXROOTS : nr;
    const int t = threadIdx.x;
    if (!general && t < nr) {  // rank sort of the few roots
        const int32_t v = a.roots[t];
        int rank = 0;
        for (int j = 0; j < nr; ++j) rank += a.roots[j] < v ? 1 : 0;
        s_roots[rank] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0 && t == 0) {
        *a.ncl_out = nr;
        if (a.err_out && (over || g.err)) *a.err_out |= (over ? 2 : 0) | (g.err & 1 ? 32 : 0) | (g.err & 2 ? 64 : 0);  // cluster limit | extent / non-finite | hash capacity
    }
    for (int64_t p = (int64_t)blockIdx.x * 256 + t; p < g.n; p += (int64_t)gridDim.x * 256) {
        if (!a.s_core[p]) continue;
        const int32_t orig = a.s_orig[p];
        const int32_t r = db_find(a.parent, orig);
        int num = -1;
        if (general) num = a.rank[r];
        else
            for (int j = 0; j < nr; ++j) num = s_roots[j] == r ? j : num;
        a.labels[orig] = num;
        const int c = __float_as_int(a.s_pt[p].w);
        if (a.cell_rep[c] == orig) a.cell_num[c] = num;
    }
}

// border points: the smallest cluster number among the core points within eps; noise otherwise.
// One WAVE per point.  The 125 cells (own cell included) are classified one per lane: a cell with core points whose whole box is
// within eps (the own cell always is) offers its cluster number outright - the wave's minimum over those is the answer unless a
// CUT cell holds a smaller number, and only those cut cells are scanned, 64 candidates at a time, any hit ends the cell.
// (Round 3: a thread per point, 4.5 ms at N = 100k; round 4: a wave per point walking the cells one by one, 1.0 ms; now 0.2.)
__global__ __launch_bounds__(256) void k_db_border(DbArgs a) {
    const DbGrid g = *a.grid;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t p = wave; p < g.n; p += nwaves) {
        if (a.s_core[p]) continue;
        const float4 me = a.s_pt[p];
        const int c = __float_as_int(me.w);
        int cx, cy, cz;
        db_cell_xyz(g, a.hkeys, c, cx, cy, cz);
        int best = 0x7fffffff;
        int c2v[2], numv[2];
        unsigned long long cut[2];
#pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            const int nbr = lane + 64 * turn;
            const int c2 = db_neighbour_cell(g, a.hkeys, c, cx, cy, cz, nbr, true);
            int num = -1, cls = 0;  // cls 1: some core point of the cell is certainly within eps, 2: cut
            if (c2 >= 0) {
                num = a.cell_num[c2];
                if (num >= 0) {
                    if (c2 == c) cls = 1;  // a core point of the own cell is within eps
                    else {
                        double mind2, maxd2;
                        db_box_bounds(a, c2, me, mind2, maxd2);
                        cls = maxd2 <= a.r2 ? 1 : (mind2 <= a.r2 ? 2 : 0);
                    }
                }
            }
            c2v[turn] = c2; numv[turn] = num;
            int mine = cls == 1 ? num : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mine, o); mine = v < mine ? v : mine; }
            best = mine < best ? mine : best;
            cut[turn] = __ballot(cls == 2);
        }
#pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            unsigned long long m = cut[turn];
            while (m) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1;
                const int num = __shfl(numv[turn], l);
                if (num >= best) continue;
                const int c2 = __shfl(c2v[turn], l);
                const int e = a.cell_start[c2 + 1], b0 = a.cell_start[c2];
                for (int q0 = b0; q0 < e; q0 += 64) {
                    const int q = q0 + lane, qc = q < e ? q : e - 1;
                    const bool hit = q < e && a.s_core[qc] && db_within(me, a.s_pt[qc], a.r2);
                    if (__any(hit)) { best = num; break; }
                }
            }
        }
        if (lane == 0) a.labels[a.s_orig[p]] = best == 0x7fffffff ? -1 : best;
    }
}

int launch_dbscan(midas_ctx* ctx, int64_t cap, const int32_t* n_dev, const float* poses, double eps, int64_t min_samples,
                  int32_t* labels_out, int32_t* ncl_out, int32_t* err_out, int32_t max_clusters) {
    if (cap <= 0) return MIDAS_OK;
    hipStream_t st = ctx->stream;
    DbArgs a;
    a.N = cap; a.n_dev = n_dev; a.poses = poses; a.eps = eps; a.r2 = eps * eps; a.min_samples = min_samples;
    a.labels = labels_out; a.ncl_out = ncl_out; a.err_out = err_out; a.max_clusters = max_clusters;
    const int nbb = (int)(ceil_div(cap, 256) < 256 ? ceil_div(cap, 256) : 256);
    int rc;
    void* p;
#define DB_SCRATCH(field, type, count)                                        \
    if ((rc = midas_scratch(ctx, (size_t)(count) * sizeof(type), &p))) return rc; \
    a.field = (type*)p
    DB_SCRATCH(grid, DbGrid, 1);
    DB_SCRATCH(part, float, nbb * 6);
    DB_SCRATCH(cell_count, int32_t, DB_MAXCELLS);
    DB_SCRATCH(cell_start, int32_t, DB_MAXCELLS + 1);
    DB_SCRATCH(cell_rep, int32_t, DB_MAXCELLS);
    DB_SCRATCH(cell_num, int32_t, DB_MAXCELLS);
    DB_SCRATCH(cell_box, uint32_t, 6 * (size_t)DB_MAXCELLS);
    DB_SCRATCH(cid, int32_t, cap);
    DB_SCRATCH(s_orig, int32_t, cap);
    DB_SCRATCH(s_pt, float4, cap);
    DB_SCRATCH(s_core, uint8_t, cap);
    DB_SCRATCH(parent, int32_t, cap);
    DB_SCRATCH(roots, int32_t, DB_MAXROOTS + 1);
    DB_SCRATCH(work, int32_t, cap);
    DB_SCRATCH(core_cells, int32_t, cap);
    DB_SCRATCH(hkeys, unsigned long long, DB_MAXCELLS);
    DB_SCRATCH(rank, int32_t, cap + 1);
#undef DB_SCRATCH
    const unsigned gp = (unsigned)ceil_div(cap, 256);
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(a.cell_count, 0, (size_t)DB_MAXCELLS * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_db_bounds, dim3(nbb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_setup, dim3(1), dim3(64), 0, st, a, nbb);
    hipLaunchKernelGGL(k_db_hclear, dim3(1024), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_count, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_scan, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_db_scatter, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_core, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_core_count, dim3(gp < 2048 ? gp : 2048), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_clique, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_link, dim3(gp < 2048 ? gp : 2048), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_roots, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_rank, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_db_number, dim3(gp < 1024 ? gp : 1024), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_db_border, dim3(gp), dim3(256), 0, st, a);
    DB_LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

MIDAS_WARM_TU(dbscan, k_db_setup)

}  // namespace midas
