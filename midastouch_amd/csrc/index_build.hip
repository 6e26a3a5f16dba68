// index_build.hip - the per-entry lists of the hint fast paths built ON THE DEVICE: for every codebook entry its NBR_M
// nearest entries in the 6-d feature space (the neighbour graph nn6_hint_scan walks) and the MESH_M mesh vertices nearest
// to its translation (the list mesh_list_check walks).  The reference has no counterpart (pynanoflann / sklearn build
// their trees on the host, tactile_tree/tactile_tree.py:34-41, modules/particle_filter.py:108-110); round 1 built these
// lists with host threads (1.4 s at K = 50k, 17 s at K = 500k) - here one 256-thread workgroup per entry streams the
// points (brute force: exact by construction, no tree logic to get wrong), keeps the best candidates in LDS and cuts
// them back with a bitonic sort whenever the buffer fills.  Distances are float64 sums in the host builder's order and
// the order is (distance, index), so the lists are the host builder's bit for bit (tests/test_gpu_index_build.py
// compares every byte); the host builder stays behind MIDAS_HOST_INDEX=1.
#include <cmath>
#include <cstdlib>

#include "midas_internal.hpp"
#include "midas_math.hpp"

namespace midas {

namespace {

constexpr int SEL_CAP = 2048;  // candidate buffer (pairs) per workgroup
constexpr int SEL_T = 256;     // threads per workgroup = points offered per round

// Selection of the m smallest (d, idx) pairs of a stream.  All 256 threads call every method together.
struct Selector {
    double* d;   // [SEL_CAP]
    int* i;      // [SEL_CAP]
    int* cnt;    // [1]
    double* T;   // [1] current bound: only pairs strictly below (T, Ti) are admitted
    int* Ti;     // [1]

    MD void init() {
        if (threadIdx.x == 0) { *cnt = 0; *T = INFINITY; *Ti = 0x7fffffff; }
        __syncthreads();
    }
    MD void offer(double dd, int id, bool valid) {
        const double t = *T;
        const int ti = *Ti;
        if (valid && (dd < t || (dd == t && id < ti))) {  // NaN never enters
            const int pos = atomicAdd(cnt, 1);
            d[pos] = dd;
            i[pos] = id;
        }
    }
    // sort what is there by (d, idx) and keep the m best; afterwards T = the m-th pair when there are m
    MD void compact(int m) {
        __syncthreads();
        const int n0 = *cnt;
        int n = 256;  // power of two >= n0 (at least one element per thread)
        while (n < n0) n <<= 1;
        for (int k = n0 + threadIdx.x; k < n; k += SEL_T) { d[k] = INFINITY; i[k] = 0x7fffffff; }
        __syncthreads();
        for (int k = 2; k <= n; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int a = threadIdx.x; a < n; a += SEL_T) {
                    const int b = a ^ j;
                    if (b > a) {
                        const double da = d[a], db = d[b];
                        const int ia = i[a], ib = i[b];
                        const bool a_after_b = da > db || (da == db && ia > ib);
                        const bool up = (a & k) == 0;
                        if (a_after_b == up) { d[a] = db; d[b] = da; i[a] = ib; i[b] = ia; }
                    }
                }
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) {
            const int keep = n0 < m ? n0 : m;
            *cnt = keep;
            if (keep == m) { *T = d[m - 1]; *Ti = i[m - 1]; }
        }
        __syncthreads();
    }
    // between rounds: make room for the next 256 offers
    MD void make_room(int m) {
        __syncthreads();
        const int c = *cnt;
        __syncthreads();  // everybody has read the count before the next round's offers move it
        if (c > SEL_CAP - SEL_T) compact(m);
    }
};

MD float round_down_f32_dev(double v) {  // (float)v rounded toward zero when the cast rounded up (v >= 0)
    float f = (float)v;
    if ((double)f > v) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

// ---- neighbour graph ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SEL_T) void k_build_nbrs(const Point6* __restrict__ pts, const int32_t* __restrict__ inv_perm,
                                                      int levels, int64_t K, Nbr6* __restrict__ nbrs, float* __restrict__ rho_out,
                                                      int32_t* __restrict__ twin) {
    __shared__ double s_d[SEL_CAP];
    __shared__ int s_i[SEL_CAP];
    __shared__ int s_cnt, s_Ti;
    __shared__ double s_T;
    __shared__ double s_td[4];
    __shared__ int s_ti[4];
    Selector sel{s_d, s_i, &s_cnt, &s_T, &s_Ti};
    const int64_t k = blockIdx.x;
    const int t = threadIdx.x;
    const Point6 self = pts[inv_perm[k]];
    double q[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) q[a] = (double)self.c[a];
    // the entry across the rotation-angle-pi cut (see build_neighbour_graph): nearest entry of the flipped feature
    const double wn = __builtin_sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5]);
    const bool want_twin = wn > 0.01 * (M_PI - 0.6);
    double qf[6] = {q[0], q[1], q[2], 0.0, 0.0, 0.0};
    if (want_twin) {
        const double sc = (wn - 0.01 * 2.0 * M_PI) / wn;
#pragma unroll
        for (int a = 3; a < 6; ++a) qf[a] = q[a] * sc;
    }
    double tw_d = INFINITY;
    int tw_i = 0x7fffffff;
    sel.init();
    const int64_t nslots = ((int64_t)LEAF_CAP) << (3 * levels);
    for (int64_t s0 = 0; s0 < nslots; s0 += SEL_T) {
        const int64_t s = s0 + t;
        const Point6 p = pts[s < nslots ? s : nslots - 1];
        const bool there = s < nslots && p.idx != 0x7fffffff;
        double dd = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) { const double x = q[a] - (double)p.c[a]; dd += x * x; }
        sel.offer(dd, p.idx, there && (int64_t)p.idx != k);
        if (want_twin && there) {
            double df = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { const double x = qf[a] - (double)p.c[a]; df += x * x; }
            if (df < tw_d || (df == tw_d && p.idx < tw_i)) { tw_d = df; tw_i = p.idx; }
        }
        sel.make_room(NBR_M + 1);
    }
    sel.compact(NBR_M + 1);
    const int have = s_cnt;
    Nbr6* out = nbrs + (size_t)k * NBR_REC;
    if (t == 0) {  // record 0 = the entry itself (rho 0): the scan needs no other lookup
        Nbr6 r;
#pragma unroll
        for (int a = 0; a < 6; ++a) r.c[a] = self.c[a];
        r.idx = (int32_t)k;
        r.rho = 0.0f;
        out[0] = r;
        rho_out[k] = have > NBR_M ? round_down_f32_dev(__builtin_sqrt(s_d[NBR_M])) : INFINITY;
    }
    for (int j = t; j < NBR_M; j += SEL_T) {
        Nbr6 r;
        if (j < have) {
            const int id = s_i[j];
            const Point6 p = pts[inv_perm[id]];
#pragma unroll
            for (int a = 0; a < 6; ++a) r.c[a] = p.c[a];
            r.idx = id;
            r.rho = round_down_f32_dev(__builtin_sqrt(s_d[j]));
        } else {
#pragma unroll
            for (int a = 0; a < 6; ++a) r.c[a] = INFINITY;
            r.idx = 0x7fffffff;
            r.rho = INFINITY;
        }
        out[1 + j] = r;
    }
    // twin: minimum of (distance to the flipped feature, index) over the workgroup
    if (want_twin) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(tw_d, o);
            const int oi = __shfl_xor(tw_i, o);
            if (od < tw_d || (od == tw_d && oi < tw_i)) { tw_d = od; tw_i = oi; }
        }
        if ((t & 63) == 0) { s_td[t >> 6] = tw_d; s_ti[t >> 6] = tw_i; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_td[w] < tw_d || (s_td[w] == tw_d && s_ti[w] < tw_i)) { tw_d = s_td[w]; tw_i = s_ti[w]; }
            twin[k] = (tw_i != 0x7fffffff && (int64_t)tw_i != k) ? tw_i : -1;
        }
    } else if (t == 0) {
        twin[k] = -1;
    }
}

// ---- mesh-vertex lists ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SEL_T) void k_build_vlists(const Point3* __restrict__ mpts, const int32_t* __restrict__ minv,
                                                        int mlevels, const float* __restrict__ cb_poses, MeshRec* __restrict__ recs) {
    __shared__ double s_d[SEL_CAP];
    __shared__ int s_i[SEL_CAP];
    __shared__ int s_cnt, s_Ti;
    __shared__ double s_T;
    Selector sel{s_d, s_i, &s_cnt, &s_T, &s_Ti};
    const int64_t k = blockIdx.x;
    const int t = threadIdx.x;
    const double q[3] = {(double)cb_poses[k * 16 + 3], (double)cb_poses[k * 16 + 7], (double)cb_poses[k * 16 + 11]};
    sel.init();
    const int64_t nslots = ((int64_t)LEAF_CAP) << (3 * mlevels);
    for (int64_t s0 = 0; s0 < nslots; s0 += SEL_T) {
        const int64_t s = s0 + t;
        const Point3 p = mpts[s < nslots ? s : nslots - 1];
        const bool there = s < nslots && p.idx != 0x7fffffff;
        double dd = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { const double x = q[a] - p.c[a]; dd += x * x; }
        sel.offer(dd, (int)p.idx, there);
        sel.make_room(MESH_M + 1);
    }
    sel.compact(MESH_M + 1);
    const int have = s_cnt;
    MeshRec* out = recs + (size_t)k * MESH_REC;
    if (t == 0) {
        MeshRec hd;
        hd.c[0] = q[0]; hd.c[1] = q[1]; hd.c[2] = q[2];
        hd.rho = have > MESH_M ? round_down_f32_dev(__builtin_sqrt(s_d[MESH_M])) : INFINITY;
        hd.pad = 0;
        out[0] = hd;
    }
    for (int j = t; j < MESH_M; j += SEL_T) {
        MeshRec r;
        if (j < have) {
            const Point3 p = mpts[minv[s_i[j]]];
            r.c[0] = p.c[0]; r.c[1] = p.c[1]; r.c[2] = p.c[2];
            r.rho = round_down_f32_dev(__builtin_sqrt(s_d[j]));
        } else {
            r.c[0] = r.c[1] = r.c[2] = INFINITY;
            r.rho = INFINITY;
        }
        r.pad = 0;
        out[1 + j] = r;
    }
}

__global__ __launch_bounds__(256) void k_vertex_screen(int64_t n, const MeshRec* __restrict__ in, MeshScr* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const MeshRec r = in[i];
    MeshScr o;
    o.c[0] = (float)r.c[0]; o.c[1] = (float)r.c[1]; o.c[2] = (float)r.c[2];  // round to nearest; inf padding stays inf
    o.rho = r.rho;
    out[i] = o;
}

}  // namespace

int build_vertex_screen(midas_ctx* ctx, midas_tree* t6) {
    if (t6->vscr) { (void)hipFree(t6->vscr); t6->vscr = nullptr; }
    static const bool off = getenv("MIDAS_NO_VSCR") && atoi(getenv("MIDAS_NO_VSCR")) != 0;
    if (off || !t6->vlist) return MIDAS_OK;
    const int64_t n = t6->K * MESH_REC;
    if (hipMalloc(&t6->vscr, (size_t)n * sizeof(MeshScr)) != hipSuccess) {  // optional: the float64 lists decide alone
        (void)hipGetLastError();
        t6->vscr = nullptr;
        return MIDAS_OK;
    }
    hipLaunchKernelGGL(k_vertex_screen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, (const MeshRec*)t6->vlist,
                       (MeshScr*)t6->vscr);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MIDAS_OK;
}

bool index_build_on_host() {
    const char* e = getenv("MIDAS_HOST_INDEX");
    return e && atoi(e) != 0;
}

// nbrs / rho_out / twin of a 6-d tree whose boxes, pts and inv_perm are on the device already
int build_neighbour_graph_device(midas_ctx* ctx, midas_tree* t) {
    const int64_t K = t->K;
    MIDAS_HIP_CHECK(ctx, hipMalloc(&t->nbrs, (size_t)K * NBR_REC * sizeof(Nbr6)));
    MIDAS_HIP_CHECK(ctx, hipMalloc((void**)&t->rho_out, (size_t)K * sizeof(float)));
    MIDAS_HIP_CHECK(ctx, hipMalloc((void**)&t->twin, (size_t)K * sizeof(int32_t)));
    hipLaunchKernelGGL(k_build_nbrs, dim3((unsigned)K), dim3(SEL_T), 0, ctx->stream, (const Point6*)t->pts, t->inv_perm, t->levels, K,
                       (Nbr6*)t->nbrs, t->rho_out, t->twin);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MIDAS_OK;
}

int build_vertex_lists_device(midas_ctx* ctx, midas_tree* t6, const midas_tree* t3, const float* cb_poses_dev) {
    const int64_t K = t6->K;
    if (t6->vlist) { (void)hipFree(t6->vlist); t6->vlist = nullptr; }
    MIDAS_HIP_CHECK(ctx, hipMalloc(&t6->vlist, (size_t)K * MESH_REC * sizeof(MeshRec)));
    hipLaunchKernelGGL(k_build_vlists, dim3((unsigned)K), dim3(SEL_T), 0, ctx->stream, (const Point3*)t3->pts, t3->inv_perm, t3->levels,
                       cb_poses_dev, (MeshRec*)t6->vlist);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    t6->vlist_mesh = t3;
    return build_vertex_screen(ctx, t6);
}

MIDAS_WARM_TU(index_build, k_vertex_screen)

}  // namespace midas
