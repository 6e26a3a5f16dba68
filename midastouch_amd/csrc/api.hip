// api.hip - context, scratch, profiling and the extern "C" surface of libmidas_hip.so.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "midas_internal.hpp"
#include "peer_row.hpp"

using namespace midas;

namespace midas {
int tree_build_impl(midas_ctx* ctx, int32_t dim, int64_t K, const void* points_dev, midas_tree* out);
int attach_mesh_impl(midas_ctx* ctx, midas_tree* t6, const midas_tree* t3, const float* cb_poses_dev);
void tree_free_host(midas_tree* t);
}

// ---- scratch: bump allocator over library-owned device chunks, reset at every API entry ----------
// The chunks live as long as the context: a reset only rewinds to the first one, a request that does not fit the current
// chunk moves on to the next, and only a request that fits none allocates (at least twice the largest chunk so far).  The first
// form freed and re-allocated one consolidated chunk whenever a call had needed more than one - a stream synchronisation,
// hipFree and hipMalloc of tens of MB in the middle of a run: the 5 - 30 ms frames the loop showed now and then whenever a
// phase combination asked for more than any frame before it (MIDAS_SCRATCH_LOG=1 reports every new chunk).
struct ScratchChunk { void* p; size_t cap; };
struct ScratchState {
    std::vector<ScratchChunk> chunks;
    size_t cur = 0;    // chunk in use
    size_t used = 0;   // bytes taken from chunks[cur]
    size_t total = 0;  // bytes requested since the last reset
};
static ScratchState* scratch_of(midas_ctx* ctx) { return reinterpret_cast<ScratchState*>(ctx->scratch); }

static int scratch_reset(midas_ctx* ctx) {
    ScratchState* s = scratch_of(ctx);
    s->cur = 0;
    s->used = 0;
    s->total = 0;
    return MIDAS_OK;
}

int midas_scratch(midas_ctx* ctx, size_t bytes, void** out) {
    ScratchState* s = scratch_of(ctx);
    bytes = (bytes + 255) & ~(size_t)255;
    s->total += bytes;
    while (s->cur < s->chunks.size() && s->used + bytes > s->chunks[s->cur].cap) { ++s->cur; s->used = 0; }
    if (s->cur == s->chunks.size()) {
        size_t cap = bytes > (size_t)(1 << 20) ? bytes : (size_t)(1 << 20);
        size_t largest = 0;
        for (const auto& c : s->chunks) largest = c.cap > largest ? c.cap : largest;
        cap = cap > 2 * largest ? cap : 2 * largest;
        void* p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {
            cap = bytes > (size_t)(1 << 20) ? bytes : (size_t)(1 << 20);  // not the generous size then: what is needed
            if (hipMalloc(&p, cap) != hipSuccess) return midas_set_error(ctx, MIDAS_ERR_NOMEM, "hipMalloc(scratch)", "out of device memory");
        }
        static const bool log = getenv("MIDAS_SCRATCH_LOG") != nullptr;
        if (log) fprintf(stderr, "[midas] scratch: chunk %zu of %zu bytes\n", s->chunks.size(), cap);
        s->chunks.push_back({p, cap});
        s->used = 0;
    }
    *out = (char*)s->chunks[s->cur].p + s->used;
    s->used += bytes;
    return MIDAS_OK;
}

int midas_set_error(midas_ctx* ctx, int code, const char* what, const char* detail) {
    if (ctx) {
        ctx->last_error = std::string(midas_strerror(code)) + ": " + (what ? what : "") + " (" + (detail ? detail : "") + ")";
    }
    return code;
}

namespace midas {
void prof_mark(midas_ctx* ctx, int slot) {
    if (!ctx->prof || !ctx->ev_ready || slot > MIDAS_PROF_SLOTS) return;
    // prof_only >= 0: bracket a single kernel (events slot and slot+1 only) so that the other kernels run
    // back to back as in an untimed frame
    if (ctx->prof_only >= 0 && slot != ctx->prof_only && slot != ctx->prof_only + 1) return;
    (void)hipEventRecord(ctx->ev[slot], ctx->stream);
}
}  // namespace midas

static const char* kSlotNames[MIDAS_PROF_SLOTS] = {
    "score_codebook", "particle_update", "tail_a", "tail_b", "", "", "", "event_pair_overhead"};

// entry guard: bind the device, reset the scratch bump pointer
#define MIDAS_ENTER(ctx)                                                     \
    do {                                                                     \
        if (!(ctx)) return MIDAS_ERR_INVALID;                                \
        MIDAS_HIP_CHECK((ctx), hipSetDevice((ctx)->device));                 \
        int _rc = scratch_reset(ctx);                                        \
        if (_rc) return _rc;                                                 \
    } while (0)

extern "C" {

#define MIDAS_EXPORT __attribute__((visibility("default")))

MIDAS_EXPORT const char* midas_version(void) { return "midas-hip 0.1 (gfx950)"; }

MIDAS_EXPORT const char* midas_strerror(int code) {
    switch (code) {
        case MIDAS_OK: return "ok";
        case MIDAS_ERR_INVALID: return "invalid argument";
        case MIDAS_ERR_HIP: return "HIP runtime error";
        case MIDAS_ERR_NOMEM: return "out of device memory";
        case MIDAS_ERR_NODEVICE: return "no usable HIP device";
        default: return "unknown error";
    }
}

MIDAS_EXPORT const char* midas_last_error(const midas_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

MIDAS_EXPORT int midas_ctx_create(int device, void* hip_stream, midas_ctx** out) {
    if (!out) return MIDAS_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MIDAS_ERR_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return MIDAS_ERR_NODEVICE;
    midas_ctx* ctx = new (std::nothrow) midas_ctx();
    if (!ctx) return MIDAS_ERR_NOMEM;
    ctx->device = device;
    ctx->scratch = new (std::nothrow) ScratchState();
    // NULL selects the device's default (null) stream - torch's default stream on ROCm - so that
    // kernels stay ordered with the caller's other work.  The only stream the library creates is the side stream of
    // the batch step, forked from and joined back into this stream by events inside midas_filter_step_batch.
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    if (const char* ov = getenv("MIDAS_OVERLAP")) ctx->overlap = atoi(ov) != 0;
    // every translation unit's code object is loaded now, not by the first frame that happens to need it (midas_internal.hpp)
    const char* lazy = getenv("MIDAS_LAZY_MODULES");
    if (!(lazy && lazy[0] == '1')) {
        int (*const warm[])() = {warm_score, warm_particles, warm_resample, warm_cluster, warm_topn, warm_selfsim, warm_loop,
                                 warm_dbscan, warm_dbscan_nd, warm_index_build, warm_mt19937, warm_topk_aten};
        for (auto w : warm)
            if (w() != 0) { (void)hipGetLastError(); }  // not fatal: the unit then loads at its first launch, as before
    }
    // hand-over records of the grouped tail (4 KB per 4096 particles; without them the tail takes its one-workgroup-per-block form)
    {
        void* p = nullptr;
        const int blocks = midas::TAIL_GROUP_MAX_BLOCKS;
        // (hipMemset is not ordered against a non-blocking caller stream: the device is drained before anybody can launch on the records)
        if (hipMalloc(&p, (size_t)blocks * midas::TAIL_GROUP_BLOCK_BYTES) == hipSuccess && hipMemset(p, 0, (size_t)blocks * midas::TAIL_GROUP_BLOCK_BYTES) == hipSuccess &&
            hipDeviceSynchronize() == hipSuccess) {
            ctx->tail_rec = (unsigned long long*)p;
            ctx->tail_rec_blocks = blocks;
        } else {
            if (p) (void)hipFree(p);
            (void)hipGetLastError();
        }
    }
    *out = ctx;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_scratch_reserve(midas_ctx* ctx, int64_t bytes) {
    if (!ctx || bytes < 0) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ScratchState* s = scratch_of(ctx);
    size_t have = 0;
    for (const auto& c : s->chunks) have = c.cap > have ? c.cap : have;
    const size_t want = ((size_t)bytes + 255) & ~(size_t)255;
    if (s->chunks.size() == 1 && have >= want) return MIDAS_OK;
    if (s->chunks.size() >= 1 && have >= want && s->chunks[0].cap == have) return MIDAS_OK;  // the first chunk already holds it
    // one chunk that holds everything, in front of the others (a call takes the first chunk its request fits): nothing in
    // flight may still use the old ones when they are freed
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->side));
    void* p = nullptr;
    const size_t cap = want > have ? want : have;
    if (hipMalloc(&p, cap) != hipSuccess) return midas_set_error(ctx, MIDAS_ERR_NOMEM, "hipMalloc(scratch reserve)", "out of device memory");
    for (auto& c : s->chunks) (void)hipFree(c.p);
    s->chunks.clear();
    s->chunks.push_back({p, cap});
    s->cur = 0; s->used = 0; s->total = 0;
    static const bool log = getenv("MIDAS_SCRATCH_LOG") != nullptr;
    if (log) fprintf(stderr, "[midas] scratch: reserved one chunk of %zu bytes\n", cap);
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_ctx_destroy(midas_ctx* ctx) {
    if (!ctx) return MIDAS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ScratchState* s = scratch_of(ctx);
    for (auto& c : s->chunks) (void)hipFree(c.p);
    delete s;
    if (ctx->tail_rec) (void)hipFree(ctx->tail_rec);
    if (ctx->ev_ready)
        for (auto& e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipEventDestroy(ctx->ev_fork);
        (void)hipEventDestroy(ctx->ev_join);
        (void)hipStreamDestroy(ctx->side);
    }
    delete ctx;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_ctx_set_stream(midas_ctx* ctx, void* hip_stream) {
    if (!ctx) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) {
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    ctx->stream = (hipStream_t)hip_stream;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_sync(midas_ctx* ctx) {
    if (!ctx) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MIDAS_OK;
}

// ---- codebook ------------------------------------------------------------------------------------
MIDAS_EXPORT int midas_codebook_create(midas_ctx* ctx, int64_t K, int32_t D, const void* emb_dev, int32_t dtype,
                                       midas_codebook** out) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, out && emb_dev && K > 0 && D > 0 && (dtype == MIDAS_F32 || dtype == MIDAS_F64));
    midas_codebook* cb = new (std::nothrow) midas_codebook();
    if (!cb) return midas_set_error(ctx, MIDAS_ERR_NOMEM, "new midas_codebook", "");
    cb->ctx = ctx; cb->K = K; cb->D = D; cb->dtype = dtype; cb->emb = emb_dev; cb->norms = nullptr;
    if (hipMalloc((void**)&cb->norms, (size_t)K * sizeof(double)) != hipSuccess) {
        delete cb;
        return midas_set_error(ctx, MIDAS_ERR_NOMEM, "hipMalloc(norms)", "");
    }
    int rc = launch_row_norms(ctx, K, D, emb_dev, dtype, cb->norms);
    if (rc) { (void)hipFree(cb->norms); delete cb; return rc; }
    *out = cb;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_codebook_destroy(midas_codebook* cb) {
    if (!cb) return MIDAS_OK;
    (void)hipStreamSynchronize(cb->ctx->stream);
    (void)hipFree(cb->norms);
    delete cb;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_score(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes_dev,
                             double* scores_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && B >= 1 && codes_dev && scores_dev);
    return launch_score(ctx, cb, B, codes_dev, scores_dev);
}

MIDAS_EXPORT int midas_score_batch(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes_dev,
                                   double* scores_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && B >= 1 && codes_dev && scores_dev);
    return launch_score_batch(ctx, cb, B, codes_dev, scores_dev);
}

// ---- features / trees ----------------------------------------------------------------------------
MIDAS_EXPORT int midas_se3_feature(midas_ctx* ctx, int64_t N, const float* poses_dev, float w, float* feat6_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && (N == 0 || (poses_dev && feat6_dev)));
    return launch_se3_feature(ctx, N, poses_dev, w, feat6_dev);
}

MIDAS_EXPORT int midas_tree_build(midas_ctx* ctx, int32_t dim, int64_t K, const void* points_dev, midas_tree** out) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, out && points_dev && K > 0 && K < ((int64_t)1 << 30) && (dim == 6 || dim == 3));
    midas_tree* t = new (std::nothrow) midas_tree();
    if (!t) return midas_set_error(ctx, MIDAS_ERR_NOMEM, "new midas_tree", "");
    std::memset(t, 0, sizeof(*t));
    t->ctx = ctx;
    t->dim = dim;
    int rc = tree_build_impl(ctx, dim, K, points_dev, t);
    if (rc) { midas_tree_destroy(t); return rc; }
    *out = t;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_tree_attach_mesh(midas_ctx* ctx, midas_tree* tree6, const midas_tree* tree3,
                                        const float* cb_poses_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree6 && tree3 && cb_poses_dev && tree6->dim == 6 && tree3->dim == 3);
    return attach_mesh_impl(ctx, tree6, tree3, cb_poses_dev);
}

MIDAS_EXPORT int midas_tree_export(midas_ctx* ctx, const midas_tree* tree, int32_t what, void* dst_host, int64_t bytes) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree && dst_host && bytes >= 0 && what >= 0 && what <= 3);
    const void* src = nullptr;
    int64_t have = 0;
    switch (what) {
        case 0: src = tree->nbrs; have = tree->nbrs ? tree->K * NBR_REC * (int64_t)sizeof(Nbr6) : 0; break;
        case 1: src = tree->rho_out; have = tree->rho_out ? tree->K * (int64_t)sizeof(float) : 0; break;
        case 2: src = tree->twin; have = tree->twin ? tree->K * (int64_t)sizeof(int32_t) : 0; break;
        default: src = tree->vlist; have = tree->vlist ? tree->K * MESH_REC * (int64_t)sizeof(MeshRec) : 0; break;
    }
    MIDAS_REQUIRE(ctx, src && bytes == have);
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MIDAS_HIP_CHECK(ctx, hipMemcpy(dst_host, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_tree_destroy(midas_tree* t) {
    if (!t) return MIDAS_OK;
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->boxes) (void)hipFree(t->boxes);
    if (t->pts) (void)hipFree(t->pts);
    if (t->inv_perm) (void)hipFree(t->inv_perm);
    if (t->nbrs) (void)hipFree(t->nbrs);
    if (t->rho_out) (void)hipFree(t->rho_out);
    if (t->twin) (void)hipFree(t->twin);
    if (t->vlist) (void)hipFree(t->vlist);
    if (t->vscr) (void)hipFree(t->vscr);
    if (t->field.d) (void)hipFree(const_cast<float*>(t->field.d));
    tree_free_host(t);
    delete t;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_nn6(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev,
                           const int32_t* hint_dev, int32_t* idx_dev, float* d2_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree && tree->dim == 6 && N >= 0 && (N == 0 || (feat6_dev && idx_dev)));
    return launch_nn6(ctx, tree, N, feat6_dev, hint_dev, idx_dev, d2_dev);
}

MIDAS_EXPORT int midas_knn6(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev, int32_t k,
                            int32_t* idx_dev, float* d2_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree && tree->dim == 6 && N >= 0 && k >= 1 && k <= 64 && (int64_t)k <= tree->K &&
                           (N == 0 || (feat6_dev && idx_dev)));
    return launch_knn6(ctx, tree, N, feat6_dev, k, idx_dev, d2_dev);
}

MIDAS_EXPORT int midas_nn6_stats(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev,
                                 const int32_t* hint_dev, int32_t* leaves_dev, int32_t* nodes_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree && tree->dim == 6 && N >= 0 && (N == 0 || (feat6_dev && leaves_dev && nodes_dev)));
    return launch_nn6_stats(ctx, tree, N, feat6_dev, hint_dev, leaves_dev, nodes_dev);
}

MIDAS_EXPORT int midas_nn3(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* poses_dev, double* dist_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, tree && tree->dim == 3 && N >= 0 && (N == 0 || (poses_dev && dist_dev)));
    return launch_nn3(ctx, tree, N, poses_dev, dist_dev);
}

// ---- motion model --------------------------------------------------------------------------------
MIDAS_EXPORT int midas_propagate(midas_ctx* ctx, int64_t N, const float* poses_in_dev, float* poses_out_dev,
                                 const float* odom16_dev, const float* tn_dev, const float* rot_dev, float std_t,
                                 float std_r, uint64_t seed, uint64_t step) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && odom16_dev && (N == 0 || (poses_in_dev && poses_out_dev)));
    MIDAS_REQUIRE(ctx, (tn_dev == nullptr) == (rot_dev == nullptr));
    return launch_propagate(ctx, N, poses_in_dev, poses_out_dev, odom16_dev, tn_dev, rot_dev, std_t, std_r, seed, step);
}

MIDAS_EXPORT int midas_check_poses(midas_ctx* ctx, int64_t N, const float* poses_dev, uint8_t* flag_dev,
                                   int32_t* count_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && count_dev && (N == 0 || (poses_dev && flag_dev)));
    return launch_check_poses(ctx, N, poses_dev, flag_dev, count_dev);
}

// ---- weights -------------------------------------------------------------------------------------
MIDAS_EXPORT int midas_gather_f64(midas_ctx* ctx, int64_t N, const double* table_dev, const int32_t* idx_dev,
                                  double* out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && (N == 0 || (table_dev && idx_dev && out_dev)));
    return launch_gather_f64(ctx, N, table_dev, idx_dev, out_dev);
}

MIDAS_EXPORT int midas_softmax(midas_ctx* ctx, int64_t N, const double* x_dev, int32_t softmax, double* w_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && (N == 0 || (x_dev && w_dev)));
    return launch_softmax(ctx, N, x_dev, softmax, w_dev);
}

MIDAS_EXPORT int midas_prune(midas_ctx* ctx, int64_t N, double* w_dev, const double* dist_dev, double thr,
                             int32_t* nvalid_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && nvalid_dev && (N == 0 || (w_dev && dist_dev)));
    return launch_prune(ctx, N, w_dev, dist_dev, thr, nvalid_dev);
}

// ---- resample ------------------------------------------------------------------------------------
MIDAS_EXPORT int midas_cdf(midas_ctx* ctx, int64_t N, const double* w_dev, double* cdf_dev, int32_t* status_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N >= 0 && status_dev && (N == 0 || (w_dev && cdf_dev)));
    return launch_cdf(ctx, N, w_dev, cdf_dev, status_dev);
}

MIDAS_EXPORT int midas_mt19937_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, state_dev != nullptr);
    return launch_mt_seed(ctx, seed, state_dev);
}

MIDAS_EXPORT int midas_mt19937_rand64(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t N, double* out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, state_dev != nullptr && skip_words >= 0 && N >= 0 && (N == 0 || out_dev != nullptr));
    return launch_mt_rand64(ctx, state_dev, skip_words, N, out_dev, nullptr);
}

MIDAS_EXPORT int midas_mt19937_normal32(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t numel, float mean, float std,
                                        const float* radius_dev, const float* cos_dev, const float* sin_dev, float* out_dev, uint32_t* hist_dev,
                                        const uint32_t* polys_dev, int32_t pieces) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, state_dev != nullptr && skip_words >= 0 && numel >= 16 && radius_dev && cos_dev && sin_dev && out_dev);
    const bool chunked = polys_dev && pieces > 0;
    MIDAS_REQUIRE(ctx, !chunked || (hist_dev != nullptr && pieces <= 1024 && numel >= MIDAS_MT19937_HIST_WORDS));
    return launch_mt_normal32(ctx, state_dev, skip_words, numel, mean, std, radius_dev, cos_dev, sin_dev, out_dev, hist_dev,
                              chunked ? polys_dev : nullptr, chunked ? pieces : 0);
}

MIDAS_EXPORT int midas_mt19937_rand64_chunked(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t N, double* out_dev,
                                              uint32_t* hist_dev, const uint32_t* polys_dev, int32_t pieces) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, state_dev != nullptr && skip_words >= 0 && N >= 0 && (N == 0 || out_dev != nullptr));
    if (!polys_dev || pieces <= 0)  // the sequential walk; leaves the history for a chunked call to follow
        return launch_mt_rand64(ctx, state_dev, skip_words, N, out_dev, hist_dev);
    MIDAS_REQUIRE(ctx, hist_dev != nullptr && N > 0 && pieces <= 1024 && 2 * N >= MIDAS_MT19937_HIST_WORDS);
    return launch_mt_rand64_chunked(ctx, state_dev, N, out_dev, hist_dev, polys_dev, pieces);
}

MIDAS_EXPORT int midas_mt19937_draws(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int32_t nseg, const midas_mt_segment* segs,
                                     const float* radius_dev, const float* cos_dev, const float* sin_dev, uint32_t* hist_dev,
                                     const uint32_t* polys_dev, int32_t pieces) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, state_dev != nullptr && skip_words >= 0 && nseg >= 1 && nseg <= 8 && segs != nullptr);
    int64_t total = 0;
    for (int i = 0; i < nseg; ++i) {
        const midas_mt_segment& g = segs[i];
        MIDAS_REQUIRE(ctx, (g.kind == MIDAS_MT_SEGMENT_RAND64 && g.count >= 0 && (g.count == 0 || g.out_dev)) ||
                               (g.kind == MIDAS_MT_SEGMENT_NORMAL32 && g.count >= 16 && g.out_dev && radius_dev && cos_dev && sin_dev));
        total += g.kind == MIDAS_MT_SEGMENT_RAND64 ? 2 * g.count : g.count + ((g.count & 15) ? 16 : 0);
    }
    const bool chunked = polys_dev && pieces > 0;
    MIDAS_REQUIRE(ctx, !chunked || (hist_dev != nullptr && pieces <= 1024 && total >= MIDAS_MT19937_HIST_WORDS));
    return launch_mt_draws(ctx, state_dev, skip_words, nseg, segs, radius_dev, cos_dev, sin_dev, hist_dev, chunked ? polys_dev : nullptr,
                           chunked ? pieces : 0);
}

MIDAS_EXPORT int midas_resample_search(midas_ctx* ctx, int64_t N, const double* cdf_dev, int64_t M, int32_t mode,
                                       const double* u_dev, float u32, uint64_t seed, uint64_t step, int32_t* idx_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && M >= 0 && cdf_dev && (M == 0 || idx_dev));
    MIDAS_REQUIRE(ctx, mode == MIDAS_RESAMPLE_MULTINOMIAL || mode == MIDAS_RESAMPLE_SYSTEMATIC);
    return launch_search(ctx, N, cdf_dev, M, mode, u_dev, u32, seed, step, idx_dev);
}

MIDAS_EXPORT int midas_gather_rows(midas_ctx* ctx, int64_t M, const int32_t* idx_dev, const void* src_dev, void* dst_dev,
                                   int32_t row_bytes) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, M >= 0 && row_bytes > 0 && (M == 0 || (idx_dev && src_dev && dst_dev)));
    return launch_gather_rows(ctx, M, idx_dev, src_dev, dst_dev, row_bytes);
}

MIDAS_EXPORT int midas_rmse(midas_ctx* ctx, int64_t N, const float* poses_dev, const float* gt16_dev, double* out2_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && poses_dev && gt16_dev && out2_dev);
    return launch_rmse(ctx, N, poses_dev, gt16_dev, out2_dev);
}

// ---- fused step ----------------------------------------------------------------------------------
// largest float64 t2 with sqrt(t2) <= thr, so that  sqrt(d2) > thr  <=>  d2 > t2  exactly
static double squared_threshold(double thr) {
    if (!(thr >= 0.0)) return -1.0;  // nothing is within a negative / NaN threshold
    if (std::isinf(thr)) return INFINITY;
    double t = thr * thr;
    while (std::sqrt(t) > thr) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) <= thr) t = std::nextafter(t, INFINITY);
    return t;
}

static int filter_step_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                            const midas_step_args* args, int32_t B);

MIDAS_EXPORT int midas_cluster_centers(midas_ctx* ctx, int64_t N, const float* poses_dev, const double* weights64_dev,
                                       const float* weights32_dev, const int64_t* labels_dev, int32_t C,
                                       const int64_t* label_values_dev, float* centers_dev, float* stds_dev,
                                       int64_t* counts_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && poses_dev && labels_dev && label_values_dev && centers_dev && stds_dev && C >= 1 && C <= 64);
    MIDAS_REQUIRE(ctx, (weights64_dev == nullptr) != (weights32_dev == nullptr));
    MIDAS_REQUIRE(ctx, (uintptr_t)poses_dev % 16 == 0);
    return launch_cluster_centers(ctx, N, poses_dev, weights64_dev, weights32_dev, labels_dev, C, label_values_dev, centers_dev,
                                  stds_dev, counts_dev);
}

MIDAS_EXPORT int midas_topn_pose_error(midas_ctx* ctx, int32_t B, int64_t K, const double* scores_dev, int64_t row0, int32_t n,
                                       const double* feat_dev, int32_t d, double* err_dev, int32_t* idx_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, B >= 1 && K >= 1 && scores_dev && feat_dev && err_dev && n >= 1 && n <= 256 && d >= 1 && d <= 16);
    MIDAS_REQUIRE(ctx, row0 >= 0 && row0 + B <= K);
    return launch_topn_pose_error(ctx, B, K, scores_dev, row0, n, feat_dev, d, err_dev, idx_dev);
}

MIDAS_EXPORT int midas_filter_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6,
                                   const midas_tree* tree3, const midas_step_args* args) {
    MIDAS_ENTER(ctx);
    return filter_step_impl(ctx, cb, tree6, tree3, args, 1);
}

MIDAS_EXPORT int midas_filter_step_batch(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6,
                                         const midas_tree* tree3, const midas_step_args* args, int32_t B) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, B >= 1 && B <= 65535);
    return filter_step_impl(ctx, cb, tree6, tree3, args, B);
}

static int filter_step_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                            const midas_step_args* args, int32_t B) {
    MIDAS_REQUIRE(ctx, cb && tree6 && tree3 && args && tree6->dim == 6 && tree3->dim == 3);
    const midas_step_args& s = *args;
    MIDAS_REQUIRE(ctx, s.N > 0 && s.poses_in_dev && s.poses_prop_dev && s.poses_out_dev && s.weights_dev &&
                           s.weights_out_dev && s.nn_idx_dev && s.hint_out_dev && s.ridx_dev && s.odom16_dev &&
                           s.code_dev && s.status_dev);
    MIDAS_REQUIRE(ctx, s.poses_prop_dev != s.poses_in_dev && s.poses_prop_dev != s.poses_out_dev);
    MIDAS_REQUIRE(ctx, (s.tn_dev == nullptr) == (s.rot_dev == nullptr));
    MIDAS_REQUIRE(ctx, tree6->K == cb->K);
    const int64_t N = s.N;
    const int npart = particle_update_blocks(N);
    void *scores, *x, *e, *valid, *pmax, *pmin, *prm = nullptr, *cdf;
    int rc;
    const size_t Bz = (size_t)B;
    if ((rc = midas_scratch(ctx, Bz * cb->K * sizeof(double), &scores))) return rc;
    if ((rc = midas_scratch(ctx, Bz * N * sizeof(double), &x))) return rc;
    if ((rc = midas_scratch(ctx, Bz * N * sizeof(double), &e))) return rc;
    if ((rc = midas_scratch(ctx, Bz * N, &valid))) return rc;
    if ((rc = midas_scratch(ctx, Bz * npart * sizeof(double), &pmax))) return rc;
    if ((rc = midas_scratch(ctx, Bz * npart * sizeof(double), &pmin))) return rc;
    if (s.gt16_dev && s.rmse_dev)
        if ((rc = midas_scratch(ctx, Bz * npart * 2 * sizeof(double), &prm))) return rc;
    if ((rc = midas_scratch(ctx, Bz * N * sizeof(double), &cdf))) return rc;
    // Single trajectory: the codebook scoring and the particle update share one launch (k_frame_front); the
    // tail then gathers the scores.  Other layouts / batches: scoring, then the particle update with the scores.
    void* lp_raw = nullptr;
    // a batch scores all its codes in one pass over the codebook on the matrix cores when the layout allows it
    const bool mfma = B > 1 && cb->dtype == MIDAS_F32 && cb->D % 16 == 0 && (uintptr_t)cb->emb % 16 == 0;
    // Batch: that pass (a separate kernel shape: 1024-thread workgroups, 132 KB of LDS) runs on a side stream
    // concurrently with the particle update, which does not need the scores; the fork / join events cost ~8 us,
    // the overlap saves the ~60 us of the scoring.
    const bool defer_batch = mfma && ctx->overlap;
    if ((B == 1 && ctx->overlap) || defer_batch || (B > 1 && s.score_stamps_dev))
        if ((rc = midas_scratch(ctx, Bz * N * sizeof(double), &lp_raw))) return rc;
    if (defer_batch && !ctx->side) {
        MIDAS_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        MIDAS_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        MIDAS_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }

    if (ctx->prof && ctx->ev_ready) {  // calibration: an empty event pair measures the bracket overhead itself
        (void)hipEventRecord(ctx->ev[6], ctx->stream);
        (void)hipEventRecord(ctx->ev[7], ctx->stream);
    }
    ParticleUpdateArgs pa;
    pa.batch = B;
    pa.score_stride = cb->K;
    pa.N = N;
    pa.poses_in = s.poses_in_dev;
    pa.poses_prop = s.poses_prop_dev;
    pa.odom16 = s.odom16_dev;
    pa.tn = s.tn_dev;
    pa.rot = s.rot_dev;
    pa.std_t = s.std_t;
    pa.std_r = s.std_r;
    pa.seed = s.seed;
    pa.step = s.step;
    pa.hint_in = s.hint_in_dev;
    pa.nn_idx = s.nn_idx_dev;
    pa.scores = (const double*)scores;
    pa.x = (double*)x;
    pa.e = (double*)e;
    pa.valid = (uint8_t*)valid;
    pa.t2 = squared_threshold(s.prune_thr);
    pa.thr = s.prune_thr;
    pa.vlist = (tree6->vlist && tree6->vlist_mesh == tree3) ? (const MeshRec*)tree6->vlist : nullptr;
    pa.vscr = pa.vlist ? (const MeshScr*)tree6->vscr : nullptr;
    pa.field = tree3->field;
    pa.telemetry = (unsigned long long*)s.telemetry_dev;
    pa.status_reset = s.status_dev;
    pa.part_max = (double*)pmax;
    pa.part_min = (double*)pmin;
    pa.gt16 = prm ? s.gt16_dev : nullptr;
    pa.part_rmse = (double*)prm;
    if (B == 1 && s.score_stamps_dev && s.score_epoch) { pa.sp.stamps = s.score_stamps_dev; pa.sp.epoch = s.score_epoch; }
    bool defer = false;
    // a batch with stamps (B x K of them): every trajectory's particle waves score the rows they need from its own code -
    // the float64 arithmetic of the single-trajectory step, no matrix-core pass, no side stream
    const bool sparse_batch = B > 1 && s.score_stamps_dev && s.score_epoch && cb->dtype == MIDAS_F32 &&
                              (cb->D == 128 || cb->D == 256 || cb->D == 512 || cb->D == 1024) && (uintptr_t)cb->emb % 16 == 0 &&
                              (uintptr_t)s.code_dev % 16 == 0;
    if (sparse_batch) {
        pa.sp.stamps = s.score_stamps_dev; pa.sp.epoch = s.score_epoch;
        pa.sp.emb = (const float*)cb->emb; pa.sp.norms = cb->norms; pa.sp.code = s.code_dev; pa.sp.scores = (double*)scores;
        pa.sp.nj = cb->D / 64;
        pa.scores = nullptr;  // deferred: the tail gathers the scores
        prof_mark(ctx, 1);
        if ((rc = launch_particle_update(ctx, tree6, tree3, pa))) return rc;
        defer = true;
    }
    if (B == 1 && ctx->overlap) {
        prof_mark(ctx, 1);  // fused front: reported in the particle_update slot, the score slot stays empty
        if ((rc = launch_frame_front(ctx, tree6, tree3, pa, cb, s.code_dev, (double*)scores, &defer))) return rc;
    }
    if (defer_batch && !sparse_batch) {
        hipStream_t main_stream = ctx->stream;
        MIDAS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, main_stream));  // the codes, and last frame's readers of `scores`
        MIDAS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        ctx->stream = ctx->side;
        rc = launch_score_batch(ctx, cb, B, s.code_dev, (double*)scores);
        ctx->stream = main_stream;
        if (rc) return rc;
        MIDAS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, ctx->side));
        prof_mark(ctx, 1);
        pa.scores = nullptr;  // deferred: the tail gathers the scores
        if ((rc = launch_particle_update(ctx, tree6, tree3, pa))) return rc;
        MIDAS_HIP_CHECK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        defer = true;
    } else if (!defer) {
        prof_mark(ctx, 0);
        if ((rc = mfma ? launch_score_batch(ctx, cb, B, s.code_dev, (double*)scores)
                       : launch_score(ctx, cb, B, s.code_dev, (double*)scores)))
            return rc;
        prof_mark(ctx, 1);
        pa.sp.stamps = nullptr;  // scored densely just above
        if ((rc = launch_particle_update(ctx, tree6, tree3, pa))) return rc;
    }
    prof_mark(ctx, 2);

    StepTailArgs ta;
    ta.batch = B;
    ta.N = N;
    ta.npart = npart;
    ta.x = defer ? nullptr : (const double*)x;
    ta.scores = (const double*)scores;
    ta.score_stride = cb->K;
    ta.x_raw = (double*)x;
    ta.lp_raw = (double*)lp_raw;
    ta.e = (double*)e;
    ta.valid = (const uint8_t*)valid;
    ta.part_max = (const double*)pmax;
    ta.part_min = (const double*)pmin;
    ta.softmax = s.softmax;
    ta.weights = s.weights_dev;
    ta.cdf = (double*)cdf;
    ta.status = s.status_dev;
    ta.mode = s.resample_mode;
    ta.u = s.u_dev;
    ta.u32 = s.u32;
    ta.seed = s.seed;
    ta.step = s.step;
    ta.ridx = s.ridx_dev;
    ta.poses_prop = s.poses_prop_dev;
    ta.poses_out = s.poses_out_dev;
    ta.weights_out = s.weights_out_dev;
    ta.nn_idx = s.nn_idx_dev;
    ta.hint_out = s.hint_out_dev;
    ta.part_rmse = (const double*)prm;
    ta.rmse_out = s.rmse_dev;
    if ((rc = launch_step_tail(ctx, ta, 2))) return rc;

    if (ctx->prof && ctx->ev_ready) {
        const int lo = ctx->prof_only >= 0 ? ctx->prof_only : 0, hi = ctx->prof_only >= 0 ? ctx->prof_only + 1 : 4;
        MIDAS_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[hi]));
        for (int i = lo; i < hi; ++i) {
            float ms = 0.f;
            if (i == 0 && defer) continue;  // no separate scoring kernel in the fused front
            MIDAS_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
            ctx->prof_ms[i] += (double)ms;
        }
        float cal = 0.f;
        MIDAS_HIP_CHECK(ctx, hipEventElapsedTime(&cal, ctx->ev[6], ctx->ev[7]));
        ctx->prof_ms[7] += (double)cal;
        ctx->prof_calls += 1;
    }
    return MIDAS_OK;
}


// ---- pipelined single-trajectory step ----------------------------------------------------------------
// Layout of the caller's table block (doubles).  The per-slot and per-chunk arrays are padded to multiples of 16 so
// that the lazy front may fetch whole 16-value lines with aligned 16-byte loads (values past the data are ignored).
static TailTables tables_of(double* t, int64_t N) {
    const int64_t ng = ceil_div(N, SCAN_CHUNK), nb = ceil_div(N, SCAN_BLOCK);
    const int64_t Np = ceil_div(N, 16) * 16, ngp = ceil_div(ng, 16) * 16;
    TailTables tb;
    tb.e = t; tb.x_raw = tb.e + Np; tb.lp = tb.x_raw + Np; tb.lp_raw = tb.lp + Np;
    tb.gend = tb.lp_raw + Np; tb.gend_raw = tb.gend + ngp;
    tb.ggend = tb.gend_raw + ngp; tb.ggend_raw = tb.ggend + 16 * nb;
    tb.bsum_e = tb.ggend_raw + 16 * nb; tb.btot = tb.bsum_e + nb; tb.btot_raw = tb.btot + nb; tb.bmax = tb.btot_raw + nb; tb.bmin = tb.bmax + nb;
    return tb;
}

static int lazy_step_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                          const midas_lazy_args& s, double* rmse_out, int32_t B = 1, int64_t tstride = 0);
static int64_t tables_doubles(int64_t N) {  // size of one trajectory's table block (tables_of), padded to whole 128-byte lines
    const int64_t ng = ceil_div(N, SCAN_CHUNK), nb = ceil_div(N, SCAN_BLOCK);
    const int64_t raw = 4 * (ceil_div(N, 16) * 16) + 2 * (ceil_div(ng, 16) * 16) + 37 * nb;
    return ceil_div(raw, 16) * 16;
}

MIDAS_EXPORT int midas_lazy_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                                 const midas_lazy_args* args) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && tree6 && tree3 && args && tree6->dim == 6 && tree3->dim == 3 && tree6->K == cb->K);
    return lazy_step_impl(ctx, cb, tree6, tree3, *args, (args->gt16_dev && args->part_rmse_dev) ? args->rmse_dev : nullptr);
}

MIDAS_EXPORT int midas_lazy_run(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                                const midas_lazy_args* first, int32_t T, double* rmse_log_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && tree6 && tree3 && first && tree6->dim == 6 && tree3->dim == 3 && tree6->K == cb->K && T >= 1);
    MIDAS_REQUIRE(ctx, first->poses_prop_prev_dev && first->nn_idx_prev_dev && first->status_prev_dev && !first->tn_dev &&
                           !first->rot_dev && !first->u_prev_dev);
    MIDAS_REQUIRE(ctx, !rmse_log_dev || (first->gt16_dev && first->part_rmse_dev));
    midas_lazy_args a = *first;
    if (a.score_stamps_dev) {  // every epoch of the run is checked BEFORE anything is enqueued (the last frame uses first + inc (T - 1));
                               // the caller restarts the epochs (and zeroes the stamps) long before the limit
        const uint64_t inc = a.score_list_dev ? 2u : 1u;
        MIDAS_REQUIRE(ctx, (uint64_t)a.score_epoch + inc * (uint64_t)(T - 1) < (a.score_list_dev ? (uint64_t)MIDAS_EPOCH_LIMIT : 0xFFFFFFF0ull));
    }
    for (int32_t f = 0; f < T; ++f) {
        int rc = f ? scratch_reset(ctx) : MIDAS_OK;  // frames are ordered on the stream: each may reuse the scratch
        if (rc) return rc;
        rc = lazy_step_impl(ctx, cb, tree6, tree3, a, rmse_log_dev ? rmse_log_dev + 3 * f : nullptr);
        if (rc) return rc;
        // next frame: the buffer sets swap, the resample of this frame is folded in, the inputs advance
        float* pp = const_cast<float*>(a.poses_prop_prev_dev);
        int32_t* np = const_cast<int32_t*>(a.nn_idx_prev_dev);
        int32_t* sp = const_cast<int32_t*>(a.status_prev_dev);
        a.poses_prop_prev_dev = a.poses_prop_dev; a.nn_idx_prev_dev = a.nn_idx_dev; a.status_prev_dev = a.status_dev;
        a.poses_prop_dev = pp; a.nn_idx_dev = np; a.status_dev = sp;
        a.resample_prev = 1;
        a.u32_prev = -1.0f;
        a.step_prev = a.step;
        a.step += 1;
        if (a.score_stamps_dev) {  // never 0; two per frame with a prediction list (the tag between two epochs marks its rows)
            a.score_epoch += a.score_list_dev ? 2u : 1u;  // (range checked above, for the whole run)
        }
        a.odom16_dev += 16;
        a.code_dev += cb->D;
        if (a.gt16_dev) a.gt16_dev += 16;
    }
    return MIDAS_OK;
}

static int lazy_step_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                          const midas_lazy_args& s, double* rmse_out, int32_t B, int64_t tstride) {
    MIDAS_REQUIRE(ctx, s.N > 0 && ceil_div(s.N, SCAN_BLOCK) <= LAZY_MAX_BLOCKS && s.poses_prop_dev && s.nn_idx_dev && s.valid_dev &&
                           s.status_dev && s.tables_dev && (uintptr_t)s.tables_dev % 128 == 0 && s.scores_dev && s.odom16_dev && s.code_dev);
    MIDAS_REQUIRE(ctx, s.resample_prev ? (s.poses_prop_prev_dev && s.nn_idx_prev_dev && s.status_prev_dev &&
                                          s.poses_prop_prev_dev != s.poses_prop_dev && s.nn_idx_prev_dev != s.nn_idx_dev &&
                                          s.status_prev_dev != s.status_dev)
                                       : (s.poses_in_dev && s.poses_in_dev != s.poses_prop_dev));
    MIDAS_REQUIRE(ctx, (s.tn_dev == nullptr) == (s.rot_dev == nullptr));
    MIDAS_REQUIRE(ctx, s.resample_mode == MIDAS_RESAMPLE_MULTINOMIAL || s.resample_mode == MIDAS_RESAMPLE_SYSTEMATIC);
    const int64_t N = s.N;
    TailTables tb = tables_of(s.tables_dev, N);
    // guide tables of the summation blocks (GUIDE_BINS, midas_internal.hpp): softmax variant | raw variant
    if (s.guide_dev && B == 1) {
        MIDAS_REQUIRE(ctx, (uintptr_t)s.guide_dev % 16 == 0);
        tb.guide = reinterpret_cast<guide_t*>(s.guide_dev);
        tb.guide_raw = tb.guide + ceil_div(N, SCAN_BLOCK) * GUIDE_STRIDE;
    }
    ParticleUpdateArgs pa;
    pa.N = N;
    pa.batch = B;
    pa.score_stride = cb->K;
    pa.poses_in = s.poses_in_dev;
    pa.poses_prop = s.poses_prop_dev;
    pa.odom16 = s.odom16_dev;
    pa.tn = s.tn_dev;
    pa.rot = s.rot_dev;
    pa.std_t = s.std_t;
    pa.std_r = s.std_r;
    pa.seed = s.seed;
    pa.step = s.step;
    pa.hint_in = s.hint_in_dev;
    pa.nn_idx = s.nn_idx_dev;
    pa.scores = nullptr;
    pa.valid = s.valid_dev;
    pa.t2 = squared_threshold(s.prune_thr);
    pa.thr = s.prune_thr;
    pa.vlist = (tree6->vlist && tree6->vlist_mesh == tree3) ? (const MeshRec*)tree6->vlist : nullptr;
    pa.vscr = pa.vlist ? (const MeshScr*)tree6->vscr : nullptr;
    pa.field = tree3->field;
    pa.telemetry = (unsigned long long*)s.telemetry_dev;
    pa.status_reset = s.status_dev;
    pa.gt16 = (s.gt16_dev && s.part_rmse_dev) ? s.gt16_dev : nullptr;
    pa.part_rmse = s.part_rmse_dev;
    if (s.score_stamps_dev && s.score_epoch) { pa.sp.stamps = s.score_stamps_dev; pa.sp.epoch = s.score_epoch; }
    ScorePredict predict;
    if (pa.sp.stamps && s.score_list_dev && B == 1 && s.score_epoch >= 2 && N >= SCAN_CHUNK) {
        MIDAS_REQUIRE(ctx, s.score_epoch < MIDAS_EPOCH_LIMIT);  // (bit 31 of a stamp flags a listed row's second chance)
        const int par = (int)((s.score_epoch >> 1) & 1u);
        int32_t* base = s.score_list_dev;
        pa.sp.pred_tag = s.score_epoch - 1u;
        pa.sp.list_count = base + par;
        pa.sp.list = base + 2 + (int64_t)par * cb->K;
        pa.sp.list_cap = (int32_t)(cb->K < 0x7fffffff ? cb->K : 0x7fffffff);
        pa.sp.next_count = base + (par ^ 1);
        predict.stamps = s.score_stamps_dev; predict.epoch = s.score_epoch; predict.K = cb->K;
        predict.count = base + (par ^ 1);
        predict.list = base + 2 + (int64_t)(par ^ 1) * cb->K;
    }
    if (s.resample_prev) {
        LazyResample& r = pa.rs;
        r.enabled = true;
        r.e = tb.e; r.x_raw = tb.x_raw; r.lp = tb.lp; r.lp_raw = tb.lp_raw; r.gend = tb.gend; r.gend_raw = tb.gend_raw;
        r.ggend = tb.ggend; r.ggend_raw = tb.ggend_raw;
        r.guide = tb.guide; r.guide_raw = tb.guide_raw;
        r.bsum_e = tb.bsum_e; r.btot = tb.btot; r.btot_raw = tb.btot_raw; r.bmax = tb.bmax; r.bmin = tb.bmin;
        r.poses_prev = s.poses_prop_prev_dev; r.nn_prev = s.nn_idx_prev_dev; r.status_prev = s.status_prev_dev;
        r.ridx_out = s.ridx_dev;
        r.nb = (int)ceil_div(N, SCAN_BLOCK); r.ng = (int)ceil_div(N, SCAN_CHUNK);
        r.softmax = s.softmax; r.mode = s.resample_mode; r.u = s.u_prev_dev; r.u32 = s.u32_prev;
        r.seed = s.seed; r.step = s.step_prev;
        r.tstride = tstride;
    }
    if (ctx->prof && ctx->ev_ready) {
        (void)hipEventRecord(ctx->ev[6], ctx->stream);
        (void)hipEventRecord(ctx->ev[7], ctx->stream);
    }
    prof_mark(ctx, 1);
    bool launched = false;
    int rc;
    if (B > 1 && !s.resample_prev) {
        // a batch's first frame (nothing to fold in yet): the plain particle update over grid.y, sparse scoring per trajectory
        MIDAS_REQUIRE(ctx, pa.sp.stamps && cb->dtype == MIDAS_F32 && (cb->D == 128 || cb->D == 256 || cb->D == 512 || cb->D == 1024) &&
                               (uintptr_t)cb->emb % 16 == 0 && (uintptr_t)s.code_dev % 16 == 0);
        pa.sp.emb = (const float*)cb->emb; pa.sp.norms = cb->norms; pa.sp.code = s.code_dev; pa.sp.scores = s.scores_dev; pa.sp.nj = cb->D / 64;
        if ((rc = launch_particle_update(ctx, tree6, tree3, pa))) return rc;
    } else {
        if ((rc = launch_frame_front(ctx, tree6, tree3, pa, cb, s.code_dev, s.scores_dev, &launched))) return rc;
        if (!launched)
            return midas_set_error(ctx, MIDAS_ERR_INVALID, "codebook", B > 1 ? "the pipelined batch step needs a float32 codebook with D in {128,256,512,1024}, score stamps and N <= 262144"
                                                                              : "the pipelined step needs a float32 codebook with D in {128,256,512,1024}");
    }
    prof_mark(ctx, 2);
    if ((rc = launch_tail_a2(ctx, N, s.scores_dev, s.nn_idx_dev, s.valid_dev, s.softmax, tb, s.status_dev, B, cb->K, true,
                             pa.gt16 ? s.part_rmse_dev : nullptr, rmse_out, B > 1 ? tstride : 0, predict.stamps ? &predict : nullptr)))
        return rc;
    prof_mark(ctx, 3);
    if (ctx->prof && ctx->ev_ready) {
        const int lo = ctx->prof_only >= 0 ? ctx->prof_only : 1, hi = ctx->prof_only >= 0 ? ctx->prof_only + 1 : 3;
        if (lo >= 1 && hi <= 3) {
            MIDAS_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[hi]));
            for (int i = lo; i < hi; ++i) {
                float ms = 0.f;
                MIDAS_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
                ctx->prof_ms[i] += (double)ms;
            }
        } else {
            MIDAS_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[7]));
        }
        float cal = 0.f;
        MIDAS_HIP_CHECK(ctx, hipEventElapsedTime(&cal, ctx->ev[6], ctx->ev[7]));
        ctx->prof_ms[7] += (double)cal;
        ctx->prof_calls += 1;
    }
    return MIDAS_OK;
}

static int lazy_flush_impl(midas_ctx* ctx, const midas_lazy_flush_args& s, int32_t B, int64_t tstride);

MIDAS_EXPORT int midas_lazy_flush(midas_ctx* ctx, const midas_lazy_flush_args* args) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, args != nullptr);
    return lazy_flush_impl(ctx, *args, 1, 0);
}

static int lazy_flush_impl(midas_ctx* ctx, const midas_lazy_flush_args& s, int32_t B, int64_t tstride) {
    MIDAS_REQUIRE(ctx, s.N > 0 && s.tables_dev && s.valid_dev && s.nn_idx_dev && s.poses_prop_dev && s.status_dev && s.weights_dev &&
                           s.ridx_dev && s.poses_out_dev && s.weights_out_dev && s.hint_out_dev && s.poses_out_dev != s.poses_prop_dev);
    MIDAS_REQUIRE(ctx, s.resample_mode == MIDAS_RESAMPLE_MULTINOMIAL || s.resample_mode == MIDAS_RESAMPLE_SYSTEMATIC);
    const TailTables tb = tables_of(const_cast<double*>(s.tables_dev), s.N);
    StepTailArgs ta;
    ta.batch = B;
    ta.tstride = B > 1 ? tstride : 0;
    ta.N = s.N;
    ta.npart = 0;
    ta.x = nullptr; ta.e = nullptr; ta.cdf = nullptr; ta.part_max = nullptr; ta.part_min = nullptr;
    ta.valid = s.valid_dev;
    ta.softmax = s.softmax;
    ta.weights = s.weights_dev;
    ta.status = s.status_dev;
    ta.mode = s.resample_mode;
    ta.u = s.u_dev;
    ta.u32 = s.u32;
    ta.seed = s.seed;
    ta.step = s.step;
    ta.ridx = s.ridx_dev;
    ta.poses_prop = s.poses_prop_dev;
    ta.poses_out = s.poses_out_dev;
    ta.weights_out = s.weights_out_dev;
    ta.nn_idx = s.nn_idx_dev;
    ta.hint_out = s.hint_out_dev;
    ta.part_rmse = (s.part_rmse_dev && s.rmse_dev) ? s.part_rmse_dev : nullptr;
    ta.rmse_out = s.rmse_dev;
    return launch_tail_b2(ctx, ta, tb);
}

MIDAS_EXPORT int midas_score_list_seed(midas_ctx* ctx, int64_t K, uint32_t* score_stamps_dev, uint32_t score_epoch, int32_t* score_list_dev,
                                       int64_t N, const int32_t* nn_idx_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, K > 0 && score_stamps_dev && score_epoch >= 2 && score_epoch < MIDAS_EPOCH_LIMIT && score_list_dev && N > 0 && nn_idx_dev);
    const int par = (int)((score_epoch >> 1) & 1u);
    ScorePredict pr;
    pr.stamps = score_stamps_dev; pr.epoch = score_epoch; pr.K = K;
    pr.count = score_list_dev + (par ^ 1);
    pr.list = score_list_dev + 2 + (int64_t)(par ^ 1) * K;
    return launch_predict_seed(ctx, N, nn_idx_dev, pr);
}

// ---- pipelined batch (config 5): B trajectories, grid.y, one table block per trajectory -------------------------------
MIDAS_EXPORT int64_t midas_lazy_tables_doubles(int64_t N) { return N > 0 ? tables_doubles(N) : 0; }
MIDAS_EXPORT int midas_lazy_guide_layout(int32_t* bins_out, int32_t* unit_out, int32_t* stride_out) {
    if (bins_out) *bins_out = GUIDE_BINS;
    if (unit_out) *unit_out = GUIDE_UNIT;
    if (stride_out) *stride_out = GUIDE_STRIDE;
    return MIDAS_OK;
}
MIDAS_EXPORT int64_t midas_lazy_guide_bytes(int64_t N) { return N > 0 ? 2 * ceil_div(N, SCAN_BLOCK) * (int64_t)GUIDE_STRIDE * (int64_t)sizeof(guide_t) : 0; }

MIDAS_EXPORT int midas_lazy_step_batch(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                                       const midas_lazy_args* args, int32_t B) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && tree6 && tree3 && args && tree6->dim == 6 && tree3->dim == 3 && tree6->K == cb->K && B >= 1);
    MIDAS_REQUIRE(ctx, args->score_stamps_dev && args->score_epoch && ceil_div(args->N, SCAN_BLOCK) <= 64 && args->N >= SCAN_CHUNK);
    return lazy_step_impl(ctx, cb, tree6, tree3, *args, (args->gt16_dev && args->part_rmse_dev) ? args->rmse_dev : nullptr, B,
                          tables_doubles(args->N));
}

MIDAS_EXPORT int midas_lazy_flush_batch(midas_ctx* ctx, const midas_lazy_flush_args* args, int32_t B) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, args != nullptr && B >= 1);
    return lazy_flush_impl(ctx, *args, B, tables_doubles(args->N));
}

// ---- particle-sharded step pieces -------------------------------------------------------------------
static int shard_front_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                            const midas_shard_front_args* args, void** part_rmse_out = nullptr, int32_t* score_list = nullptr,
                            ScorePredict* predict_out = nullptr, const PeerInboxSrc* inbox = nullptr);
MIDAS_EXPORT int midas_shard_front(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6,
                                   const midas_tree* tree3, const midas_shard_front_args* args) {
    MIDAS_ENTER(ctx);
    return shard_front_impl(ctx, cb, tree6, tree3, args);
}

// part_rmse_out (C-side frame): the per-wave rmse sums are left in scratch for the tail to add up (no k_reduce_partials launch);
// score_list / predict_out (C-side frame): prediction lists of the sparse scoring as in midas_lazy_args.score_list_dev
// inbox (midas_shard_run, frames after the first): the particles are the rows of the rank's inbox - the previous frame's unpack
// folded into this front (poses_in / hint_in are not read)
static int shard_front_impl(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                            const midas_shard_front_args* args, void** part_rmse_out, int32_t* score_list, ScorePredict* predict_out,
                            const PeerInboxSrc* inbox) {
    MIDAS_REQUIRE(ctx, tree6 && tree3 && args && tree6->dim == 6 && tree3->dim == 3);
    const midas_shard_front_args& s = *args;
    MIDAS_REQUIRE(ctx, s.scores_ready || (cb && tree6->K == cb->K && s.code_dev));
    MIDAS_REQUIRE(ctx, s.N > 0 && s.poses_in_dev && s.poses_prop_dev && s.nn_idx_dev && s.valid_dev && s.scores_dev &&
                           s.odom16_dev && s.status_dev && s.flags_dev && s.poses_in_dev != s.poses_prop_dev);
    MIDAS_REQUIRE(ctx, (s.tn_dev == nullptr) == (s.rot_dev == nullptr));
    const int npart = particle_update_blocks(s.N);
    void* prm = nullptr;
    int rc;
    if (s.gt16_dev && s.rmse_sums_dev)
        if ((rc = midas_scratch(ctx, (size_t)npart * 2 * sizeof(double), &prm))) return rc;
    ParticleUpdateArgs pa;
    pa.N = s.N;
    pa.poses_in = s.poses_in_dev;
    pa.poses_prop = s.poses_prop_dev;
    pa.odom16 = s.odom16_dev;
    pa.tn = s.tn_dev;
    pa.rot = s.rot_dev;
    pa.std_t = s.std_t;
    pa.std_r = s.std_r;
    pa.seed = s.seed;
    pa.step = s.step;
    pa.slot_base = s.slot_base;
    pa.hint_in = s.hint_in_dev;
    pa.nn_idx = s.nn_idx_dev;
    pa.scores = nullptr;  // deferred: midas_shard_tail_a gathers the scores
    pa.valid = s.valid_dev;
    pa.t2 = squared_threshold(s.prune_thr);
    pa.thr = s.prune_thr;
    pa.vlist = (tree6->vlist && tree6->vlist_mesh == tree3) ? (const MeshRec*)tree6->vlist : nullptr;
    pa.vscr = pa.vlist ? (const MeshScr*)tree6->vscr : nullptr;
    pa.field = tree3->field;
    pa.telemetry = (unsigned long long*)s.telemetry_dev;
    pa.status_reset = s.status_dev;
    pa.flags_reset = s.flags_dev;
    pa.gt16 = prm ? s.gt16_dev : nullptr;
    pa.part_rmse = (double*)prm;
    if (inbox) pa.inbox = *inbox;
    if (!s.scores_ready && s.score_stamps_dev && s.score_epoch) { pa.sp.stamps = s.score_stamps_dev; pa.sp.epoch = s.score_epoch; }
    // (an epoch at the limit is an error here as in lazy_step_impl - bits 31:30 of a stamp are a listed row's age -, not a frame
    // that silently runs without its list)
    MIDAS_REQUIRE(ctx, !(pa.sp.stamps && score_list && predict_out) || s.score_epoch < MIDAS_EPOCH_LIMIT);
    if (pa.sp.stamps && score_list && predict_out && s.score_epoch >= 2 && s.N >= SCAN_CHUNK && cb) {
        const int par = (int)((s.score_epoch >> 1) & 1u);
        pa.sp.pred_tag = s.score_epoch - 1u;
        pa.sp.list_count = score_list + par;
        pa.sp.list = score_list + 2 + (int64_t)par * cb->K;
        pa.sp.list_cap = (int32_t)(cb->K < 0x7fffffff ? cb->K : 0x7fffffff);
        pa.sp.next_count = score_list + (par ^ 1);
        predict_out->stamps = s.score_stamps_dev; predict_out->epoch = s.score_epoch; predict_out->K = cb->K;
        predict_out->count = score_list + (par ^ 1);
        predict_out->list = score_list + 2 + (int64_t)(par ^ 1) * cb->K;
    }
    bool fused = false;
    if (!s.scores_ready && ctx->overlap)
        if ((rc = launch_frame_front(ctx, tree6, tree3, pa, cb, s.code_dev, s.scores_dev, &fused))) return rc;
    if (!fused) {
        pa.sp = SparseScore();  // the unfused form scores every row first
        if (!s.scores_ready)
            if ((rc = launch_score(ctx, cb, 1, s.code_dev, s.scores_dev))) return rc;
        if ((rc = launch_particle_update(ctx, tree6, tree3, pa))) return rc;
    }
    if (!fused && predict_out) *predict_out = ScorePredict();  // the unfused form scored every row: no list for the next frame
    if (part_rmse_out) { *part_rmse_out = prm; return MIDAS_OK; }
    if (prm) return launch_reduce_partials(ctx, npart, nullptr, nullptr, (const double*)prm, nullptr, s.rmse_sums_dev);
    return MIDAS_OK;
}

// tables block of one shard: the lazy layout without the per-block records (those live in the exchange record r1)
static TailTables shard_tables_of(double* t, int64_t N) {
    const int64_t ng = ceil_div(N, SCAN_CHUNK), nb = ceil_div(N, SCAN_BLOCK);
    const int64_t Np = ceil_div(N, 16) * 16, ngp = ceil_div(ng, 16) * 16;
    TailTables tb;
    tb.e = t; tb.x_raw = tb.e + Np; tb.lp = tb.x_raw + Np; tb.lp_raw = tb.lp + Np;
    tb.gend = tb.lp_raw + Np; tb.gend_raw = tb.gend + ngp;
    tb.ggend = tb.gend_raw + ngp; tb.ggend_raw = tb.ggend + 16 * nb;
    tb.bsum_e = tb.btot = tb.btot_raw = tb.bmax = tb.bmin = nullptr;
    return tb;
}

MIDAS_EXPORT int midas_shard_tail_a(midas_ctx* ctx, int64_t N, const double* scores_dev, const int32_t* nn_idx_dev,
                                    const uint8_t* valid_dev, int32_t softmax, double* tables_dev, double* r1_dev,
                                    int32_t* status_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && scores_dev && nn_idx_dev && valid_dev && tables_dev && (uintptr_t)tables_dev % 128 == 0 && r1_dev &&
                           status_dev);
    return launch_shard_tail_a(ctx, N, scores_dev, nn_idx_dev, valid_dev, softmax, shard_tables_of(tables_dev, N), r1_dev, status_dev);
}

MIDAS_EXPORT int midas_shard_tail_fin(midas_ctx* ctx, int64_t N, const double* tables_dev, const uint8_t* valid_dev,
                                      double* weights_dev, double* cdf_dev, int32_t G, const double* r1_all_dev, int32_t rank,
                                      int64_t N_total, int32_t softmax, double* rmse_dev, int32_t* status_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && G > 0 && rank >= 0 && rank < G && tables_dev && valid_dev && weights_dev && cdf_dev && r1_all_dev &&
                           N_total >= N && status_dev);
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    const TailTables tb = shard_tables_of(const_cast<double*>(tables_dev), N);
    return launch_tail_fin(ctx, N, tb.e, tb.x_raw, tb.lp, tb.lp_raw, valid_dev, weights_dev, cdf_dev, G, nb, r1_all_dev, rank,
                           (double)N_total, softmax, rmse_dev, status_dev);
}

static int shard_route(midas_ctx* ctx, const midas_shard_route_args* args, bool pack, const PeerRouteSync* sync = nullptr) {
    MIDAS_REQUIRE(ctx, args != nullptr);
    const midas_shard_route_args& s = *args;
    MIDAS_REQUIRE(ctx, s.N >= 256 && s.G > 0 && s.G <= 64 && s.rank >= 0 && s.rank < s.G && s.r1_all_dev && s.tables_dev &&
                           (uintptr_t)s.tables_dev % 128 == 0 && s.valid_dev && s.nn_idx_dev && s.poses_prop_dev && s.status_dev &&
                           s.counts_dev);
    MIDAS_REQUIRE(ctx, !pack || (s.weights_dev && (s.peers_dev || (s.send_dev && (uintptr_t)s.send_dev % 8 == 0))));
    MIDAS_REQUIRE(ctx, !pack || s.peers_dev || s.fixed_cap == 0 || (s.fixed_cap > 0 && s.ovf_cap > 0 && s.ovf_dev && (uintptr_t)s.ovf_dev % 8 == 0 && s.self_dev && (uintptr_t)s.self_dev % 8 == 0 &&
                                                     s.G * s.fixed_cap < ((int64_t)1 << 31)));
    MIDAS_REQUIRE(ctx, s.resample_mode == MIDAS_RESAMPLE_MULTINOMIAL || s.resample_mode == MIDAS_RESAMPLE_SYSTEMATIC);
    TailTables tb = shard_tables_of(const_cast<double*>(s.tables_dev), s.N);
    if (s.guide_dev) {  // (read only by the peer-mapped form's searches; the table layout of midas_lazy_args.guide_dev)
        MIDAS_REQUIRE(ctx, (uintptr_t)s.guide_dev % 16 == 0);
        tb.guide = reinterpret_cast<guide_t*>(const_cast<uint8_t*>(s.guide_dev));
        tb.guide_raw = tb.guide + ceil_div(s.N, SCAN_BLOCK) * GUIDE_STRIDE;
    }
    return launch_shard_route(ctx, s, tb, pack, sync);
}

MIDAS_EXPORT int midas_shard_route_count(midas_ctx* ctx, const midas_shard_route_args* args) {
    MIDAS_ENTER(ctx);
    return shard_route(ctx, args, false);
}

MIDAS_EXPORT int midas_shard_route_pack(midas_ctx* ctx, const midas_shard_route_args* args) {
    MIDAS_ENTER(ctx);
    return shard_route(ctx, args, true);
}

MIDAS_EXPORT int midas_shard_unpack(midas_ctx* ctx, int64_t N, const void* recv_dev, int32_t* ridx_dev, float* poses_out_dev,
                                    double* weights_out_dev, int32_t* hint_out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && recv_dev && (uintptr_t)recv_dev % 8 == 0 && ridx_dev && poses_out_dev && weights_out_dev && hint_out_dev);
    return launch_shard_unpack(ctx, N, recv_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev);
}

MIDAS_EXPORT int midas_shard_unpack_rows(midas_ctx* ctx, int64_t rows, const void* recv_dev, int32_t dest, int32_t* ridx_dev,
                                         float* poses_out_dev, double* weights_out_dev, int32_t* hint_out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, rows > 0 && recv_dev && (uintptr_t)recv_dev % 8 == 0 && ridx_dev && poses_out_dev && weights_out_dev && hint_out_dev);
    return launch_shard_unpack(ctx, rows, recv_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev, dest);
}

MIDAS_EXPORT int midas_shard_unpack_fixed(midas_ctx* ctx, int64_t rows_recv, const void* recv_dev, int64_t rows_ovf,
                                          const void* ovf_all_dev, int32_t rank, int64_t rows_self, const void* self_dev,
                                          int32_t* ridx_dev, float* poses_out_dev, double* weights_out_dev, int32_t* hint_out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, rows_recv >= 0 && rows_ovf >= 0 && rows_self >= 0 && rank >= 0 && ridx_dev && poses_out_dev && weights_out_dev &&
                           hint_out_dev);
    MIDAS_REQUIRE(ctx, (rows_recv == 0 || (recv_dev && (uintptr_t)recv_dev % 8 == 0)) && (rows_ovf == 0 || (ovf_all_dev && (uintptr_t)ovf_all_dev % 8 == 0)) &&
                           (rows_self == 0 || (self_dev && (uintptr_t)self_dev % 8 == 0)));
    int rc = MIDAS_OK;
    if (rows_recv > 0) rc = launch_shard_unpack(ctx, rows_recv, recv_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev, -1);
    if (rc == MIDAS_OK && rows_ovf > 0) rc = launch_shard_unpack(ctx, rows_ovf, ovf_all_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev, rank);
    if (rc == MIDAS_OK && rows_self > 0) rc = launch_shard_unpack(ctx, rows_self, self_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev, -1);
    return rc;
}

MIDAS_EXPORT int midas_shard_unpack_peer(midas_ctx* ctx, int64_t N, const void* inbox_dev, int32_t* ridx_dev, float* poses_out_dev,
                                         double* weights_out_dev, int32_t* hint_out_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && inbox_dev && (uintptr_t)inbox_dev % 8 == 0 && ridx_dev && poses_out_dev && weights_out_dev && hint_out_dev);
    return launch_shard_unpack_peer(ctx, N, inbox_dev, ridx_dev, poses_out_dev, weights_out_dev, hint_out_dev);
}

MIDAS_EXPORT int midas_peer_alloc(midas_ctx* ctx, int64_t bytes, void** ptr_out, void* handle64_out) {
    MIDAS_ENTER(ctx);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the interprocess handle is 64 bytes");
    MIDAS_REQUIRE(ctx, bytes > 0 && ptr_out && handle64_out);
    void* p = nullptr;
    MIDAS_HIP_CHECK(ctx, hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        MIDAS_HIP_CHECK(ctx, e);
    }
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(p, 0, (size_t)bytes, ctx->stream));
    memcpy(handle64_out, &h, 64);
    *ptr_out = p;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_peer_free(midas_ctx* ctx, void* ptr) {
    MIDAS_ENTER(ctx);
    if (ptr) MIDAS_HIP_CHECK(ctx, hipFree(ptr));
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_peer_open(midas_ctx* ctx, const void* handle64, void** ptr_out) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, handle64 && ptr_out);
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    MIDAS_HIP_CHECK(ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *ptr_out = p;
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_peer_close(midas_ctx* ctx, void* ptr) {
    MIDAS_ENTER(ctx);
    if (ptr) MIDAS_HIP_CHECK(ctx, hipIpcCloseMemHandle(ptr));
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_peer_probe_write(midas_ctx* ctx, void* const* peers_dev, int32_t G, int32_t rank, int32_t nonce) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, peers_dev && G > 0 && G <= 64 && rank >= 0 && rank < G);
    return launch_peer_probe(ctx, peers_dev, nullptr, G, rank, nonce, nullptr);
}

MIDAS_EXPORT int midas_peer_probe_check(midas_ctx* ctx, const void* inbox_dev, int32_t G, int32_t nonce, int32_t* ok_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, inbox_dev && G > 0 && G <= 64 && ok_dev);
    return launch_peer_probe(ctx, nullptr, inbox_dev, G, 0, nonce, ok_dev);
}

// ---- the sharded frame enqueued by ONE call, on a library-owned RCCL communicator ------------------------------------------
struct midas_comm;
extern "C" int midas_comm_all_gather(midas_comm* c, const void* send_dev, void* recv_dev, int64_t bytes);

// from_inbox (midas_shard_run): the front takes its particles from the rows of the inbox (the previous frame ran without its
// UNPACK phase; its route kernel ended with the inbox complete)
static int shard_step_impl(midas_ctx* ctx, midas_comm* comm, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                           const midas_shard_step_args& s, int32_t phases, bool from_inbox = false) {
    MIDAS_REQUIRE(ctx, phases != 0 && (phases & ~31) == 0);
    MIDAS_REQUIRE(ctx, s.front.N >= 256 && s.G >= 1 && s.G <= 64 && s.rank >= 0 && s.rank < s.G && s.tables_dev && s.r1_dev);
    MIDAS_REQUIRE(ctx, s.r1_all_dev || !(phases & (MIDAS_SHARD_PHASE_GATHER | MIDAS_SHARD_PHASE_ROUTE)));
    const int64_t N = s.front.N;
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    const int64_t rec = 5 * (int64_t)nb + 4;
    int rc;
    if (phases & MIDAS_SHARD_PHASE_LOCAL) {  // propagate / NN / prune / scoring, then the shard's softmax tables and its record
        void* prm = nullptr;
        ScorePredict predict;
        PeerInboxSrc src;
        if (from_inbox) {
            MIDAS_REQUIRE(ctx, s.inbox_dev && s.flag_offset >= N * PEER_ROW);
            src.rows = (const char*)s.inbox_dev;
        }
        if ((rc = shard_front_impl(ctx, cb, tree6, tree3, &s.front, &prm, s.score_list_dev, &predict, from_inbox ? &src : nullptr))) return rc;
        MIDAS_REQUIRE(ctx, (uintptr_t)s.tables_dev % 128 == 0);
        TailTables tbs = shard_tables_of(s.tables_dev, N);
        if (s.guide_dev) {
            MIDAS_REQUIRE(ctx, (uintptr_t)s.guide_dev % 16 == 0);
            tbs.guide = reinterpret_cast<guide_t*>(s.guide_dev);
            tbs.guide_raw = tbs.guide + nb * GUIDE_STRIDE;
        }
        if ((rc = launch_shard_tail_a(ctx, N, s.front.scores_dev, s.front.nn_idx_dev, s.front.valid_dev, s.softmax,
                                      tbs, s.r1_dev, s.front.status_dev, (const double*)prm,
                                      predict.stamps ? &predict : nullptr)))
            return rc;
    }
    if (phases & MIDAS_SHARD_PHASE_GATHER) {  // one record per rank, in rank order, to every rank
        MIDAS_REQUIRE(ctx, comm != nullptr);
        if ((rc = midas_comm_all_gather(comm, s.r1_dev, s.r1_all_dev, rec * (int64_t)sizeof(double)))) return rc;
    }
    if (phases & (MIDAS_SHARD_PHASE_ROUTE | MIDAS_SHARD_PHASE_UNPACK))
        MIDAS_REQUIRE(ctx, s.peers_dev && s.inbox_dev && s.flag_offset >= N * PEER_ROW && s.flag_offset % 8 == 0 && s.frame_tag != 0 &&
                               s.counts_dev && s.weights_dev && s.ridx_dev && s.poses_out_dev && s.weights_out_dev && s.hint_out_dev);
    if (phases & MIDAS_SHARD_PHASE_ROUTE) {  // owner-side resample into the peers' inboxes, then the completion flags
        midas_shard_route_args r;
        memset(&r, 0, sizeof(r));
        r.N = N; r.G = s.G; r.rank = s.rank;
        r.r1_all_dev = s.r1_all_dev; r.tables_dev = s.tables_dev; r.valid_dev = s.front.valid_dev; r.nn_idx_dev = s.front.nn_idx_dev;
        r.poses_prop_dev = s.front.poses_prop_dev; r.status_dev = s.front.status_dev; r.rmse_dev = s.rmse_dev;
        r.softmax = s.softmax; r.resample_mode = s.resample_mode; r.u_all_dev = s.u_all_dev; r.u32 = s.u32;
        r.seed = s.front.seed; r.step = s.front.step;
        r.counts_dev = s.counts_dev; r.weights_dev = s.weights_dev; r.peers_dev = s.peers_dev;
        r.guide_dev = s.guide_dev;
        // without FLAG the route kernel's last workgroup publishes this rank's flag and waits for every rank's: when the kernel
        // ends the inbox is complete (one polling wave; the word behind the 64 flags is its workgroup counter)
        const PeerRouteSync sync{(const char*)s.inbox_dev, s.flag_offset, s.frame_tag};
        if ((rc = shard_route(ctx, &r, true, (phases & MIDAS_SHARD_PHASE_FLAG) ? nullptr : &sync))) return rc;
        if (phases & MIDAS_SHARD_PHASE_FLAG)  // shards of one process on one stream: the flags must be out before ANY shard waits
            if ((rc = launch_peer_flag_write(ctx, s.peers_dev, s.G, s.rank, s.flag_offset, s.frame_tag))) return rc;
    }
    if (phases & MIDAS_SHARD_PHASE_UNPACK) {  // inbox -> slots; with FLAG behind a wait for every rank's flag in the own inbox
        if (phases & MIDAS_SHARD_PHASE_FLAG)
            rc = launch_shard_unpack_peer_wait(ctx, N, s.inbox_dev, s.ridx_dev, s.poses_out_dev, s.weights_out_dev, s.hint_out_dev,
                                               s.G, s.flag_offset, s.frame_tag, s.front.status_dev, nullptr, s.rank);
        else
            rc = launch_shard_unpack_peer(ctx, N, s.inbox_dev, s.ridx_dev, s.poses_out_dev, s.weights_out_dev, s.hint_out_dev);
        if (rc) return rc;
    }
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_shard_step(midas_ctx* ctx, midas_comm* comm, const midas_codebook* cb, const midas_tree* tree6,
                                  const midas_tree* tree3, const midas_shard_step_args* args, int32_t phases) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, args != nullptr);
    return shard_step_impl(ctx, comm, cb, tree6, tree3, *args, phases);
}

MIDAS_EXPORT int midas_shard_run(midas_ctx* ctx, midas_comm* comm, const midas_codebook* cb, const midas_tree* tree6,
                                 const midas_tree* tree3, const midas_shard_step_args* first, int32_t T) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, first != nullptr && comm != nullptr && cb != nullptr && T >= 1);
    MIDAS_REQUIRE(ctx, !first->front.tn_dev && !first->front.rot_dev && !first->u_all_dev && !first->front.scores_ready);
    midas_shard_step_args a = *first;
    if (a.front.score_stamps_dev)  // every epoch of the run checked before anything is enqueued (see midas_lazy_run)
        MIDAS_REQUIRE(ctx, (uint64_t)a.front.score_epoch + (a.score_list_dev ? 2ull : 1ull) * (uint64_t)(T - 1) < (uint64_t)MIDAS_EPOCH_LIMIT);
    // The unpack of every frame but the last is folded into the NEXT frame's front: the rows other ranks stored into this rank's
    // inbox are read there, behind the same flag wait (MIDAS_SHARD_FOLD=0: every frame unpacks into the particle arrays).
    // Safe with one inbox: a peer stores the rows of frame f + 1 behind its record all_gather of frame f + 1, which completes only
    // when every rank has joined it - and a rank joins behind its own front of frame f + 1, the reader of the rows of frame f.
    static const bool fold = !(getenv("MIDAS_SHARD_FOLD") && getenv("MIDAS_SHARD_FOLD")[0] == '0');
    for (int32_t f = 0; f < T; ++f) {
        int rc = f ? scratch_reset(ctx) : MIDAS_OK;
        if (rc) return rc;
        const bool last = f == T - 1;
        const int32_t phases = MIDAS_SHARD_PHASE_LOCAL | MIDAS_SHARD_PHASE_GATHER | MIDAS_SHARD_PHASE_ROUTE | ((last || !fold) ? MIDAS_SHARD_PHASE_UNPACK : 0);
        if ((rc = shard_step_impl(ctx, comm, cb, tree6, tree3, a, phases, fold && f > 0))) return rc;
        // next frame: the resampled particles are in poses_out / hint_out (= the front's inputs: the engine passes the same buffers)
        a.front.step += 1;
        a.frame_tag += 1;
        a.u32 = -1.0f;
        if (a.front.score_stamps_dev) a.front.score_epoch += a.score_list_dev ? 2u : 1u;
        a.front.odom16_dev += 16;
        a.front.code_dev += cb->D;
        if (a.front.gt16_dev) a.front.gt16_dev += 16;
    }
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_tail_resample(midas_ctx* ctx, const midas_tail_resample_args* args) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, args && args->N > 0 && args->n_per_rank > 0 && args->N_all >= args->N && args->slot_base >= 0 &&
                           args->slot_base + args->N <= args->N_all && args->pack_all_dev &&
                           args->rank_stride >= 84 * args->n_per_rank && args->rank_stride % 16 == 0 &&
                           args->n_per_rank % 2 == 0 && args->status_dev && args->ridx_dev && args->poses_out_dev &&
                           args->weights_out_dev && args->hint_out_dev);
    MIDAS_REQUIRE(ctx, args->mode == MIDAS_RESAMPLE_MULTINOMIAL || args->mode == MIDAS_RESAMPLE_SYSTEMATIC);
    return launch_tail_resample(ctx, *args);
}

// ---- the whole loop body on a variable-size particle set ----------------------------------------------
MIDAS_EXPORT int midas_loop_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                                 const midas_loop_args* args, int32_t phases) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, args != nullptr && phases != 0 && (phases & ~15) == 0);
    const midas_loop_args& s = *args;
    MIDAS_REQUIRE(ctx, s.cap > 0 && ceil_div(s.cap, SCAN_BLOCK) <= LAZY_MAX_BLOCKS && s.ctl_i_dev && s.ctl_d_dev);
    MIDAS_REQUIRE(ctx, s.poses_dev && s.poses_prop_dev && s.poses_dev != s.poses_prop_dev && s.hint_dev && s.nn_idx_dev && s.valid_dev &&
                           s.x_dev && s.e_dev && s.weights_dev && s.weights_out_dev && s.labels_dev && s.labels_out_dev &&
                           s.labels_dev != s.labels_out_dev && s.src_dev && s.ridx_dev && s.scores_dev && s.cluster_poses_dev &&
                           s.cluster_stds_dev);
    MIDAS_REQUIRE(ctx, (uintptr_t)s.poses_dev % 16 == 0 && (uintptr_t)s.poses_prop_dev % 16 == 0);
    if (phases & MIDAS_LOOP_FRONT) {
        MIDAS_REQUIRE(ctx, cb && tree6 && tree3 && tree6->dim == 6 && tree3->dim == 3 && tree6->K == cb->K);
        MIDAS_REQUIRE(ctx, s.odom16_dev && s.code_dev && s.cb_poses_dev && (uintptr_t)s.cb_poses_dev % 16 == 0);
        MIDAS_REQUIRE(ctx, (s.tn_dev == nullptr) == (s.rot_dev == nullptr));
    }
    if (phases & MIDAS_LOOP_DBSCAN) MIDAS_REQUIRE(ctx, s.eps > 0.0);
    if (phases & MIDAS_LOOP_RESAMPLE)
        MIDAS_REQUIRE(ctx, s.resample_mode == MIDAS_RESAMPLE_MULTINOMIAL || s.resample_mode == MIDAS_RESAMPLE_SYSTEMATIC);
    return launch_loop_step(ctx, cb, tree6, tree3, s, phases);
}

MIDAS_EXPORT int midas_anneal_select(midas_ctx* ctx, int64_t N, const double* weights_dev, int32_t mode, int64_t k,
                                     int32_t* src_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && ceil_div(N, SCAN_BLOCK) <= LAZY_MAX_BLOCKS && weights_dev && src_dev && (mode == 1 || mode == 2) &&
                           k >= 0 && k <= N / 3);
    return launch_anneal_select(ctx, N, weights_dev, k == 0 ? 0 : mode, k, MIDAS_TOPK_TIES_INDEX, src_dev, nullptr);
}

MIDAS_EXPORT int midas_anneal_select_ties(midas_ctx* ctx, int64_t N, const double* weights_dev, int32_t mode, int64_t k, int32_t ties,
                                          int32_t* src_dev, int32_t* info_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && ceil_div(N, SCAN_BLOCK) <= LAZY_MAX_BLOCKS && weights_dev && src_dev && (mode == 1 || mode == 2) &&
                           k >= 0 && k <= N / 3 && (ties == MIDAS_TOPK_TIES_INDEX || ties == MIDAS_TOPK_TIES_ATEN_CPU));
    return launch_anneal_select(ctx, N, weights_dev, k == 0 ? 0 : mode, k, ties, src_dev, info_dev);
}

MIDAS_EXPORT int midas_dbscan(midas_ctx* ctx, int64_t N, const float* poses_dev, double eps, int64_t min_samples,
                              int32_t* labels_dev, int32_t* ncl_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && N < ((int64_t)1 << 31) && poses_dev && labels_dev && ncl_dev && eps > 0.0);
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(ncl_dev, 0, 2 * sizeof(int32_t), ctx->stream));
    return launch_dbscan(ctx, N, nullptr, poses_dev, eps, min_samples, labels_dev, ncl_dev, ncl_dev + 1, 0);
}

MIDAS_EXPORT int midas_dbscan_points(midas_ctx* ctx, int64_t N, int32_t dim, const double* points_dev, double eps, int64_t min_samples,
                                     int32_t* labels_dev, int32_t* info_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, N > 0 && N < ((int64_t)1 << 31) && dim >= 2 && dim <= 6 && points_dev && labels_dev && info_dev && eps > 0.0);
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(info_dev, 0, 2 * sizeof(int32_t), ctx->stream));
    return launch_dbscan_points(ctx, N, dim, points_dev, eps, min_samples, labels_dev, info_dev);
}

MIDAS_EXPORT int midas_selfsim_panel(midas_ctx* ctx, const midas_codebook* cb, int64_t i0, int64_t R, float* panel_dev, int64_t ldo) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && cb->dtype == MIDAS_F32 && cb->D % 32 == 0 && (uintptr_t)cb->emb % 16 == 0 && panel_dev && (uintptr_t)panel_dev % 16 == 0);
    MIDAS_REQUIRE(ctx, i0 >= 0 && R >= 1 && i0 + R <= cb->K && ldo >= ceil_div(cb->K, 128) * 128 && ldo % 4 == 0);
    return launch_selfsim_panel(ctx, cb, i0, R, panel_dev, ldo);
}

MIDAS_EXPORT int midas_selfsim_topn(midas_ctx* ctx, const midas_codebook* cb, int32_t n, const double* feat_dev, int32_t d,
                                    int64_t rows_per_panel, double* err_dev, int32_t* idx_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, cb && cb->dtype == MIDAS_F32 && cb->D % 32 == 0 && (uintptr_t)cb->emb % 16 == 0 && feat_dev && err_dev);
    MIDAS_REQUIRE(ctx, n >= 1 && n <= 256 && d >= 1 && d <= 16 && rows_per_panel >= 128);
    const int64_t K = cb->K, ldo = ceil_div(K, 128) * 128;
    const int64_t R = ceil_div(rows_per_panel < K ? rows_per_panel : K, 128) * 128;
    const int64_t npanels = ceil_div(K, R);
    // Two panels: the selection of panel p (bound by its reads of the panel and by LDS sorts) runs on a side stream beside
    // the GEMM of panel p + 1 (bound by the matrix pipe).  Events hand the panels back and forth.
    const int nbuf = npanels > 1 ? 2 : 1;
    void* panel;
    int rc = midas_scratch(ctx, ((size_t)nbuf * R + 1) * ldo * sizeof(float), &panel);  // + one row: float32 reciprocal norms (the selection's screen)
    if (rc) return rc;
    float* rinv = (float*)panel + (size_t)nbuf * R * ldo;
    rc = launch_topn_rinv(ctx, K, ldo, cb->norms, rinv);
    if (rc) return rc;
    const char* stream_env = getenv("MIDAS_TOPN_STREAM");  // 1: the streaming selection kernel for every row (A/B runs and tests)
    if (stream_env && stream_env[0] == '1') rinv = nullptr;
    if (!ctx->side) MIDAS_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    hipEvent_t ev_gemm[2] = {nullptr, nullptr}, ev_sel[2] = {nullptr, nullptr};
    hipStream_t main_stream = ctx->stream;
    // every exit goes through `finish`: the side stream is joined behind the main stream again (the scratch panels may be handed
    // to the next API call) and the events are destroyed, whatever failed on the way
    auto finish = [&](int code) {
        ctx->stream = main_stream;
        hipEvent_t join = nullptr;
        if (hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(join, ctx->side) == hipSuccess) (void)hipStreamWaitEvent(main_stream, join, 0);
            (void)hipEventDestroy(join);
        }
        for (int k = 0; k < 2; ++k) {
            if (ev_gemm[k]) (void)hipEventDestroy(ev_gemm[k]);
            if (ev_sel[k]) (void)hipEventDestroy(ev_sel[k]);
        }
        return code;
    };
#define TOPN_CHECK(expr)                                                                                              \
    do {                                                                                                              \
        hipError_t _e = (expr);                                                                                       \
        if (_e != hipSuccess) return finish(midas_set_error(ctx, MIDAS_ERR_HIP, #expr, hipGetErrorString(_e)));       \
    } while (0)
    for (int k = 0; k < nbuf; ++k) {
        TOPN_CHECK(hipEventCreateWithFlags(&ev_gemm[k], hipEventDisableTiming));
        TOPN_CHECK(hipEventCreateWithFlags(&ev_sel[k], hipEventDisableTiming));
    }
    // the side stream starts behind whatever the main stream holds (the caller's inputs)
    TOPN_CHECK(hipEventRecord(ev_sel[0], main_stream));
    TOPN_CHECK(hipStreamWaitEvent(ctx->side, ev_sel[0], 0));
    for (int64_t p = 0; p < npanels; ++p) {
        const int k = (int)(p % nbuf);
        const int64_t i0 = p * R, rows = K - i0 < R ? K - i0 : R;
        float* pan = (float*)panel + (size_t)k * R * ldo;
        if (p >= nbuf) TOPN_CHECK(hipStreamWaitEvent(main_stream, ev_sel[k], 0));  // the panel's previous tenant has been consumed
        rc = launch_selfsim_panel(ctx, cb, i0, rows, pan, ldo);
        if (rc) return finish(rc);
        TOPN_CHECK(hipEventRecord(ev_gemm[k], main_stream));
        TOPN_CHECK(hipStreamWaitEvent(ctx->side, ev_gemm[k], 0));
        ctx->stream = ctx->side;  // the launcher enqueues on ctx->stream
        rc = launch_topn_pose_error_dots(ctx, (int32_t)rows, K, pan, ldo, cb->norms, rinv, i0, n, feat_dev, d, err_dev + i0,
                                         idx_dev ? idx_dev + i0 * n : nullptr);
        ctx->stream = main_stream;
        if (rc) return finish(rc);
        TOPN_CHECK(hipEventRecord(ev_sel[k], ctx->side));
    }
#undef TOPN_CHECK
    return finish(MIDAS_OK);  // the results are ordered behind the main stream again
}

MIDAS_EXPORT int midas_selftest_wave_sums(midas_ctx* ctx, const double* in64_dev, double* out256_dev) {
    MIDAS_ENTER(ctx);
    MIDAS_REQUIRE(ctx, in64_dev && out256_dev);
    return midas::launch_selftest_wave_sums(ctx, in64_dev, out256_dev);
}

#ifdef MIDAS_DEBUG_CLOCKS
MIDAS_EXPORT int midas_debug_tb2_clocks(long long* out16) { return midas::debug_tb2_clocks(out16); }
MIDAS_EXPORT int midas_debug_ta_clocks(long long* out16) { return midas::debug_ta_clocks(out16); }
MIDAS_EXPORT int midas_debug_tg_clocks(long long* io64, int reset) { return midas::debug_tg_clocks(io64, reset); }
MIDAS_EXPORT int midas_debug_tg_waves(long long* out4096) { return midas::debug_tg_waves(out4096); }
MIDAS_EXPORT int midas_debug_ff_clocks(long long* io8192, int reset) { return midas::debug_ff_clocks(io8192, reset); }
#endif

// ---- profiling -----------------------------------------------------------------------------------
MIDAS_EXPORT int midas_profile_enable(midas_ctx* ctx, int32_t on) {
    if (!ctx) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->ev_ready) {
        for (auto& e : ctx->ev) MIDAS_HIP_CHECK(ctx, hipEventCreate(&e));
        ctx->ev_ready = true;
    }
    ctx->prof = on != 0;
    ctx->prof_only = on >= 2 ? on - 2 : -1;  // 1: every kernel of the step; 2 + k: only kernel slot k
    return MIDAS_OK;
}

MIDAS_EXPORT int midas_profile_read(midas_ctx* ctx, double* ms_out, int64_t* calls_out, int32_t reset) {
    if (!ctx || !ms_out || !calls_out) return MIDAS_ERR_INVALID;
    MIDAS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < MIDAS_PROF_SLOTS; ++i) ms_out[i] = ctx->prof_ms[i];
    *calls_out = ctx->prof_calls;
    if (reset) {
        for (auto& v : ctx->prof_ms) v = 0.0;
        ctx->prof_calls = 0;
    }
    return MIDAS_OK;
}

MIDAS_EXPORT const char* midas_profile_slot_name(int32_t slot) {
    return (slot >= 0 && slot < MIDAS_PROF_SLOTS) ? kSlotNames[slot] : "";
}

}  // extern "C"
