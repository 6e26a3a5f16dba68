// midas_internal.hpp - shared declarations of the libmidas_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/midas_hip.h"

namespace midas {

// ---- spatial index layouts (shared host/device) ------------------------------------------------
// Complete 8-ary tree of axis-aligned boxes in 0-based heap order: node n has children 8n+1 .. 8n+8,
// level l starts at id (8^l - 1)/7; the 8^L nodes of level L are leaves.  Every node stores the box of
// its points, so the eight child boxes of a node are 384 contiguous bytes (one 48-byte box per lane of
// an octet).  Leaf i owns the LEAF_CAP fixed point slots pts[i*LEAF_CAP ..); unused slots hold +inf
// coordinates.  Built from balanced median splits (three binary splits per 8-ary level).
struct alignas(16) Box6 { float lo[6], hi[6]; };
struct alignas(16) Point6 { float c[6]; int32_t idx; int32_t pad; };
struct alignas(16) Box3 { double lo[3], hi[3]; };
struct alignas(16) Point3 { double c[3]; int64_t idx; };

struct Kd6 {
    using T = float;
    using Box = Box6;
    using Point = Point6;
    static constexpr int DIM = 6;
};
struct Kd3 {
    using T = double;
    using Box = Box3;
    using Point = Point3;
    static constexpr int DIM = 3;
};

constexpr int LEAF_CAP = 16;      // point slots per leaf (two per lane of an octet)
constexpr int KD_MAX_LEVELS = 10; // 8-ary levels (8^10 leaves x 16 slots is far beyond any codebook)

// Neighbour record of the hint fast path (dim 6 only): entry k owns NBR_M records sorted by the
// distance rho from F_k to that neighbour (rounded down), coordinates inlined so that one 32-byte
// read is one candidate.  Unused records are sentinels (+inf coordinates, rho = +inf).
struct alignas(16) Nbr6 { float c[6]; int32_t idx; float rho; };
#ifndef MIDAS_NBR_M
#define MIDAS_NBR_M 512
#endif
constexpr int NBR_M = MIDAS_NBR_M;
constexpr int NBR_REC = NBR_M + 1;  // record 0 = the entry itself

// Mesh-vertex record of the prune fast path: the vertices nearest to a codebook entry's translation.
struct alignas(16) MeshRec { double c[3]; float rho; int32_t pad; };
// float32 screening copy of a MeshRec (vertex rounded to nearest; the header's translation is a float32 value already):
// half the bytes and float32 arithmetic for the decisions that are not within rounding of the threshold (mesh_screen_check)
struct alignas(16) MeshScr { float c[3]; float rho; };
constexpr int MESH_M = 256;
constexpr int MESH_REC = MESH_M + 1;  // record 0 = header: c = the entry's translation, rho = distance of the
                                      // first vertex NOT in the list

// Distance field of a mesh (dim-3 trees): the exact distance from the CENTRE of every cell of a uniform grid to its nearest mesh
// vertex, as float32.  The distance to the nearest vertex is 1-Lipschitz, so a particle at distance rho from its cell's centre has
// d in [v - rho, v + rho]: the prune ("some vertex within thr", modules/particle_filter.py:386-391) is DECIDED for every particle
// outside a shell of half-width rho (< 0.87 cells) around the threshold surface - by one 4-byte read requested before the
// nearest-neighbour search - and only the shell goes on to the vertex lists / the tree, which remain the exact decision.
// A point outside the grid (the vertices' bounding box grown by `expand`) is farther than `expand` from every vertex.
struct MeshField {
    const float* d = nullptr;  // [n[2]][n[1]][n[0]]
    float lo[3] = {0.f, 0.f, 0.f};
    float h = 0.f, inv_h = 0.f;
    int32_t n[3] = {0, 0, 0};
    float expand = 0.f;
};

template <class KD>
struct TreeView {
    const typename KD::Box* boxes;  // [(8^(L+1) - 1)/7]
    const typename KD::Point* pts;  // [8^L * LEAF_CAP]
    const int32_t* inv_perm;        // [K] original index -> slot in pts
    const Nbr6* nbrs;               // [K * NBR_REC] (dim 6) or nullptr
    const float* rho_out;           // [K] distance from F_k to its (NBR_M+1)-th neighbour, rounded down
    const int32_t* twin;            // [K] entry across the |log R| = pi cut (or -1) - second hint of the scan
    int32_t levels;                 // L: number of 8-ary levels above the leaves
    int64_t K;
};


}  // namespace midas

// ---- opaque handles ---------------------------------------------------------------------------
struct midas_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // growable scratch
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    std::string last_error;
    // profiling
    bool prof = false;
    int prof_only = -1;
    hipEvent_t ev[MIDAS_PROF_SLOTS + 1] = {};
    bool ev_ready = false;
    double prof_ms[MIDAS_PROF_SLOTS] = {};
    int64_t prof_calls = 0;
    bool overlap = true;  // MIDAS_OVERLAP=0: separate scoring and particle-update launches (the pre-fusion path)
    // side stream of the batch step: the matrix-core scoring of the B codes runs there, concurrently with the particle
    // update (created on first use, non-blocking, forked from / joined into `stream` by the two events)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // hand-over records of the grouped tail (tail_group.hpp): [tail_rec_blocks x TG_BLOCK_WORDS] words, zeroed at allocation
    // (midas_ctx_create); tail_tag numbers its launches (a record is current when its pairs carry the launch's key)
    unsigned long long* tail_rec = nullptr;
    int tail_rec_blocks = 0;
    uint32_t tail_tag = 0;
};

struct midas_codebook {
    midas_ctx* ctx;
    int64_t K;
    int32_t D;
    int32_t dtype;
    const void* emb;  // caller-owned
    double* norms;    // [K] max(|C_k|, 1e-8), library-owned
};

struct midas_tree {
    midas_ctx* ctx;
    int32_t dim;
    int64_t K;
    int32_t levels;
    void* boxes;
    void* pts;
    int32_t* inv_perm;
    void* nbrs;      // Nbr6[K * NBR_REC] (dim 6)
    float* rho_out;  // [K] (dim 6)
    int32_t* twin;   // [K] (dim 6)
    void* vlist;     // MeshRec[K * MESH_REC] (dim 6, after midas_tree_attach_mesh)
    void* vscr;      // MeshScr[K * MESH_REC]: float32 screening copy of vlist (nullable)
    const midas_tree* vlist_mesh;  // the mesh tree the lists were built from
    void* host;      // host copy of the tree (dim 3: used to build the lists)
    midas::MeshField field;  // dim 3: distance field of the vertices (d == nullptr: none); library-owned
};

// ---- error helpers ----------------------------------------------------------------------------
int midas_set_error(midas_ctx* ctx, int code, const char* what, const char* detail);

#define MIDAS_HIP_CHECK(ctx, expr)                                                     \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return midas_set_error((ctx), MIDAS_ERR_HIP, #expr, hipGetErrorString(_e)); \
    } while (0)

#define MIDAS_REQUIRE(ctx, cond)                                                       \
    do {                                                                               \
        if (!(cond)) return midas_set_error((ctx), MIDAS_ERR_INVALID, #cond, "invalid argument"); \
    } while (0)

// Code objects load lazily, per translation unit, at the first launch of one of its kernels (tens of ms for the large ones, in
// the middle of whatever frame happens to be the first to need them: the "cold frame" of a run).  Every unit exports a function
// that makes the runtime load it now - hipFuncGetAttributes resolves a kernel without launching it - and midas_ctx_create calls
// them all (MIDAS_LAZY_MODULES=1 keeps the lazy behaviour).
#define MIDAS_WARM_TU(name, kernel)                                                          \
    int warm_##name() {                                                                      \
        hipFuncAttributes attr;                                                              \
        return (int)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&kernel));     \
    }
#define MIDAS_WARM_DECL(name) int warm_##name();
namespace midas {
MIDAS_WARM_DECL(score) MIDAS_WARM_DECL(particles) MIDAS_WARM_DECL(resample) MIDAS_WARM_DECL(cluster) MIDAS_WARM_DECL(topn)
MIDAS_WARM_DECL(selfsim) MIDAS_WARM_DECL(loop) MIDAS_WARM_DECL(dbscan) MIDAS_WARM_DECL(dbscan_nd) MIDAS_WARM_DECL(index_build)
MIDAS_WARM_DECL(mt19937) MIDAS_WARM_DECL(topk_aten)
}  // namespace midas

// scratch carve-out (stream-ordered reuse; one stream per context)
int midas_scratch(midas_ctx* ctx, size_t bytes, void** out);

namespace midas {

// control block of the loop engine (midas_loop_args.ctl_i_dev / ctl_d_dev; include/midas_hip.h MIDAS_LOOP_*)
enum : int {
    LOOP_I_N = MIDAS_LOOP_I_N, LOOP_I_NSET = MIDAS_LOOP_I_NSET, LOOP_I_MODE = MIDAS_LOOP_I_MODE, LOOP_I_K = MIDAS_LOOP_I_K,
    LOOP_I_INIT = MIDAS_LOOP_I_INIT, LOOP_I_VARSET = MIDAS_LOOP_I_VARSET, LOOP_I_KEPT = MIDAS_LOOP_I_KEPT,
    LOOP_I_DRIFT = MIDAS_LOOP_I_DRIFT, LOOP_I_STATUS = MIDAS_LOOP_I_STATUS, LOOP_I_RAW = MIDAS_LOOP_I_RAW,
    LOOP_I_NCL = MIDAS_LOOP_I_NCL, LOOP_I_NPRES = MIDAS_LOOP_I_NPRES, LOOP_I_FRAME = MIDAS_LOOP_I_FRAME,
    LOOP_I_NAN = MIDAS_LOOP_I_NAN, LOOP_I_ERR = MIDAS_LOOP_I_ERR,
    LOOP_D_S = MIDAS_LOOP_D_S, LOOP_D_VARPREV = MIDAS_LOOP_D_VARPREV, LOOP_D_VAR = MIDAS_LOOP_D_VAR,
    LOOP_D_RMSE_T = MIDAS_LOOP_D_RMSE_T, LOOP_D_RMSE_R = MIDAS_LOOP_D_RMSE_R, LOOP_D_XMAX = MIDAS_LOOP_D_XMAX,
    LOOP_D_XMIN = MIDAS_LOOP_D_XMIN, LOOP_D_TOTAL = MIDAS_LOOP_D_TOTAL,
    LOOP_MAX_CLUSTERS = MIDAS_LOOP_MAX_CLUSTERS,
};

// blocked-scan spec constants (DESIGN.md "Summation order")
constexpr int SCAN_CHUNK = 16;
constexpr int SCAN_TPB = 256;
constexpr int SCAN_BLOCK = SCAN_CHUNK * SCAN_TPB;  // 4096 values per block
// Guide table of a summation block (single-trajectory pipelined step, optional): GUIDE_BINS equal bins over the block's masked
// total W, edge k at fl(k * (W / GUIDE_BINS)) (a power of two: the bin width is exact).  The block's slots are taken in units of
// GUIDE_UNIT (16: the chunks of the summation spec; 8 / 4: halves / quarters of them); entry k = min(number of the block's unit
// ends < edge k, units of the block - 1), entry GUIDE_BINS = units of the block - 1.  A draw whose block-local target lies in
// bin k finds its slot in unit entry[k] .. entry[k + 1]: one entry pair and GUIDE_UNIT (or twice that) prefix values instead of
// the group-end, chunk-end and slot lines of search_in_block (a hint: the exact fix-up behind it decides).
#ifndef MIDAS_GUIDE_BINS
#define MIDAS_GUIDE_BINS 2048
#endif
#ifndef MIDAS_GUIDE_UNIT
#define MIDAS_GUIDE_UNIT 8
#endif
typedef uint16_t guide_t;
constexpr int GUIDE_BINS = MIDAS_GUIDE_BINS, GUIDE_STRIDE = GUIDE_BINS + 16, GUIDE_UNIT = MIDAS_GUIDE_UNIT;  // (stride in entries)
constexpr double GUIDE_WIDTH = 1.0 / GUIDE_BINS;
constexpr int TAIL_GUIDE_LDS = GUIDE_BINS / 2 + 8;  // 32-bit words of LDS the tail's guide pass takes: the edge histogram (16-bit counters) + 4 wave totals
static_assert((GUIDE_BINS & (GUIDE_BINS - 1)) == 0 && GUIDE_BINS >= 2048, "bins: a power of two, eight or more per chunk");
static_assert(GUIDE_UNIT == 16 || GUIDE_UNIT == 8 || GUIDE_UNIT == 4, "units: whole chunks, halves or quarters");

// grouped tail (tail_group.hpp): one wave per 256-slot group while the whole grid is resident - up to this many 4096-slot blocks
// (128 four-wave workgroups a 32 blocks: 512 workgroups on 256 CUs); larger sets keep one workgroup per block (they fill the chip)
constexpr int TAIL_GROUP_MAX_BLOCKS = 128;
constexpr size_t TAIL_GROUP_BLOCK_BYTES = 2 * 16 * 16 * 8;  // TG_ROUNDS x 16 groups x one 128-byte record

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- launchers (each enqueues on ctx->stream and returns a status) --------------------------------
// score.hip
int launch_row_norms(midas_ctx* ctx, int64_t K, int32_t D, const void* emb, int32_t dtype, double* norms);
int launch_score(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes, double* scores);
int launch_score_batch(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes, double* scores);

// particles.hip
int launch_se3_feature(midas_ctx* ctx, int64_t N, const float* poses, float w, float* feat6);
int launch_propagate(midas_ctx* ctx, int64_t N, const float* in, float* out, const float* odom,
                     const float* tn, const float* rot, float std_t, float std_r, uint64_t seed, uint64_t step);
int launch_check_poses(midas_ctx* ctx, int64_t N, const float* poses, uint8_t* flag, int32_t* count);
int launch_nn6(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, const int32_t* hint,
               int32_t* idx, float* d2);
int launch_nn3(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* poses, double* dist);
int launch_nn6_stats(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, const int32_t* hint,
                     int32_t* leaves, int32_t* nodes);
int launch_rmse(midas_ctx* ctx, int64_t N, const float* poses, const float* gt16, double* out2);
// Resample of the PREVIOUS frame folded into the front kernel (k_frame_front): slot n of this frame is particle
// src(n) of the previous one - a per-slot dependence - so the search and the gather of the resampler run as a
// prologue of the particle update and the resampled poses never travel through HBM.  The tables are the ones
// k_tail_a2 wrote for the previous frame; src is exactly what k_tail_b2 computes (same predicates).
constexpr int LAZY_MAX_BLOCKS = 256;  // 1 M particles
struct LazyResample {
    bool enabled = false;
    const double *e, *x_raw, *lp, *lp_raw, *gend, *gend_raw;       // [N] x4, [ng] x2
    const double *ggend, *ggend_raw;                                // [16 nb] x2
    const guide_t *guide = nullptr, *guide_raw = nullptr;           // nullable [nb x GUIDE_STRIDE] x2 (see GUIDE_BINS)
    const double *bsum_e, *btot, *btot_raw, *bmax, *bmin;           // [nb] each
    const float* poses_prev;                                         // [N x 16] propagated poses of the previous frame
    const int32_t* nn_prev;                                          // [N]
    const int32_t* status_prev;                                      // [2]
    int32_t* ridx_out;                                               // nullable [N]
    int nb, ng;
    int32_t softmax, mode;
    const double* u;                                                 // nullable: the previous frame's uniforms
    float u32;
    uint64_t seed, step;                                             // Philox key / counter of the previous frame's draws
    int64_t tstride = 0;      // batch of trajectories: doubles between two trajectories' table blocks
    int64_t key_base = 0;     // Philox key of this trajectory's slot 0 (trajectory * N), set per trajectory in the kernel
    int32_t traj = 0;         // trajectory (the systematic draw's key is seed + trajectory, as in k_tail_b2)
};

// sparse scoring inside the particle kernels (score_body.hpp score_claimed_rows)
struct SparseScore {
    uint32_t* stamps = nullptr;   // [K] epoch of the last frame that scored row k (nullptr: dense scoring)
    uint32_t epoch = 0;
    const float* emb = nullptr;
    const double* norms = nullptr;
    const double* code = nullptr;
    double* scores = nullptr;
    int nj = 0;                   // D / 64
    // prediction list (single-trajectory fused front): the rows the previous frame used, listed and stamped `pred_tag`
    // (= epoch - 1) by that frame's tail kernel; scored by streaming waves of the front launch (score_list_wave)
    uint32_t pred_tag = 0;              // 0: no prediction in this launch (every stamp other than `epoch` is stale)
    const int32_t* list = nullptr;      // [list_cap]
    const int32_t* list_count = nullptr;
    int32_t* next_count = nullptr;      // the other list's counter, zeroed by the front for this frame's tail
    int32_t list_cap = 0;
    // dense_thr > 0: a frame whose list holds more rows than this scores ALL rows with the streaming waves instead (K rows in one
    // coalesced stream cost less than a few thousand scattered ones plus the claims of the rows the list missed - the frames after
    // a wide start); the particle waves then only mark the rows they use (for the next frame's list) and score nothing
    int32_t dense_thr = 0;
    int64_t K = 0;
};
// what the tail kernel needs to build the next frame's list: rows whose stamp is `epoch` (claimed or confirmed in this
// frame) are appended to `list` and re-stamped epoch + 1, the next frame's pred_tag (its epoch is epoch + 2).
// Further chances (round 5): a row that was on this frame's list but that nobody needed (stamp still epoch - 1, whatever its age
// bits) is listed again, up to MIDAS_PRED_CHANCES (3) frames in a row, stamped (epoch + 1) | age << 30 - in the frames after a wide start half of the rows a frame needs and the frame before
// did not were in use two frames back (the cloud's fringe flickers: tools/diag_flicker.py), and every such row was a claim: a
// stamp exchange and a cold 2 KB fetch inside a particle wave.  Epochs stay below 2^30 (MIDAS_EPOCH_LIMIT), bits 31:30 count the
// frames a listed row went unused (MIDAS_PRED_CHANCES of them are allowed).
#ifndef MIDAS_PRED_CHANCES
#define MIDAS_PRED_CHANCES 3  // further frames a listed row stays on the list unused (0 .. 3)
#endif
constexpr uint32_t PRED_SECOND = 0xC0000000u;  // bits 31:30 of a listed row's stamp: frames it has been listed without being used
constexpr uint32_t PRED_AGE1 = 0x40000000u;
constexpr uint32_t MIDAS_EPOCH_LIMIT = 0x3FFFFFF0u;
struct ScorePredict {
    uint32_t* stamps = nullptr;
    uint32_t epoch = 0;
    int64_t K = 0;
    int32_t* list = nullptr;
    int32_t* count = nullptr;
};

// Where a front kernel takes its particles from when the unpack of the previous frame is folded into it (midas_shard_run):
// rows == nullptr: the particle arrays.
struct PeerInboxSrc {
    const char* rows = nullptr;        // this rank's inbox
    long long flag_off = 0;            // bytes from the inbox to its flag block
    unsigned long long tag = 0;        // the frame tag the rows must carry
    int G = 0, rank = 0;
    char* const* peers = nullptr;      // non-null: the launch publishes this rank's flag (its first wave) before anybody waits
    int32_t* status = nullptr;         // bit 16 when a flag did not arrive within the bound
};

struct ParticleUpdateArgs {
    int64_t N;
    const float* poses_in;
    float* poses_prop;
    const float* odom16;
    const float* tn;
    const float* rot;
    float std_t, std_r;
    uint64_t seed, step;
    const int32_t* n_live = nullptr;  // nullable: the live particle count in device memory (loop engine); N is then
                                      // the capacity the grid was sized for and slots >= *n_live are skipped
    int64_t slot_base = 0;  // global index of local particle 0 (Philox key)
    int32_t batch = 1;      // trajectories (grid.y); per-trajectory arrays are (batch, ...) contiguous
    int64_t score_stride = 0;  // K: scores are (batch, K)
    const int32_t* hint_in;
    int32_t* nn_idx;
    const double* scores;  // [K]
    double* x;             // [N] gathered score
    double* e;             // [N] exp(x - 1): softmax numerator with the constant shift 1 (scores are cosines)
    uint8_t* valid;        // [N] prune mask
    double t2;             // squared prune threshold (exact: sqrt(d2) > thr  <=>  d2 > t2)
    double thr;            // the threshold itself (triangle-inequality tests of the vertex lists)
    const MeshRec* vlist;  // nullable: per-codebook-entry mesh vertex lists
    const MeshScr* vscr = nullptr;  // nullable: their float32 screening copy (only read when vlist is set)
    MeshField field;                // d != nullptr: the mesh's distance field decides the prune wherever it can (see MeshField)
    double* part_max;      // [nblocks]
    double* part_min;      // [nblocks]
    const float* gt16;     // nullable
    double* part_rmse;     // [2*nblocks] when gt16
    int ablate = 0;        // profiling only (MIDAS_ABLATE)
    unsigned long long* telemetry = nullptr;  // nullable: cumulative [NN tree searches, mesh tree searches]
    int32_t* status_reset = nullptr;          // nullable: status[0..1] zeroed here for the tail kernels' atomics
    double* flags_reset = nullptr;            // nullable: two float64 counters zeroed here (sharded exchange record)
    LazyResample rs;                          // fused front only
    SparseScore sp;                           // stamps != nullptr: only the rows that are some particle's nearest entry are scored
    PeerInboxSrc inbox;                       // rows != nullptr: particles come from the rank's inbox (unpack folded in; forms without folded resample)
    // presorted form of the folded resample (k_presort_search / k_presort_group in front of the launch): the waves take their
    // particles in an order that puts slots with the same nearest-entry hint side by side, and find the slot's source ready
    const int32_t* pre_order = nullptr;       // [batch x N] rank -> slot
    const int32_t* pre_src = nullptr;         // [batch x N] rank -> source particle of that slot (what lazy_source returns)
    double* pre_rmse_terms = nullptr;         // [batch x N x 2] with gt16: the particles' rmse terms BY SLOT - the per-wave sums the tail
                                              // reads are formed from them in slot order (k_rmse_parts), as the unsorted launch forms them
};
int particle_update_blocks(int64_t N);
bool index_build_on_host();  // MIDAS_HOST_INDEX=1: the host builders of round 1 (checkers of the device builders)
int build_neighbour_graph_device(midas_ctx* ctx, midas_tree* t);
int build_vertex_screen(midas_ctx* ctx, midas_tree* t6);  // t6->vscr from t6->vlist (MIDAS_NO_VSCR=1: none)
int build_vertex_lists_device(midas_ctx* ctx, midas_tree* t6, const midas_tree* t3, const float* cb_poses_dev);
int launch_knn6(midas_ctx* ctx, const midas_tree* t, int64_t N, const float* feat6, int32_t k, int32_t* idx, float* d2);
int launch_frame_front(midas_ctx* ctx, const midas_tree* t6, const midas_tree* t3, const ParticleUpdateArgs& a,
                       const midas_codebook* cb, const double* code, double* scores, bool* launched);
int launch_particle_update(midas_ctx* ctx, const midas_tree* t6, const midas_tree* t3, const ParticleUpdateArgs& a);

// resample.hip
int launch_gather_f64(midas_ctx* ctx, int64_t N, const double* table, const int32_t* idx, double* out);
int launch_softmax(midas_ctx* ctx, int64_t N, const double* x, int32_t softmax, double* w);
int launch_prune(midas_ctx* ctx, int64_t N, double* w, const double* dist, double thr, int32_t* nvalid);
int launch_cdf(midas_ctx* ctx, int64_t N, const double* w, double* cdf, int32_t* status);
int launch_search(midas_ctx* ctx, int64_t N, const double* cdf, int64_t M, int32_t mode, const double* u,
                  float u32, uint64_t seed, uint64_t step, int32_t* idx);
int launch_gather_rows(midas_ctx* ctx, int64_t M, const int32_t* idx, const void* src, void* dst, int32_t row_bytes);
// fused tail of the step: x,valid,partials -> weights (masked) -> cdf -> search -> gather
struct StepTailArgs {
    int32_t batch = 1;       // trajectories (grid.y)
    int64_t N;
    int npart;               // number of part_max/part_min entries
    const double* x;
    double* e;               // exp(x - 1) from the particle update (replaced by x when the softmax is skipped)
    const uint8_t* valid;
    const double* part_max;
    const double* part_min;
    int32_t softmax;
    double* weights;         // [N] masked weights (pre-resample)
    double* cdf;             // [N] scratch
    int32_t* status;         // [2]
    int32_t mode;
    const double* u;
    float u32;
    uint64_t seed, step;
    int32_t* ridx;
    const float* poses_prop;
    float* poses_out;
    double* weights_out;
    const int32_t* nn_idx;
    int32_t* hint_out;
    const double* part_rmse;  // nullable
    double* rmse_out;
    // deferred mode (x == nullptr): the particle update ran without the scores (concurrently with the scoring);
    // the tail gathers x = scores[nn_idx], takes the exponentials and decides the isclose guard itself
    const double* scores = nullptr;
    double* x_raw = nullptr;   // [N] scratch: raw scores, written only where the guard may fire
    double* lp_raw = nullptr;  // [N] scratch: block-local prefix of x*valid, likewise
    int64_t score_stride = 0;  // K: scores are (batch, K)
    int64_t tstride = 0;       // pipelined batch: doubles between two trajectories' table blocks (tables_of layout per trajectory)
};
int launch_step_tail(midas_ctx* ctx, const StepTailArgs& a, int prof_slot_base);
// the deferred tail on explicit tables (what k_tail_a2 writes and k_tail_b2 / the lazy front read)
struct TailTables {
    double *e, *x_raw, *lp, *lp_raw, *gend, *gend_raw;       // [N] x4, [ceil(N/16)] x2
    double *bsum_e, *btot, *btot_raw, *bmax, *bmin;           // [ceil(N/4096)] each
    double *ggend, *ggend_raw;                                // [16 ceil(N/4096)]: block-local prefix at the end of each 256-slot group
    guide_t *guide = nullptr, *guide_raw = nullptr;           // nullable [ceil(N/4096) x GUIDE_STRIDE] each (see GUIDE_BINS)
};
int launch_tail_a2(midas_ctx* ctx, int64_t N, const double* scores, const int32_t* nn_idx, const uint8_t* valid,
                   int32_t softmax, const TailTables& tb, int32_t* status, int batch = 1, int64_t score_stride = 0,
                   bool padded_tables = false, const double* part_rmse = nullptr, double* rmse_out = nullptr, int64_t tstride = 0,
                   const ScorePredict* predict = nullptr);  // padded: per-slot tables hold a multiple of 16 values (tables_of, api.hip)
int launch_predict_seed(midas_ctx* ctx, int64_t N, const int32_t* idx, const ScorePredict& pr);
int launch_tail_b2(midas_ctx* ctx, const StepTailArgs& a, const TailTables& tb);  // a.x, a.e, a.cdf, a.lp_raw unused
int launch_shard_tail_a(midas_ctx* ctx, int64_t N, const double* scores, const int32_t* nn_idx, const uint8_t* valid,
                        int32_t softmax, const TailTables& tb, double* r1, int32_t* status, const double* part_rmse = nullptr,
                        const ScorePredict* predict = nullptr);
// sync (peer form, C-side frame): the route kernel's last workgroup publishes the frame's completion flag and waits for every
// rank's (the word behind the 64 flags of the own inbox counts the finished workgroups: zero between launches)
struct PeerRouteSync { const char* inbox; long long flag_off; unsigned long long tag; };
int launch_shard_route(midas_ctx* ctx, const midas_shard_route_args& r, const TailTables& tb, bool pack, const PeerRouteSync* sync = nullptr);
int launch_shard_unpack(midas_ctx* ctx, int64_t N, const void* recv, int32_t* ridx, float* poses_out, double* weights_out,
                        int32_t* hint_out, int32_t dest = -1);
int launch_shard_unpack_peer(midas_ctx* ctx, int64_t N, const void* inbox, int32_t* ridx, float* poses_out, double* weights_out,
                             int32_t* hint_out);
int launch_peer_probe(midas_ctx* ctx, void* const* peers, const void* inbox, int G, int rank, int nonce, int32_t* ok);
int launch_peer_flag_write(midas_ctx* ctx, void* const* peers, int G, int rank, int64_t flag_off, uint64_t tag);
int launch_shard_unpack_peer_wait(midas_ctx* ctx, int64_t N, const void* inbox, int32_t* ridx, float* poses_out, double* weights_out,
                                  int32_t* hint_out, int G, int64_t flag_off, uint64_t tag, int32_t* status, void* const* peers, int rank);
int launch_selftest_wave_sums(midas_ctx* ctx, const double* in64, double* out256);
int debug_tb2_clocks(long long* out16);
int debug_ta_clocks(long long* out16);
int debug_tg_clocks(long long* io64, int reset);
int debug_tg_waves(long long* out4096);
int debug_ff_clocks(long long* io8192, int reset);
int launch_tail_a(midas_ctx* ctx, int64_t N, const double* x, const uint8_t* valid, int np, int pstride,
                  const double* pmax_all, const double* pmin_all, int32_t softmax, double* e_io, double* lp_out,
                  double* block_sums_e, double* block_totals_em, double* flags_out, int32_t* flag, int32_t* status,
                  int batch = 1);
int launch_tail_fin(midas_ctx* ctx, int64_t N, const double* e, const double* x_raw, const double* lp, const double* lp_raw,
                    const uint8_t* valid, double* weights, double* cdf_io, int G, int nb, const double* r1_all, int rank,
                    double n_total, int32_t softmax, double* rmse_out, int32_t* status);
int launch_tail_resample(midas_ctx* ctx, const midas_tail_resample_args& r);
int launch_reduce_partials(midas_ctx* ctx, int np, const double* pmax, const double* pmin, const double* prm,
                           double* extrema2, double* rmse_sums2);

// cluster.hip
int launch_cluster_centers(midas_ctx* ctx, int64_t N, const float* poses, const double* w64, const float* w32,
                           const int64_t* labels, int32_t C, const int64_t* label_values, float* centers, float* stds,
                           int64_t* counts);

struct LoopWeightsArgs;  // loop_weights.hpp
int launch_loop_cluster(midas_ctx* ctx, int64_t cap, const int32_t* ctl_i, const float* poses, const double* w64,
                        const int32_t* labels, double* part, float* centers, float* stds, int64_t* counts, double* rot,
                        const LoopWeightsArgs* weights = nullptr);  // weights: computed at the head of the moment launch (loop.hip)

// loop.hip / dbscan.hip - the reference's whole loop body on a variable-size particle set (midas_loop_step)
int launch_loop_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* t6, const midas_tree* t3,
                     const midas_loop_args& a, int32_t phases);
// src[0 .. n_set): the annealed particle set as indices - mode 1: the N particles minus the k of smallest weight, in
// order; mode 2: all N followed by the k of largest weight, best first; ties to the smaller index
int launch_anneal_select(midas_ctx* ctx, int64_t N, const double* w, int32_t mode, int64_t k, int32_t ties, int32_t* src, int32_t* info);
int launch_topk_aten(midas_ctx* ctx, int64_t cap, const int32_t* ci, const double* w, int32_t* src, int32_t* info);
// labels_out[i] in [-1, ncl) for the n = *n_dev (or N when n_dev is null) poses; min_samples < 0 -> n / 5 (cluster_particles);
// ncl_out[0] = number of clusters; err_out (nullable) |= 2 when the grid / cluster limits were exceeded
int launch_dbscan_points(midas_ctx* ctx, int64_t N, int32_t dim, const double* pts, double eps, int64_t min_samples, int32_t* labels, int32_t* info);
int launch_dbscan(midas_ctx* ctx, int64_t cap, const int32_t* n_dev, const float* poses, double eps, int64_t min_samples,
                  int32_t* labels_out, int32_t* ncl_out, int32_t* err_out, int32_t max_clusters);  // max_clusters 0: any number

// mt19937.hip - torch's CPU generator stream on the device
int launch_mt_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state);
int launch_mt_rand64(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t N, double* out, uint32_t* hist);
int launch_mt_rand64_chunked(midas_ctx* ctx, uint32_t* state, int64_t N, double* out, uint32_t* hist, const uint32_t* polys, int32_t G);
int launch_mt_draws(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int32_t nseg, const midas_mt_segment* segs, const float* R,
                    const float* C, const float* S, uint32_t* hist, const uint32_t* polys, int32_t G);
int launch_mt_normal32(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t numel, float mean, float std, const float* R, const float* C,
                       const float* S, float* out, uint32_t* hist, const uint32_t* polys, int32_t G);

// topn.hip
int launch_topn_pose_error(midas_ctx* ctx, int32_t B, int64_t K, const double* scores, int64_t row0, int32_t n,
                           const double* feat, int32_t d, double* err_out, int32_t* idx_out);

int launch_topn_rinv(midas_ctx* ctx, int64_t K, int64_t ld, const double* norms, float* rinv);
int launch_topn_pose_error_dots(midas_ctx* ctx, int32_t B, int64_t K, const float* panel, int64_t ld, const double* norms, const float* rinv,
                                int64_t row0, int32_t n, const double* feat, int32_t d, double* err_out, int32_t* idx_out);
// selfsim.hip
int launch_selfsim_panel(midas_ctx* ctx, const midas_codebook* cb, int64_t i0, int64_t R, float* panel, int64_t ldo);

// profiling hook used by the step: record event `slot` on the stream when profiling is on
void prof_mark(midas_ctx* ctx, int slot);

}  // namespace midas
