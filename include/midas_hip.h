/*
 * midas_hip.h - C ABI of libmidas_hip.so, the MI355X (gfx950) implementation of the MidasTouch
 * particle-filter hot path.
 *
 * The reference (facebookresearch/MidasTouch) has NO FFI for this path: it is reached by direct
 * Python method calls on two classes from the Hydra runner (midastouch/filter/filter.py:82,89-93,
 * 155,159-160,170-173,176,183-190).  This header is therefore the boundary a maintainer binds with
 * ctypes underneath those classes; each entry point cites the reference
 * call site whose arithmetic it replaces (paths relative to /root/reference/midastouch).
 *
 * Conventions
 *  - every pointer named *_dev is a DEVICE pointer owned by the caller (e.g. torch tensor data_ptr());
 *    the library allocates only its own handles and scratch;
 *  - every call is asynchronous on the context's HIP stream unless stated otherwise;
 *  - every function returns an int status: 0 = MIDAS_OK, < 0 = error (midas_strerror); nothing throws
 *    across the boundary; midas_last_error(ctx) holds the detailed text of the last failure;
 *  - a context is not thread-safe; use one context per thread / stream.  Calls may come from any
 *    host thread (the reference runs its filter on a worker thread, filter/filter.py:269-273).
 */
#ifndef MIDAS_HIP_H
#define MIDAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIDAS_OK 0
#define MIDAS_ERR_INVALID (-1)  /* bad argument */
#define MIDAS_ERR_HIP (-2)      /* a HIP runtime call failed */
#define MIDAS_ERR_NOMEM (-3)    /* device allocation failed */
#define MIDAS_ERR_NODEVICE (-4) /* no usable gfx950 device */

#define MIDAS_F32 0
#define MIDAS_F64 1

#define MIDAS_RESAMPLE_MULTINOMIAL 0 /* "weighted_random" (default), modules/particle_filter.py:243-249 */
#define MIDAS_RESAMPLE_SYSTEMATIC 1  /* "low_var" / "low_var_batch", modules/particle_filter.py:251-307 */

typedef struct midas_ctx midas_ctx;
typedef struct midas_codebook midas_codebook;
typedef struct midas_tree midas_tree;

/* ---- context --------------------------------------------------------------------------------- */
/* `hip_stream` is a hipStream_t; NULL = the device's default (null) stream, which is what torch's
 * default stream is on ROCm.  The library never creates streams of its own. */
int midas_ctx_create(int device, void* hip_stream, midas_ctx** out);
int midas_ctx_destroy(midas_ctx* ctx);
int midas_ctx_set_stream(midas_ctx* ctx, void* hip_stream);
int midas_sync(midas_ctx* ctx); /* hipStreamSynchronize */
/* The library's scratch (per-call work arrays: DBSCAN's cell tables, selection histograms, the batch step's score panels) is a
 * bump allocator over library-owned chunks that GROWS when a call needs more than any call before it - a hipMalloc in the middle
 * of a run (milliseconds: the cold frame of a loop whose 50th frame is the first to cluster).  A caller that knows its capacity
 * reserves once, up front: afterwards no call below that need allocates (MIDAS_SCRATCH_LOG=1 reports every allocation).
 * Synchronises the context's stream when it has to replace chunks.  No counterpart in the reference (torch's caching allocator
 * plays this role for modules/particle_filter.py's temporaries). */
int midas_scratch_reserve(midas_ctx* ctx, int64_t bytes);
/* Self-test of the float64 sums whose ORDER is part of the arithmetic spec (the wave's xor butterfly 32 .. 1 and the 16-lane row's
 * 8 .. 1: get_similarity's dot products and the CDF's block sums, modules/particle_filter.py:449-469, :237-252, restated in
 * oracle/midas_oracle.c): the kernels form them with register moves (csrc/midas_math.hpp); this entry returns, for 64 doubles in,
 * per lane {butterfly wave sum, register-move wave sum, butterfly row sum, register-move row sum} (256 doubles out), enqueued on the
 * context's stream.  tests/test_gpu_sums.py compares the pairs bit for bit. */
int midas_selftest_wave_sums(midas_ctx* ctx, const double* in64_dev, double* out256_dev);
const char* midas_strerror(int code);
const char* midas_last_error(const midas_ctx* ctx);
const char* midas_version(void);

/* ---- codebook: embeddings + cosine scores  (K1) ---------------------------------------------- */
/* Wraps (does not copy) a K x D row-major embedding matrix and precomputes max(|C_k|, 1e-8).
 * Replaces tactile_tree.embeddings (tactile_tree/tactile_tree.py:17-19,29-32). dtype MIDAS_F32/F64. */
int midas_codebook_create(midas_ctx* ctx, int64_t K, int32_t D, const void* emb_dev, int32_t dtype,
                          midas_codebook** out);
int midas_codebook_destroy(midas_codebook* cb);
/* scores[b*K + k] = cos(codes[b], C_k), float64.  One pass over the codebook per call.
 * Replaces cosine_similarity over gathered rows (modules/particle_filter.py:455-457) and the
 * heat-map call (filter/filter.py:213-215).  B >= 1 tactile codes (B*D doubles). */
int midas_score(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes_dev,
                double* scores_dev);

/* Batched form for B concurrent trajectories (BASELINE config 5): one pass over the codebook on the matrix
 * cores (v_mfma_f32_16x16x4_f32).  The codes are rounded to float32 (they are float32 network outputs) and
 * the dot products are float32 fma chains in a fixed order, then divided by the float64 norms:
 * scores agree with midas_score to ~1e-7.  Needs float32 embeddings and D % 16 == 0.  The 64 codes of a pass sit in LDS whole
 * up to D = 636 (one persistent workgroup per CU); beyond, the waves read the float32 code rows from memory - same arithmetic. */
int midas_score_batch(midas_ctx* ctx, const midas_codebook* cb, int32_t B, const double* codes_dev,
                      double* scores_dev);

/* ---- SE(3) feature and exact nearest neighbour  (K3, K4) ------------------------------------- */
/* feat6 = [ (1-w) t , w log(R) ]  - R3_SE3 (tactile_tree/tactile_tree.py:73-77, modules/pose.py:19-23) */
int midas_se3_feature(midas_ctx* ctx, int64_t N, const float* poses_dev, float w, float* feat6_dev);
/* Static KD-tree over K points: dim 6 -> float32 (codebook features, replaces pynanoflann
 * tactile_tree.init_tree, tactile_tree/tactile_tree.py:34-41); dim 3 -> float64 (mesh vertices, replaces
 * sklearn KDTree, modules/particle_filter.py:108-110).  Synchronous (built on the host once). */
int midas_tree_build(midas_ctx* ctx, int32_t dim, int64_t K, const void* points_dev, midas_tree** out);
int midas_tree_destroy(midas_tree* tree);
/* Optional accelerator of the fused step's prune: for every codebook entry, the mesh vertices nearest to
 * its translation (cb_poses_dev: K x 16 float32), so that most particles are classified from the list of
 * their NN entry with the same exact predicate and only the rest search the mesh tree.  Synchronous. */
int midas_tree_attach_mesh(midas_ctx* ctx, midas_tree* tree6, const midas_tree* tree3,
                           const float* cb_poses_dev);
/* Test / inspection hook: copies one of the tree's per-entry lists to host memory.  what = 0 neighbour records
 * (K x 513 x 32 B), 1 rho_out (K floats), 2 twin (K int32), 3 mesh-vertex records (K x 257 x 32 B, after
 * midas_tree_attach_mesh); bytes must be the array's size.  The lists are built on the device (index_build.hip; brute
 * force, float64 distances, (distance, index) order) - MIDAS_HOST_INDEX=1 selects the host builder they are checked against.
 * No reference counterpart (pynanoflann / sklearn trees: tactile_tree/tactile_tree.py:34-41, modules/particle_filter.py:108-110). */
int midas_tree_export(midas_ctx* ctx, const midas_tree* tree, int32_t what, void* dst_host, int64_t bytes);
/* idx[n] = argmin_k |feat6[n] - F_k|^2 (ties -> smallest k); hint_dev (nullable) = a candidate index
 * per query that seeds the search bound; d2_dev nullable.  Replaces kneighbors (tactile_tree.py:50-52). */
int midas_nn6(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev,
              const int32_t* hint_dev, int32_t* idx_dev, float* d2_dev);
/* idx[n*k + r] = the r-th nearest codebook entry of query n by (squared distance, index), exact (brute force over the
 * codebook, one wave per query); 1 <= k <= 64, k <= K; d2_dev nullable.  Replaces kneighbors(n_neighbors = nn) for nn > 1
 * (tactile_tree/tactile_tree.py:43-58; the filter itself only asks for nn = 1). */
int midas_knn6(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev, int32_t k,
               int32_t* idx_dev, float* d2_dev);
/* Diagnostic twin of midas_nn6: leaves / tree nodes visited per query (used to tune the tree). */
int midas_nn6_stats(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* feat6_dev,
                    const int32_t* hint_dev, int32_t* leaves_dev, int32_t* nodes_dev);
/* dist[n] = float64 distance from pose n's translation to the nearest vertex.
 * Replaces mesh_kdtree.query (modules/particle_filter.py:386-392). */
int midas_nn3(midas_ctx* ctx, const midas_tree* tree, int64_t N, const float* poses_dev, double* dist_dev);

/* ---- motion model  (K2) ---------------------------------------------------------------------- */
/* poses_out[n] = poses_in[n] @ (odom @ Tn(tn[n], Rz Ry Rx(deg2rad(rot[n])))).
 * tn_dev/rot_dev (N x 3 float32 each) are the already-scaled host draws of add_noise_to_odom
 * (modules/particle_filter.py:326-335); when both are NULL the kernel draws them itself from the
 * Philox spec streams keyed by (seed, step) and scales by std_t / std_r.
 * Replaces add_noise_to_odom + the compose in motionModel (modules/particle_filter.py:319-345,370-375). */
int midas_propagate(midas_ctx* ctx, int64_t N, const float* poses_in_dev, float* poses_out_dev,
                    const float* odom16_dev, const float* tn_dev, const float* rot_dev, float std_t,
                    float std_r, uint64_t seed, uint64_t step);
/* flag[n] = 1 when pose n's rotation gives a NaN / zero-norm quaternion (check_quats,
 * modules/particle_filter.py:347-357); count_dev[0] = number of flagged particles. */
int midas_check_poses(midas_ctx* ctx, int64_t N, const float* poses_dev, uint8_t* flag_dev,
                      int32_t* count_dev);

/* ---- weights  (K5) --------------------------------------------------------------------------- */
/* out[n] = table[idx[n]] (float64): x_n = scores[nn_idx[n]] */
int midas_gather_f64(midas_ctx* ctx, int64_t N, const double* table_dev, const int32_t* idx_dev,
                     double* out_dev);
/* get_similarity tail (modules/particle_filter.py:459-468): if softmax and |max-min| > 1e-8,
 * w = exp(x-max)/sum exp(x-max); else w = x.  In place allowed. */
int midas_softmax(midas_ctx* ctx, int64_t N, const double* x_dev, int32_t softmax, double* w_dev);
/* remove_invalid_particles tail (modules/particle_filter.py:394-402): w[n] *= !(dist[n] > thr);
 * nvalid_dev[0] = number of particles kept (0 => "drifted"). */
int midas_prune(midas_ctx* ctx, int64_t N, double* w_dev, const double* dist_dev, double thr,
                int32_t* nvalid_dev);

/* ---- torch's CPU random stream on the device ---------------------------------------------------- */
/* The reference draws on torch's default CPU generator (at::mt19937).  Under torch.manual_seed(s) the resampler's
 * WeightedRandomSampler / torch.multinomial(weights.double(), N, True) (modules/particle_filter.py:245) consumes two 32-bit
 * outputs per sample - the stream of torch.rand(N, dtype=float64) - and each torch.normal of add_noise_to_odom (:326-335)
 * one output per float32 value (+ 16 when the size is not a multiple of 16).  These two calls reproduce that stream from a
 * generator state kept in device memory (state_dev: 626 uint32, caller-owned), so "bit-exact resample indices under a
 * fixed seed" needs neither a host generator nor a per-frame upload of uniforms.
 * midas_mt19937_seed: the state of torch.manual_seed(seed) (low 32 bits, as at::mt19937 takes them).
 * midas_mt19937_rand64: discards skip_words 32-bit outputs, then writes the next N float64 uniforms
 * ((hi << 32 | lo) & (2^53 - 1)) * 2^-53 to out_dev (NULL with N == 0: skip only), and leaves the state advanced.
 * One workgroup walks the 624-word blocks (one barrier a block, ~310 ns), a second kernel tempers and converts; the call uses
 * (2 N + 1872) x 4 bytes of the context's scratch.  A caller that wants it beside other work gives it a context / stream of its
 * own (midastouch_amd/torch_rng.py does). */
int midas_mt19937_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state_dev);
int midas_mt19937_rand64(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t N, double* out_dev);
/* The same stream (same reference call: torch.multinomial's draws, modules/particle_filter.py:245) with the call's 2 N words
 * generated in `pieces` pieces side by side.  mt19937 is linear over GF(2): x[k + J] = XOR of x[k + i] over the exponents i of
 * t^J mod phi(t) (phi: the generator's characteristic polynomial, degree 19937), so the 624 words that start a piece follow
 * from MIDAS_MT19937_HIST_WORDS consecutive words generated earlier and one polynomial per piece:
 *   hist_dev  (in/out) MIDAS_MT19937_HIST_WORDS uint32: on entry the first such words the PREVIOUS call handed out (raw,
 *             untempered - every call with 2 N >= MIDAS_MT19937_HIST_WORDS leaves them), on exit this call's;
 *   polys_dev pieces x 624 uint32: bit b of word w of polynomial c = coefficient of t^(32 w + b) of t^(J_c) mod phi, J_c = distance
 *             in words from hist_dev's first word to the first word of piece c = (words the previous call handed out) + skip_words
 *             + c x 624 x ceil(ceil(2 N / 624) / pieces)  (host-side set-up: midastouch_amd/mt_jump.py; a wrong table gives wrong
 *             numbers, nothing else - tests/test_torch_stream.py holds the chunked stream against torch.rand);
 * polys_dev NULL or pieces <= 0: the sequential walk of midas_mt19937_rand64, which also leaves hist_dev (if given).
 * state_dev is left as after midas_mt19937_rand64.  Needs 2 N >= MIDAS_MT19937_HIST_WORDS. */
/* torch.normal(mean, std, size) of `numel` float32 values (numel >= 16) from the same stream - the motion noise of
 * add_noise_to_odom (modules/particle_filter.py:326-335: two calls of 3 N values per frame, in front of the resampler's draws).
 * ATen turns float32 uniforms ((w & 0xFFFFFF) 2^-24, one generator output each) into normals sixteen at a time (Box-Muller:
 * radius(u1) cos / sin(theta(u2)), times std plus mean by one fused multiply-add) and draws the last sixteen again when numel is
 * not a multiple of 16.  radius_dev / cos_dev / sin_dev: the three functions as tables over the 2^24 uniforms (float32[2^24] each),
 * read off torch.normal itself by the host (midastouch_amd/torch_normal.py) - whatever math library ATen uses on the machine, the
 * values are torch.normal's bit for bit (tests/test_torch_stream.py).  Consumes numel (+ 16) outputs behind skip_words.
 * hist_dev / polys_dev / pieces: as midas_mt19937_rand64_chunked (J_c counts this call's words: numel (+ 16)); NULL / 0: sequential. */
int midas_mt19937_normal32(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t numel, float mean, float std,
                           const float* radius_dev, const float* cos_dev, const float* sin_dev, float* out_dev, uint32_t* hist_dev,
                           const uint32_t* polys_dev, int32_t pieces);
#define MIDAS_MT19937_HIST_WORDS 20560
int midas_mt19937_rand64_chunked(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int64_t N, double* out_dev,
                                 uint32_t* hist_dev, const uint32_t* polys_dev, int32_t pieces);
/* Several consecutive draws of the stream by ONE walk of the generator - a seeded frame of the reference takes torch.normal (N, 3)
 * twice (add_noise_to_odom, modules/particle_filter.py:326-335) and N float64 uniforms (the resampler, :245): `segs` (host array of
 * nseg <= 8 entries, in the stream's order) names each draw - MIDAS_MT_SEGMENT_RAND64: count values as midas_mt19937_rand64 into
 * out_dev (double); MIDAS_MT_SEGMENT_NORMAL32: count (>= 16) values as midas_mt19937_normal32 with mean / std into out_dev (float;
 * the three tables are then required).  The results are those of the separate calls, number for number.  hist_dev / polys_dev /
 * pieces as midas_mt19937_rand64_chunked, J_c counting the words of ALL segments (2 per float64, count (+ 16 when count is not a
 * multiple of 16) per normal draw); in pieces the sum must be >= MIDAS_MT19937_HIST_WORDS.  NULL / 0: the sequential walk, which
 * leaves the history (when hist_dev is given and the sum is long enough) for a following call in pieces. */
#define MIDAS_MT_SEGMENT_RAND64 0
#define MIDAS_MT_SEGMENT_NORMAL32 1
typedef struct midas_mt_segment {
    int32_t kind;
    int32_t pad_;
    int64_t count;
    float mean, std;
    void* out_dev;
} midas_mt_segment;
int midas_mt19937_draws(midas_ctx* ctx, uint32_t* state_dev, int64_t skip_words, int32_t nseg, const midas_mt_segment* segs,
                        const float* radius_dev, const float* cos_dev, const float* sin_dev, uint32_t* hist_dev,
                        const uint32_t* polys_dev, int32_t pieces);

/* ---- resample  (K6, K7, K8) ------------------------------------------------------------------ */
/* cdf = blocked_prefix(w) / total, cdf[N-1] = 1 (float64, fixed summation order - DESIGN.md).
 * status_dev[0] = 0 ok, 1 all weights zero, 2 NaN present (resampler returns its input unchanged,
 * modules/particle_filter.py:240-241).  Replaces :237-239,252 and the cumsum inside torch.multinomial. */
int midas_cdf(midas_ctx* ctx, int64_t N, const double* w_dev, double* cdf_dev, int32_t* status_dev);
/* idx[i] for M output slots. MULTINOMIAL: first j with cdf[j] >= u[i]; u_dev = M float64 uniforms
 * (NULL -> Philox spec stream (seed, step)).  SYSTEMATIC: first j with cdf[j] > fmod(i/M + u32/M, 1);
 * u32 < 0 -> Philox.  Replaces torch.multinomial / the low_var loop (particle_filter.py:245,295-303). */
int midas_resample_search(midas_ctx* ctx, int64_t N, const double* cdf_dev, int64_t M, int32_t mode,
                          const double* u_dev, float u32, uint64_t seed, uint64_t step,
                          int32_t* idx_dev);
/* dst[i] = src[idx[i]] for rows of row_bytes bytes (poses 64, weights 8, labels 4/8).
 * Replaces the fancy-index gathers (modules/particle_filter.py:246-248, tactile_tree.py:54-58). */
int midas_gather_rows(midas_ctx* ctx, int64_t M, const int32_t* idx_dev, const void* src_dev,
                      void* dst_dev, int32_t row_bytes);

/* ---- metric epilogue ------------------------------------------------------------------------- */
/* out2 = { rmse_t [m], rmse_r [deg] } float64 - particle_rmse (modules/particle_filter.py:472-496). */
int midas_rmse(midas_ctx* ctx, int64_t N, const float* poses_dev, const float* gt16_dev,
               double* out2_dev);

/* ---- cluster centres (K9) -------------------------------------------------------------------- */
/* particle_filter.get_cluster_centers(method="quat_avg") (modules/particle_filter.py:153-206) with pose.xyz_quat_averaged
 * (modules/pose.py:112-147; its removed Tensor.eig is a symmetric 4x4 eigenproblem; its top eigenvector is found here by repeated squaring in float64).
 * For each of the C label values: members = particles carrying it; weights taken as float32 (:161) and flattened to 1
 * when isclose(max - min, 0) (:178-184); centre rotation = principal eigenvector of sum w q q^T / sum w over sign-fixed
 * unit quaternions, centre translation = weighted mean; std = sqrt(sum w (t - centre)^2 / sum w) per axis.
 * weights: float64 (weights64_dev) or float32 (weights32_dev), exactly one non-NULL.  centers_dev: C x 16 float32,
 * stds_dev: C x 3 float32, counts_dev: NULL or C int64 (members per label; a label nobody carries gives NaN rows).
 * 1 <= C <= 64. */
int midas_cluster_centers(midas_ctx* ctx, int64_t N, const float* poses_dev, const double* weights64_dev,
                          const float* weights32_dev, const int64_t* labels_dev, int32_t C, const int64_t* label_values_dev,
                          float* centers_dev, float* stds_dev, int64_t* counts_dev);

/* ---- single-touch evaluation: top-n of similarity rows + best pose error (next-3) ------------- */
/* eval/single_touch_test.py:35-73 top_n_error: for each of the B query rows of scores_dev (B x K float64, row b = the
 * similarities of codebook entry row0 + b to every entry, e.g. from midas_score / midas_score_batch) the diagonal entry is
 * set to 0 (:64), the n best-scoring entries are selected (value descending, smaller index first on ties; np.argpartition
 * leaves ties unspecified) and err_dev[b] = min over them of |feat[j] - feat[row0 + b]|_2 (feat_dev: K x d float64, d <= 16).
 * idx_dev: NULL or B x n int32, the selected entries best first (-1 padded when K < n).  1 <= n <= 256. */
/* The whole top_n_error of a codebook against itself (single_touch_test.py:35-73) by one call: the K x K x D self-similarity
 * as a float32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32: exact float32 fma chains in midas_score_batch's order),
 * in panels of rows_per_panel query rows (>= 128; a panel is rows x K float32 of scratch), each consumed by the selection
 * above (cosine = (double)dot / (|E_i| |E_j|), diagonal 0, n best, best pose error).  float32 embeddings, D % 32 == 0.
 * err_dev: K float64; idx_dev: NULL or K x n int32. */
/* one panel of it: panel_dev[(i - i0) * ldo + j] = <E_i, E_j> (raw float32 dot products) for i in [i0, i0 + R), all j;
 * ldo >= ceil(K / 128) * 128, the panel holds ceil(R / 128) * 128 rows (whole tiles are written) */
int midas_selfsim_panel(midas_ctx* ctx, const midas_codebook* cb, int64_t i0, int64_t R, float* panel_dev, int64_t ldo);
int midas_selfsim_topn(midas_ctx* ctx, const midas_codebook* cb, int32_t n, const double* feat_dev, int32_t d,
                       int64_t rows_per_panel, double* err_dev, int32_t* idx_dev);
int midas_topn_pose_error(midas_ctx* ctx, int32_t B, int64_t K, const double* scores_dev, int64_t row0, int32_t n,
                          const double* feat_dev, int32_t d, double* err_dev, int32_t* idx_dev);

/* ---- the fused per-frame step ---------------------------------------------------------------- */
/* One call = filter/filter.py:150-190 minus clustering/annealing:
 *   score codebook -> propagate -> feature -> NN -> x = s[idx] -> softmax -> prune -> cdf ->
 *   resample search -> gather (poses, weights, hints).  All device-resident, no host sync. */
typedef struct midas_step_args {
    int64_t N;
    const float* poses_in_dev;   /* N x 16 */
    float* poses_prop_dev;       /* N x 16 scratch: propagated, pre-resample poses */
    float* poses_out_dev;        /* N x 16 resampled poses (may alias poses_in_dev) */
    double* weights_dev;         /* N: softmax weights x prune mask, BEFORE resampling */
    double* weights_out_dev;     /* N: the same, gathered by the resample indices */
    const int32_t* hint_in_dev;  /* N or NULL: NN index of each particle's ancestor (search seed) */
    int32_t* nn_idx_dev;         /* N: nearest codebook entry of each propagated particle */
    int32_t* hint_out_dev;       /* N: nn_idx gathered by the resample indices */
    int32_t* ridx_dev;           /* N: resample indices */
    const float* odom16_dev;     /* 16 */
    const double* code_dev;      /* D: tactile code of this frame */
    const float* gt16_dev;       /* 16 or NULL: ground truth pose for the rmse epilogue */
    double* rmse_dev;            /* 2 or NULL */
    const float* tn_dev;         /* parity mode: host draws (see midas_propagate); NULL -> Philox */
    const float* rot_dev;
    const double* u_dev;         /* parity mode: N float64 uniforms; NULL -> Philox */
    float u32;                   /* systematic offset draw; < 0 -> Philox */
    float std_t, std_r;
    uint64_t seed, step;
    double prune_thr;            /* pen_max (config/tdn/default.yaml:18) */
    int32_t softmax;
    int32_t resample_mode;
    int32_t* status_dev;         /* [0] cdf status (see midas_cdf), [1] particles kept by the prune */
    uint64_t* telemetry_dev;     /* NULL or 16 cumulative counters: [0],[1] particles whose NN / prune needed the tree search; [2] codebook rows scored by particle waves (sparse scoring: first particle on a row), [3] rows scored off the prediction list; the rest reserved (profiling builds keep per-wave statistics behind them, MIDAS_ABLATE=4) */
    uint32_t* score_stamps_dev;  /* NULL: every codebook row is scored every frame.  Else K uint32 stamps, zeroed once by the caller, together with
                                    * score_epoch (non-zero, different on every frame that uses these stamps): only the rows that are
                                    * some particle's nearest entry are scored - by the particle kernels themselves, same arithmetic,
                                    * same scores; scores_dev then holds the frame's scores at those rows only */
    uint32_t score_epoch;
} midas_step_args;

int midas_filter_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6,
                      const midas_tree* tree3, const midas_step_args* args);


/* ---- pipelined single-trajectory step ------------------------------------------------------------------------
 * Slot n of frame t+1 is particle src(n) of frame t: the resampler's search and gather (particle_filter.py:230-307)
 * depend on nothing but the slot, so midas_lazy_step folds the resample of the PREVIOUS frame into the front kernel of
 * this one - the resampled poses never travel through HBM and one launch per frame disappears.  A frame is then
 *     midas_lazy_step  = [resample frame t-1 | propagate + NN + prune | score codebook] (one launch) + softmax/CDF tables
 * and the resampled particle set of the last frame exists only implicitly until midas_lazy_flush materialises it
 * (same kernel as the tail of midas_filter_step: identical indices, poses, weights).  After a flush the next
 * midas_lazy_step starts from the materialised particles (resample_prev = 0).  The caller owns every buffer and
 * alternates two sets of the per-frame state.  Requires a float32 codebook with D in {128, 256, 512, 1024} and
 * N <= 1 M; MIDAS_ERR_INVALID otherwise (use midas_filter_step). */
typedef struct midas_lazy_args {
    int64_t N;
    const float* poses_prop_prev_dev;  /* N x 16: propagated poses of the previous frame (resample_prev) */
    const int32_t* nn_idx_prev_dev;    /* N */
    const int32_t* status_prev_dev;    /* 2 */
    float* poses_prop_dev;             /* N x 16 out */
    int32_t* nn_idx_dev;               /* N out */
    uint8_t* valid_dev;                /* N out */
    int32_t* status_dev;               /* 2 out: [0] = 2 on NaN weights, [1] = particles kept by the prune.  [0] bit 4 (value 16): a wave
                                        * of the grouped tail (one wave per 256 slots, the waves of a 4096-slot block hand their sums to
                                        * each other) did not see its block's records within 0.2 s - the frame's tables are UNDEFINED.
                                        * The form is only taken while the whole grid is resident on the device (queried occupancy), so
                                        * this means foreign work held the compute units; MIDAS_TAIL_GROUPED=0 selects the form without
                                        * waits.  midastouch_amd's engines raise on it (`check()`) */
    double* tables_dev;                /* 128-byte aligned, 4 N16 + 2 G16 + 37 ceil(N/4096) doubles with N16 = N and G16 = ceil(N/16),
                                        * each rounded up to a multiple of 16: on entry the previous frame's softmax /
                                        * CDF tables (read when resample_prev), on exit this frame's */
    double* scores_dev;                /* K: this frame's codebook scores */
    double* part_rmse_dev;             /* NULL or 2 ceil(N/64): per-wave rmse sums of this frame (gt16_dev) */
    int32_t resample_prev;             /* 1: particles = resample of the previous frame; 0: poses_in_dev / hint_in_dev */
    const float* poses_in_dev;         /* N x 16 (resample_prev == 0) */
    const int32_t* hint_in_dev;        /* N or NULL */
    int32_t resample_mode;             /* draws of the previous frame's resample: */
    const double* u_prev_dev;          /*   N uniforms or NULL -> Philox(seed, step_prev) */
    float u32_prev;                    /*   systematic offset, < 0 -> Philox */
    uint64_t step_prev;
    int32_t* ridx_dev;                 /* NULL or N out: the previous frame's resample indices (resample_prev) */
    const float* odom16_dev;
    const double* code_dev;
    const float* gt16_dev;             /* NULL or 16 */
    const float* tn_dev;               /* host draws or NULL -> Philox(seed, step) */
    const float* rot_dev;
    float std_t, std_r;
    uint64_t seed, step;
    double prune_thr;
    int32_t softmax;
    uint64_t* telemetry_dev;           /* NULL or 16 cumulative counters (see midas_step_args) */
    uint32_t* score_stamps_dev;        /* sparse scoring, see midas_step_args (midas_lazy_run advances the epoch by one per frame) */
    uint32_t score_epoch;
    double* rmse_dev;                  /* NULL or 3 out (needs gt16_dev, part_rmse_dev): this frame's {rmse_t, rmse_r, device clock
                                        * in us} - particle_rmse is taken on the propagated particles (filter.py:164), so the
                                        * frame's own statistics exist without materialising its resample */
    int32_t* score_list_dev;           /* NULL or 2 + 2 K int32, zero-initialised by the caller: prediction lists of the sparse
                                        * scoring (single trajectory).  [0], [1] = the two lists' lengths, then two lists of K
                                        * rows.  The frame with score_epoch e scores list (e >> 1) & 1 - the rows the frame
                                        * before it used, stamped e - 1 by that frame's tail, and the rows that frame had on its
                                        * list without using them (a listed row stays up to THREE further frames unused - compile knob MIDAS_PRED_CHANCES, 0 .. 3;
                                        * its stamp is (e + 1) | age << 30, bits 31:30 = frames listed without use) - with streaming
                                        * workgroups of its front launch, and its own tail writes the other list.  With a list
                                        * the caller advances score_epoch by TWO per frame (midas_lazy_run does), keeps it below
                                        * 0x3FFFFFF0 and zeroes the two lengths whenever it zeroes the stamps.  Same scores as without (which rows are scored by whom is all
                                        * that changes); replaces nothing in the reference - it gathers all N rows every frame
                                        * (tactile_tree/tactile_tree.py:54-58) */
    uint8_t* guide_dev;                /* NULL or midas_lazy_guide_bytes(N) bytes, 16-byte aligned (single trajectory; ignored by the
                                        * batch form): guide tables of the summation blocks, written by the frame's tail next to
                                        * tables_dev and read by the NEXT frame's folded resample.  Per 4096-slot block `bins`
                                        * equal bins over its masked total W (edge k = fl(k * (W / bins))), a 16-bit entry each:
                                        * min(number of the block's `unit`-slot pieces whose last prefix value lies left of edge k,
                                        * pieces of the block - 1) - the piece a draw falling into bin k starts its search from;
                                        * 0xFFFF in every entry of a block that holds a negative weight (raw scores of mixed sign:
                                        * its prefix values do not rise; the front searches such a block through the table lines)
                                        * (layout: softmax variant | raw-score variant, each ceil(N/4096) rows of `stride`
                                        * entries; midas_lazy_guide_layout).  One entry pair and 8 - 16 prefix values replace the
                                        * three 128-byte table lines of a draw's search; a hint only - the exact comparison on
                                        * (BP + lp_i) / total decides the index as without it (torch.multinomial's lower bound,
                                        * modules/particle_filter.py:245) */
} midas_lazy_args;
int64_t midas_lazy_guide_bytes(int64_t N);
int midas_lazy_guide_layout(int32_t* bins_out, int32_t* unit_out, int32_t* stride_out);
int midas_lazy_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                    const midas_lazy_args* args);
/* The prediction list from scratch, for a particle set that did not come out of a frame (after the projection onto the
 * codebook, filter/filter.py:159-160, where every particle's nearest entry nn_idx_dev[n] is known): lists the distinct rows
 * and tags them for the frame that runs with score_epoch + 2.  score_epoch: an epoch of the caller's sequence (even step
 * of 2) that no frame has used. */
int midas_score_list_seed(midas_ctx* ctx, int64_t K, uint32_t* score_stamps_dev, uint32_t score_epoch, int32_t* score_list_dev,
                          int64_t N, const int32_t* nn_idx_dev);
/* T consecutive frames of midas_lazy_step enqueued by one call (device Philox draws): frame f takes odom16_dev + 16 f,
 * code_dev + D f and (when given) gt16_dev + 16 f, alternates the two buffer sets of `first` (frame 0 uses first's own
 * assignment, frame 1 the swapped one, ...) and folds the resample of the frame before it; rmse_log_dev (NULL or 3 T
 * doubles) receives every frame's {rmse_t, rmse_r, device wall clock in us at the end of the frame}.  The reference's loop body is called once per sensor frame
 * (filter/filter.py:131-233); replaying a recorded sequence needs no host turn-around between frames.  After the call the
 * latest frame's buffers are first's (poses_prop, nn_idx, status) when T is even, its *_prev ones when T is odd.
 * The frame's tail numbers its launches (the waves of a summation block hand their totals to each other through records that
 * carry the launch's number, csrc/tail_group.hpp): the calls of the step family (midas_filter_step, midas_lazy_step / _run,
 * midas_shard_*) must be ISSUED, not replayed from a captured hipGraph - a replay would repeat a number. */
int midas_lazy_run(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                   const midas_lazy_args* first, int32_t T, double* rmse_log_dev);

typedef struct midas_lazy_flush_args {
    int64_t N;
    const double* tables_dev;          /* the frame's tables, valid mask, NN indices, propagated poses (midas_lazy_step) */
    const uint8_t* valid_dev;
    const int32_t* nn_idx_dev;
    const float* poses_prop_dev;
    int32_t* status_dev;               /* [0] completed with the cdf status (see midas_cdf) */
    const double* part_rmse_dev;       /* NULL or the frame's per-wave sums */
    int32_t softmax, resample_mode;
    const double* u_dev;               /* N uniforms or NULL -> Philox(seed, step) */
    float u32;
    uint64_t seed, step;
    double* weights_dev;               /* N out: masked weights (pre-resample) */
    int32_t* ridx_dev;                 /* N out */
    float* poses_out_dev;              /* N x 16 out */
    double* weights_out_dev;           /* N out */
    int32_t* hint_out_dev;             /* N out */
    double* rmse_dev;                  /* NULL or 2 out */
} midas_lazy_flush_args;
int midas_lazy_flush(midas_ctx* ctx, const midas_lazy_flush_args* args);

/* The pipelined step for a BATCH of B independent trajectories against one codebook (BASELINE config 5; the reference runs one
 * trajectory per process, filter/filter.py:150-190 - here they share a launch, trajectory = grid.y).  Same argument structs; every
 * per-trajectory array gains a leading batch dimension, contiguous: poses (B, N, 16), nn_idx / valid / ridx / hint / weights
 * (B, N), status (B, 2), scores and score_stamps (B, K), odom16 / gt16 (B, 16), code (B, D), part_rmse (B, 2 * ceil(N / 64)),
 * rmse (B, 3) for the step [rmse_t, rmse_r, device clock] and (B, 2) for the flush, u (B, N); `tables_dev` holds B blocks of
 * midas_lazy_tables_doubles(N) doubles, 128-byte aligned.  Draws are keyed per trajectory as in midas_filter_step_batch
 * (Philox key = trajectory * N + slot; the systematic offset's key = seed + trajectory), so every trajectory is bit-identical
 * to a single-trajectory run of the batch step.  Needs score stamps (sparse scoring), a float32 codebook with D in
 * {128, 256, 512, 1024}, 16 <= N <= 262144.
 * The batch step runs PRESORTED by default (MIDAS_PRESORT=0: off): in front of the front kernel the folded resample's sources are
 * computed and the slots of every trajectory are put in an execution order that keeps equal nearest-entry hints together (dealt
 * to the waves in runs of MIDAS_PRESORT_RUN = 8 slots); the particles' arithmetic and every output are unchanged (a particle
 * does not know its lane), the rmse partial sums are formed in slot order. */
int64_t midas_lazy_tables_doubles(int64_t N);
int midas_lazy_step_batch(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                          const midas_lazy_args* args, int32_t B);
int midas_lazy_flush_batch(midas_ctx* ctx, const midas_lazy_flush_args* args, int32_t B);


/* ---- particle-sharded step (one process per GPU; the caller runs the collectives between the calls) ---- */
/* The frame of midas_filter_step split at its three global reductions so that N_total particles can be
 * sharded across ranks: shard r owns the global slots [slot_base, slot_base + N) and the
 * global summation blocks [block_base, ...) (4096 slots per block; slot_base must be a multiple of 4096
 * on every rank but the last for results identical to a single-GPU run).  Arrays named *_all span all
 * shards in rank order and are assembled by the caller (torch.distributed all_gather over RCCL). */
typedef struct midas_shard_front_args {
    int64_t N;                  /* local particles */
    int64_t slot_base;          /* global index of local particle 0 (keys the Philox streams) */
    const float* poses_in_dev;  /* N x 16 */
    float* poses_prop_dev;      /* N x 16 out */
    const int32_t* hint_in_dev; /* N or NULL */
    int32_t* nn_idx_dev;        /* N out */
    uint8_t* valid_dev;         /* N out: prune mask */
    const float* odom16_dev;
    const double* code_dev;     /* D: tactile code (unused when scores_ready) */
    double* scores_dev;         /* K: the frame's codebook scores - written here unless scores_ready (codebook rows
                                 * sharded across ranks: the caller gathered the slices into it) */
    int32_t scores_ready;
    const float* gt16_dev;      /* NULL or 16 */
    double* rmse_sums_dev;      /* NULL or 2 out: sum |dt|^2, sum angle^2 over the local particles (r1 + 5 nb + 2) */
    const float* tn_dev;        /* local host draws or NULL -> Philox */
    const float* rot_dev;
    float std_t, std_r;
    uint64_t seed, step;
    double prune_thr;
    uint64_t* telemetry_dev;    /* NULL or 16 cumulative counters (see midas_step_args) */
    int32_t* status_dev;        /* 2: zeroed here, filled by midas_shard_tail_a / midas_shard_tail_fin */
    double* flags_dev;          /* the NaN / kept counters of this rank's exchange record (r1 + 5 nb): zeroed here */
    uint32_t* score_stamps_dev; /* NULL or K: sparse scoring as in midas_step_args (the local particles' waves score the rows
                                 * they need; ignored when scores_ready) */
    uint32_t score_epoch;       /* this frame's stamp value, != 0 and different from every value still in score_stamps_dev */
} midas_shard_front_args;
/* propagate + feature + NN + prune for the local particles and - in the same launch when the codebook is
 * replicated - the codebook scores (particle_filter.py:359-403, tactile_tree.py:43-58) */
int midas_shard_front(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                      const midas_shard_front_args* args);
/* Exchange record (float64), ONE per rank and frame, gathered by the caller in rank order (nb = ceil(N / 4096)):
 *   r1 = { nb block sums of e | nb block totals of e*valid | nb block totals of x*valid | nb block max x | nb block min x |
 *          NaN count | kept count | sum |dt|^2 | sum angle^2 }
 * midas_shard_tail_a: x = scores[nn_idx], e = exp(x - 1); fills this rank's softmax / CDF tables (tables_dev: the layout
 * of midas_lazy_args.tables_dev minus its 5 nb block records, i.e. 4 N16 + 2 G16 + 32 nb doubles, 128-byte aligned: e | x |
 * lp | lp_raw | chunk ends x2 | group ends x2; the raw variants only where a block's own score range is within the isclose
 * tolerance - the guard of get_similarity is global and decided after the exchange) and r1[0 .. 5 nb + 2);
 * status[0] = 2 on NaN, status[1] = local particles kept. */
int midas_shard_tail_a(midas_ctx* ctx, int64_t N, const double* scores_dev, const int32_t* nn_idx_dev,
                       const uint8_t* valid_dev, int32_t softmax, double* tables_dev, double* r1_dev, int32_t* status_dev);
/* Owner-side resample.  The draw of a slot is a pure function of the slot (Philox keyed by the global slot, replicated host
 * uniforms, or the systematic comb) and the gathered records give every rank the global block prefix, so every rank can
 * tell for EVERY slot of the filter which rank owns its source particle; the owner resolves the source inside its own
 * tables and sends that one row to the rank holding the slot.  Per frame:
 *   midas_shard_route_count  -> counts_dev[0 .. G) rows this rank sends to each rank, [G .. 2G) rows it receives from each
 *                               (the caller reads them back: split sizes of the all_to_all); finalises status / rmse
 *   midas_shard_route_pack   -> send_dev: sum(send counts) records of 88 bytes, segment of rank d at the exclusive prefix
 *                               of the send counts; weights_dev: this rank's masked weights
 *   all_to_all (caller), then midas_shard_unpack scatters the N received records to their slots.
 * Record: int32 slot (local at the destination) | int32 global source index | int32 NN index | int32 0 | f64 weight |
 * 16 x f32 pose.  N must be the same on every rank and >= 256. */
typedef struct midas_shard_route_args {
    int64_t N;
    int32_t G, rank;
    const double* r1_all_dev;       /* G x (5 nb + 4) */
    const double* tables_dev;       /* this rank's tables (midas_shard_tail_a) */
    const uint8_t* valid_dev;
    const int32_t* nn_idx_dev;
    const float* poses_prop_dev;
    int32_t* status_dev;            /* 2 out: global cdf status, total particles kept */
    double* rmse_dev;               /* NULL or 2 out */
    int32_t softmax, resample_mode;
    const double* u_all_dev;        /* NULL -> Philox, or G * N uniforms (the same on every rank) */
    float u32;
    uint64_t seed, step;
    int32_t* counts_dev;            /* 3 G + 1 ints: send counts | receive counts | scratch */
    void* send_dev;                 /* pack: sum(send counts) x 88 bytes (fixed_cap > 0: G x fixed_cap x 88) */
    double* weights_dev;            /* pack: N out */
    /* Fixed-capacity form of midas_shard_route_pack - no count pass, nothing read back: the rows for rank d go to rows
     * [d * fixed_cap, (d + 1) * fixed_cap) of send_dev (unused rows keep slot = -1), the all_to_all has equal splits; rows
     * that do not fit a segment go to ovf_dev (ovf_cap rows, gathered by every rank, destination rank in the record's
     * fourth int); counts_dev[3 G] = rows sent there (> ovf_cap: rows were lost, the frame is invalid).  The receiver scans all G x fixed_cap + G x ovf_cap rows
     * (midas_shard_unpack_rows).  0: the counted form above. */
    int64_t fixed_cap, ovf_cap;
    void* ovf_dev;
    void* self_dev;                 /* fixed form: N x 88 bytes - the rows whose slot and source both live on this rank stay
                                     * here instead of travelling (unpacked like the others) */
    /* Peer-mapped form of midas_shard_route_pack - no send buffer, no collective for the rows: peers_dev[d] is rank d's
     * inbox (N rows of 128 bytes, midas_peer_alloc; csrc/peer_row.hpp) as mapped into THIS process (midas_peer_open; the own
     * inbox for d == rank); the owner of a slot's source stores the row straight into row `slot` of the destination's inbox
     * (system-scope stores over xGMI, sixteen adjacent lanes per row: one whole 128-byte line).  Every slot of the filter has exactly one owner, so after a barrier across the ranks each inbox
     * holds its N rows (midas_shard_unpack_peer).  NULL: the forms above. */
    void* const* peers_dev;
    /* NULL or the shard's guide tables (midas_lazy_guide_bytes(N) bytes, 16-byte aligned; layout and meaning as
     * midas_lazy_args.guide_dev), written by the shard's tail next to tables_dev: the owner-side search of a draw then starts
     * from one entry pair instead of two table lines.  Same indices either way. */
    const uint8_t* guide_dev;
} midas_shard_route_args;
int midas_shard_route_count(midas_ctx* ctx, const midas_shard_route_args* args);
int midas_shard_route_pack(midas_ctx* ctx, const midas_shard_route_args* args);
int midas_shard_unpack(midas_ctx* ctx, int64_t N, const void* recv_dev, int32_t* ridx_dev, float* poses_out_dev,
                       double* weights_out_dev, int32_t* hint_out_dev);
/* the same over `rows` records of which some are padding (slot -1); dest >= 0: only the records whose destination rank is
 * dest (the gathered overflow blocks) */
int midas_shard_unpack_rows(midas_ctx* ctx, int64_t rows, const void* recv_dev, int32_t dest, int32_t* ridx_dev,
                            float* poses_out_dev, double* weights_out_dev, int32_t* hint_out_dev);
/* the three unpack passes of a fixed-capacity frame in ONE call (the frame is bound by the host's enqueueing, ~14 us per call):
 * the received segments (every record), the gathered overflow blocks (records for `rank`), the rows this rank owned and
 * needed itself (every record); any of the three may have 0 rows.  Same kernels, same results as three midas_shard_unpack_rows
 * calls (midastouch_amd/dist.py unpack_fixed; no counterpart in the reference, whose filter is a single process). */
int midas_shard_unpack_fixed(midas_ctx* ctx, int64_t rows_recv, const void* recv_dev, int64_t rows_ovf, const void* ovf_all_dev,
                             int32_t rank, int64_t rows_self, const void* self_dev, int32_t* ridx_dev, float* poses_out_dev,
                             double* weights_out_dev, int32_t* hint_out_dev);
/* ---- peer-mapped inboxes (the fourth exchange form; RCCL only carries the 1 KB block records and a barrier) ----
 * midas_peer_alloc: `bytes` of fine-grained device memory (hipExtMallocWithFlags, what RCCL uses for buffers its peers
 * write) + the 64-byte interprocess handle other ranks open it with.  midas_peer_open maps another process's inbox
 * (hipIpcOpenMemHandle, lazy peer access over xGMI; a process on the SAME device works too - the shared-GPU test).
 * midas_peer_probe_write / _check: start-up self test of the data path - every rank stores {nonce, rank} into row
 * `rank` of every inbox, barrier, every rank checks its G rows (ok_dev[0] = 1 when all match) - run twice with different
 * nonces so that a stale cached line would be caught; the caller falls back to the collective forms if any rank fails. */
int midas_peer_alloc(midas_ctx* ctx, int64_t bytes, void** ptr_out, void* handle64_out);
int midas_peer_free(midas_ctx* ctx, void* ptr);
int midas_peer_open(midas_ctx* ctx, const void* handle64, void** ptr_out);
int midas_peer_close(midas_ctx* ctx, void* ptr);
int midas_peer_probe_write(midas_ctx* ctx, void* const* peers_dev, int32_t G, int32_t rank, int32_t nonce);
int midas_peer_probe_check(midas_ctx* ctx, const void* inbox_dev, int32_t G, int32_t nonce, int32_t* ok_dev);
/* the N rows of this rank's inbox -> slots (reads that bypass the non-coherent caches) */
int midas_shard_unpack_peer(midas_ctx* ctx, int64_t N, const void* inbox_dev, int32_t* ridx_dev, float* poses_out_dev,
                            double* weights_out_dev, int32_t* hint_out_dev);
/* ---- the sharded frame by ONE call, on a library-owned RCCL communicator ------------------------------------------------
 * The reference has no distributed code; BASELINE.json's north_star asks for the particle shards of a node to combine
 * their per-shard weight sums "with an RCCL all-reduce ... over xGMI".  midas_comm wraps an ncclComm_t the LIBRARY owns
 * (librccl opened with dlopen: rccl_path = the copy to use, e.g. the one torch has loaded; NULL = the loader's default):
 *   midas_comm_unique_id   rank 0 obtains the 128-byte id and hands it to the others (any channel: a store, a file);
 *   midas_comm_create      collective over the ranks (ncclCommInitRank);  midas_comm_all_gather: bytes % 8 == 0.
 * midas_shard_step enqueues, on the context's stream and without returning to the host in between, the phases selected:
 *   MIDAS_SHARD_PHASE_LOCAL     midas_shard_front + midas_shard_tail_a on this rank's particles (record r1_dev)
 *   MIDAS_SHARD_PHASE_GATHER    ncclAllGather of the records -> r1_all_dev (G x (5 nb + 4) doubles); needs `comm`
 *   MIDAS_SHARD_PHASE_ROUTE     owner-side resample straight into the peers' inboxes (midas_shard_route_pack, peer form).  The
 *                               LAST workgroup of the kernel to finish - every row this rank stores is out then - stores
 *                               frame_tag into slot `rank` of the flag block of every inbox (G uint64 at inbox + flag_offset,
 *                               flag_offset >= 128 N) and waits until all G slots of its OWN inbox carry the tag (bounded: 2 s,
 *                               then status[0] |= 16): when the kernel ends, the inbox holds its N rows - device-side flags
 *                               over the mapped inboxes instead of a second collective, polled by one wave
 *   MIDAS_SHARD_PHASE_UNPACK    inbox -> slots (ridx / poses_out / weights_out / hint_out)
 *   MIDAS_SHARD_PHASE_FLAG      shards of ONE process on one stream need every shard's flag out before any waits: with ROUTE
 *                               the route kernel does not wait and the flags are published by a kernel of their own right
 *                               behind it; with UNPACK the unpack kernel waits for the flags first.
 * A caller without RCCL between its ranks (tests: two processes sharing one GPU) runs LOCAL, gathers the records itself,
 * then runs ROUTE | UNPACK; shards of ONE process on one stream run every shard's ROUTE | FLAG before any UNPACK | FLAG (a
 * waiting kernel in front of the kernel it waits for would never end).  frame_tag must grow from frame to frame (the flag slots are
 * never reset).
 * midas_shard_run: T frames by one call (device draws; odom16_dev / code_dev / gt16_dev advance by one frame each, step,
 * frame_tag and score_epoch by one).  The resampled particles land in poses_out_dev / hint_out_dev, which the engine
 * passes as the next frame's poses_in_dev / hint_in_dev (the same buffers) - after the LAST frame of the call: in between the
 * UNPACK phase is folded into the next frame's front kernel, which waits for the flags and takes particle `slot` from row
 * `slot` of the inbox (one launch fewer per frame; ridx / weights_out are those of the call's last frame.
 * MIDAS_SHARD_FOLD=0 in the environment: every frame unpacks). */
typedef struct midas_comm midas_comm;
int midas_comm_unique_id(midas_ctx* ctx, const char* rccl_path, void* id128_out);
int midas_comm_create(midas_ctx* ctx, const char* rccl_path, const void* id128, int32_t world, int32_t rank, midas_comm** out);
int midas_comm_destroy(midas_comm* comm);
int midas_comm_all_gather(midas_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes);
#define MIDAS_SHARD_PHASE_LOCAL 1
#define MIDAS_SHARD_PHASE_GATHER 2
#define MIDAS_SHARD_PHASE_ROUTE 4
#define MIDAS_SHARD_PHASE_UNPACK 8
#define MIDAS_SHARD_PHASE_FLAG 16  /* with ROUTE: publish the flags by a kernel of their own right behind the route kernel */
typedef struct midas_shard_step_args {
    midas_shard_front_args front;   /* as midas_shard_front (rmse_sums_dev = r1_dev + 5 nb + 2, flags_dev = r1_dev + 5 nb) */
    int32_t softmax;
    double* tables_dev;             /* as midas_shard_tail_a */
    double* r1_dev;                 /* 5 nb + 4: this rank's record */
    double* r1_all_dev;             /* G x (5 nb + 4) */
    int32_t G, rank, resample_mode;
    const double* u_all_dev;        /* NULL -> Philox, or G * N uniforms (the same on every rank) */
    float u32;
    int32_t* counts_dev;            /* 3 G + 1 ints of scratch */
    double* weights_dev;            /* N out: masked weights before the resample */
    double* rmse_dev;               /* NULL or 2 out */
    void* const* peers_dev;         /* G inbox addresses as mapped into this process (midas_peer_open) */
    void* inbox_dev;                /* this rank's inbox: N x 128 bytes of rows, then the flag block: 64 uint64 flags and one
                                     * zero-initialised 64-byte line (the route kernel's workgroup counter): flag_offset + 576 bytes */
    int64_t flag_offset;
    uint64_t frame_tag;
    int32_t* ridx_dev;              /* N out */
    float* poses_out_dev;           /* N x 16 out */
    double* weights_out_dev;        /* N out */
    int32_t* hint_out_dev;          /* N out */
    int32_t* score_list_dev;        /* NULL or 2 + 2 K int32, zero-initialised: prediction lists of the sparse scoring, as in
                                     * midas_lazy_args (front.score_epoch then advances by TWO per frame; midas_shard_run does) */
    uint8_t* guide_dev;             /* NULL or midas_lazy_guide_bytes(front.N) bytes, 16-byte aligned: the shard's guide tables, written
                                     * by the LOCAL phase's tail and read by the ROUTE phase's searches (midas_shard_route_args.guide_dev) */
} midas_shard_step_args;
int midas_shard_step(midas_ctx* ctx, midas_comm* comm, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                     const midas_shard_step_args* args, int32_t phases);
int midas_shard_run(midas_ctx* ctx, midas_comm* comm, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                    const midas_shard_step_args* first, int32_t T);

/* ---- the all_gather form of the exchange (every rank materialises its slice of the global CDF and gathers every
 * shard's packed block; G-1 times the bytes of the owner-side form, no read-back of counts) ---- */
/* midas_shard_tail_fin: softmax applied unless softmax == 0 or |max x - min x| over r1_all <= 1e-8 (then e := x);
 * weights = e / S * valid; cdf_dev = (BP + lp) / total with S, BP, total summed sequentially in global
 * block order over r1_all; the globally last slot is forced to 1; status = global cdf status + total kept count;
 * rmse_dev (nullable, 2 doubles) from the sums in r1_all over N_total particles. */
int midas_shard_tail_fin(midas_ctx* ctx, int64_t N, const double* tables_dev, const uint8_t* valid_dev,
                         double* weights_dev, double* cdf_dev, int32_t G,
                         const double* r1_all_dev, int32_t rank, int64_t N_total, int32_t softmax, double* rmse_dev,
                         int32_t* status_dev);
/* The cross-rank resample reads any shard's particles from ONE gathered buffer: every rank contributes a
 * record block of rank_stride bytes laid out as
 *     [ cdf: n x f64 | weights: n x f64 | propagated poses: n x 16 f32 | nn_idx: n x i32 ]   (n = n_per_rank,
 * even; rank_stride >= 84 n and a multiple of 16), blocks in rank order (torch.distributed all_gather). */
typedef struct midas_tail_resample_args {
    int64_t N, N_all, slot_base;
    const void* pack_all_dev;       /* world x rank_stride bytes */
    int64_t rank_stride, n_per_rank;
    const int32_t* status_dev;      /* [0] != 0 -> identity resample */
    int32_t mode;
    const double* u_dev;            /* N local uniforms or NULL -> Philox keyed by the global slot */
    float u32;
    uint64_t seed, step;
    int32_t* ridx_dev;              /* N out: GLOBAL source index of each local slot */
    float* poses_out_dev;           /* N x 16 */
    double* weights_out_dev;        /* N */
    int32_t* hint_out_dev;          /* N */
} midas_tail_resample_args;
int midas_tail_resample(midas_ctx* ctx, const midas_tail_resample_args* args);

/* ---- the whole loop body on a variable-size particle set ------------------------------------------------------
 * One call = filter/filter.py:150-190 INCLUDING clustering and annealing: motion model (particle_filter.py:359-377), rmse
 * (:472-496), SE3_NN + get_similarity (:449-469, tactile_tree.py:43-58), remove_invalid_particles (:379-403) with the
 * re-projection when every particle drifted (filter.py:176-179), cluster_particles (DBSCAN, :208-228) on the frames the
 * caller asks for, get_cluster_centers("quat_avg") (:153-206), annealing (:405-447: device-side top-k selection, compaction
 * or duplication) and resampler (:230-307) - with the particle count in device memory, so that nothing is read back
 * between frames.  Every per-particle array holds `cap` entries (the initial particle count: annealing never grows the set
 * beyond it, :439-440); the live count is ctl_i[MIDAS_LOOP_I_N].
 *
 * phases (bit mask, executed in this order; a frame is all of them, possibly split over several calls when the host
 * wants to look at or replace something in between - e.g. draw the resampler's uniforms once the annealed size is known):
 *   MIDAS_LOOP_FRONT     propagate + NN + prune + codebook scores; x, e, masked weights w, S, guard, drift re-projection, rmse
 *   MIDAS_LOOP_DBSCAN    labels_dev <- DBSCAN(translations of the propagated poses, eps, n / 5); ctl_i[NCL]
 *   MIDAS_LOOP_ANNEAL    cluster centres of labels_dev, var = mean(stds), annealing decision, src_dev = annealed set
 *                        (without this phase the annealed set is the identity: n_set = n)
 *   MIDAS_LOOP_RESAMPLE  blocked CDF of (e * valid)[src], n_set draws, gathers into poses_dev / weights_out_dev /
 *                        hint_dev / labels_out_dev; ctl_i[N] <- n_set; the frame's log row
 * Summation order, draws and guards as in midas_filter_step; ties in the annealing's top-k go to the smaller index
 * (what torch.topk does on CUDA; its CPU kernel makes another, equally arbitrary choice). */
#define MIDAS_LOOP_FRONT 1
#define MIDAS_LOOP_DBSCAN 2
#define MIDAS_LOOP_ANNEAL 4
#define MIDAS_LOOP_RESAMPLE 8
#define MIDAS_LOOP_MAX_CLUSTERS 64
/* ctl_i (32 x int32) */
#define MIDAS_LOOP_I_N 0       /* live particle count (set by the caller before the first frame) */
#define MIDAS_LOOP_I_NSET 1    /* size of the annealed set of this frame = draws of its resample */
#define MIDAS_LOOP_I_MODE 2    /* annealing of this frame: 0 none, 1 removed K particles, 2 duplicated K */
#define MIDAS_LOOP_I_K 3
#define MIDAS_LOOP_I_INIT 4    /* init_particles (particle_filter.py:416) */
#define MIDAS_LOOP_I_VARSET 5  /* 0 until the first annealing call stored its variance (:413-417) */
#define MIDAS_LOOP_I_KEPT 6    /* particles kept by the prune */
#define MIDAS_LOOP_I_DRIFT 7   /* 1: every particle was pruned, poses re-projected onto the codebook */
#define MIDAS_LOOP_I_STATUS 8  /* cdf status of the resample (see midas_cdf): != 0 -> the annealed set went on unresampled */
#define MIDAS_LOOP_I_RAW 9     /* 1: the weights are raw scores (softmax off or skipped by the isclose guard) */
#define MIDAS_LOOP_I_NCL 10    /* labels are in [-1, NCL) (DBSCAN phase, or set by the caller with its own labels) */
#define MIDAS_LOOP_I_NPRES 11  /* rows of cluster_poses_dev / cluster_stds_dev: labels present, ascending */
#define MIDAS_LOOP_I_FRAME 12  /* frames completed */
#define MIDAS_LOOP_I_NAN 13    /* NaN among the scores */
#define MIDAS_LOOP_I_ERR 14    /* conditions of the CURRENT frame (cleared once its log row holds them): bit 0 / 1: more than
                                * MIDAS_LOOP_MAX_CLUSTERS - 1 clusters (decide / DBSCAN), bit 2: live count above the launches' bound (particles were
                                * not processed), bit 5: DBSCAN saw non-finite translations or more than 2^21 cells per axis, bit 6: a
                                * cloud wider than 128 cells per axis with more than 2^20 particles (bits 5 / 6: labels undefined), bit 7: anneal_frozen was
                                * set and the annealing rule wanted to act (the set was left as it was: not the reference's) */
/* ctl_d (16 x float64) */
#define MIDAS_LOOP_D_S 0        /* softmax denominator (1 when raw) */
#define MIDAS_LOOP_D_VARPREV 1  /* particle_var (float32 value) */
#define MIDAS_LOOP_D_VAR 2      /* this frame's mean cluster spread */
#define MIDAS_LOOP_D_RMSE_T 3
#define MIDAS_LOOP_D_RMSE_R 4
#define MIDAS_LOOP_D_XMAX 5
#define MIDAS_LOOP_D_XMIN 6
#define MIDAS_LOOP_D_TOTAL 7    /* total of the resample's CDF */
#define MIDAS_LOOP_LOG_DOUBLES 168 /* log row: 16 scalars {frame, n, n_set, rmse_t, rmse_r, kept, drifted, status, mode, k,
                                    * npres, var, S, raw, ncl, err} + 8 x (16 pose + 3 std) of the first clusters present */
typedef struct midas_loop_args {
    int64_t cap;
    int32_t* ctl_i_dev;            /* 32 */
    double* ctl_d_dev;             /* 16 */
    float* poses_dev;              /* cap x 16: particle set at the frame start; the resampled set on return */
    float* poses_prop_dev;         /* cap x 16: propagated (and possibly re-projected) poses of the frame */
    int32_t* hint_dev;             /* cap: NN index of each particle's ancestor, in / out (-1 = none) */
    int32_t* nn_idx_dev;           /* cap */
    uint8_t* valid_dev;            /* cap: prune mask */
    double* x_dev;                 /* cap: scores[nn] */
    double* e_dev;                 /* cap: exp(x - 1), or x when the softmax is off */
    double* weights_dev;           /* cap: masked weights before annealing / resampling */
    double* weights_out_dev;       /* cap: the same gathered by the resample */
    int32_t* labels_dev;           /* cap: cluster labels of the particles of this frame (in, or written by the DBSCAN phase) */
    int32_t* labels_out_dev;       /* cap: labels gathered by the resample (the caller swaps the two) */
    int32_t* src_dev;              /* cap: annealed set -> index into the frame's particles */
    int32_t* ridx_dev;             /* cap: resample indices into the annealed set */
    double* scores_dev;            /* K */
    double* part_rmse_dev;         /* NULL or 2 ceil(cap / 64) */
    const float* cb_poses_dev;     /* K x 16: codebook poses (drift re-projection, filter.py:176-179) */
    float* cluster_poses_dev;      /* MIDAS_LOOP_MAX_CLUSTERS x 16 out */
    float* cluster_stds_dev;       /* MIDAS_LOOP_MAX_CLUSTERS x 3 out */
    double* log_dev;               /* NULL or MIDAS_LOOP_LOG_DOUBLES: this frame's log row */
    const float* odom16_dev;
    const double* code_dev;
    const float* gt16_dev;         /* NULL or 16 */
    const float* tn_dev;           /* host draws for the live particles or NULL -> Philox(seed, step) */
    const float* rot_dev;
    const double* u_dev;           /* NULL -> Philox, or >= n_set uniforms */
    float u32;
    float std_t, std_r;
    uint64_t seed, step;
    double prune_thr;
    int32_t softmax, resample_mode;
    int32_t floor;                 /* annealing floor (particle_filter.py:406) */
    double eps;                    /* DBSCAN radius (particle_filter.py:209) */
    int32_t unit_weights;          /* 1: a frame without measurement update - every particle scores 1, so the weights are the
                                    * prune mask (filter/filter_real.py:205-212, `update_freq`) */
    uint64_t* telemetry_dev;       /* NULL or 16 counters (see midas_step_args) */
    uint32_t* score_stamps_dev;    /* sparse scoring, see midas_step_args */
    uint32_t score_epoch;
    /* What lets the caller size the launches to the live set without reading anything back: */
    int32_t* host_mirror;          /* NULL or 2 int32 in PINNED HOST memory: the RESAMPLE phase stores {frames completed, live
                                    * count} there when the frame is done; the host may look at it whenever it likes */
    int64_t grid_n;                /* 0 (= cap) or an UPPER BOUND of the live count at the start of this frame: the grids cover
                                    * grid_n particles (annealing grows the set by at most n / 3 per frame, so a count seen L frames
                                    * ago bounds today's by (4/3)^L); ctl_i[ERR] |= 4 if the bound was wrong */
    int32_t anneal_small;          /* 1: grid_n <= 16384 - one workgroup runs decide + select + compaction + sort (same results) */
    int32_t topk_ties;             /* MIDAS_TOPK_TIES_INDEX (0): ties of annealing's torch.topk go to the smaller index (torch's CUDA
                                    * kernel; the fast radix select).  MIDAS_TOPK_TIES_ATEN_CPU (1): the members - and for duplicates
                                    * the order - ATen's CPU kernel picks (std::partial_sort when k * 64 <= n, else std::nth_element +
                                    * std::sort, walked move for move by one wave: topk_aten.hip), i.e. what the reference keeps when it
                                    * runs on the CPU under a fixed seed (modules/particle_filter.py:433-441) */
    int32_t anneal_frozen;         /* 1: the caller states that annealing cannot change the set - the live count equals `floor` AND the
                                    * count annealing started from (modules/particle_filter.py:421-446: a removal needs |n - floor| > 0,
                                    * a duplication k + n <= init_particles; both stay impossible once they are).  The ANNEAL phase then
                                    * runs its decision only (cluster rows, variance: two small workgroups) and none of the selection's
                                    * launches; the decision checks the statement and raises ctl_i[ERR] bit 7 if it finds work to do. */
    int32_t pad2_;
} midas_loop_args;
int midas_loop_step(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6, const midas_tree* tree3,
                    const midas_loop_args* args, int32_t phases);
/* cluster_particles(method="euclidean") alone (particle_filter.py:208-217): labels_dev[i] = DBSCAN label of pose i's
 * translation, eps as given, min_samples < 0 -> N / 5.  Exact float64 predicate |dx|^2 <= eps^2, clusters numbered by their
 * first core point, border points to the smallest adjacent cluster - what sklearn's DBSCAN returns.  Any extent (dense cell grid
 * up to 128 cells of 0.577 eps per axis, a hash table of the occupied cells beyond) and any number of clusters.
 * ncl_dev: 2 x int32 out {number of clusters, flag: 32 = non-finite coordinates / more than 2^21 cells per axis, 64 = a cloud wider
 * than 128 cells per axis with more than 2^20 points (the hash table of occupied cells holds 2^21 slots): labels undefined}. */
int midas_dbscan(midas_ctx* ctx, int64_t N, const float* poses_dev, double eps, int64_t min_samples, int32_t* labels_dev,
                 int32_t* ncl_dev);
/* cluster_particles(method="logmap") (particle_filter.py:218-223): DBSCAN of N points of `dim` (2 .. 6) float64 coordinates
 * (row-major) - the 6-d SE(3) logarithms there -, all pairs, the same predicate and numbering as midas_dbscan (sklearn's
 * labels).  min_samples < 0 -> N / 5.  info_dev: 2 x int32 out {number of clusters, spread steps taken (-1: not settled)}.
 * Synchronises the stream (one look per spread step); not on the filter's loop. */
int midas_dbscan_points(midas_ctx* ctx, int64_t N, int32_t dim, const double* points_dev, double eps, int64_t min_samples,
                        int32_t* labels_dev, int32_t* info_dev);

/* The selection step of particle_filter.annealing alone (modules/particle_filter.py:421-446, torch.topk + Particles.remove /
 * add): src_dev[0 .. n_set) = the annealed set as indices into the N particles.  mode 1: the N particles minus the k of
 * smallest weight, in their order (n_set = N - k); mode 2: all N followed by the k of largest weight, largest first
 * (n_set = N + k; src_dev holds N + k entries).  Ties go to the smaller index.  0 <= k <= N / 3.  weights_dev: N float64. */
int midas_anneal_select(midas_ctx* ctx, int64_t N, const double* weights_dev, int32_t mode, int64_t k, int32_t* src_dev);
/* The same with the tie rule chosen (modules/particle_filter.py:433-441, `torch.topk(particles.weights, k, largest=...)`):
 * ties = MIDAS_TOPK_TIES_INDEX is midas_anneal_select; MIDAS_TOPK_TIES_ATEN_CPU returns exactly the set (mode 1) / the list
 * (mode 2) torch.topk returns on the CPU, whichever members of a tie that is.  info_dev: NULL or 1 int32 the call ADDS the
 * number of depth-limit fallbacks to (heap select inside nth_element, heap sort inside sort; tests). */
#define MIDAS_TOPK_TIES_INDEX 0
#define MIDAS_TOPK_TIES_ATEN_CPU 1
int midas_anneal_select_ties(midas_ctx* ctx, int64_t N, const double* weights_dev, int32_t mode, int64_t k, int32_t ties,
                             int32_t* src_dev, int32_t* info_dev);

/* B concurrent trajectories against one codebook (BASELINE config 5, "throughput mode"): every per-trajectory
 * array of `args` carries a leading batch dimension, contiguous - poses (B,N,16), weights (B,N), hints (B,N),
 * odom16 (B,16), code (B,D), gt16 (B,16), rmse (B,2), status (B,2), tn/rot (B,N,3), u (B,N); scalars are shared.
 * The B codes are scored in ONE pass over the codebook on the matrix cores (midas_score_batch) when the
 * embeddings are float32 and D % 16 == 0; the remaining kernels run with the trajectory as grid.y.  Philox
 * streams are keyed by the flattened particle index b*N + n, so trajectories draw independent noise. */
int midas_filter_step_batch(midas_ctx* ctx, const midas_codebook* cb, const midas_tree* tree6,
                            const midas_tree* tree3, const midas_step_args* args, int32_t B);

/* per-kernel timing of midas_filter_step (HIP events on the context stream).  When enabled every
 * kernel of the step is bracketed by events; midas_profile_read synchronises and returns the
 * accumulated milliseconds per kernel slot since the last reset. */
#define MIDAS_PROF_SLOTS 8
/* on: 0 = off, 1 = bracket every kernel, 2 + k = bracket only kernel slot k (least perturbation) */
int midas_profile_enable(midas_ctx* ctx, int32_t on);
int midas_profile_read(midas_ctx* ctx, double* ms_out /*MIDAS_PROF_SLOTS*/, int64_t* calls_out,
                       int32_t reset);
const char* midas_profile_slot_name(int32_t slot);

#ifdef __cplusplus
}
#endif
#endif /* MIDAS_HIP_H */
