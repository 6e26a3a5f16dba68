"""Python face of the CPU oracle (TEST INFRASTRUCTURE - see oracle/midas_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps oracle/libmidas_oracle.so (built by `make -C oracle`) with numpy in / numpy out
functions named after the reference operations they restate, plus `OracleFilter`, the
reference's per-frame loop body (filter/filter.py:150-190) as one `step()`.

Reference citations are relative to /root/reference/midastouch.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmidas_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms)."""
    srcs = [os.path.join(_HERE, f) for f in ("midas_oracle.c", "aten_topk.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.mo_atan2f.restype = C.c_float
        _lib.mo_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.mo_logf.restype = C.c_float
        _lib.mo_logf.argtypes = [C.c_float]
        _lib.mo_philox_uniform32.restype = C.c_float
        _lib.mo_philox_uniform32.argtypes = [C.c_uint64, C.c_uint64]
        _lib.mo_blocked_scan.restype = C.c_double
        _lib.mo_softmax.restype = C.c_int
        _lib.mo_cdf.restype = C.c_int
        _lib.mo_exp.restype = C.c_double
        _lib.mo_exp.argtypes = [C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------------------------------
# elementary spec functions (exposed for the HIP bit-exactness tests)
# ------------------------------------------------------------------------------------------
def sincosf(a):
    a = _f32(a).ravel()
    s, c = np.empty_like(a), np.empty_like(a)
    sv, cv = C.c_float(), C.c_float()
    L = lib()
    for i, x in enumerate(a):
        L.mo_sincosf(C.c_float(float(x)), C.byref(sv), C.byref(cv))
        s[i], c[i] = sv.value, cv.value
    return s, c


def atan2f(y, x):
    y, x = _f32(y).ravel(), _f32(x).ravel()
    L = lib()
    return np.array([L.mo_atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)


def logf(x):
    L = lib()
    return np.array([L.mo_logf(float(a)) for a in _f32(x).ravel()], dtype=np.float32)


def euler_zyx_rad(ang):
    """pose.euler_angles_to_matrix(ang, "ZYX") (modules/pose.py:215-269)."""
    ang = _f32(ang).reshape(-1, 3)
    R = np.empty((ang.shape[0], 3, 3), dtype=np.float32)
    lib().mo_euler_zyx_rad(C.c_int64(ang.shape[0]), _p(ang), _p(R))
    return R


# ------------------------------------------------------------------------------------------
# propagate
# ------------------------------------------------------------------------------------------
def propagate(poses, odom, tn, rot_deg):
    """motionModel arithmetic: poses @ (odom @ Tn(tn, Rz Ry Rx(deg2rad(rot)))) (particle_filter.py:319-375)."""
    poses, odom, tn, rot_deg = _f32(poses), _f32(odom), _f32(tn), _f32(rot_deg)
    out = np.empty_like(poses)
    lib().mo_propagate(C.c_int64(poses.shape[0]), _p(poses), _p(odom), _p(tn), _p(rot_deg), _p(out))
    return out


def philox_noise(N, seed, step, std_t, std_r):
    tn = np.empty((N, 3), dtype=np.float32)
    rot = np.empty((N, 3), dtype=np.float32)
    lib().mo_philox_noise(C.c_int64(N), C.c_uint64(seed), C.c_uint64(step), C.c_float(std_t),
                          C.c_float(std_r), _p(tn), _p(rot))
    return tn, rot


def philox_uniform64(N, seed, step):
    u = np.empty(N, dtype=np.float64)
    lib().mo_philox_uniform64(C.c_int64(N), C.c_uint64(seed), C.c_uint64(step), _p(u))
    return u


def torch_rand64(seed: int, N: int, skip_words: int = 0):
    """torch.rand(N, dtype=float64) after torch.manual_seed(seed) and `skip_words` earlier 32-bit draws - the stream
    torch.multinomial consumes in the resampler (particle_filter.py:245); restated MT19937 (mo_mt19937_rand64)."""
    u = np.empty(N, dtype=np.float64)
    lib().mo_mt19937_rand64(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_int64(skip_words), C.c_int64(N), _p(u))
    return u


def torch_normal_words(numel: int) -> int:
    """32-bit outputs one torch.normal(mean, std, size) of `numel` float32 values takes from the CPU generator (ATen
    normal_fill: one per value, and the last 16 are drawn again when numel is not a multiple of 16; numel >= 16)."""
    assert numel >= 16
    return numel + (16 if numel % 16 else 0)


def philox_raw(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().mo_philox_raw(_p(ctr), _p(key), _p(out))
    return out


def philox_uniform32(seed, step) -> float:
    return float(lib().mo_philox_uniform32(C.c_uint64(seed), C.c_uint64(step)))


# ------------------------------------------------------------------------------------------
# feature + NN
# ------------------------------------------------------------------------------------------
def so3_log(poses):
    poses = _f32(poses).reshape(-1, 4, 4)
    out = np.empty((poses.shape[0], 3), dtype=np.float32)
    lib().mo_so3_log(C.c_int64(poses.shape[0]), _p(poses), _p(out))
    return out


def R3_SE3(poses, w: float = 0.01):
    """tactile_tree.R3_SE3 (tactile_tree/tactile_tree.py:73-77)."""
    poses = _f32(poses).reshape(-1, 4, 4)
    out = np.empty((poses.shape[0], 6), dtype=np.float32)
    lib().mo_se3_feature(C.c_int64(poses.shape[0]), _p(poses), C.c_float(np.float32(1.0 - w)),
                         C.c_float(np.float32(w)), _p(out))
    return out


def nn6(query_feat, cb_feat):
    q, c = _f32(query_feat), _f32(cb_feat)
    idx = np.empty(q.shape[0], dtype=np.int32)
    d2 = np.empty(q.shape[0], dtype=np.float32)
    lib().mo_nn6(C.c_int64(q.shape[0]), C.c_int64(c.shape[0]), _p(q), _p(c), _p(idx), _p(d2))
    return idx, d2


def knn6(query_feat, cb_feat, k):
    """(N, k) indices / squared distances of the k nearest entries by (distance, index) (tactile_tree.py:50-52, n_neighbors = k)."""
    q, c = _f32(query_feat), _f32(cb_feat)
    idx = np.empty((q.shape[0], k), dtype=np.int32)
    d2 = np.empty((q.shape[0], k), dtype=np.float32)
    lib().mo_knn6(C.c_int64(q.shape[0]), C.c_int64(c.shape[0]), C.c_int32(k), _p(q), _p(c), _p(idx), _p(d2))
    return idx, d2


def nn3_dist(poses, verts):
    poses, verts = _f32(poses).reshape(-1, 4, 4), _f64(verts)
    dist = np.empty(poses.shape[0], dtype=np.float64)
    lib().mo_nn3(C.c_int64(poses.shape[0]), C.c_int64(verts.shape[0]), _p(poses), _p(verts), _p(dist))
    return dist


# ------------------------------------------------------------------------------------------
# scores / weights
# ------------------------------------------------------------------------------------------
def score_codebook(emb, code):
    """cosine of one code against every row (get_similarity(..., softmax=False) on all embeddings)."""
    code = _f64(code).ravel()
    emb = np.ascontiguousarray(emb)
    out = np.empty(emb.shape[0], dtype=np.float64)
    if emb.dtype == np.float32:
        lib().mo_score_f32(C.c_int64(emb.shape[0]), C.c_int64(emb.shape[1]), _p(emb), _p(code), _p(out))
    else:
        emb = _f64(emb)
        lib().mo_score_f64(C.c_int64(emb.shape[0]), C.c_int64(emb.shape[1]), _p(emb), _p(code), _p(out))
    return out


def score_codebook_batch(emb, codes):
    """Batched scoring spec of the MFMA kernel: float32 fma chains in the matrix-core order -> (B, K) float64."""
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    codes = _f64(np.atleast_2d(codes))
    out = np.empty((codes.shape[0], emb.shape[0]), dtype=np.float64)
    lib().mo_score_batch_f32(C.c_int64(emb.shape[0]), C.c_int64(emb.shape[1]), C.c_int64(codes.shape[0]), _p(emb),
                             _p(codes), _p(out))
    return out


def softmax_weights(x, softmax: bool = True):
    x = _f64(x).ravel()
    w = np.empty_like(x)
    applied = lib().mo_softmax(C.c_int64(x.shape[0]), _p(x), C.c_int(int(softmax)), _p(w))
    return w, bool(applied)


def softmax_numerators(x, softmax: bool = True, shift=None, exp: str = "spec"):
    """e = exp(x - shift) (the spec exponential mo_exp; exp="libm": the C library's, for the mismatch probes), or x itself
    when the softmax is skipped; returns (e, applied).

    shift None -> max(x) (torch's Softmax).  The fused step uses the constant shift 1.0: the scores are
    cosines (<= 1), the softmax is shift-invariant, and a constant lets every particle take its exponential
    without waiting for a global maximum (csrc/particles.hip k_particle_update)."""
    x = _f64(x).ravel()
    mx, mn = float(np.max(x)), float(np.min(x))
    applied = bool(softmax) and not (abs(mx - mn) <= 1e-8)
    if not applied:
        return x.copy(), False
    c = mx if shift is None else float(shift)
    if exp == "libm":
        import math
        return np.array([math.exp(v) for v in (x - c)], dtype=np.float64), True
    return exp_spec(x, c), True


def exp_spec(x, shift: float = 0.0):
    """mo_exp(x - shift) elementwise: the float64 exponential of the arithmetic spec (midas_math.hpp exp_spec on the device)."""
    x = _f64(x).ravel()
    out = np.empty_like(x)
    lib().mo_exp_vec(C.c_int64(x.shape[0]), _p(x), C.c_double(float(shift)), _p(out))
    return out


def get_similarity(code, targets, softmax: bool = True):
    """particle_filter.get_similarity (particle_filter.py:449-469): targets (N,D) gathered rows."""
    x = score_codebook(np.atleast_2d(targets), code)
    if x.shape[0] == 1:
        return x.reshape(())  # .squeeze() of a single target; softmax is skipped (max == min)
    return softmax_weights(x, softmax)[0]


def blocked_scan(w):
    w = _f64(w).ravel()
    out = np.empty_like(w)
    total = lib().mo_blocked_scan(C.c_int64(w.shape[0]), _p(w), _p(out))
    return out, float(total)


def cdf(w):
    w = _f64(w).ravel()
    out = np.empty_like(w)
    status = lib().mo_cdf(C.c_int64(w.shape[0]), _p(w), _p(out))
    return out, int(status)


def cdf_sequential(w):
    """The reference-true order: ATen's multinomial does a sequential float64 running sum,
    divides by the total and forces the last bucket to 1 (torch.multinomial CPU kernel)."""
    w = _f64(w).ravel()
    p = w / w.sum()
    c = np.cumsum(p)
    c = c / c[-1]
    c[-1] = 1.0
    return c


def search_lower(cdf_arr, u):
    cdf_arr, u = _f64(cdf_arr), _f64(u).ravel()
    idx = np.empty(u.shape[0], dtype=np.int32)
    lib().mo_search_lower(C.c_int64(cdf_arr.shape[0]), _p(cdf_arr), C.c_int64(u.shape[0]), _p(u), _p(idx))
    return idx


def search_systematic(cdf_arr, M, u32):
    cdf_arr = _f64(cdf_arr)
    idx = np.empty(M, dtype=np.int32)
    lib().mo_search_systematic(C.c_int64(cdf_arr.shape[0]), _p(cdf_arr), C.c_int64(M), C.c_float(u32), _p(idx))
    return idx


def resample_indices(weights, mode: str = "weighted_random", u=None, u32=None):
    """particle_filter.resampler index selection (particle_filter.py:230-307).

    Returns (idx or None, status); status != 0 means the reference returns its input unchanged.
    `u`: N float64 uniforms (torch.rand(N, dtype=float64) stream == torch.multinomial's);
    `u32`: the single float32 torch.rand(1) of the low-variance modes.
    """
    c, status = cdf(weights)
    if status:
        return None, status
    n = c.shape[0]
    if mode == "weighted_random":
        return search_lower(c, u), 0
    if mode in ("low_var", "low_var_batch"):
        return search_systematic(c, n, float(u32)), 0
    raise ValueError(mode)


def particle_rmse(poses, gt):
    poses, gt = _f32(poses).reshape(-1, 4, 4), _f32(gt)
    out = np.empty(2, dtype=np.float64)
    lib().mo_rmse(C.c_int64(poses.shape[0]), _p(poses), _p(gt), _p(out))
    return float(out[0]), float(out[1])


# ------------------------------------------------------------------------------------------
# host logic restated: init_filter, annealing
# ------------------------------------------------------------------------------------------
def init_filter_compose(gt, tn, rot_deg):
    """init_filter (particle_filter.py:129-145): gt @ T(from_euler('zyx', rot, degrees), tn).

    The reference builds Rn with scipy in float64 and torch promotes gt(f32) @ Tn(f32 storage of
    the f64 matrix): Tn is allocated with gt's dtype, so Rn is rounded to float32 first.
    """
    from scipy.spatial.transform import Rotation
    n = tn.shape[0]
    Rn = Rotation.from_euler("zyx", np.asarray(rot_deg, dtype=np.float32), degrees=True).as_matrix()
    Tn = np.zeros((n, 4, 4), dtype=np.float32)
    Tn[:, :3, :3], Tn[:, :3, 3], Tn[:, 3, 3] = Rn.astype(np.float32), _f32(tn), 1.0
    return (np.asarray(gt, dtype=np.float32)[None] @ Tn).astype(np.float32)


def cluster_centers(poses, weights, labels):
    """particle_filter.get_cluster_centers(method="quat_avg") (particle_filter.py:153-206) + pose.xyz_quat_averaged
    (pose.py:112-147) in numpy float64: per unique label, float32 weights flattened to 1 when isclose(max - min, 0);
    Markley mean = principal eigenvector of sum w q q^T / sum w over quaternions with qw >= 0 (numpy eigh stands in for the
    removed Tensor.eig: the matrix is symmetric); weighted mean translation; std about the float32 centre.
    -> (label values (C,), centres (C,4,4) f32, stds (C,3) f32)."""
    from scipy.spatial.transform import Rotation
    poses = np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
    w32 = np.asarray(weights).astype(np.float32)
    labels = np.asarray(labels)
    uniq = np.unique(labels)
    centers = np.zeros((len(uniq), 4, 4), dtype=np.float32)
    stds = np.zeros((len(uniq), 3), dtype=np.float32)
    for i, lab in enumerate(uniq):
        sel = labels == lab
        tp, tw = poses[sel].astype(np.float64), w32[sel]
        if abs(np.float32(tw.max() - tw.min())) <= 1e-8:
            tw = np.ones_like(tw)
        tw = tw.astype(np.float64)
        q = Rotation.from_matrix(tp[:, :3, :3]).as_quat()  # x, y, z, w
        q[q[:, 3] < 0] *= -1.0
        M = np.einsum("n,ni,nj->ij", tw, q, q) / tw.sum()
        evals, evecs = np.linalg.eigh(M)
        aq = evecs[:, np.argmax(evals)]
        if aq[3] < 0:
            aq = -aq
        centers[i, :3, :3] = Rotation.from_quat(aq).as_matrix()
        centers[i, :3, 3] = (tp[:, :3, 3] * tw[:, None]).sum(axis=0) / tw.sum()
        centers[i, 3, 3] = 1.0
        d = tp[:, :3, 3] - centers[i, :3, 3].astype(np.float64)
        stds[i] = np.sqrt((d * d * tw[:, None]).sum(axis=0) / tw.sum())
    return uniq, centers, stds


def dbscan(points, eps: float = 1e-2, min_samples: int = 1):
    """particle_filter.cluster_particles(method="euclidean") labels (particle_filter.py:208-217): sklearn DBSCAN on the
    (N,3) float32 translations, restated in C (mo_dbscan).  Returns (labels int32 (N,), number of clusters)."""
    X = _f32(points).reshape(-1, 3)
    labels = np.empty(X.shape[0], dtype=np.int32)
    L = lib()
    L.mo_dbscan.restype = C.c_int32
    ncl = L.mo_dbscan(C.c_int64(X.shape[0]), _p(X), C.c_double(float(eps)), C.c_int64(int(min_samples)), _p(labels))
    return labels, int(ncl)


def cluster_var(stds):
    """`torch.mean(cluster_stds)` (filter/filter.py:189) of the float32 (C,3) spreads as the spec states it: float32
    running sum in row-major order divided by the float32 count."""
    s = np.float32(0.0)
    flat = np.asarray(stds, dtype=np.float32).ravel()
    for v in flat:
        s = np.float32(s + v)
    return np.float32(s / np.float32(flat.shape[0]))


def aten_topk(values, k: int, largest: bool = True, sorted: bool = True, return_fallbacks: bool = False):
    """`torch.topk(values, k, largest=..., sorted=...).indices` of a 1-d float64 tensor ON THE CPU, ties included:
    oracle/aten_topk.c restates ATen's `topk_impl_loop` over libstdc++'s partial_sort / nth_element / sort
    (particle_filter.py:433-441 is the caller).  Pinned against torch.topk by tests/test_aten_topk.py."""
    v = _f64(values).ravel()
    out = np.empty(int(k), dtype=np.int64)
    fb = C.c_int64(0)
    rc = lib().mo_aten_topk(_p(v), C.c_int64(v.shape[0]), C.c_int64(int(k)), C.c_int(int(bool(largest))), C.c_int(int(bool(sorted))),
                            _p(out), None, C.byref(fb))
    if rc != 0:
        raise ValueError("aten_topk: k out of range")
    return (out, fb.value) if return_fallbacks else out


def aten_topk_killer(n: int, nth: int = 0, for_sort: bool = False):
    """A permutation of 0..n-1 (float64) on which the median-of-three partition of nth_element (at position `nth`) or of
    sort degenerates until the depth limit is spent (McIlroy's adversary played against the restatement)."""
    out = np.empty(int(n), dtype=np.float64)
    if lib().mo_aten_topk_killer(C.c_int64(int(n)), C.c_int64(int(nth)), C.c_int(int(bool(for_sort))), _p(out)) != 0:
        raise MemoryError
    return out


class Annealer:
    """particle_filter.annealing (particle_filter.py:405-447) on index sets.

    `step(weights, var, floor)` returns the index array (into the current particles) of the
    particles that survive, with duplicates appended for growth - same order as the reference
    (`Particles.remove` keeps the original order, `add` appends in topk order).

    ties: which members of a tie `torch.topk` takes (and in which order it lists them) is the one thing the values do not
    decide.  "index" = smaller index first (torch's CUDA kernel, the device's default rule); "aten_cpu" = what ATen's CPU
    kernel does (`aten_topk`: the reference as it runs on the CPU, i.e. what the G13 fixture holds).
    """

    def __init__(self, ties: str = "index"):
        assert ties in ("index", "aten_cpu")
        self.ties = ties
        self.particle_var = float("inf")
        self.init_particles = None

    def step(self, weights, var: float, floor: int = 1000):
        w = np.asarray(weights)
        n = w.shape[0]
        keep = np.arange(n)
        # `var` is a float32 torch scalar in the reference (torch.mean(cluster_stds)), so the
        # ratio and the counts derived from it are float32 arithmetic.
        var = np.float32(var)
        if np.isinf(self.particle_var):
            self.particle_var = var
            self.init_particles = n
            return keep
        if var == 0.0:
            return keep
        ratio = np.float32(var / np.float32(self.particle_var))
        self.particle_var = var
        one = np.float32(1.0)
        if ratio < 1:
            num_remove = min(int(np.float32(one - ratio) * np.float32(n)), abs(n - floor), n // 3)
            if not num_remove:
                return keep
            order = np.argsort(w, kind="stable")[:num_remove] if self.ties == "index" else aten_topk(w, num_remove, largest=False)
            mask = np.ones(n, dtype=bool)
            mask[order] = False
            return keep[mask]
        if ratio > 1:
            num_increase = min(int(np.float32(ratio - one) * np.float32(n)), n // 3)
            if num_increase + n > self.init_particles:
                return keep
            order = np.argsort(-w, kind="stable")[:num_increase] if self.ties == "index" else aten_topk(w, num_increase, largest=True)
            return np.concatenate([keep, order])
        return keep


# ------------------------------------------------------------------------------------------
# the per-frame loop body as one object
# ------------------------------------------------------------------------------------------
class OracleLoop:
    """The whole loop body filter/filter.py:150-190 - motion, rmse, weights, prune (+ re-projection when every particle
    drifted, :176-179), DBSCAN every 50th frame (:182-183), cluster centres (:184-186), annealing (:189), resampling
    (:190) - as one `step()` over a particle set whose size changes from frame to frame.  The spec of the device loop
    engine (midas_loop_step): same arithmetic as OracleFilter.step plus
      labels   = dbscan(translations, eps, N // 5) on frames with count % 50 == 0, carried through the resample otherwise;
      var      = cluster_var(stds of the clusters present), annealing on it (Annealer; ties in the top-k by index);
      resample = N' draws over the blocked CDF of (e * mask)[keep], N' = size of the annealed set; the weights that
                 travel on are e / S * mask with S the blocked sum of e over the N particles BEFORE annealing."""

    def __init__(self, cb_poses, cb_embeddings, mesh_verts, pen_max=0.002, floor=1000, eps=1e-2, softmax=True, cluster=True,
                 cluster_every=50, ties="index"):
        self.f = OracleFilter(cb_poses, cb_embeddings, mesh_verts, pen_max)
        self.annealer = Annealer(ties)
        self.floor, self.eps, self.softmax, self.cluster = int(floor), float(eps), bool(softmax), bool(cluster)
        self.cluster_every = int(cluster_every)
        self.count = 0

    def step(self, poses, labels, odom, code, tn, rot_deg, u=None, gt=None, mode="weighted_random", u32=None,
             keep_override=None, draws=None):
        """`u`: uniforms for the N' draws (only the first N' are used) or `draws(N')` -> uniforms, called once N' is known
        (the reference draws them after annealing).  keep_override: teacher-forced annealed index list (tests of tie frames)."""
        f, out = self.f, {}
        N = poses.shape[0]
        p1 = propagate(poses, odom, tn, rot_deg)
        if gt is not None:
            out["rmse"] = particle_rmse(p1, gt)
        idx = nn6(R3_SE3(p1), f.cb_feat)[0]
        scores = score_codebook(f.emb, code)
        x = scores[idx]
        mask = ~(nn3_dist(p1, f.verts) > f.pen_max)
        e, applied = softmax_numerators(x, self.softmax, shift=1.0)
        S = blocked_scan(e)[1] if applied else 1.0
        w = e / S * mask
        out.update(poses_prop=p1, nn_idx=idx, mask=mask, weights=w, drifted=bool(mask.sum() == 0))
        if out["drifted"]:  # every particle is off the surface: back onto the codebook (:176-179)
            p1 = f.cb_poses[idx].copy()
            out["poses_prop"] = p1
        if self.cluster:
            if self.count % self.cluster_every == 0:
                labels = dbscan(p1[:, :3, 3], self.eps, N // 5)[0]
            uniq, centers, stds = cluster_centers(p1, w, labels)
            var = cluster_var(stds)
            keep = self.annealer.step(w, var, self.floor) if keep_override is None else np.asarray(keep_override)
            if keep_override is not None:  # keep the annealer's state in step with the forced decision
                self.annealer.step(w, var, self.floor)
            out.update(labels_frame=labels, cluster_labels=uniq, cluster_poses=centers, cluster_stds=stds, var=var)
        else:
            keep = np.arange(N)
        out["keep"] = keep.astype(np.int32)
        n2 = keep.shape[0]
        em = (e * mask)[keep]
        if u is None and draws is not None:
            u = draws(n2)
        ridx, status = resample_indices(em, mode, u=None if u is None else np.asarray(u)[:n2], u32=u32)
        out["status"] = status
        if status:
            ridx = np.arange(n2, dtype=np.int32)
        src = keep[ridx]
        out.update(ridx=ridx.astype(np.int32), src=src.astype(np.int32), poses=p1[src], weights_res=w[src],
                   nn_idx_res=idx[src], labels=np.asarray(labels)[src], N=n2)
        self.count += 1
        return out


class OracleFilter:
    """filter/filter.py:150-190 without clustering/annealing (the fixed-N headline step).

    State: poses (N,4,4) f32, weights (N,) f64.  Random draws are supplied by the caller
    (host mt19937 draws in parity mode, or the Philox spec streams in device mode).
    """

    def __init__(self, cb_poses, cb_embeddings, mesh_verts, pen_max=0.002):
        self.cb_poses = _f32(cb_poses)
        self.cb_feat = R3_SE3(self.cb_poses)
        self.emb = np.ascontiguousarray(cb_embeddings)
        self.verts = _f64(mesh_verts)
        self.pen_max = float(pen_max)

    def SE3_NN_idx(self, poses):
        return nn6(R3_SE3(poses), self.cb_feat)[0]

    def step(self, poses, odom, code, tn, rot_deg, u=None, mode="weighted_random", u32=None, softmax=True, scores=None,
             prop_override=None):
        """Returns dict with every intermediate the parity tests compare.  prop_override: continue from these propagated
        poses instead of the step's own (teacher forcing against a trace of the reference)."""
        out = {}
        p1 = propagate(poses, odom, tn, rot_deg) if prop_override is None else _f32(prop_override)
        out["poses_prop"] = p1
        feat = R3_SE3(p1)
        idx, d2 = nn6(feat, self.cb_feat)
        out["feat"], out["nn_idx"], out["nn_d2"] = feat, idx, d2
        if scores is None:  # batch mode passes the matrix-core scores (score_codebook_batch)
            scores = score_codebook(self.emb, code)
        out["scores"] = scores
        x = scores[idx]
        dist = nn3_dist(p1, self.verts)
        mask = ~(dist > self.pen_max)
        out["dist"], out["mask"] = dist, mask
        # fused-step spec (csrc/resample.hip k_tail_a/k_tail_b): e = exp(x - 1) (or x when the softmax is
        # skipped); weights = e / blocked_sum(e) * mask; the CDF is built from e * mask directly - the
        # normalisation by sum(e) cancels in prefix / total
        e, applied = softmax_numerators(x, softmax, shift=1.0)
        S = blocked_scan(e)[1] if applied else 1.0
        w_pre = e / S
        out["weights_pre"] = w_pre.copy()
        w = w_pre * mask
        out["weights"] = w
        out["drifted"] = bool(mask.sum() == 0)
        ridx, status = resample_indices(e * mask, mode, u=u, u32=u32)
        out["status"] = status
        if status:
            out["ridx"] = np.arange(len(w), dtype=np.int32)
        else:
            out["ridx"] = ridx
        out["poses"] = p1[out["ridx"]]
        out["weights_res"] = w[out["ridx"]]
        out["nn_idx_res"] = idx[out["ridx"]]
        return out
