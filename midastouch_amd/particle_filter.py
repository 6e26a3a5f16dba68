"""Particles / particle_filter / particle_rmse - the reference's filter-core surface on MI355X kernels.

Same class and method names, argument meaning, return types and guard behaviour as
`midastouch/modules/particle_filter.py` (citations below are to that file unless stated), so the
Hydra-driven runner can import these instead.  Every arithmetic step runs in libmidas_hip.so; host
code only draws the random numbers the reference draws (torch CPU generator, same order) and does the
annealing rule's three lines of scalar arithmetic.  No CPU fallback.
"""
from __future__ import annotations

import copy
import struct
from typing import Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import MidasError
from .tactile_tree import EmbeddingMatrix, NNCodes


class Particles:
    """[poses, weights, cluster labels]  (:33-78)"""

    poses = None
    weights = None
    labels = None

    def __init__(self, poses: torch.Tensor, weights: torch.Tensor = None, labels: torch.Tensor = None):
        self.poses = poses
        self.weights = weights if weights is not None else torch.ones(self.poses.shape[0], device=poses.device)
        self.labels = labels if labels is not None else torch.zeros(self.poses.shape[0], device=poses.device)

    def __len__(self):
        return self.poses.shape[0]

    def remove(self, idxs: torch.Tensor) -> None:
        self.poses = torch_delete(self.poses, idxs, dim=0)
        self.weights = torch_delete(self.weights, idxs)
        self.labels = torch_delete(self.labels, idxs)

    def add(self, poses: torch.Tensor, weights: torch.Tensor, labels: torch.Tensor) -> None:
        self.poses = torch.cat((self.poses, poses), dim=0)
        self.weights = torch.cat((self.weights, weights))
        self.labels = torch.cat((self.labels, labels))


def torch_delete(arr: torch.Tensor, idxs: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """np.delete for torch (:81-90) - via a keep-mask instead of the reference's N x r comparison matrix."""
    if idxs.nelement():
        keep = torch.ones(arr.size(dim), dtype=torch.bool, device=arr.device)
        keep[idxs.reshape(-1).to(arr.device)] = False
        return arr[keep]
    return arr


# ---- mesh input ---------------------------------------------------------------------------------
def load_mesh_vertices(mesh) -> np.ndarray:
    """Unique vertices (M,3) float64 of a mesh given as an STL path, an .npy/.npz path or an array.

    The reference uses trimesh.load(...).vertices (:108-109); trimesh is not a dependency here, so
    binary/ASCII STL is parsed directly and duplicate vertices are merged in first-occurrence order.
    """
    if isinstance(mesh, (np.ndarray, torch.Tensor)):
        return np.asarray(mesh.cpu() if isinstance(mesh, torch.Tensor) else mesh, dtype=np.float64).reshape(-1, 3)
    path = str(mesh)
    if path.endswith(".npy"):
        return np.load(path).astype(np.float64).reshape(-1, 3)
    if path.endswith(".npz"):
        return np.load(path)["vertices"].astype(np.float64).reshape(-1, 3)
    raw = open(path, "rb").read()
    ntri = struct.unpack_from("<I", raw, 80)[0] if len(raw) >= 84 else -1
    if len(raw) == 84 + 50 * ntri:  # binary STL
        rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), offset=84, count=ntri)
        tri = rec["v"].reshape(-1, 3)
    else:  # ASCII STL
        tri = np.array([[float(x) for x in line.split()[1:4]] for line in raw.decode(errors="ignore").splitlines()
                        if line.strip().startswith("vertex")], dtype=np.float32)
    _, first = np.unique(tri, axis=0, return_index=True)
    return tri[np.sort(first)].astype(np.float64)


def _cfg_get(node, *path):
    for key in path:
        node = node[key] if isinstance(node, dict) else getattr(node, key)
    return node


class particle_filter:
    """Update and propagation of SE(3) particles on a mesh (:93-469)."""

    def __init__(self, cfg, mesh_path, noise: float = 1.0, real: bool = False, downsample: int = 10, device=None):
        self.pen_max = float(_cfg_get(cfg, "tdn", "render", "pen", "max"))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        verts = load_mesh_vertices(mesh_path)
        self.mesh_scale = float(np.linalg.norm(verts.max(axis=0) - verts.min(axis=0)))  # trimesh mesh.scale
        self.mesh_vertices = verts[::downsample, :]
        self._mesh_tree = None  # built on first use (needs the GPU)
        which = "real" if real else "sim"

        def _noise(key):  # accepts {sim, real} (expt/ycb.yaml) and the scalar form of expt/mcmaster.yaml:19-20
            v = _cfg_get(cfg, "expt", "params", key)
            try:
                return float(_cfg_get(v, which))
            except (TypeError, KeyError, AttributeError):
                return float(v)

        self.motion_noise = {"mu": 0, "sig_r": _noise("noise_r"), "sig_t": _noise("noise_t")}
        self.particle_var = torch.tensor([float("inf")])
        self.topk_ties = "index"  # annealing(): whom torch.topk takes inside a tie ("aten_cpu": as the reference on the CPU)
        self.init_noise = [self.mesh_diagonal() / 3.0 * noise, 180.0 / 3.0 * noise]  # (:124-127)

    # ---------------------------------------------------------------------------------------------
    @property
    def mesh_kdtree(self) -> ops.Tree:
        if self._mesh_tree is None:
            self._mesh_tree = ops.Tree(torch.as_tensor(self.mesh_vertices).to(self.device, torch.float64))
        return self._mesh_tree

    def mesh_diagonal(self):
        return self.mesh_scale

    def init_filter(self, gt_pose: torch.Tensor = torch.eye(4), N: int = 10000) -> Particles:
        """gt (x) N  @  T(from_euler('zyx', N(0, s_r) deg), N(0, s_t))   (:129-145)"""
        from scipy.spatial.transform import Rotation as R

        dev = gt_pose.device
        tn = self._normal(0.0, self.init_noise[0], N).cpu()
        rotNoise = self._normal(0.0, self.init_noise[1], N).cpu()
        Rn = torch.tensor(R.from_euler("zyx", rotNoise, degrees=True).as_matrix())
        Tn = torch.zeros((N, 4, 4), dtype=gt_pose.dtype)
        Tn[:, :3, :3], Tn[:, :3, 3], Tn[:, 3, 3] = Rn, tn, 1
        initPoses = gt_pose.cpu()[None, :, :] @ Tn  # one-off at t = 0, on the host like the reference's CPU path
        return Particles(initPoses.to(dev))

    # ---------------------------------------------------------------------------------------------
    def motionModel(self, _particles: Particles, odom: torch.Tensor, multiplier: float = 1.0) -> Particles:
        """Odometry update with per-particle SE(3) noise (:359-377, :319-345)."""
        if multiplier < 1.0:
            multiplier = 1.0
        particles = copy.copy(_particles)
        N = particles.poses.shape[0]
        # same draws, same order, same generator as add_noise_to_odom (:326-335)
        tn = self._normal(self.motion_noise["mu"], float(multiplier) * self.motion_noise["sig_t"], N)
        rotNoise = self._normal(self.motion_noise["mu"], float(multiplier) * self.motion_noise["sig_r"], N)
        particles.poses = ops.propagate(particles.poses, odom, tn, rotNoise)
        return self.check_quats(particles)

    def _normal(self, mean: float, std: float, N: int) -> torch.Tensor:
        """torch.normal(mean, std, size=(N, 3)) as the reference draws it: on torch's CPU generator, or - after
        seed_device_stream() - from that generator's replica on the device (the same numbers, torch_rng.py; below 16 values ATen takes
        another path, which stays on the host: the stream is handed over for that call and taken back)."""
        stream = getattr(self, "torch_stream", None)
        if stream is None:
            return torch.normal(mean=mean, std=std, size=(N, 3))
        if 3 * N >= 16:
            return stream.normal(mean, std, (N, 3))
        g = torch.Generator()
        stream.to_host(g)
        z = torch.normal(mean=mean, std=std, size=(N, 3), generator=g)
        stream.from_host(g)
        return z

    def check_quats(self, particles: Particles) -> Particles:
        """Prune particles whose rotation gives a NaN / zero-norm quaternion (:347-357)."""
        flag, count = ops.check_poses(particles.poses)
        if int(count.item()):
            particles.remove(flag.nonzero())
        return particles

    # ---------------------------------------------------------------------------------------------
    def get_similarity(self, queries: torch.Tensor, targets, softmax=True) -> torch.Tensor:
        """Cosine score of the tactile code against target embeddings, optionally softmax-ed (:449-469).

        `targets` may be the NNCodes view returned by tactile_tree.SE3_NN (the codebook is scored once
        and the scalar gathered per particle), the EmbeddingMatrix of get_embeddings() (heat-map,
        filter/filter.py:213-215) or any (M, D) tensor.
        """
        if isinstance(targets, NNCodes):
            scores = targets.tree.codebook.score(queries)[0]
            x = ops.gather_f64(scores, targets.idx)
        elif isinstance(targets, EmbeddingMatrix):
            x = targets.tree.codebook.score(queries)[0]
        else:
            targets = torch.atleast_2d(targets)
            if not targets.is_cuda:
                raise MidasError("get_similarity needs targets on a HIP device; there is no CPU fallback")
            x = ops.Codebook(targets).score(queries)[0]
        if x.shape[0] == 1:
            return x.reshape(())  # .squeeze() of a single target; max == min so the softmax is skipped
        return ops.softmax_weights(x, softmax)

    def remove_invalid_particles(self, _particles: Particles, invalid_dist=None) -> Tuple[Particles, torch.Tensor]:
        """weights *= (distance to the mesh <= pen_max), in place like the reference (:379-403)."""
        particles = copy.copy(_particles)
        dist = ops.nn3_dist(self.mesh_kdtree, particles.poses)
        thr = self.pen_max if invalid_dist is None else float(invalid_dist)
        w = particles.weights
        if w.dtype == torch.float64 and w.is_contiguous():
            kept = ops.prune_(w, dist, thr)
        else:  # float32 weights of a never-updated Particles: same in-place semantics through torch
            m = (~(dist > thr)).to(w.dtype)
            w *= m
            kept = m.sum().to(torch.int32).reshape(1)
        drifted = kept[0] == 0
        return particles, drifted

    # ---------------------------------------------------------------------------------------------
    def cluster_particles(self, _particles: Particles, method: str = "euclidean", eps: float = 1e-2) -> Particles:
        """DBSCAN labels of the particles, min_samples = N / 5 (:208-228).  "euclidean" (what the loop uses, filter.py:183)
        runs on the device (midas_dbscan: exact float64 predicate on a uniform grid - dense up to 128 cells per axis, a hash
        table of the occupied cells beyond -, clusters numbered by their first core point like sklearn's scan) - the reference's host call takes seconds to minutes at 100k particles; "logmap" clusters
        the 6-d SE(3) logarithms on the device too (midas_dbscan_points: all pairs, the same predicate and numbering)."""
        particles = copy.copy(_particles)
        if method == "euclidean":
            # any extent (a hash table of the occupied cells takes over from the dense grid beyond 128 cells per axis) and any
            # number of clusters: there is no host fallback.  Flag 32: coordinates at infinity / beyond 2^21 cells per axis (6 km at
            # eps = 1e-2), where no cell structure applies; flag 64: a wide cloud of more than 2^20 particles (hash table capacity).
            labels, info = ops.dbscan(particles.poses, eps)
            flag = int(info[1].item())
            if flag & 32:
                raise ops.MidasError(f"cluster_particles: particle translations are not finite or span more than 2^21 cells of "
                                     f"{0.577 * eps:.3g} m; DBSCAN labels undefined")
            if flag & 64:
                raise ops.MidasError(f"cluster_particles: {len(particles)} particles spread over more than 128 cells of {0.577 * eps:.3g} m per axis - the "
                                     "hash table of occupied cells holds 2^20 particles; cluster a subset, raise eps, or shard the set")
            particles.labels = labels.to(torch.int64)
            return particles
        if method != "logmap":
            raise ValueError(method)
        from .pose import se3_log

        data = se3_log(particles.poses)  # [V^-1 t, omega], the embedding th.SE3.log_map gives (:219-220)
        # six dimensions: all pairs on the device (midas_dbscan_points, csrc/dbscan_nd.hip), sklearn's labels
        labels, info = ops.dbscan_points(data, eps, int(len(particles) / 5))
        if int(info[1].item()) < 0:
            raise ops.MidasError("cluster_particles(logmap): the cluster spread did not settle")
        particles.labels = labels.to(torch.int64).to(particles.labels.device)
        return particles

    def get_cluster_centers(self, _particles: Particles, method: str = "logmap") -> Tuple[torch.Tensor, torch.Tensor]:
        """Weighted pose mean + translation std per cluster label (:153-206).

        "quat_avg" (what filter.py:185 asks for) is Markley's quaternion mean (modules/pose.py:112-147) on the device:
        one pass over the particles accumulates every cluster's moments, a small kernel solves the symmetric 4x4
        eigenproblem the reference handed to the removed Tensor.eig (midas_cluster_centers, csrc/cluster.hip).
        "logmap" (pose.log_map_averaged, modules/pose.py:101-109): mean of the SE(3) logarithms, all clusters in one matrix product.
        """
        particles = copy.copy(_particles)
        poses, labels = particles.poses, particles.labels
        uniq = torch.unique(labels)
        if method == "quat_avg":  # K9: every cluster in one pass over the particles (ops.cluster_centers)
            cluster_poses, cluster_stds, _ = ops.cluster_centers(poses, particles.weights, labels, uniq)
            return cluster_poses, cluster_stds
        from .pose import logmap_cluster_centers

        # "logmap" (the signature's default): every cluster at once - one pass of SE(3) logarithms, one float64 matrix product
        # for all the clusters' sums, the exponentials batched (pose.py: logmap_cluster_centers)
        return logmap_cluster_centers(poses, particles.weights, labels)

    def _anneal_plan(self, n: int, var, floor: int):
        """The rule of :413-447 as a plan (mode, k): 1 = drop the k particles of smallest weight, 2 = duplicate the k of
        largest weight, None = leave the set alone.  `var` and `particle_var` keep whatever scalar type the caller works in
        (the loop hands in float32 tensors, so the ratio and the counts are float32 arithmetic there)."""
        if torch.isinf(torch.as_tensor(self.particle_var)).all():  # first frame: remember spread and size
            self.particle_var, self.init_particles = var, n
            return None
        if var == 0.0:  # converged to a single pose
            return None
        ratio = var / self.particle_var
        self.particle_var = var
        if ratio < 1:
            k = min(int((1.0 - ratio) * n), abs(n - floor), n // 3)
            return (1, k) if k > 0 else None
        if ratio > 1:
            k = min(int((ratio - 1.0) * n), n // 3)
            return (2, k) if k > 0 and k + n <= self.init_particles else None
        return None

    def annealing(self, _particles: Particles, var, floor: int = 1000) -> Particles:
        """Adapt the particle count to the cluster spread (:405-447).  The selection runs on the device and comes back as one
        index list the three arrays are gathered through.  `self.topk_ties`: "index" (default) - radix select of the k-th
        weight, ties to the smaller index, as torch.topk resolves them on CUDA; "aten_cpu" (set by seed_device_stream, the
        mode that replays the reference's CPU run) - the members and the order torch.topk returns on the CPU (topk_aten.hip)."""
        particles = copy.copy(_particles)
        plan = self._anneal_plan(len(particles.weights), var, floor)
        if plan is None:
            return particles
        src = ops.anneal_select(particles.weights, *plan, ties=self.topk_ties)
        self.last_anneal_indices = src
        particles.poses = ops.gather_rows(particles.poses, src)
        particles.weights = ops.gather_rows(particles.weights, src)
        particles.labels = ops.gather_rows(particles.labels, src)
        return particles

    # ---------------------------------------------------------------------------------------------
    def seed_device_stream(self, seed: int):
        """torch.manual_seed(seed), kept on the device: every draw of the class surface - `init_filter`'s and `motionModel`'s
        torch.normal calls (:129-130, :326-335), the uniforms `resampler("weighted_random")`'s torch.multinomial would consume (:245) -
        then comes from the device replica of torch's CPU generator (midastouch_amd/torch_rng.py, torch_normal.py): the numbers of a
        run of the reference under that seed, bit for bit, nothing generated on the host.  `seed=None` returns to the host generator.  A seeded run replays the reference on the CPU, so `annealing` then
        also takes torch.topk's CPU choice inside a tie (`topk_ties`)."""
        from .torch_rng import TorchCpuStream
        self.torch_stream = None if seed is None else TorchCpuStream(seed, self.device)
        self.topk_ties = "index" if seed is None else "aten_cpu"
        return self.torch_stream

    def resampler(self, _particles: Particles, resample: str = "weighted_random") -> Particles:
        """Importance resampling (:230-307).

        "weighted_random" == torch.multinomial(p, N, replacement=True): a float64 CDF and one binary
        search per draw, the draws being torch.rand(N, dtype=float64) of the CPU generator (the stream
        torch.multinomial itself consumes).  "low_var"/"low_var_batch" == systematic resampling with the
        single float32 torch.rand(1) offset (the batch variant's int16 counter overflow for N > 32767 and
        its first-element miscount, :271,:279, are not reproduced).
        """
        particles = copy.copy(_particles)
        nSamples = len(particles)
        if nSamples == 0:
            return particles
        cdf, status = ops.cdf(particles.weights)
        if int(status.item()) != 0:  # all-zero or NaN weights: return the input (:240-241)
            return particles
        if resample == "weighted_random":
            # the draws torch.multinomial would take from the CPU generator: from that generator itself, or - after
            # seed_device_stream() - from its replica on the device (no host generator, no upload of N float64)
            stream = getattr(self, "torch_stream", None)
            u = stream.rand64(nSamples) if stream is not None else torch.rand(nSamples, dtype=torch.float64)
            idxs = ops.resample_search(cdf, nSamples, _lib.RESAMPLE_MULTINOMIAL, u=u)
        elif resample in ("low_var", "low_var_batch"):
            offset = torch.rand(1)
            idxs = ops.resample_search(cdf, nSamples, _lib.RESAMPLE_SYSTEMATIC, u32=float(offset.item()))
        else:
            raise ValueError(f"unknown resampling mode {resample!r}")
        self.last_resample_indices = idxs
        return Particles(ops.gather_rows(particles.poses, idxs), ops.gather_rows(particles.weights, idxs),
                         ops.gather_rows(particles.labels, idxs))


def particle_rmse(_particles, gt_pose: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """RMSE of [translation (m), rotation angle (deg)] of the particles w.r.t. gt (:472-496)."""
    poses = _particles.poses if isinstance(_particles, Particles) else _particles
    out = ops.rmse(poses, gt_pose)
    return out[0].float(), out[1].float()
