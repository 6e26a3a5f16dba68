"""LoopEngine: the reference's whole loop body (`filter/filter.py:150-190`) as one C-ABI call per frame, on a particle
set whose size lives in device memory.

FilterEngine / PipelinedFilterEngine run the fixed-N part of the frame (motion, NN, similarity, prune, resample).  The
reference's loop also clusters (DBSCAN every 50th frame, `particle_filter.py:208-228`), takes the cluster centres
(:153-206) and anneals the particle count on their spread (:405-447) between the update and the resample, so N changes
every frame.  `midas_loop_step` does all of it on the device - DBSCAN included, top-k selection / compaction /
duplication included - with the live count in the control block, every array sized to the initial particle count
(annealing never grows the set beyond it, :439-440), so a frame is enqueued without reading anything back.  Per-frame
results (rmse, N, cluster centres ...) go to a device log that is read once, when somebody asks.

Draws: device Philox streams keyed by (seed, frame) by default; `tn` / `rot` / `u` take the host draws of the reference
(torch CPU generator, its order: tn, rot, then the resampler's uniforms) - the host then has to know the particle count,
i.e. `n` (one small read-back per frame), and may split the frame with `phases` to draw the uniforms once the annealed
size is known, as the reference does.  Such a replay also wants `topk_ties="aten_cpu"`: inside a tie (the normal case:
particles on one codebook entry share a weight) annealing's `torch.topk` keeps whomever ATen's CPU kernel happens to reach,
and the device then walks that kernel's algorithm (topk_aten.hip) instead of its own radix select (ties by index).
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import LoopArgs, MidasError, _ptr
from .engine import operand

ALL_PHASES = _lib.LOOP_FRONT | _lib.LOOP_DBSCAN | _lib.LOOP_ANNEAL | _lib.LOOP_RESAMPLE


class LoopEngine:
    def __init__(self, cb_poses, cb_embeddings, mesh_vertices, num_particles: int, *, sig_t=2e-4, sig_r=0.5, pen_max=0.002,
                 seed=4000, softmax=True, resample="weighted_random", floor: int = 1000, eps: float = 1e-2, cluster: bool = True,
                 cluster_every: int = 50, log_frames: int = 4096, device=None,
                 topk_ties: str = "index"):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = _lib.context(dev)
        self.device = d = self.ctx.device
        if hasattr(cb_poses, "SE3_NN") and cb_embeddings is None:  # a tactile_tree already on the device: share its index
            tt = cb_poses
            self.cb_poses, self.cb_feat, self.tree6, self.codebook = tt.poses, tt.logmap_pose, tt.tree, tt.codebook
        else:
            self.cb_poses = torch.as_tensor(cb_poses).to(d, torch.float32).contiguous()
            self.cb_feat = ops.se3_feature(self.cb_poses)
            self.tree6 = ops.Tree(self.cb_feat)
            self.codebook = ops.Codebook(torch.as_tensor(cb_embeddings).to(d))
        self.tree3 = mesh_vertices if isinstance(mesh_vertices, ops.Tree) else ops.Tree(torch.as_tensor(mesh_vertices).to(d, torch.float64))
        if getattr(self.tree6, "_mesh", None) is not self.tree3:  # vertex lists of this mesh not yet on the codebook index
            self.tree6.attach_mesh(self.tree3, self.cb_poses)
        self.K, self.D = self.codebook.K, self.codebook.D
        self.sig_t, self.sig_r, self.pen_max = float(sig_t), float(sig_r), float(pen_max)
        self.seed, self.softmax, self.floor, self.eps = int(seed), bool(softmax), int(floor), float(eps)
        self.cluster, self.cluster_every = bool(cluster), max(int(cluster_every), 1)
        # whom annealing's torch.topk takes inside a tie: "index" (torch's CUDA rule, the radix select) or "aten_cpu" (the
        # reference as it runs on the CPU - the rule of a seeded replay; topk_aten.hip)
        self.topk_ties = {"index": _lib.TOPK_TIES_INDEX, "aten_cpu": _lib.TOPK_TIES_ATEN_CPU}[topk_ties]
        self.mode = {"weighted_random": _lib.RESAMPLE_MULTINOMIAL, "low_var": _lib.RESAMPLE_SYSTEMATIC,
                     "low_var_batch": _lib.RESAMPLE_SYSTEMATIC}[resample]
        self.cap = cap = int(num_particles)
        if cap < 1 or cap > (1 << 20):
            raise MidasError("LoopEngine holds 1 .. 2^20 particles")
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=d)  # noqa: E731
        self.ctl_i, self.ctl_d = z(32, torch.int32), z(104 if os.environ.get("MIDAS_ANNEAL_CLOCKS") else 16, torch.float64)  # (104: profiling builds)
        self._poses, self.poses_prop = z((cap, 4, 4), torch.float32), z((cap, 4, 4), torch.float32)
        self._hint = torch.full((cap,), -1, dtype=torch.int32, device=d)
        self._nn_idx, self._valid = z(cap, torch.int32), z(cap, torch.uint8)
        self._x, self._e, self._w, self._w_res = (z(cap, torch.float64) for _ in range(4))
        self._labels, self._labels_next = z(cap, torch.int32), z(cap, torch.int32)
        self._src, self._ridx = z(cap, torch.int32), z(cap, torch.int32)
        self._scores = z(self.K, torch.float64)
        self._part_rmse = z(2 * ((cap + 63) // 64), torch.float64)
        self._cl_poses, self._cl_stds = z((_lib.LOOP_MAX_CLUSTERS, 4, 4), torch.float32), z((_lib.LOOP_MAX_CLUSTERS, 3), torch.float32)
        self.log_frames = int(log_frames)
        self._log = z((self.log_frames, _lib.LOOP_LOG_DOUBLES), torch.float64)
        self.telemetry = z(16, torch.int64)
        import os as _os
        self.sparse_scores = self.codebook.emb.dtype == torch.float32 and self.D in (128, 256, 512, 1024) and \
            _os.environ.get("MIDAS_DENSE_SCORES", "0") != "1"
        self._stamps, self._epoch = z(self.K, torch.int32), 0  # sparse scoring (include/midas_hip.h score_stamps_dev)
        # {frames completed, live count} as the device last reported them, in pinned host memory: lets step() size its launches
        # to the live set without waiting for anything (an upper bound is all it needs, see _count_bound)
        self._mirror = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._frames_at_reset, self._n_at_reset, self._steps_at_reset, self._grid_n = 0, cap, 0, cap
        self.max_ahead = 1         # frames the host may enqueue ahead of the device's last report (None: no limit)
        self.allow_frozen = True   # see _frozen()
        self._init_known = None    # init_particles as the host knows it (None: annealing's first call will take the live count)
        self.step_count = 0        # frames enqueued (Philox counter, log row)
        self._n_host = None        # particle count as last known by the host (None: ask the device)
        self._pending_phases = 0
        self.use_hint = True
        # the scratch every phase combination of a frame at this capacity can ask for (DBSCAN's cell tables 84 MB + 41 B per
        # particle, the cluster-centre partials 72 B per particle, selection tables) reserved now: no frame of the run
        # allocates (MIDAS_SCRATCH_LOG=1 prints nothing after this line; the cold frame of BENCH_r03's floor_N run)
        self.ctx.call("midas_scratch_reserve", (128 << 20) + 256 * cap)

    # ---- state ----------------------------------------------------------------------------------------------------
    def set_particles(self, poses: torch.Tensor, labels: torch.Tensor = None, reset_annealing: bool = True):
        """Start (or restart) from `poses` (n <= capacity); labels default to 0 like `Particles` (particle_filter.py:47)."""
        poses = torch.as_tensor(poses).to(self.device, torch.float32).reshape(-1, 4, 4)
        n = poses.shape[0]
        if n < 1 or n > self.cap:
            raise MidasError(f"{n} particles do not fit the engine's capacity of {self.cap}")
        self._poses[:n].copy_(poses)
        self._hint.fill_(-1)
        if labels is None:
            self._labels.zero_()
            ncl = 1  # label 0 everywhere
        else:
            labels = torch.as_tensor(labels).to(self.device).to(torch.int32)
            self._labels[:n].copy_(labels)
            ncl = int(labels.max().item()) + 1
        ci = torch.zeros(32, dtype=torch.int32)
        if not reset_annealing:
            old = self.ctl_i.cpu()
            for k in (_lib.LOOP_I_INIT, _lib.LOOP_I_VARSET, _lib.LOOP_I_FRAME):
                ci[k] = old[k]
            self._init_known = int(old[_lib.LOOP_I_INIT]) if int(old[_lib.LOOP_I_VARSET]) else None
        else:
            self.ctl_d.zero_()
            self._init_known = None
        ci[_lib.LOOP_I_N] = n
        ci[_lib.LOOP_I_NSET] = n
        ci[_lib.LOOP_I_NCL] = ncl
        self.ctl_i.copy_(ci)
        self._n_host = n
        self._pending_phases = 0
        torch.cuda.current_stream(self.device).synchronize()  # the control block is in place, nothing older is in flight
        self._mirror[0], self._mirror[1] = int(ci[_lib.LOOP_I_FRAME]), n
        self._frames_at_reset, self._n_at_reset, self._steps_at_reset = int(ci[_lib.LOOP_I_FRAME]), n, self.step_count

    def _count_bound(self) -> int:
        """An upper bound of the live count at the start of the frame about to be enqueued, from what the device reported
        last: annealing adds at most n // 3 particles per frame (particle_filter.py:437) and never exceeds the capacity."""
        if not self.cluster:
            return self._n_at_reset
        enq = self.step_count - self._steps_at_reset
        # stay at most `max_ahead` frames in front of the device: the bound below grows by 4/3 per unreported frame, and a
        # host that runs dozens of frames ahead could only ever assume the full capacity.  One frame in the queue while the
        # next is being enqueued keeps the device busy (enqueueing a frame takes less than running it); the wait is a look
        # at pinned memory.
        if self.max_ahead is not None and enq - (int(self._mirror[0]) - self._frames_at_reset) > self.max_ahead:
            import time
            t0 = time.perf_counter()
            while enq - (int(self._mirror[0]) - self._frames_at_reset) > self.max_ahead and time.perf_counter() - t0 < 0.05:
                pass
        done = int(self._mirror[0]) - self._frames_at_reset   # frames of this run the device has finished
        n = int(self._mirror[1]) if done > 0 else self._n_at_reset
        lag = (self.step_count - self._steps_at_reset) - max(done, 0)  # frames enqueued since that report
        for _ in range(max(lag, 0)):
            n += n // 3
            if n >= self.cap:
                return self.cap
        return min(n, self.cap)

    def _frozen(self) -> bool:
        """Annealing cannot change the set: the live count equals `floor` and the count annealing starts (started) from
        (particle_filter.py:421-446: a removal needs |n - floor| > 0, a duplication k + n <= init_particles) - true from a start with
        n == floor on, for good.  The ANNEAL phase then runs its decision only, not the selection's ten launches; the device checks
        the statement (log row err bit 7).  `allow_frozen = False` keeps the launches (tests)."""
        if not (self.cluster and self.allow_frozen):
            return False
        n0 = self._n_at_reset
        return self.floor == n0 and self._init_known in (None, n0)

    def set_annealing_state(self, particle_var: float, init_particles: int):
        """particle_filter.particle_var / init_particles (particle_filter.py:413-417) - for restarts and tests."""
        self._init_known = int(init_particles)
        ci = self.ctl_i.cpu()
        ci[_lib.LOOP_I_VARSET] = 0 if np.isinf(particle_var) else 1
        ci[_lib.LOOP_I_INIT] = int(init_particles)
        self.ctl_i.copy_(ci)
        cd = self.ctl_d.cpu()
        cd[_lib.LOOP_D_VARPREV] = float(np.float32(particle_var)) if not np.isinf(particle_var) else 0.0
        self.ctl_d.copy_(cd)

    def project_to_codebook(self):
        """poses := codebook pose nearest to each particle (filter/filter.py:159-160)."""
        n = self.n
        idx = ops.nn6(self.tree6, ops.se3_feature(self._poses[:n]))
        self._poses[:n].copy_(ops.gather_rows(self.cb_poses, idx))
        self._hint[:n].copy_(idx)
        return idx

    @property
    def n(self) -> int:
        """Live particle count (reads the control block when the host does not know it)."""
        if self._n_host is None:
            self._n_host = int(self.ctl_i[_lib.LOOP_I_N].item())
        return self._n_host

    # views of the live particle set / the latest frame (each reads the count: a synchronisation)
    poses = property(lambda self: self._poses[:self.n])
    weights_res = property(lambda self: self._w_res[:self.n])
    labels = property(lambda self: self._labels[:self.n])
    hint = property(lambda self: self._hint[:self.n])
    ridx = property(lambda self: self._ridx[:self.n])

    def frame_view(self):
        """The latest COMPLETED frame, for tests and inspection (one synchronisation): its log record plus views of the
        per-particle arrays - before annealing (n entries: poses_prop, nn_idx, valid, weights, labels_frame), the annealed set
        (n_after entries: src, ridx) and the resampled particle set (n_after entries: poses, weights_res, labels, hint)."""
        if self.step_count == 0 or self._pending_phases:
            raise MidasError("frame_view needs a completed frame")
        rec = self.read_log(self.step_count - 1, self.step_count)[0]
        nb, ns = rec["n"], rec["n_after"]
        rec.update(poses_prop=self.poses_prop[:nb], nn_idx=self._nn_idx[:nb], valid=self._valid[:nb], weights=self._w[:nb],
                   labels_frame=self._labels_prev[:nb], src=self._src[:ns], ridx=self._ridx[:ns], poses=self._poses[:ns],
                   weights_res=self._w_res[:ns], labels=self._labels[:ns], hint=self._hint[:ns],
                   ctl_i=self.ctl_i.cpu().numpy(), ctl_d=self.ctl_d.cpu().numpy())
        self._n_host = ns
        return rec

    # ---- one frame ------------------------------------------------------------------------------------------------
    def step(self, odom, code, gt=None, tn=None, rot=None, u=None, u32=-1.0, multiplier: float = 1.0, dbscan=None,
             phases: int = None, std_override=None, unit_weights: bool = False):
        """Enqueues one frame (or the given phases of it).  dbscan: None = on every `cluster_every`-th frame (count % 50 ==
        0, filter.py:182), True / False to force.  Host draws tn / rot (n, 3) and u (>= n_set,) as in FilterEngine.step."""
        d = self.device
        frame_start = self._pending_phases == 0
        if phases is None:
            phases = ALL_PHASES & ~self._pending_phases if not frame_start else ALL_PHASES
        if not self.cluster:
            phases &= ~(_lib.LOOP_DBSCAN | _lib.LOOP_ANNEAL)
        elif dbscan is False or (dbscan is None and self.step_count % self.cluster_every != 0):
            phases &= ~_lib.LOOP_DBSCAN
        a = LoopArgs()
        a.cap = self.cap
        a.ctl_i, a.ctl_d = _ptr(self.ctl_i), _ptr(self.ctl_d)
        a.poses, a.poses_prop = _ptr(self._poses), _ptr(self.poses_prop)
        a.hint, a.nn_idx, a.valid = _ptr(self._hint), _ptr(self._nn_idx), _ptr(self._valid)
        a.x, a.e, a.weights, a.weights_out = _ptr(self._x), _ptr(self._e), _ptr(self._w), _ptr(self._w_res)
        a.labels, a.labels_out = _ptr(self._labels), _ptr(self._labels_next)
        a.src, a.ridx, a.scores = _ptr(self._src), _ptr(self._ridx), _ptr(self._scores)
        a.cb_poses = _ptr(self.cb_poses)
        a.cluster_poses, a.cluster_stds = _ptr(self._cl_poses), _ptr(self._cl_stds)
        a.log = C.c_void_p(self._log.data_ptr() + (self.step_count % self.log_frames) * _lib.LOOP_LOG_DOUBLES * 8)
        keep = []
        if phases & _lib.LOOP_FRONT:
            if (tn is None) != (rot is None):
                raise MidasError("tn and rot (the motion model's host draws) come together or not at all")
            odom = operand(odom, "odom", torch.float32, (4, 4), d)
            code = operand(code, "tactile code", torch.float64, (self.D,), d)
            gt = operand(gt, "gt pose", torch.float32, (4, 4), d)
            if tn is not None:
                n = self.n
                tn, rot = operand(tn, "tn", torch.float32, (n, 3), d), operand(rot, "rot", torch.float32, (n, 3), d)
            a.odom16, a.code, a.gt16, a.tn, a.rot = _ptr(odom), _ptr(code), _ptr(gt), _ptr(tn), _ptr(rot)
            a.part_rmse = _ptr(self._part_rmse) if gt is not None else None
            keep += [odom, code, gt, tn, rot]
        if phases & _lib.LOOP_RESAMPLE and u is not None:
            u = torch.as_tensor(u).to(d, torch.float64).contiguous().reshape(-1)
            a.u = _ptr(u)
            keep.append(u)
        a.u32 = float(u32)
        mul = max(float(multiplier), 1.0)
        a.std_t, a.std_r = (mul * self.sig_t, mul * self.sig_r) if std_override is None else std_override
        a.seed, a.step = self.seed, self.step_count
        a.prune_thr, a.softmax, a.resample_mode = self.pen_max, int(self.softmax), self.mode
        a.floor, a.eps = self.floor, self.eps
        a.unit_weights = int(bool(unit_weights))
        if frame_start:
            self._grid_n = self._count_bound()
        a.host_mirror = C.c_void_p(self._mirror.data_ptr())
        a.grid_n = self._grid_n
        a.anneal_small = int(self._grid_n <= 16384)
        a.anneal_frozen = int(self._frozen())
        a.topk_ties = self.topk_ties
        a.telemetry = _ptr(self.telemetry)
        if self.sparse_scores and phases & _lib.LOOP_FRONT:
            from .engine import advance_epoch
            a.score_stamps, a.score_epoch = _ptr(self._stamps), advance_epoch(self)
        self._keep = keep
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_loop_step(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a), int(phases)))
        self._pending_phases |= phases
        if phases & _lib.LOOP_RESAMPLE:  # frame complete
            self._labels_prev = self._labels
            self._labels, self._labels_next = self._labels_next, self._labels
            self._pending_phases = 0
            self.step_count += 1
            self._n_host = None if self.cluster else self._n_host

    # ---- results --------------------------------------------------------------------------------------------------
    def read_log(self, first: int = 0, last: int = None, strict: bool = True):
        """Per-frame records of frames [first, last) as a list of dicts (one read-back): frame, n (before annealing),
        n_after, rmse_t, rmse_r, kept, drifted, status, mode, k, clusters, var, cluster_poses (C,4,4), cluster_stds (C,3), err.
        Conditions that leave a frame's particles or labels UNDEFINED (err bit 2: the live count exceeded the bound the launches
        were sized for; bits 5 / 6: DBSCAN's cell structure did not apply) raise MidasError once every row is parsed (the records
        are attached to the exception as `.records`); strict=False reports them as warnings like the benign ones (cluster limits)."""
        last = self.step_count if last is None else min(last, self.step_count)
        first = max(first, last - self.log_frames)
        rows = self._log.cpu().numpy()
        out, fatal = [], []
        import warnings
        for f in range(first, last):
            L = rows[f % self.log_frames]
            npres = int(L[10])
            cl = L[16:16 + 19 * min(npres, 8)].reshape(-1, 19)
            err = int(L[15])
            # condition bits of THIS frame (the device clears the word once the row holds it)
            if err & (1 | 2):
                warnings.warn(f"frame {f}: more than {_lib.LOOP_MAX_CLUSTERS - 2} clusters in one frame - the labels beyond that limit, and the "
                              "annealing driven by them, are not the reference's (min_samples = n / 5 allows about five)")
            if err & 8:
                warnings.warn(f"frame {f}: {npres} cluster labels present, the log row keeps the centres of the first 8")
            if err & 4:
                fatal.append(f"frame {f}: the live particle count exceeded the bound the launches were sized for; "
                             "the particles beyond it were not processed in this frame")
            if err & 32:
                fatal.append(f"frame {f}: DBSCAN saw non-finite particle translations or a cloud of more than 2^21 cells per axis; labels undefined")
            if err & 64:
                fatal.append(f"frame {f}: DBSCAN on a cloud wider than 128 cells per axis with more than 2^20 particles (hash table capacity); labels undefined")
            if err & 128:
                fatal.append(f"frame {f}: the engine stated that annealing could not act (live count == floor == init_particles) and the "
                             "rule wanted to: the set was left as it was")
            out.append(dict(frame=f, n=int(L[1]), n_after=int(L[2]), rmse_t=float(L[3]), rmse_r=float(L[4]), kept=int(L[5]),
                            drifted=bool(L[6]), status=int(L[7]), mode=int(L[8]), k=int(L[9]), clusters=npres, var=float(L[11]),
                            S=float(L[12]), raw=bool(L[13]), ncl=int(L[14]), err=int(L[15]),
                            cluster_poses=cl[:, :16].reshape(-1, 4, 4).astype(np.float32), cluster_stds=cl[:, 16:].astype(np.float32)))
        if fatal and strict:
            e = MidasError("; ".join(fatal))
            e.records = out
            raise e
        for msg in fatal:
            warnings.warn(msg)
        return out
