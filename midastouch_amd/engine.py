"""Device-resident filter engine: the reference's per-frame loop body as ONE C-ABI call per frame.

`FilterEngine.step()` is the fast path behind the north-star aliases `step()/update_weights()/
resample()` (midastouch_amd/filter.py): score codebook -> propagate -> feature -> NN -> gather score ->
softmax -> prune -> CDF -> resample -> gather, all on the GPU with no host synchronisation
(reference loop body: filter/filter.py:150-190, clustering/annealing excluded = fixed N).

Two random-draw modes:
  * parity mode  - the caller supplies the host draws of the reference (torch CPU mt19937:
                   tn, rot from torch.normal in that order, then N float64 uniforms);
  * device mode  - the kernels draw from the Philox spec streams keyed by (seed, step).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import LazyArgs, LazyFlushArgs, MidasError, StepArgs, _ptr


EPOCH_LIMIT = 0x3FFFFFF0  # (bits 31:30 of a stamp count the frames a listed row went unused: csrc/midas_internal.hpp)


def advance_epoch(eng, n: int = 1) -> int:
    """First of n consecutive sparse-scoring epochs of an engine (`_epoch`, `_stamps`, optionally `_score_list`): non-zero,
    spaced by 2 when the engine keeps a prediction list (the value between two epochs tags the listed rows), never reused
    while the stamps live - before the 32-bit counter could wrap (2.5 days at 20k frames/s) the stamps and the list lengths
    are zeroed and the count restarts, so a stale stamp can never equal a current epoch."""
    lst = getattr(eng, "_score_list", None)
    inc = 2 if lst is not None else 1
    if eng._epoch + inc * n >= EPOCH_LIMIT:
        eng._stamps.zero_()
        if lst is not None:
            lst[:2].zero_()
        eng._epoch = 0
    first = eng._epoch + inc
    eng._epoch += inc * n
    return first


def operand(t, name: str, dtype, shape, device):
    """A frame operand as the C ABI reads it: on `device`, `dtype`, contiguous, exactly `shape` elements (the kernels
    take raw pointers and check nothing).  Host tensors (the reference's CPU-generator draws), float32 tactile codes
    (what the TCN emits before its .double()) and strided views are converted; a wrong element count raises."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor) and t.dtype == dtype and t.device == device and t.shape == tuple(shape) and t.is_contiguous():
        return t  # (already what the kernels read: the common case of a run() fed from device tensors)
    t = torch.as_tensor(t)
    n = 1
    for s_ in shape:
        n *= int(s_)
    if t.numel() != n:
        raise MidasError(f"{name}: expected shape {tuple(shape)} ({n} values), got {tuple(t.shape)}")
    if t.device != device or t.dtype != dtype:
        t = t.to(device=device, dtype=dtype)
    return t.contiguous().reshape(shape)


class FilterEngine:
    def __init__(self, cb_poses, cb_embeddings, mesh_vertices, num_particles: int, *, sig_t=2e-4, sig_r=0.5,
                 pen_max=0.002, seed=4000, softmax=True, resample="weighted_random", device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = _lib.context(dev)
        self.device = self.ctx.device
        f32 = dict(dtype=torch.float32, device=self.device)
        if hasattr(cb_poses, "SE3_NN") and cb_embeddings is None:  # a tactile_tree already on the device: share its index
            tt = cb_poses
            self.cb_poses, self.cb_feat, self.tree6, self.codebook = tt.poses, tt.logmap_pose, tt.tree, tt.codebook
        else:
            self.cb_poses = torch.as_tensor(cb_poses).to(**f32).contiguous()
            self.cb_feat = ops.se3_feature(self.cb_poses)
            self.tree6 = ops.Tree(self.cb_feat)
            self.codebook = ops.Codebook(torch.as_tensor(cb_embeddings).to(self.device))
        self.tree3 = mesh_vertices if isinstance(mesh_vertices, ops.Tree) else ops.Tree(torch.as_tensor(mesh_vertices).to(self.device, torch.float64))
        if getattr(self.tree6, "_mesh", None) is not self.tree3:
            self.tree6.attach_mesh(self.tree3, self.cb_poses)
        self.K, self.D = self.codebook.K, self.codebook.D
        self.sig_t, self.sig_r, self.pen_max = float(sig_t), float(sig_r), float(pen_max)
        self.seed, self.softmax = int(seed), bool(softmax)
        self.mode = {"weighted_random": _lib.RESAMPLE_MULTINOMIAL, "low_var": _lib.RESAMPLE_SYSTEMATIC,
                     "low_var_batch": _lib.RESAMPLE_SYSTEMATIC}[resample]
        self.N = int(num_particles)
        N = self.N
        self.poses = torch.zeros((N, 4, 4), **f32)
        self.poses_prop = torch.zeros((N, 4, 4), **f32)
        self.weights = torch.zeros(N, dtype=torch.float64, device=self.device)      # pre-resample, masked
        self.weights_res = torch.ones(N, dtype=torch.float64, device=self.device)   # gathered by resample
        self.nn_idx = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.hint = torch.full((N,), -1, dtype=torch.int32, device=self.device)
        self.hint_next = torch.full((N,), -1, dtype=torch.int32, device=self.device)
        self.ridx = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.rmse = torch.zeros(2, dtype=torch.float64, device=self.device)
        # 16 cumulative counters ([0], [1] = tree-search fallbacks); with MIDAS_ABLATE=4 (profiling) the kernel
        # also keeps 16 statistics slots per wave behind them
        extra = 16 * ((N + 15) // 16) if int(os.environ.get("MIDAS_ABLATE", "0")) & 4 else 0
        self.telemetry = torch.zeros(16 + extra, dtype=torch.int64, device=self.device)
        self.step_count = 0
        self.use_hint = True
        # sparse scoring: only the rows that are some particle's nearest entry are scored, by the particle kernels
        # themselves (stamps of the frame that last scored a row; include/midas_hip.h score_stamps_dev).  Same scores.
        _os = os
        self.sparse_scores = self.codebook.emb.dtype == torch.float32 and self.D in (128, 256, 512, 1024) and \
            _os.environ.get("MIDAS_DENSE_SCORES", "0") != "1"
        self._stamps = torch.zeros(self.K, dtype=torch.int32, device=self.device)
        self._epoch = 0

    def _next_epoch(self, n: int = 1) -> int:
        """First of n consecutive score epochs (see advance_epoch)."""
        return advance_epoch(self, n)

    # ---- state ----------------------------------------------------------------------------------
    def set_particles(self, poses: torch.Tensor):
        poses = torch.as_tensor(poses).to(self.device, torch.float32).contiguous()
        if poses.shape != (self.N, 4, 4):
            raise MidasError(f"expected ({self.N},4,4) poses, got {tuple(poses.shape)}")
        self.poses.copy_(poses)
        self.hint.fill_(-1)

    def project_to_codebook(self):
        """poses := codebook pose nearest to each particle (filter/filter.py:159-160)."""
        idx = ops.nn6(self.tree6, ops.se3_feature(self.poses))
        self.poses.copy_(ops.gather_rows(self.cb_poses, idx))
        self.hint.copy_(idx)
        return idx

    # ---- one frame ------------------------------------------------------------------------------
    def step(self, odom, code, gt=None, tn=None, rot=None, u=None, u32=-1.0, multiplier: float = 1.0):
        """Runs one frame; results stay on the device (self.poses, self.weights, self.ridx ...)."""
        odom, code, gt, tn, rot, u = self._operands(odom, code, gt, tn, rot, u)
        a = StepArgs()
        a.N = self.N
        a.poses_in, a.poses_prop, a.poses_out = _ptr(self.poses), _ptr(self.poses_prop), _ptr(self.poses)
        a.weights, a.weights_out = _ptr(self.weights), _ptr(self.weights_res)
        a.hint_in = _ptr(self.hint) if self.use_hint else None
        a.nn_idx, a.hint_out, a.ridx = _ptr(self.nn_idx), _ptr(self.hint_next), _ptr(self.ridx)
        a.odom16, a.code = _ptr(odom), _ptr(code)
        a.gt16 = _ptr(gt)
        a.rmse = _ptr(self.rmse) if gt is not None else None
        a.tn, a.rot, a.u = _ptr(tn), _ptr(rot), _ptr(u)
        a.u32 = float(u32)
        mul = max(float(multiplier), 1.0)  # motionModel clamps multiplier >= 1 (particle_filter.py:365-366)
        a.std_t, a.std_r = mul * self.sig_t, mul * self.sig_r
        a.seed, a.step = self.seed, self.step_count
        a.prune_thr = self.pen_max
        a.softmax, a.resample_mode = int(self.softmax), self.mode
        a.status = _ptr(self.status)
        a.telemetry = _ptr(self.telemetry)
        if self.sparse_scores:
            a.score_stamps, a.score_epoch = _ptr(self._stamps), self._next_epoch()
        self._keep = (odom, code, gt, tn, rot, u)  # keep operands alive until the stream has consumed them
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_filter_step(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h,
                                                      C.byref(a)))
        self.hint, self.hint_next = self.hint_next, self.hint
        self.step_count += 1

    def _operands(self, odom, code, gt, tn, rot, u):
        d, N = self.device, self.N
        if (tn is None) != (rot is None):
            raise MidasError("tn and rot (the motion model's host draws) come together or not at all")
        return (operand(odom, "odom", torch.float32, (4, 4), d), operand(code, "tactile code", torch.float64, (self.D,), d),
                operand(gt, "gt pose", torch.float32, (4, 4), d), operand(tn, "tn", torch.float32, (N, 3), d),
                operand(rot, "rot", torch.float32, (N, 3), d), operand(u, "u", torch.float64, (N,), d))

    # ---- profiling --------------------------------------------------------------------------------
    def profile(self, on, only_slot: int = None):
        """HIP-event timing of the step's kernels: all of them, or only slot `only_slot` (least perturbation)."""
        self.ctx.call("midas_profile_enable", 0 if not on else (1 if only_slot is None else 2 + int(only_slot)))

    def profile_read(self, reset=True):
        ms = (C.c_double * _lib.PROF_SLOTS)()
        calls = C.c_int64()
        self.ctx.call("midas_profile_read", ms, C.byref(calls), int(reset))
        names = [self.ctx.lib.midas_profile_slot_name(i).decode() for i in range(_lib.PROF_SLOTS)]
        return {n: ms[i] for i, n in enumerate(names) if n}, calls.value


def _materialised(name):
    """Attribute that belongs to the resampled particle set: reading it materialises a pending resample first."""

    def get(self):
        self.flush()
        return getattr(self, "_" + name)

    def put(self, value):
        setattr(self, "_" + name, value)

    return property(get, put)


class PipelinedFilterEngine(FilterEngine):
    """FilterEngine with the resample of frame t folded into the front kernel of frame t+1 (midas_lazy_step).

    Slot n of the next frame is particle src(n) of this one, a per-slot dependence: the resampler's search and
    gather run as a prologue of the next particle update, the resampled poses never travel through HBM and a
    frame is two launches instead of three.  The resampled particle set of the latest frame therefore exists only
    implicitly (tables + draws) until somebody reads it: `poses`, `weights`, `weights_res`, `hint`, `ridx`, `status`
    and `rmse` materialise it on access (`flush()`, the same kernel as the eager engine's tail - bit-identical
    results); `nn_idx` and `poses_prop` of the latest frame are always there.  A caller that looks at the particles
    every frame gets the eager engine's launches; one that only steps gets the pipelined ones.
    Needs a float32 codebook with D in {128, 256, 512, 1024} and N <= 1 M (MidasError otherwise: use FilterEngine).
    """

    poses = _materialised("poses")
    weights = _materialised("weights")
    weights_res = _materialised("weights_res")
    hint = _materialised("hint")
    ridx = _materialised("ridx")

    @property
    def rmse(self):
        """rmse of the latest frame's propagated particles (filter.py:164): written by the frame itself, no materialisation."""
        return self._rmse_last[:2] if self._pending else self._rmse

    @rmse.setter
    def rmse(self, v):
        self._rmse = v

    def __init__(self, *args, **kw):
        self._pending = False
        self._flushed = True
        super().__init__(*args, **kw)
        N, dev = self.N, self.device
        if self.codebook.emb.dtype != torch.float32 or self.D not in (128, 256, 512, 1024) or N > (1 << 20):
            raise MidasError("PipelinedFilterEngine needs a float32 codebook with D in {128,256,512,1024} and N <= 2^20")
        f64 = dict(dtype=torch.float64, device=dev)
        self._prop = [self.poses_prop, torch.zeros_like(self.poses_prop)]
        self._nn = [self.nn_idx, torch.zeros_like(self.nn_idx)]
        self._st = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(2)]
        self._valid = torch.zeros(N, dtype=torch.uint8, device=dev)
        ng, nb = (N + 15) // 16, (N + 4095) // 4096
        self._tables = torch.zeros(4 * (-(-N // 16) * 16) + 2 * (-(-ng // 16) * 16) + 37 * nb, **f64)
        # guide tables of the folded resample's search (midas_lazy_args.guide_dev; MIDAS_GUIDE=0: the three-line search alone)
        self._guide = (torch.zeros(int(self.ctx.lib.midas_lazy_guide_bytes(N)), dtype=torch.uint8, device=dev)
                       if os.environ.get("MIDAS_GUIDE", "1") != "0" else None)
        self._scores = torch.zeros(self.K, **f64)
        self._part_rmse = torch.zeros(2 * ((N + 63) // 64), **f64)
        self._cur = 0
        self._draw = (None, -1.0, 0)
        self._had_gt = False
        self._rmse_frame = torch.zeros(3, **f64)
        self._rmse_last = self._rmse_frame   # where the latest frame left {rmse_t, rmse_r, clock}: _rmse_frame or a row of the run log
        # prediction lists of the sparse scoring (include/midas_hip.h score_list_dev): the rows a frame used are scored for
        # the next frame by streaming workgroups of its front launch.  MIDAS_SCORE_LIST=0: every row by its first particle.
        self._score_list = torch.zeros(2 + 2 * self.K, dtype=torch.int32, device=dev) \
            if self.sparse_scores and os.environ.get("MIDAS_SCORE_LIST", "1") != "0" and N >= 16 else None

    def check(self):
        """Raises if a frame's tail reported its tables undefined (status bit 4, value 16: a wave of the grouped tail waited
        0.2 s for its block's records - include/midas_hip.h, midas_lazy_args.status_dev).  One small read-back."""
        st = torch.stack([self._st[0][0], self._st[1][0]]).cpu()
        if int(st[0]) & 16 or int(st[1]) & 16:
            raise MidasError("grouped tail: a wave did not receive its block's records in time - the frame's CDF tables are undefined "
                             "(foreign work on the device?); MIDAS_TAIL_GROUPED=0 selects the form without waits")

    # the latest frame's own outputs
    @property
    def status(self):
        self.flush()
        return self._st[self._cur]

    @status.setter
    def status(self, v):
        pass  # the base constructor's tensor is not used

    @property
    def nn_idx(self):
        return self._nn[self._cur] if hasattr(self, "_nn") else self._nn0

    @nn_idx.setter
    def nn_idx(self, v):
        self._nn0 = v

    @property
    def poses_prop(self):
        return self._prop[self._cur] if hasattr(self, "_prop") else self._prop0

    @poses_prop.setter
    def poses_prop(self, v):
        self._prop0 = v

    def set_particles(self, poses):
        self._pending, self._flushed = False, True
        super().set_particles(poses)

    def project_to_codebook(self):
        self.flush()
        idx = super().project_to_codebook()
        if self._score_list is not None:
            # every particle's nearest entry is known: the first frame's rows go on the prediction list right away (they would
            # otherwise all be claimed by whichever particle wave touches them first - up to 64 rows a wave after a wide start)
            self.ctx.bind_current_stream()
            self.ctx.call("midas_score_list_seed", self.K, _ptr(self._stamps), self._next_epoch(), _ptr(self._score_list), self.N, _ptr(idx))
        return idx

    def seed_torch_stream(self, seed, motion: bool = False):
        """Resample draws from the device replica of torch's CPU generator under torch.manual_seed(seed) (torch_rng.py):
        every step() without explicit `u` then resamples with the uniforms torch.multinomial would consume
        (modules/particle_filter.py:245).  With host motion noise (tn, rot given) the stream first steps over the words those
        two torch.normal calls took, so it stays aligned with a host generator seeded alike.
        motion=True: the motion noise comes from the stream too - `torch.normal(0, sig_t, (N, 3))`, `torch.normal(0, sig_r, (N, 3))`
        in front of the resampler's uniforms, the reference's order (:326-335, :245) - i.e. EVERY draw of a frame is the one a
        seeded run of the reference takes, generated on the device (unit normals a frame ahead, scaled where they are used).
        seed=None: back to Philox."""
        from .torch_rng import TorchCpuStream
        self.torch_stream = None if seed is None else TorchCpuStream(seed, self.device)
        self.torch_motion = bool(motion) and seed is not None
        self._unit_noise = None  # (tn, rot, event): unit normals of the NEXT frame, drawn behind this frame's uniforms
        return self.torch_stream

    def step(self, odom, code, gt=None, tn=None, rot=None, u=None, u32=-1.0, multiplier: float = 1.0):
        stream = getattr(self, "torch_stream", None)
        own_u, u_event = False, None  # own_u: generated here (a buffer nobody else holds): kept for the folded frame without a copy
        if stream is not None and u is None and self.mode == _lib.RESAMPLE_MULTINOMIAL:
            stream_motion = False
            if tn is not None:
                stream.skip_normal(3 * self.N).skip_normal(3 * self.N)
            elif getattr(self, "torch_motion", False):
                stream_motion = True
                # the frame's own normals: std = 1 draws scaled here - fl(n x std) is what ATen's fused multiply-add with mean 0 gives -
                # so that they can be drawn a frame ahead whatever `multiplier` the caller passes then
                if self._unit_noise is None:
                    a, _ = stream.normal_async(0.0, 1.0, (self.N, 3))
                    b, ev = stream.normal_async(0.0, 1.0, (self.N, 3))
                    self._unit_noise = (a, b, ev)
                a, b, ev = self._unit_noise
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                mul = max(float(multiplier), 1.0)
                tn, rot = a * (mul * self.sig_t), b * (mul * self.sig_r)
            # this frame's draws are consumed by the NEXT launch (the folded resample) or by flush(): generated beside this
            # frame's kernels on the generator's own stream, waited for where they are read
            own_u = True
            if stream_motion:
                # ... together with the next frame's unit normals, which follow them in the stream: one walk of the generator
                # (midas_mt19937_draws) instead of three
                (u, a, b), ev = stream.draws_async([("rand64", self.N), ("normal", 0.0, 1.0, (self.N, 3)), ("normal", 0.0, 1.0, (self.N, 3))])
                u_event = ev
                self._unit_noise = (a, b, ev)
            else:
                u, u_event = stream.rand64_async(self.N)
        self._wait_draws()
        odom, code, gt, tn, rot, u = self._operands(odom, code, gt, tn, rot, u)
        cur, nxt = self._cur, self._cur ^ 1
        fold = self._pending and not self._flushed
        a = LazyArgs()
        a.N = self.N
        a.poses_prop_prev, a.nn_idx_prev, a.status_prev = _ptr(self._prop[cur]), _ptr(self._nn[cur]), _ptr(self._st[cur])
        a.poses_prop, a.nn_idx, a.valid, a.status = _ptr(self._prop[nxt]), _ptr(self._nn[nxt]), _ptr(self._valid), _ptr(self._st[nxt])
        a.tables, a.scores = _ptr(self._tables), _ptr(self._scores)
        a.guide = _ptr(self._guide) if self._guide is not None else None
        a.part_rmse = _ptr(self._part_rmse) if gt is not None else None
        a.resample_prev = int(fold)
        a.poses_in = _ptr(self._poses)
        a.hint_in = _ptr(self._hint) if self.use_hint else None
        pu, pu32, pstep = self._draw
        a.resample_mode, a.u_prev, a.u32_prev, a.step_prev = self.mode, _ptr(pu), float(pu32), int(pstep)
        a.ridx = _ptr(self._ridx) if fold else None
        a.odom16, a.code, a.gt16 = _ptr(odom), _ptr(code), _ptr(gt)
        a.tn, a.rot = _ptr(tn), _ptr(rot)
        mul = max(float(multiplier), 1.0)
        a.std_t, a.std_r = mul * self.sig_t, mul * self.sig_r
        a.seed, a.step = self.seed, self.step_count
        a.prune_thr, a.softmax = self.pen_max, int(self.softmax)
        a.telemetry = _ptr(self.telemetry)
        if self.sparse_scores:
            a.score_stamps, a.score_epoch = _ptr(self._stamps), self._next_epoch()
            a.score_list = _ptr(self._score_list)
        a.rmse = _ptr(self._rmse_frame) if gt is not None else None
        self._keep = (odom, code, gt, tn, rot, pu)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_lazy_step(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a)))
        # this frame's resample draws, consumed by the next step or by flush()
        # (a caller's tensor is copied: it may be mutated before the next step() / flush() consumes it)
        self._draw = (None if u is None else (u if own_u else u.clone()), float(u32), self.step_count)
        self._draw_event = u_event
        self._rmse_last = self._rmse_frame
        self._had_gt = gt is not None
        self._pending, self._flushed, self._cur = True, False, nxt
        self.step_count += 1

    def run(self, odoms, codes, gts=None):
        """T frames by ONE C-ABI call (midas_lazy_run; device draws): odoms (T,4,4) f32, codes (T,D) f64, gts (T,4,4) f32 or
        None.  Returns the (T,3) float64 device tensor of per-frame {rmse_t, rmse_r, device clock in us at the frame's end}
        when gts is given.  Equivalent to T
        calls of step() - same kernels, same results - without the per-frame turn-around through Python."""
        d = self.device
        T = int(torch.as_tensor(odoms).shape[0])
        odoms = operand(odoms, "odoms", torch.float32, (T, 4, 4), d)
        codes = operand(codes, "tactile codes", torch.float64, (T, self.D), d)
        gts = operand(gts, "gt poses", torch.float32, (T, 4, 4), d)
        cur, nxt = self._cur, self._cur ^ 1
        fold = self._pending and not self._flushed
        # the argument block with every pointer that does not change from call to call, one per buffer parity, built once (the
        # timed region of a caller starts before this call: what is set up here is time the device idles)
        # (keyed on the buffers' addresses: a caller that rebinds one - `eng.poses = t`, `eng.telemetry = ...` - gets a fresh block)
        bufs = (self._prop[cur], self._nn[cur], self._st[cur], self._prop[nxt], self._nn[nxt], self._valid, self._st[nxt], self._tables,
                self._scores, self._guide, self._poses, self.telemetry)
        sig = tuple(0 if b is None else b.data_ptr() for b in bufs)
        cache = self.__dict__.setdefault("_run_args", {})
        a, have = cache.get(cur, (None, None))
        if have != sig:
            a = LazyArgs()
            a.N = self.N
            a.poses_prop_prev, a.nn_idx_prev, a.status_prev = _ptr(self._prop[cur]), _ptr(self._nn[cur]), _ptr(self._st[cur])
            a.poses_prop, a.nn_idx, a.valid, a.status = _ptr(self._prop[nxt]), _ptr(self._nn[nxt]), _ptr(self._valid), _ptr(self._st[nxt])
            a.tables, a.scores = _ptr(self._tables), _ptr(self._scores)
            a.guide = _ptr(self._guide) if self._guide is not None else None
            a.poses_in = _ptr(self._poses)
            a.telemetry = _ptr(self.telemetry)
            a.ridx = None
            a.u_prev = None
            cache[cur] = (a, sig)
        a.part_rmse = _ptr(self._part_rmse) if gts is not None else None
        a.resample_prev = int(fold)
        a.hint_in = _ptr(self._hint) if self.use_hint else None
        self._wait_draws()
        pu, pu32, pstep = self._draw
        if pu is not None:
            raise MidasError("run() continues with device draws: the pending frame was stepped with host uniforms - flush() first")
        a.resample_mode, a.u32_prev, a.step_prev = self.mode, float(pu32), int(pstep)
        a.odom16, a.code, a.gt16 = _ptr(odoms), _ptr(codes), _ptr(gts)
        a.std_t, a.std_r = self.sig_t, self.sig_r
        a.seed, a.step = self.seed, self.step_count
        a.prune_thr, a.softmax = self.pen_max, int(self.softmax)
        if self.sparse_scores:  # (a caller may switch the scoring form between calls)
            a.score_stamps, a.score_epoch, a.score_list = _ptr(self._stamps), self._next_epoch(T), _ptr(self._score_list)
        else:
            a.score_stamps, a.score_epoch, a.score_list = None, 0, None
        # a FRESH tensor per call (caching allocator: no launch, no fill - every row is written by its frame): the log belongs
        # to the caller and stays valid across later run() / step() calls; the engine keeps a reference for `rmse`
        log = torch.empty((T, 3), dtype=torch.float64, device=d) if gts is not None else None
        self._keep = (odoms, codes, gts, log)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_lazy_run(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a), T, _ptr(log)))
        self.step_count += T
        if log is not None:
            self._rmse_last = log[T - 1]
        self._draw = (None, -1.0, self.step_count - 1)
        self._had_gt = gts is not None
        self._pending, self._flushed = True, False
        self._cur = nxt if T % 2 else cur
        return log

    def _wait_draws(self):
        """The pending frame's draws may still be in flight on the generator's stream (TorchCpuStream): order this stream behind them."""
        ev = getattr(self, "_draw_event", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._draw_event = None

    def flush(self):
        """Materialise the latest frame's resample (poses, weights, weights_res, hint, ridx, status, rmse)."""
        if not self._pending or self._flushed:
            return
        self._wait_draws()
        cur = self._cur
        u, u32, stp = self._draw
        a = LazyFlushArgs()
        a.N = self.N
        a.tables, a.valid, a.nn_idx, a.poses_prop = _ptr(self._tables), _ptr(self._valid), _ptr(self._nn[cur]), _ptr(self._prop[cur])
        a.status = _ptr(self._st[cur])
        a.part_rmse = _ptr(self._part_rmse) if self._had_gt else None
        a.softmax, a.resample_mode, a.u, a.u32 = int(self.softmax), self.mode, _ptr(u), float(u32)
        a.seed, a.step = self.seed, int(stp)
        a.weights, a.ridx, a.poses_out = _ptr(self._weights), _ptr(self._ridx), _ptr(self._poses)
        a.weights_out, a.hint_out = _ptr(self._weights_res), _ptr(self._hint)
        a.rmse = _ptr(self._rmse) if self._had_gt else None
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_lazy_flush(self.ctx.h, C.byref(a)))
        self._flushed = True


class BatchFilterEngine:
    """B independent trajectories against one codebook, one C-ABI call per frame of the whole batch
    (BASELINE config 5, "throughput mode": SURVEY.md 8(e) batch mode).

    State tensors carry a leading batch dimension.  The B tactile codes of a frame are scored in one pass
    over the codebook on the matrix cores (float32 fma chains, `midas_score_batch`), so a trajectory's
    scores differ from the single-trajectory engine's float64 GEMV in the 7th digit; everything else is the
    same kernels with the trajectory as grid.y.  Device-mode Philox streams are keyed by b*N + n.
    """

    def __init__(self, cb_poses, cb_embeddings, mesh_vertices, batch: int, num_particles: int, *, sig_t=1e-4, sig_r=0.5,
                 pen_max=0.002, seed=4000, softmax=True, resample="weighted_random", device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = _lib.context(dev)
        self.device = self.ctx.device
        f32 = dict(dtype=torch.float32, device=self.device)
        self.cb_poses = torch.as_tensor(cb_poses).to(**f32).contiguous()
        self.cb_feat = ops.se3_feature(self.cb_poses)
        self.tree6 = ops.Tree(self.cb_feat)
        self.codebook = ops.Codebook(torch.as_tensor(cb_embeddings).to(self.device))
        self.tree3 = ops.Tree(torch.as_tensor(mesh_vertices).to(self.device, torch.float64))
        self.tree6.attach_mesh(self.tree3, self.cb_poses)
        self.B, self.N = int(batch), int(num_particles)
        self.sig_t, self.sig_r, self.pen_max = float(sig_t), float(sig_r), float(pen_max)
        self.seed, self.softmax = int(seed), bool(softmax)
        self.mode = {"weighted_random": _lib.RESAMPLE_MULTINOMIAL, "low_var": _lib.RESAMPLE_SYSTEMATIC,
                     "low_var_batch": _lib.RESAMPLE_SYSTEMATIC}[resample]
        B, N, d = self.B, self.N, self.device
        self.poses = torch.zeros((B, N, 4, 4), **f32)
        self.poses_prop = torch.zeros((B, N, 4, 4), **f32)
        self.weights = torch.zeros((B, N), dtype=torch.float64, device=d)
        self.weights_res = torch.ones((B, N), dtype=torch.float64, device=d)
        self.nn_idx = torch.zeros((B, N), dtype=torch.int32, device=d)
        self.hint = torch.full((B, N), -1, dtype=torch.int32, device=d)
        self.hint_next = torch.full((B, N), -1, dtype=torch.int32, device=d)
        self.ridx = torch.zeros((B, N), dtype=torch.int32, device=d)
        self.status = torch.zeros((B, 2), dtype=torch.int32, device=d)
        self.rmse = torch.zeros((B, 2), dtype=torch.float64, device=d)
        extra = 16 * B * ((N + 63) // 64) if int(os.environ.get("MIDAS_ABLATE", "0")) & 4 else 0  # per-wave statistics (profiling)
        self.telemetry = torch.zeros(16 + extra, dtype=torch.int64, device=d)
        self.step_count = 0
        # sparse scoring per trajectory (B x K stamps): the particle waves score the rows their trajectory needs with the
        # float64 arithmetic of the single-trajectory step; MIDAS_DENSE_SCORES=1 keeps the matrix-core pass over all rows
        _os = os
        self.sparse_scores = self.codebook.emb.dtype == torch.float32 and self.codebook.D in (128, 256, 512, 1024) and \
            _os.environ.get("MIDAS_DENSE_SCORES", "0") != "1"
        self._stamps = torch.zeros((B, self.codebook.K), dtype=torch.int32, device=d) if self.sparse_scores else None
        self._epoch = 0

    def set_particles(self, poses):
        poses = torch.as_tensor(poses).to(self.device, torch.float32)
        if tuple(poses.shape) != (self.B, self.N, 4, 4):
            raise MidasError(f"expected ({self.B},{self.N},4,4) poses, got {tuple(poses.shape)}")
        self.poses.copy_(poses)
        self.hint.fill_(-1)

    def project_to_codebook(self):
        flat = self.poses.view(-1, 4, 4)
        idx = ops.nn6(self.tree6, ops.se3_feature(flat))
        self.poses.copy_(ops.gather_rows(self.cb_poses, idx).view_as(self.poses))
        self.hint.copy_(idx.view_as(self.hint))

    def step(self, odoms, codes, gts=None, tn=None, rot=None, u=None, u32=-1.0):
        """odoms (B,4,4) f32, codes (B,D) f64, gts (B,4,4) f32 or None; optional host draws tn/rot (B,N,3), u (B,N)."""
        d, B, N = self.device, self.B, self.N
        if (tn is None) != (rot is None):
            raise MidasError("tn and rot (the motion model's host draws) come together or not at all")
        odoms, gts = operand(odoms, "odoms", torch.float32, (B, 4, 4), d), operand(gts, "gt poses", torch.float32, (B, 4, 4), d)
        codes = operand(codes, "tactile codes", torch.float64, (B, self.codebook.D), d)
        tn, rot = operand(tn, "tn", torch.float32, (B, N, 3), d), operand(rot, "rot", torch.float32, (B, N, 3), d)
        u = operand(u, "u", torch.float64, (B, N), d)
        a = StepArgs()
        a.N = self.N
        a.poses_in, a.poses_prop, a.poses_out = _ptr(self.poses), _ptr(self.poses_prop), _ptr(self.poses)
        a.weights, a.weights_out = _ptr(self.weights), _ptr(self.weights_res)
        a.hint_in, a.nn_idx, a.hint_out, a.ridx = _ptr(self.hint), _ptr(self.nn_idx), _ptr(self.hint_next), _ptr(self.ridx)
        a.odom16, a.code, a.gt16 = _ptr(odoms), _ptr(codes), _ptr(gts)
        a.rmse = _ptr(self.rmse) if gts is not None else None
        a.tn, a.rot, a.u, a.u32 = _ptr(tn), _ptr(rot), _ptr(u), float(u32)
        a.std_t, a.std_r, a.seed, a.step = self.sig_t, self.sig_r, self.seed, self.step_count
        a.prune_thr, a.softmax, a.resample_mode = self.pen_max, int(self.softmax), self.mode
        a.status, a.telemetry = _ptr(self.status), _ptr(self.telemetry)
        if self.sparse_scores:
            a.score_stamps, a.score_epoch = _ptr(self._stamps), advance_epoch(self)
        self._keep = (odoms, codes, gts, tn, rot, u)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_filter_step_batch(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h,
                                                            C.byref(a), self.B))
        self.hint, self.hint_next = self.hint_next, self.hint
        self.step_count += 1


class PipelinedBatchFilterEngine(BatchFilterEngine):
    """BatchFilterEngine with every trajectory's resample folded into the next frame's front kernel (midas_lazy_step_batch):
    two launches per batch frame - the front with the trajectory as grid.y (one-wave workgroups, per-wave resample tables)
    and the LDS-free tail - instead of particle update + tail + resample-gather.  As with PipelinedFilterEngine the
    resampled particle set of the latest frame is implicit until somebody reads it: `poses`, `weights`, `weights_res`,
    `hint`, `ridx`, `status` materialise it (`flush()`, bit-identical to BatchFilterEngine's), `nn_idx` / `poses_prop` /
    `rmse` of the latest frame are always there.  Needs sparse scoring (float32 codebook, D in {128, 256, 512, 1024}) and
    16 <= N <= 262144 particles per trajectory."""

    poses = _materialised("poses")
    weights = _materialised("weights")
    weights_res = _materialised("weights_res")
    hint = _materialised("hint")
    ridx = _materialised("ridx")

    def __init__(self, *args, **kw):
        self._pending, self._flushed = False, True
        super().__init__(*args, **kw)
        B, N, d = self.B, self.N, self.device
        if not self.sparse_scores or N < 16 or N > 262144:
            raise MidasError("PipelinedBatchFilterEngine needs a float32 codebook with D in {128,256,512,1024} (sparse scoring) "
                             "and 16 <= N <= 262144")
        f64 = dict(dtype=torch.float64, device=d)
        self._prop = [self.poses_prop, torch.zeros_like(self.poses_prop)]
        self._nn = [self.nn_idx, torch.zeros_like(self.nn_idx)]
        self._st = [torch.zeros((B, 2), dtype=torch.int32, device=d) for _ in range(2)]
        self._valid = torch.zeros((B, N), dtype=torch.uint8, device=d)
        self._tstride = int(self.ctx.lib.midas_lazy_tables_doubles(N))
        self._tables = torch.zeros(B * self._tstride, **f64)
        self._scores = torch.zeros((B, self.codebook.K), **f64)
        self._part_rmse = torch.zeros((B, 2 * ((N + 63) // 64)), **f64)
        self._rmse_frame = torch.zeros((B, 3), **f64)
        self._cur, self._draw, self._had_gt = 0, (None, -1.0, 0), False

    # storage behind the materialised properties (the base constructor assigns them)
    @property
    def status(self):
        self.flush()
        return self._st[self._cur]

    @status.setter
    def status(self, v):
        pass

    @property
    def nn_idx(self):
        return self._nn[self._cur] if hasattr(self, "_nn") else self._nn0

    @nn_idx.setter
    def nn_idx(self, v):
        self._nn0 = v

    @property
    def poses_prop(self):
        return self._prop[self._cur] if hasattr(self, "_prop") else self._prop0

    @poses_prop.setter
    def poses_prop(self, v):
        self._prop0 = v

    @property
    def rmse(self):
        return self._rmse_frame[:, :2] if self._pending else self._rmse

    @rmse.setter
    def rmse(self, v):
        self._rmse = v

    def set_particles(self, poses):
        self._pending, self._flushed = False, True
        poses = torch.as_tensor(poses).to(self.device, torch.float32)
        if tuple(poses.shape) != (self.B, self.N, 4, 4):
            raise MidasError(f"expected ({self.B},{self.N},4,4) poses, got {tuple(poses.shape)}")
        self._poses.copy_(poses)
        self._hint.fill_(-1)

    def project_to_codebook(self):
        self.flush()
        idx = ops.nn6(self.tree6, ops.se3_feature(self._poses.view(-1, 4, 4)))
        self._poses.copy_(ops.gather_rows(self.cb_poses, idx).view_as(self._poses))
        self._hint.copy_(idx.view_as(self._hint))

    def step(self, odoms, codes, gts=None, tn=None, rot=None, u=None, u32=-1.0):
        d, B, N = self.device, self.B, self.N
        if (tn is None) != (rot is None):
            raise MidasError("tn and rot (the motion model's host draws) come together or not at all")
        odoms, gts = operand(odoms, "odoms", torch.float32, (B, 4, 4), d), operand(gts, "gt poses", torch.float32, (B, 4, 4), d)
        codes = operand(codes, "tactile codes", torch.float64, (B, self.codebook.D), d)
        tn, rot = operand(tn, "tn", torch.float32, (B, N, 3), d), operand(rot, "rot", torch.float32, (B, N, 3), d)
        u = operand(u, "u", torch.float64, (B, N), d)
        cur, nxt = self._cur, self._cur ^ 1
        fold = self._pending and not self._flushed
        a = LazyArgs()
        a.N = N
        a.poses_prop_prev, a.nn_idx_prev, a.status_prev = _ptr(self._prop[cur]), _ptr(self._nn[cur]), _ptr(self._st[cur])
        a.poses_prop, a.nn_idx, a.valid, a.status = _ptr(self._prop[nxt]), _ptr(self._nn[nxt]), _ptr(self._valid), _ptr(self._st[nxt])
        a.tables, a.scores = _ptr(self._tables), _ptr(self._scores)
        a.part_rmse = _ptr(self._part_rmse) if gts is not None else None
        a.resample_prev = int(fold)
        a.poses_in, a.hint_in = _ptr(self._poses), _ptr(self._hint)
        pu, pu32, pstep = self._draw
        a.resample_mode, a.u_prev, a.u32_prev, a.step_prev = self.mode, _ptr(pu), float(pu32), int(pstep)
        a.ridx = _ptr(self._ridx) if fold else None
        a.odom16, a.code, a.gt16 = _ptr(odoms), _ptr(codes), _ptr(gts)
        a.tn, a.rot = _ptr(tn), _ptr(rot)
        a.std_t, a.std_r, a.seed, a.step = self.sig_t, self.sig_r, self.seed, self.step_count
        a.prune_thr, a.softmax = self.pen_max, int(self.softmax)
        a.telemetry = _ptr(self.telemetry)
        a.score_stamps, a.score_epoch = _ptr(self._stamps), advance_epoch(self)
        a.rmse = _ptr(self._rmse_frame) if gts is not None else None
        self._keep = (odoms, codes, gts, tn, rot, pu)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_lazy_step_batch(self.ctx.h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a), B))
        self._draw = (None if u is None else u.clone(), float(u32), self.step_count)
        self._had_gt = gts is not None
        self._pending, self._flushed, self._cur = True, False, nxt
        self.step_count += 1

    def flush(self):
        """Materialise the latest frame's resample of every trajectory."""
        if not self._pending or self._flushed:
            return
        cur = self._cur
        u, u32, stp = self._draw
        a = LazyFlushArgs()
        a.N = self.N
        a.tables, a.valid, a.nn_idx, a.poses_prop = _ptr(self._tables), _ptr(self._valid), _ptr(self._nn[cur]), _ptr(self._prop[cur])
        a.status = _ptr(self._st[cur])
        a.part_rmse = _ptr(self._part_rmse) if self._had_gt else None
        a.softmax, a.resample_mode, a.u, a.u32 = int(self.softmax), self.mode, _ptr(u), float(u32)
        a.seed, a.step = self.seed, int(stp)
        a.weights, a.ridx, a.poses_out = _ptr(self._weights), _ptr(self._ridx), _ptr(self._poses)
        a.weights_out, a.hint_out = _ptr(self._weights_res), _ptr(self._hint)
        a.rmse = _ptr(self._rmse) if self._had_gt else None
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_lazy_flush_batch(self.ctx.h, C.byref(a), self.B))
        self._flushed = True
