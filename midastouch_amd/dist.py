"""Particle-sharded filter across the GPUs of one node (one process per GPU, RCCL over xGMI).

SURVEY.md 8(e): rank r owns the global particle slots [r*N, (r+1)*N) and a full replica of the codebook,
the NN index and the mesh index.  One frame is the single-GPU frame (engine.FilterEngine.step) cut at
its global reductions; between the local kernels the ranks exchange:

  R1  all_gather, 5nb+4 doubles/rank: per 4096-slot block the sum of exp(x - 1) (softmax denominator), the totals of
      exp * mask and of x * mask (CDF offsets / total of the softmax and of the raw variant), max x, min x (the isclose
      guard is global); then NaN count, kept count, rmse partial sums - all in the fixed summation order
  R2  the resampled particles.  exchange="a2a" (default): the draw of a slot is a pure function of the slot, so after
      R1 every rank can tell for every slot of the filter which rank owns its source; the OWNER resolves the source in
      its own tables and sends that one 88-byte row to the rank holding the slot - an all_to_all of N rows per rank in
      total (the split sizes are read back from the device, one small synchronisation per frame).
      exchange="allgather": every rank materialises its slice of the global CDF and all ranks gather ONE packed block
      [cdf | weights | propagated poses | NN indices] (84 N bytes per rank, G-1 times the bytes, no read-back).
      exchange="a2a_fixed": the owner-side form without the count pass (fixed-capacity segments + an overflow block).
      exchange="peer": the owner stores the row straight into the destination's inbox, device memory every process of
      the node has mapped (interprocess handles, xGMI); the ranks only gather the R1 record and a barrier word
      (DESIGN.md section 5).

The local kernels are the single-GPU ones: the fused front (particle update + codebook scoring in one launch)
and the deferred tail that gathers the scores itself.

The float64 summation order is the single-GPU one (per-block totals are gathered and every rank adds
them sequentially in global block order), so with N a multiple of 4096 the sharded run reproduces the
single-GPU run of G*N particles bit for bit - same weights, same resample indices.

The frame is written as a generator that yields at every exchange, so the same code runs under
torch.distributed (`step`) and, for tests, as several shards of one process stepped in lock-step by
`run_lockstep` (no collective library involved).  Compute goes through a backend object: the product
backend is HIP-only (`HipShardBackend`); tests may inject another one.  There is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
import sys
import os

import torch

from . import _lib, ops
from ._lib import MidasError, ShardFrontArgs, ShardRouteArgs, TailResampleArgs, _ptr

BLOCK = 4096  # summation block of the CDF spec (csrc/resample.hip)
ROUTE_REC = 88  # bytes per routed particle row in the all_to_all forms (include/midas_hip.h)
PEER_ROW = 128  # bytes per row of a peer-mapped inbox: one line, written by sixteen lanes (csrc/peer_row.hpp)


class HipShardBackend:
    """Local kernels of one shard (libmidas_hip.so).

    row_shard = (rank, world) keeps only this rank's contiguous slice of the embedding rows in HBM
    (SURVEY.md 8(e) codebook-row sharding, BASELINE config 4): the frame then starts with one extra
    all_gather of the per-rank score slices; poses, the NN index and the mesh index stay replicated.
    """

    def __init__(self, cb_poses, cb_embeddings, mesh_vertices, device, row_shard=None, share=None):
        """share: another backend of the SAME process and codebook whose replicated indices (poses, 6-d features, neighbour /
        vertex lists, mesh tree) this one uses instead of building its own (several shards on one GPU: tests, row sharding)."""
        self.ctx = _lib.context(device)
        self.device = self.ctx.device
        if share is not None:
            self.cb_poses, self.cb_feat, self.tree6 = share.cb_poses, share.cb_feat, share.tree6
        else:
            self.cb_poses = torch.as_tensor(cb_poses).to(self.device, torch.float32).contiguous()
            self.cb_feat = ops.se3_feature(self.cb_poses)
            self.tree6 = ops.Tree(self.cb_feat)
        emb = torch.as_tensor(cb_embeddings)
        self.row_shard = row_shard
        if row_shard is not None:
            r, w = row_shard
            K = emb.shape[0]
            if K % w:
                raise MidasError("codebook-row sharding needs K divisible by the number of ranks")
            emb = emb[r * (K // w):(r + 1) * (K // w)]
        self.codebook = ops.Codebook(emb.to(self.device))
        if share is not None:
            self.tree3 = share.tree3
        else:
            self.tree3 = ops.Tree(torch.as_tensor(mesh_vertices).to(self.device, torch.float64))
            self.tree6.attach_mesh(self.tree3, self.cb_poses)
        self.K = int(self.cb_poses.shape[0])
        self.D = int(emb.shape[1])
        import os
        self._sparse = (row_shard is None and self.codebook.emb.dtype == torch.float32 and self.D in (128, 256, 512, 1024) and
                        os.environ.get("MIDAS_DENSE_SCORES", "0") != "1")

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def project(self, poses):
        idx = ops.nn6(self.tree6, ops.se3_feature(poses))
        return ops.gather_rows(self.cb_poses, idx), idx

    def score_slice(self, code):
        """This rank's slice of the frame's scores (row-sharded codebook)."""
        return self.codebook.score(code)[0]

    def front(self, st, odom, code, gt, tn, rot, std_t, std_r, seed, step, prune_thr, use_hint=True, scores_ready=False):
        a = ShardFrontArgs()
        a.N, a.slot_base = st.N, st.slot_base
        a.poses_in, a.poses_prop = _ptr(st.poses), _ptr(st.poses_prop)
        a.hint_in = _ptr(st.hint) if use_hint else None
        a.nn_idx, a.valid = _ptr(st.nn_idx), _ptr(st.valid)
        a.odom16, a.code, a.gt16 = _ptr(odom), _ptr(code), _ptr(gt)
        a.scores, a.scores_ready = _ptr(st.scores), int(bool(scores_ready))
        a.rmse_sums = _ptr(st.r1[5 * st.nb + 2:]) if gt is not None else None
        a.tn, a.rot = _ptr(tn), _ptr(rot)
        a.std_t, a.std_r, a.seed, a.step, a.prune_thr = std_t, std_r, seed, step, prune_thr
        a.telemetry = _ptr(st.telemetry)
        a.status = _ptr(st.status)
        a.flags = _ptr(st.r1[5 * st.nb:])
        if not scores_ready and self._sparse:
            # sparse scoring (replicated codebook): this rank's particle waves score the rows they need, one stamp array per state
            stamps = getattr(st, "_stamps", None)
            if stamps is None:
                st._stamps = stamps = torch.zeros(self.codebook.K, dtype=torch.int32, device=st.poses.device)
                st._epoch = 0
            st._epoch = st._epoch + 1 if st._epoch < 0x7FFFFFF0 else 1
            if st._epoch == 1:
                stamps.zero_()
            a.score_stamps, a.score_epoch = _ptr(stamps), st._epoch
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_shard_front(self.ctx.h, None if scores_ready else self.codebook.h,
                                                      self.tree6.h, self.tree3.h, C.byref(a)))

    def tail_a(self, st, softmax):
        self.ctx.call("midas_shard_tail_a", st.N, _ptr(st.scores), _ptr(st.nn_idx), _ptr(st.valid), int(softmax), _ptr(st.tables),
                      _ptr(st.r1), _ptr(st.status))

    def tail_fin(self, st, r1_all, rank, world, n_total, softmax, want_rmse):
        self.ctx.call("midas_shard_tail_fin", st.N, _ptr(st.tables), _ptr(st.valid), _ptr(st.weights), _ptr(st.cdf), world,
                      _ptr(r1_all), rank, n_total, int(softmax), _ptr(st.rmse) if want_rmse else None, _ptr(st.status))

    # ---- owner-side resample ---------------------------------------------------------------------------------
    def _route_args(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse):
        a = ShardRouteArgs()
        a.N, a.G, a.rank = st.N, world, rank
        a.r1_all, a.tables, a.valid, a.nn_idx, a.poses_prop = _ptr(r1_all), _ptr(st.tables), _ptr(st.valid), _ptr(st.nn_idx), _ptr(st.poses_prop)
        a.status, a.rmse = _ptr(st.status), (_ptr(st.rmse) if want_rmse else None)
        a.softmax, a.resample_mode, a.u_all, a.u32, a.seed, a.step = int(softmax), mode, _ptr(u_all), float(u32), seed, step
        a.counts, a.weights = _ptr(st.counts), _ptr(st.weights)
        return a

    def route(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse):
        """-> (send buffer, rows sent to each rank, rows received from each rank)."""
        a = self._route_args(st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_shard_route_count(self.ctx.h, C.byref(a)))
        counts = st.counts[:2 * world].tolist()  # the split sizes: the one host read-back of the frame
        sends, recvs = counts[:world], counts[world:]
        rows = max(sum(sends), 1)
        send = getattr(st, "_send", None)
        if send is None or send.numel() < rows * ROUTE_REC:  # kept across frames; a rank owns ~N sources, up to G*N
            send = st._send = torch.empty((max(rows, 2 * st.N) * ROUTE_REC,), dtype=torch.uint8, device=self.device)
        a.send = _ptr(send)
        self.ctx.check(self.ctx.lib.midas_shard_route_pack(self.ctx.h, C.byref(a)))
        return send[:sum(sends) * ROUTE_REC], sends, recvs

    def unpack(self, st, recv):
        self.ctx.call("midas_shard_unpack", st.N, _ptr(recv), _ptr(st.ridx), _ptr(st.poses), _ptr(st.weights_res), _ptr(st.hint))

    # ---- the same without the count pass and its read-back: padded segments of fixed capacity + an overflow block ----
    def route_fixed(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse, cap, ovf_cap):
        """-> (send buffer of world x cap rows, overflow block of ovf_cap rows); unused rows carry slot -1."""
        a = self._route_args(st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse)
        send = getattr(st, "_send_fixed", None)
        if send is None or send.numel() != world * cap * ROUTE_REC:
            send = st._send_fixed = torch.empty((world * cap * ROUTE_REC,), dtype=torch.uint8, device=self.device)
            st._ovf = torch.empty((ovf_cap * ROUTE_REC,), dtype=torch.uint8, device=self.device)
            st._self = torch.empty((st.N * ROUTE_REC,), dtype=torch.uint8, device=self.device)
        a.send, a.fixed_cap, a.ovf_cap, a.ovf, a.self_rows = _ptr(send), int(cap), int(ovf_cap), _ptr(st._ovf), _ptr(st._self)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_shard_route_pack(self.ctx.h, C.byref(a)))
        return send, st._ovf

    def unpack_fixed(self, st, recv, ovf_all, rank):
        self.ctx.call("midas_shard_unpack_fixed", recv.numel() // ROUTE_REC, _ptr(recv), ovf_all.numel() // ROUTE_REC, _ptr(ovf_all), int(rank),
                      st._self.numel() // ROUTE_REC, _ptr(st._self), _ptr(st.ridx), _ptr(st.poses), _ptr(st.weights_res), _ptr(st.hint))

    def overflow_rows(self, st, world):
        """Rows the last fixed-capacity frame put into its overflow block (more than its capacity: rows were lost)."""
        return int(st.counts[3 * world].item())

    # ---- peer-mapped inboxes: the owner stores a row straight into the destination's memory ----------------------
    def peer_alloc(self, st) -> torch.Tensor:
        """This shard's inbox (N rows of fine-grained device memory) -> its 64-byte interprocess handle."""
        ptr, h = C.c_void_p(), (C.c_ubyte * 64)()
        # N rows, then the completion flags of the C-side frame (one uint64 per rank, include/midas_hip.h midas_shard_step) and the
        # route kernel's workgroup counter (one more 64-byte line; midas_peer_alloc zeroes the block)
        st._flag_off = st.N * PEER_ROW
        self.ctx.call("midas_peer_alloc", st._flag_off + 64 * 8 + 64, C.byref(ptr), h)
        st._inbox, st._opened = ptr.value, []
        return torch.tensor(list(h), dtype=torch.uint8)

    def peer_open(self, st, handle: torch.Tensor) -> int:
        ptr, h = C.c_void_p(), (C.c_ubyte * 64)(*[int(v) for v in handle.tolist()])
        self.ctx.call("midas_peer_open", h, C.byref(ptr))
        st._opened.append(ptr.value)
        return ptr.value

    def peer_table(self, st, ptrs):
        st._peers = torch.tensor([int(p) for p in ptrs], dtype=torch.int64, device=self.device)

    # ---- the whole frame by one C call (midas_shard_step / midas_shard_run; peer-mapped exchange with device-side flags) ----
    def rccl_path(self):
        """The librccl the process already has (torch's): the library opens THAT copy for its own communicator."""
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        for name in ("librccl.so", "librccl.so.1"):
            if os.path.exists(os.path.join(d, name)):
                return os.path.join(d, name).encode()
        return None

    def comm_unique_id(self) -> torch.Tensor:
        buf = (C.c_ubyte * 128)()
        self.ctx.call("midas_comm_unique_id", self.rccl_path(), buf)
        return torch.tensor(list(buf), dtype=torch.uint8)

    def comm_create(self, id128: torch.Tensor, world: int, rank: int):
        buf = (C.c_ubyte * 128)(*[int(v) for v in id128.tolist()])
        h = C.c_void_p()
        self.ctx.bind_current_stream()
        self.ctx.call("midas_comm_create", self.rccl_path(), buf, int(world), int(rank), C.byref(h))
        return h

    def comm_destroy(self, h):
        if h:
            self.ctx.lib.midas_comm_destroy(h)

    def step_args(self, st, odom, code, gt, std_t, std_r, seed, step, prune_thr, use_hint, r1_all, rank, world, softmax, mode,
                  u_all, u32, frame_tag):
        from ._lib import ShardStepArgs
        a = ShardStepArgs()
        f = a.front
        f.N, f.slot_base = st.N, st.slot_base
        f.poses_in, f.poses_prop = _ptr(st.poses), _ptr(st.poses_prop)
        f.hint_in = _ptr(st.hint) if use_hint else None
        f.nn_idx, f.valid = _ptr(st.nn_idx), _ptr(st.valid)
        f.odom16, f.code, f.gt16 = _ptr(odom), _ptr(code), _ptr(gt)
        f.scores, f.scores_ready = _ptr(st.scores), 0
        f.rmse_sums = _ptr(st.r1[5 * st.nb + 2:]) if gt is not None else None
        f.std_t, f.std_r, f.seed, f.step, f.prune_thr = std_t, std_r, seed, step, prune_thr
        f.telemetry, f.status, f.flags = _ptr(st.telemetry), _ptr(st.status), _ptr(st.r1[5 * st.nb:])
        if self._sparse:
            stamps = getattr(st, "_stamps", None)
            if stamps is None:
                st._stamps = stamps = torch.zeros(self.codebook.K, dtype=torch.int32, device=st.poses.device)
                st._epoch = 0
            f.score_stamps = _ptr(stamps)
            if getattr(st, "_score_list", None) is None:  # prediction lists (include/midas_hip.h score_list_dev)
                st._score_list = torch.zeros(2 + 2 * self.codebook.K, dtype=torch.int32, device=st.poses.device)
            a.score_list = _ptr(st._score_list)
        a.softmax, a.tables, a.r1, a.r1_all = int(softmax), _ptr(st.tables), _ptr(st.r1), _ptr(r1_all)
        if getattr(st, "_guide", None) is None and os.environ.get("MIDAS_GUIDE", "1") != "0":
            # guide tables of the owner-side searches (include/midas_hip.h guide_dev): written by the shard's tail, read by its route kernel
            st._guide = torch.zeros(int(self.ctx.lib.midas_lazy_guide_bytes(st.N)), dtype=torch.uint8, device=st.poses.device)
        a.guide = _ptr(getattr(st, "_guide", None))
        a.G, a.rank, a.resample_mode = world, rank, mode
        a.u_all, a.u32 = _ptr(u_all), float(u32)
        a.counts, a.weights = _ptr(st.counts), _ptr(st.weights)
        a.rmse = _ptr(st.rmse) if gt is not None else None
        a.peers, a.inbox, a.flag_offset, a.frame_tag = _ptr(st._peers), C.c_void_p(st._inbox), st._flag_off, int(frame_tag)
        a.ridx, a.poses_out, a.weights_out, a.hint_out = _ptr(st.ridx), _ptr(st.poses), _ptr(st.weights_res), _ptr(st.hint)
        return a

    def next_epochs(self, st, n=1) -> int:
        """First of n consecutive sparse-scoring epochs of this shard's stamps, spaced by two (the value between two epochs
        tags the rows of the prediction list); restart + zeroed stamps and list lengths long before a wrap."""
        if st._epoch + 2 * n >= 0x3FFFFFF0:  # (bits 31:30 of a stamp are the listed rows' age: csrc/midas_internal.hpp)
            st._stamps.zero_()
            st._score_list[:2].zero_()
            st._epoch = 0
        first = st._epoch + 2
        st._epoch += 2 * n
        return first

    def step_c(self, st, a, comm_h, phases, T=None):
        if self._sparse and (phases & 1):
            a.front.score_epoch = self.next_epochs(st, 1 if T is None else T)
        self.ctx.bind_current_stream()
        if T is None:
            self.ctx.check(self.ctx.lib.midas_shard_step(self.ctx.h, comm_h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a), int(phases)))
        else:
            self.ctx.check(self.ctx.lib.midas_shard_run(self.ctx.h, comm_h, self.codebook.h, self.tree6.h, self.tree3.h, C.byref(a), int(T)))

    def peer_release(self, st):
        for p in getattr(st, "_opened", []):
            self.ctx.call("midas_peer_close", C.c_void_p(p))
        if getattr(st, "_inbox", None):
            self.ctx.call("midas_peer_free", C.c_void_p(st._inbox))
        st._inbox, st._opened, st._peers = None, [], None

    def peer_probe_write(self, st, rank, world, nonce):
        self.ctx.call("midas_peer_probe_write", _ptr(st._peers), world, rank, nonce)

    def peer_probe_check(self, st, world, nonce) -> torch.Tensor:
        ok = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.ctx.call("midas_peer_probe_check", C.c_void_p(st._inbox), world, nonce, _ptr(ok))
        return ok

    def route_push(self, st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse):
        a = self._route_args(st, r1_all, rank, world, softmax, mode, u_all, u32, seed, step, want_rmse)
        a.peers = _ptr(st._peers)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_shard_route_pack(self.ctx.h, C.byref(a)))

    def unpack_peer(self, st):
        self.ctx.call("midas_shard_unpack_peer", st.N, C.c_void_p(st._inbox), _ptr(st.ridx), _ptr(st.poses), _ptr(st.weights_res),
                      _ptr(st.hint))

    def tail_resample(self, st, pack_all, n_all, mode, u, u32, seed, step):
        a = TailResampleArgs()
        a.N, a.N_all, a.slot_base = st.N, n_all, st.slot_base
        a.pack_all, a.rank_stride, a.n_per_rank = _ptr(pack_all), st.stride, st.N
        a.status, a.mode, a.u, a.u32 = _ptr(st.status), mode, _ptr(u), float(u32)
        a.seed, a.step, a.ridx = seed, step, _ptr(st.ridx)
        a.poses_out, a.weights_out, a.hint_out = _ptr(st.poses), _ptr(st.weights_res), _ptr(st.hint)
        self.ctx.bind_current_stream()
        self.ctx.check(self.ctx.lib.midas_tail_resample(self.ctx.h, C.byref(a)))


class ShardState:
    """Per-shard tensors (allocated through the backend so tests can keep them on the CPU)."""

    def __init__(self, backend, N, slot_base, K):
        e = backend.empty
        self.N, self.slot_base = int(N), int(slot_base)
        self.nb = (self.N + BLOCK - 1) // BLOCK
        if N % 2:
            raise MidasError("the sharded step needs an even number of particles per rank")
        self.poses = e((N, 4, 4), torch.float32)
        self.weights_res = e((N,), torch.float64)
        # packed record block exchanged in one all_gather: [cdf | weights | poses_prop | nn_idx] (midas_hip.h)
        self.stride = (84 * N + 255) // 256 * 256
        self.pack = e((self.stride,), torch.uint8)
        self.pack.zero_()
        self.cdf = self.pack[0:8 * N].view(torch.float64)
        self.weights = self.pack[8 * N:16 * N].view(torch.float64)
        self.poses_prop = self.pack[16 * N:80 * N].view(torch.float32).view(N, 4, 4)
        self.nn_idx = self.pack[80 * N:84 * N].view(torch.int32)
        # softmax / CDF tables of the shard (include/midas_hip.h midas_shard_tail_a): e | x | lp | lp_raw | chunk ends x2 |
        # group ends x2, per-slot and per-chunk arrays padded to multiples of 16
        ng = (N + 15) // 16
        self.tables = e((4 * (-(-N // 16) * 16) + 2 * (-(-ng // 16) * 16) + 32 * self.nb,), torch.float64)
        self.tables.zero_()
        self.counts = e((3 * 64 + 4,), torch.int32)  # send counts | receive counts | scratch of the owner-side resample | [3 G]: overflow rows (fixed-capacity form); G <= 64
        self.scores = e((int(K),), torch.float64)   # the frame's codebook scores
        self.valid = e((N,), torch.uint8)
        self.hint = e((N,), torch.int32)
        self.ridx = e((N,), torch.int32)
        self.r1 = e((5 * self.nb + 4,), torch.float64)  # exchange record (see the module docstring)
        self.status = e((2,), torch.int32)
        self.rmse = e((2,), torch.float64)
        self.telemetry = e((16,), torch.int64)
        self.telemetry.zero_()
        self.hint.fill_(-1)
        self.r1.zero_()
        self.sync = e((1,), torch.int32)            # what the ranks gather as a barrier (peer-mapped exchange)
        self.sync.zero_()
        self._inbox = self._peers = None


class TorchDistComm:
    """all_gather over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._into = dist.get_backend(group) == "nccl"
        self._bufs = {}

    def _out(self, shape, dtype, device):
        """Receive buffer for a collective: two per (shape, dtype), alternating - allocating one per call was ~5 us of host
        time per collective in a frame that is bound by its host-side enqueueing (tools/shard_host_cost.py); a buffer is
        written again two calls later, behind everything the stream was given in between."""
        key = (tuple(shape), dtype, device)
        slot = self._bufs.get(key)
        if slot is None:
            slot = self._bufs[key] = [[torch.empty(shape, dtype=dtype, device=device) for _ in range(2)], 0]
            if len(self._bufs) > 64:  # variable split sizes (exchange="a2a"): do not hoard
                self._bufs = {key: slot}
        slot[1] ^= 1
        return slot[0][slot[1]]

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        if self._into:
            out = self._out((self.world * t.shape[0],) + tuple(t.shape[1:]), t.dtype, t.device)
            self.dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        # gloo (CPU tests, or several processes sharing one GPU): through host memory
        h = t.cpu()
        out = torch.empty((self.world * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        self.dist.all_gather(list(out.chunk(self.world)), h, group=self.group)
        return out.to(t.device)

    def all_to_all(self, send: torch.Tensor, in_splits, out_splits) -> torch.Tensor:
        if self._into:
            out = self._out((sum(out_splits),), send.dtype, send.device)
            self.dist.all_to_all_single(out, send.contiguous(), list(out_splits), list(in_splits), group=self.group)
            return out
        h = send.contiguous().cpu()
        out = torch.empty((sum(out_splits),), dtype=h.dtype)
        # gloo has no all_to_all_single on every build: the same exchange as point-to-point pairs
        ins, outs = list(h.split(list(in_splits))), list(out.split(list(out_splits)))
        reqs = []
        for r in range(self.world):
            if r == self.rank:
                outs[r].copy_(ins[r])
                continue
            reqs.append(self.dist.isend(ins[r].clone(), r, group=self.group))
            reqs.append(self.dist.irecv(outs[r], r, group=self.group))
        for q in reqs:
            q.wait()
        return out.to(send.device)


class SingleComm:
    rank, world = 0, 1

    def all_gather(self, t):
        return t

    def all_to_all(self, send, in_splits, out_splits):
        return send


class ShardedFilterEngine:
    def __init__(self, cb_poses=None, cb_embeddings=None, mesh_vertices=None, num_particles: int = 0, *, sig_t=2e-4,
                 sig_r=0.5, pen_max=0.002, seed=4000, softmax=True, resample="weighted_random", device=None,
                 comm=None, backend=None, rank=None, world=None, shard_codebook_rows=False, exchange="auto"):
        if comm is None:
            import torch.distributed as dist

            comm = TorchDistComm() if dist.is_available() and dist.is_initialized() else SingleComm()
        self.comm = comm
        self.rank = comm.rank if rank is None else rank
        self.world = comm.world if world is None else world
        if backend is None:
            backend = HipShardBackend(cb_poses, cb_embeddings, mesh_vertices, device,
                                      row_shard=(self.rank, self.world) if shard_codebook_rows else None)
        self.backend = backend
        self.N = int(num_particles)
        self.N_total = self.N * self.world
        self.st = ShardState(self.backend, self.N, self.rank * self.N, self.backend.K)
        self.sig_t, self.sig_r, self.pen_max = float(sig_t), float(sig_r), float(pen_max)
        self.seed, self.softmax = int(seed), bool(softmax)
        self.mode = {"weighted_random": _lib.RESAMPLE_MULTINOMIAL, "low_var": _lib.RESAMPLE_SYSTEMATIC,
                     "low_var_batch": _lib.RESAMPLE_SYSTEMATIC}[resample]
        self.step_count = 0
        self.use_hint = True
        if exchange not in ("auto", "a2a", "a2a_fixed", "allgather", "peer", "peer_c"):
            raise MidasError("exchange must be 'auto', 'peer_c', 'peer', 'a2a', 'a2a_fixed' or 'allgather'")
        # a2a_fixed: per destination a segment of 1.5 x the expected N / G rows (after a resample every rank owns ~1 / G of
        # the weight mass), the rest through an overflow block of N / 4 rows that every rank gathers
        self.seg_cap = -(-int(1.5 * self.N / self.world + 64) // 8) * 8
        self.ovf_cap = max(self.N // 4, 256)
        # the owner-side forms move a fraction of the bytes of the all_gather form: from four ranks on.  "a2a_fixed" has no
        # host read-back inside the frame (fixed-capacity segments + overflow block); "a2a" sends exactly the rows needed
        # but reads the split sizes back
        # (multinomial draws: every rank needs ~N / G rows of every other; systematic draws walk the CDF in slot order, their
        # traffic is whatever the weight distribution makes it - counted exchange there)
        auto = "allgather" if self.world < 4 else ("a2a_fixed" if self.mode == _lib.RESAMPLE_MULTINOMIAL else "a2a")
        self.exchange = auto if exchange == "auto" else exchange
        if self.world > 64:
            raise MidasError("at most 64 particle shards (the per-pair counters of the exchange hold 64 ranks)")
        # fixed-capacity form: the overflow block's fill of every frame is copied to pinned host memory behind the frame and
        # looked at before the next one and before anything of the particle set is read (no synchronisation of its own)
        self._ovf_host = None
        self._ovf_pending = [None, None]
        # "peer": rows stored straight into the destination's memory (inboxes mapped into every process, xGMI), RCCL only
        # carries the block records and a barrier.  Ranks in separate processes connect here (a collective); it is what
        # "auto" picks under RCCL when the start-up self test of the mapped path passes on every rank.  Shards of one process
        # (tests) are wired with connect_local_peers().
        # "peer_c": the same exchange with the WHOLE frame enqueued by one C call on a communicator the library owns
        # (midas_shard_step: kernels, the record all_gather by RCCL, device-side completion flags instead of the barrier
        # collective) - what "auto" picks when it can; without RCCL between the ranks (gloo: tests) the record gather stays
        # with `comm` and the frame is two C calls around it.
        self.peer_error = None
        self._ccomm = None
        self._r1_all = None
        separate = isinstance(self.comm, TorchDistComm) and hasattr(self.backend, "peer_alloc")
        rccl = bool(getattr(self.comm, "_into", False))
        want_c = os.environ.get("MIDAS_SHARD_C", "1") != "0" and hasattr(self.backend, "step_c")
        if exchange in ("peer", "peer_c") and separate:
            if not self.connect_peers():
                raise MidasError(f"peer-mapped exchange unavailable: {self.peer_error}")
            if exchange == "peer_c" and rccl:
                self.connect_c_comm()
        elif exchange == "auto" and separate and rccl and (self.world > 1 or want_c) and os.environ.get("MIDAS_PEER_EXCHANGE", "1") != "0":
            if self.connect_peers():
                self.exchange = "peer"
                if want_c and self.connect_c_comm(required=False):
                    self.exchange = "peer_c"

    def connect_c_comm(self, required: bool = True) -> bool:
        """Collective: rank 0 obtains an RCCL unique id, every rank receives it over `comm` and joins the library-owned
        communicator (midas_comm_create).  True when every rank holds one."""
        b, dev = self.backend, self.st.poses.device
        ok, err = True, None
        try:
            ident = b.comm_unique_id() if self.rank == 0 else torch.zeros(128, dtype=torch.uint8)
        except MidasError as e:
            ok, err, ident = False, str(e), torch.zeros(128, dtype=torch.uint8)
        ident = self.comm.all_gather(ident.to(dev)).cpu().view(self.world, 128)[0]
        if ok:
            try:
                self._ccomm = b.comm_create(ident, self.world, self.rank)
            except MidasError as e:
                ok, err = False, str(e)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        ok = bool(self.comm.all_gather(flag).min().item())
        if not ok:
            if self._ccomm is not None:
                b.comm_destroy(self._ccomm)
                self._ccomm = None
            self.peer_error = err or "another rank could not create the library's communicator"
            if required:
                raise MidasError(f"library-owned RCCL communicator unavailable: {self.peer_error}")
        return ok

    def _c_args(self, odom, code, gt, u, u32, mul, r1_all, frames=1):
        st, b = self.st, self.backend
        tag = getattr(self, "_frame_tag", 0) + 1   # grows for the life of the inboxes, whatever happens to step_count
        self._frame_tag = tag + frames - 1
        return b.step_args(st, odom, code, gt, mul * self.sig_t, mul * self.sig_r, self.seed, self.step_count, self.pen_max,
                           self.use_hint, r1_all, self.rank, self.world, self.softmax, self.mode, u, u32, tag)

    def _r1_all_buf(self):
        if self._r1_all is None:
            self._r1_all = torch.empty(self.world * self.st.r1.numel(), dtype=torch.float64, device=self.st.poses.device)
        return self._r1_all

    def run(self, odoms, codes, gts=None):
        """T frames by ONE C call (midas_shard_run; exchange "peer_c" on the library's communicator, device draws)."""
        if self.exchange != "peer_c" or self._ccomm is None:
            raise MidasError("run() needs exchange='peer_c' with the library-owned RCCL communicator")
        from .engine import operand
        d, D = self.st.poses.device, int(self.backend.D)
        T = int(torch.as_tensor(odoms).shape[0])
        odoms = operand(odoms, "odoms", torch.float32, (T, 4, 4), d)
        codes = operand(codes, "tactile codes", torch.float64, (T, D), d)
        gts = operand(gts, "gt poses", torch.float32, (T, 4, 4), d)
        self._keep = (odoms, codes, gts)
        a = self._c_args(odoms, codes, gts, None, -1.0, 1.0, self._r1_all_buf(), frames=T)
        self.backend.step_c(self.st, a, self._ccomm, 15, T=T)
        self.step_count += T

    def connect_peers(self) -> bool:
        """Collective: allocate this rank's inbox, swap the interprocess handles, map the others' inboxes, run the self test
        twice.  True when EVERY rank came through (otherwise everything is released again and `peer_error` says why)."""
        b, st, G, dev = self.backend, self.st, self.world, self.st.poses.device
        msg = torch.zeros(65, dtype=torch.uint8)
        try:
            msg[:64] = b.peer_alloc(st)
            msg[64] = 1
        except MidasError as e:
            self.peer_error = f"rank {self.rank}: {e}"
        all_msg = self.comm.all_gather(msg.to(dev)).cpu().view(G, 65)
        ok = bool(all_msg[:, 64].all())
        if ok:
            try:
                b.peer_table(st, [st._inbox if r == self.rank else b.peer_open(st, all_msg[r, :64]) for r in range(G)])
            except MidasError as e:
                ok, self.peer_error = False, f"rank {self.rank}: {e}"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        ok = bool(self.comm.all_gather(flag).min().item())
        if ok:
            for nonce in (0x5EED0001, 0x5EED0002):  # twice: a line cached from the first round must not satisfy the second
                b.peer_probe_write(st, self.rank, G, nonce)
                self.comm.all_gather(st.sync)       # every rank has written
                good = self.comm.all_gather(b.peer_probe_check(st, G, nonce))
                if not bool(good.min().item()):
                    ok, self.peer_error = False, "self test of the mapped path failed on ranks " + str((good == 0).nonzero().flatten().tolist())
                    break
        if not ok:
            if self.peer_error is None:
                self.peer_error = "another rank could not map the inboxes"
            try:
                b.peer_release(st)
            except MidasError:
                pass
        return ok

    # convenience views used by bench.py / tests (same names as FilterEngine)
    def _view(name):
        def get(self):
            self._check_overflow(wait=True)  # a frame that lost rows must not be read as if it were whole
            return getattr(self.st, name)
        return property(get)

    poses = _view("poses")
    poses_prop = _view("poses_prop")
    weights = _view("weights")
    weights_res = _view("weights_res")
    nn_idx = _view("nn_idx")
    hint = _view("hint")
    ridx = _view("ridx")
    status = _view("status")
    rmse = _view("rmse")
    del _view

    # -- fixed-capacity exchange: rows beyond segment + overflow capacity are LOST (the kernel cannot grow a buffer): loud ----
    def _watch_overflow(self):
        """Behind a frame of the fixed-capacity form: its overflow count goes to one of two pinned slots; the slot's previous
        tenant (the frame two back, long finished - the host never runs that far ahead of a frame's collectives) is looked at
        first.  No synchronisation beyond that."""
        st, G = self.st, self.world
        if self._ovf_host is None:
            z = torch.zeros(2, dtype=torch.int32)
            self._ovf_host = z.pin_memory() if st.counts.is_cuda else z
            self._ovf_pending = [None, None]  # per slot: (frame, event)
        k = self.step_count & 1
        self._check_slot(k, wait=True)
        self._ovf_host[k:k + 1].copy_(st.counts[3 * G:3 * G + 1], non_blocking=True)
        ev = None
        if st.counts.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._ovf_pending[k] = (self.step_count, ev)

    def _check_slot(self, k, wait):
        pend = self._ovf_pending[k]
        if pend is None:
            return
        frame, ev = pend
        if ev is not None:
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
        self._ovf_pending[k] = None
        n = int(self._ovf_host[k])
        if n > self.ovf_cap:
            raise MidasError(f"exchange='a2a_fixed': frame {frame} needed {n} overflow rows, the block holds {self.ovf_cap} - "
                             f"{n - self.ovf_cap} resampled particles were lost (one rank owns much more than 1/{self.world} of the "
                             "weight mass).  Use exchange='a2a' (counted segments) or 'peer' for such clouds")

    def _check_overflow(self, wait: bool = False):
        """Raises when the fixed-capacity exchange of a finished frame dropped rows (one rank owned far more than 1 / G of the
        weight mass): that frame's particle set is incomplete and every later frame builds on it."""
        if self._ovf_host is None:
            return
        for k in (0, 1):
            self._check_slot(k, wait)

    def close(self):
        """Release the library's communicator, the peer-mapped inbox and the interprocess mappings (fine-grained device memory
        is not returned by the garbage collector)."""
        if getattr(self, "_ccomm", None) is not None:
            try:
                self.backend.comm_destroy(self._ccomm)
            except Exception:
                pass
            self._ccomm = None
        if getattr(self.st, "_inbox", None) or getattr(self.st, "_opened", None):
            try:
                self.backend.peer_release(self.st)
            except Exception:
                pass

    def __del__(self, _finalizing=sys.is_finalizing):  # (bound at import: module globals are gone by then)
        if _finalizing():  # the process is going away: the HIP runtime may be gone already (its calls would abort, not raise)
            return
        try:
            self.close()
        except Exception:
            pass

    def set_particles(self, poses):
        poses = torch.as_tensor(poses).to(self.st.poses.device, torch.float32)
        if tuple(poses.shape) != (self.N, 4, 4):
            raise MidasError(f"expected ({self.N},4,4) local poses, got {tuple(poses.shape)}")
        self.st.poses.copy_(poses)
        self.st.hint.fill_(-1)

    def project_to_codebook(self):
        p, idx = self.backend.project(self.st.poses)
        self.st.poses.copy_(p)
        self.st.hint.copy_(idx)

    # -- one frame as a generator: yields the local tensor of each exchange, receives the gathered one -----
    def step_gen(self, odom, code, gt=None, tn=None, rot=None, u=None, u32=-1.0, multiplier: float = 1.0):
        st, b, G = self.st, self.backend, self.world
        mul = max(float(multiplier), 1.0)
        # the kernels read raw pointers: operands on the shard's device, in the ABI's dtypes, contiguous, right sizes
        from .engine import operand
        d, D = st.poses.device, int(getattr(b, "D", torch.as_tensor(code).numel()))
        if (tn is None) != (rot is None):
            raise MidasError("tn and rot (the motion model's host draws) come together or not at all")
        odom, gt = operand(odom, "odom", torch.float32, (4, 4), d), operand(gt, "gt pose", torch.float32, (4, 4), d)
        code = operand(code, "tactile code", torch.float64, (D,), d)
        tn, rot = operand(tn, "tn", torch.float32, (self.N, 3), d), operand(rot, "rot", torch.float32, (self.N, 3), d)
        u = operand(u, "u (the uniforms of all slots of the filter)", torch.float64, (self.N_total,), d)
        self._keep = (odom, code, gt, tn, rot, u)
        if self.exchange == "peer_c":
            if getattr(b, "row_shard", None) is not None or tn is not None:
                raise MidasError("exchange='peer_c' takes a replicated codebook and device motion noise (use 'peer' otherwise)")
            if st._peers is None:
                raise MidasError("peer-mapped exchange: the inboxes are not connected (connect_peers / connect_local_peers)")
            if self._ccomm is not None:  # kernels, RCCL all_gather of the records, flags, unpack: one call
                b.step_c(st, self._c_args(odom, code, gt, u, u32, mul, self._r1_all_buf()), self._ccomm, 15)
            else:                        # no RCCL between the ranks: the record gather goes through `comm`
                a = self._c_args(odom, code, gt, u, u32, mul, None)
                b.step_c(st, a, None, 1)
                r1_all = yield st.r1
                a.r1_all = _ptr(r1_all)
                self._keep = self._keep + (r1_all,)
                if getattr(self, "_one_stream", False):  # shards of one process: every shard's rows and flags before any wait
                    b.step_c(st, a, None, 4 | 16)
                    yield st.sync
                    b.step_c(st, a, None, 8 | 16)
                else:
                    b.step_c(st, a, None, 12)
            self.step_count += 1
            return
        ready = False
        if getattr(b, "row_shard", None) is not None:  # codebook rows sharded: gather the score slices first
            st.scores.copy_((yield b.score_slice(code)))
            ready = True
        b.front(st, odom, code, gt, tn, rot, mul * self.sig_t, mul * self.sig_r, self.seed, self.step_count,
                self.pen_max, self.use_hint, scores_ready=ready)
        b.tail_a(st, self.softmax)
        r1_all = yield st.r1
        # u (parity mode): the uniforms of ALL slots of the filter, the same tensor on every rank
        if self.exchange == "a2a":
            send, sends, recvs = b.route(st, r1_all, self.rank, G, self.softmax, self.mode, u, u32, self.seed, self.step_count,
                                         gt is not None)
            recv = yield ("a2a", send, [n * ROUTE_REC for n in sends], [n * ROUTE_REC for n in recvs])
            b.unpack(st, recv)
        elif self.exchange == "a2a_fixed":
            send, ovf = b.route_fixed(st, r1_all, self.rank, G, self.softmax, self.mode, u, u32, self.seed, self.step_count, gt is not None,
                                      self.seg_cap, self.ovf_cap)
            eq = [self.seg_cap * ROUTE_REC] * G
            recv = yield ("a2a", send, eq, eq)
            ovf_all = yield ovf
            b.unpack_fixed(st, recv, ovf_all, self.rank)
            self._watch_overflow()
        elif self.exchange == "peer":
            if st._peers is None:
                raise MidasError("peer-mapped exchange: the inboxes are not connected (connect_peers / connect_local_peers)")
            b.route_push(st, r1_all, self.rank, G, self.softmax, self.mode, u, u32, self.seed, self.step_count, gt is not None)
            yield st.sync  # barrier: every rank has stored the rows it owns (the reply is not looked at)
            b.unpack_peer(st)
        else:
            b.tail_fin(st, r1_all, self.rank, G, self.N_total, self.softmax, gt is not None)
            pack_all = yield st.pack
            u_loc = None if u is None else u[self.rank * self.N:(self.rank + 1) * self.N]
            b.tail_resample(st, pack_all, self.N_total, self.mode, u_loc, u32, self.seed, self.step_count)
        self.step_count += 1

    def _exchange(self, msg):
        if isinstance(msg, tuple):
            return self.comm.all_to_all(msg[1], msg[2], msg[3])
        return self.comm.all_gather(msg)

    def step(self, odom, code, gt=None, **draws):
        gen = self.step_gen(odom, code, gt, **draws)
        try:
            msg = next(gen)
            while True:
                msg = gen.send(self._exchange(msg))
        except StopIteration:
            pass


def connect_local_peers(engines, exchange="peer"):
    """Wire the inboxes of shards that live in ONE process (tests, run_lockstep): plain pointers, nothing to map.
    exchange="peer_c": the C-side frame (midas_shard_step) with the record gather done by run_lockstep."""
    for e in engines:
        e.backend.peer_alloc(e.st)
    for e in engines:
        e.backend.peer_table(e.st, [o.st._inbox for o in engines])
        e.exchange = exchange
        e._one_stream = True


def run_lockstep(engines, per_rank_args):
    """Step several shards of ONE process in lock-step (tests): gathers are concatenations in rank order, the
    all_to_all hands every shard the segments the others addressed to it."""
    gens = [e.step_gen(*a[0], **a[1]) for e, a in zip(engines, per_rank_args)]
    msgs = [next(g) for g in gens]
    while True:
        if isinstance(msgs[0], tuple):
            G = len(msgs)
            offs = [[sum(m[2][:d]) for d in range(G)] for m in msgs]
            replies = [torch.cat([msgs[s_][1][offs[s_][d]:offs[s_][d] + msgs[s_][2][d]] for s_ in range(G)]) for d in range(G)]
            for d in range(G):
                assert [msgs[s_][2][d] for s_ in range(G)] == list(msgs[d][3]), "send / receive counts disagree"
        else:
            gathered = torch.cat([m.reshape((m.shape[0],) + tuple(m.shape[1:])) for m in msgs], dim=0)
            replies = [gathered] * len(gens)
        nxt, done = [], 0
        for g, r in zip(gens, replies):
            try:
                nxt.append(g.send(r))
            except StopIteration:
                done += 1
        if done:
            assert done == len(gens), "shards fell out of step"
            return
        msgs = nxt
