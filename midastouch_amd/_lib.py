"""ctypes binding of libmidas_hip.so (include/midas_hip.h).

The product path has no CPU fallback: if the HIP library is missing, cannot be loaded, or no
gfx950 device is visible, every entry point raises.  torch is used only as the owner of device
memory (`tensor.data_ptr()`) and of the HIP stream the kernels are enqueued on.
"""
from __future__ import annotations

import ctypes as C
import sys
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# MIDAS_HIP_LIB: another build of the same library (kernel-tuning sweeps, tools/variants.sh)
LIB_PATH = os.environ.get("MIDAS_HIP_LIB") or os.path.join(CSRC, "libmidas_hip.so")

MIDAS_F32, MIDAS_F64 = 0, 1
RESAMPLE_MULTINOMIAL, RESAMPLE_SYSTEMATIC = 0, 1
PROF_SLOTS = 8


class MidasError(RuntimeError):
    pass


class StepArgs(C.Structure):
    """midas_step_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64),
        ("poses_in", C.c_void_p),
        ("poses_prop", C.c_void_p),
        ("poses_out", C.c_void_p),
        ("weights", C.c_void_p),
        ("weights_out", C.c_void_p),
        ("hint_in", C.c_void_p),
        ("nn_idx", C.c_void_p),
        ("hint_out", C.c_void_p),
        ("ridx", C.c_void_p),
        ("odom16", C.c_void_p),
        ("code", C.c_void_p),
        ("gt16", C.c_void_p),
        ("rmse", C.c_void_p),
        ("tn", C.c_void_p),
        ("rot", C.c_void_p),
        ("u", C.c_void_p),
        ("u32", C.c_float),
        ("std_t", C.c_float),
        ("std_r", C.c_float),
        ("seed", C.c_uint64),
        ("step", C.c_uint64),
        ("prune_thr", C.c_double),
        ("softmax", C.c_int32),
        ("resample_mode", C.c_int32),
        ("status", C.c_void_p),
        ("telemetry", C.c_void_p),
        ("score_stamps", C.c_void_p),
        ("score_epoch", C.c_uint32),
    ]


class ShardFrontArgs(C.Structure):
    """midas_shard_front_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64), ("slot_base", C.c_int64),
        ("poses_in", C.c_void_p), ("poses_prop", C.c_void_p), ("hint_in", C.c_void_p), ("nn_idx", C.c_void_p),
        ("valid", C.c_void_p), ("odom16", C.c_void_p), ("code", C.c_void_p), ("scores", C.c_void_p),
        ("scores_ready", C.c_int32), ("gt16", C.c_void_p), ("rmse_sums", C.c_void_p),
        ("tn", C.c_void_p), ("rot", C.c_void_p),
        ("std_t", C.c_float), ("std_r", C.c_float), ("seed", C.c_uint64), ("step", C.c_uint64),
        ("prune_thr", C.c_double),
        ("telemetry", C.c_void_p),
        ("status", C.c_void_p),
        ("flags", C.c_void_p),
        ("score_stamps", C.c_void_p), ("score_epoch", C.c_uint32),
    ]


class LazyArgs(C.Structure):
    """midas_lazy_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64),
        ("poses_prop_prev", C.c_void_p), ("nn_idx_prev", C.c_void_p), ("status_prev", C.c_void_p),
        ("poses_prop", C.c_void_p), ("nn_idx", C.c_void_p), ("valid", C.c_void_p), ("status", C.c_void_p),
        ("tables", C.c_void_p), ("scores", C.c_void_p), ("part_rmse", C.c_void_p),
        ("resample_prev", C.c_int32), ("poses_in", C.c_void_p), ("hint_in", C.c_void_p),
        ("resample_mode", C.c_int32), ("u_prev", C.c_void_p), ("u32_prev", C.c_float), ("step_prev", C.c_uint64),
        ("ridx", C.c_void_p),
        ("odom16", C.c_void_p), ("code", C.c_void_p), ("gt16", C.c_void_p), ("tn", C.c_void_p), ("rot", C.c_void_p),
        ("std_t", C.c_float), ("std_r", C.c_float), ("seed", C.c_uint64), ("step", C.c_uint64),
        ("prune_thr", C.c_double), ("softmax", C.c_int32), ("telemetry", C.c_void_p),
        ("score_stamps", C.c_void_p), ("score_epoch", C.c_uint32), ("rmse", C.c_void_p), ("score_list", C.c_void_p),
        ("guide", C.c_void_p),
    ]


class LazyFlushArgs(C.Structure):
    """midas_lazy_flush_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64),
        ("tables", C.c_void_p), ("valid", C.c_void_p), ("nn_idx", C.c_void_p), ("poses_prop", C.c_void_p),
        ("status", C.c_void_p), ("part_rmse", C.c_void_p),
        ("softmax", C.c_int32), ("resample_mode", C.c_int32), ("u", C.c_void_p), ("u32", C.c_float),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("weights", C.c_void_p), ("ridx", C.c_void_p), ("poses_out", C.c_void_p), ("weights_out", C.c_void_p),
        ("hint_out", C.c_void_p), ("rmse", C.c_void_p),
    ]


class ShardRouteArgs(C.Structure):
    """midas_shard_route_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64), ("G", C.c_int32), ("rank", C.c_int32),
        ("r1_all", C.c_void_p), ("tables", C.c_void_p), ("valid", C.c_void_p), ("nn_idx", C.c_void_p),
        ("poses_prop", C.c_void_p), ("status", C.c_void_p), ("rmse", C.c_void_p),
        ("softmax", C.c_int32), ("resample_mode", C.c_int32), ("u_all", C.c_void_p), ("u32", C.c_float),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("counts", C.c_void_p), ("send", C.c_void_p), ("weights", C.c_void_p),
        ("fixed_cap", C.c_int64), ("ovf_cap", C.c_int64), ("ovf", C.c_void_p), ("self_rows", C.c_void_p),
        ("peers", C.c_void_p),
        ("guide", C.c_void_p),
    ]


class ShardStepArgs(C.Structure):
    """midas_shard_step_args (include/midas_hip.h)."""

    _fields_ = [
        ("front", ShardFrontArgs),
        ("softmax", C.c_int32), ("tables", C.c_void_p), ("r1", C.c_void_p), ("r1_all", C.c_void_p),
        ("G", C.c_int32), ("rank", C.c_int32), ("resample_mode", C.c_int32),
        ("u_all", C.c_void_p), ("u32", C.c_float),
        ("counts", C.c_void_p), ("weights", C.c_void_p), ("rmse", C.c_void_p),
        ("peers", C.c_void_p), ("inbox", C.c_void_p), ("flag_offset", C.c_int64), ("frame_tag", C.c_uint64),
        ("ridx", C.c_void_p), ("poses_out", C.c_void_p), ("weights_out", C.c_void_p), ("hint_out", C.c_void_p),
        ("score_list", C.c_void_p), ("guide", C.c_void_p),
    ]


class TailResampleArgs(C.Structure):
    """midas_tail_resample_args (include/midas_hip.h)."""

    _fields_ = [
        ("N", C.c_int64), ("N_all", C.c_int64), ("slot_base", C.c_int64),
        ("pack_all", C.c_void_p), ("rank_stride", C.c_int64), ("n_per_rank", C.c_int64),
        ("status", C.c_void_p), ("mode", C.c_int32), ("u", C.c_void_p), ("u32", C.c_float),
        ("seed", C.c_uint64), ("step", C.c_uint64), ("ridx", C.c_void_p),
        ("poses_out", C.c_void_p), ("weights_out", C.c_void_p), ("hint_out", C.c_void_p),
    ]


class MtSegment(C.Structure):
    """midas_mt_segment (include/midas_hip.h): one draw of a midas_mt19937_draws call."""

    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("count", C.c_int64), ("mean", C.c_float), ("std", C.c_float),
                ("out_dev", C.c_void_p)]


MT_SEGMENT_RAND64, MT_SEGMENT_NORMAL32 = 0, 1


class LoopArgs(C.Structure):
    """midas_loop_args (include/midas_hip.h)."""

    _fields_ = [
        ("cap", C.c_int64), ("ctl_i", C.c_void_p), ("ctl_d", C.c_void_p),
        ("poses", C.c_void_p), ("poses_prop", C.c_void_p), ("hint", C.c_void_p), ("nn_idx", C.c_void_p), ("valid", C.c_void_p),
        ("x", C.c_void_p), ("e", C.c_void_p), ("weights", C.c_void_p), ("weights_out", C.c_void_p),
        ("labels", C.c_void_p), ("labels_out", C.c_void_p), ("src", C.c_void_p), ("ridx", C.c_void_p),
        ("scores", C.c_void_p), ("part_rmse", C.c_void_p), ("cb_poses", C.c_void_p),
        ("cluster_poses", C.c_void_p), ("cluster_stds", C.c_void_p), ("log", C.c_void_p),
        ("odom16", C.c_void_p), ("code", C.c_void_p), ("gt16", C.c_void_p), ("tn", C.c_void_p), ("rot", C.c_void_p),
        ("u", C.c_void_p), ("u32", C.c_float), ("std_t", C.c_float), ("std_r", C.c_float),
        ("seed", C.c_uint64), ("step", C.c_uint64), ("prune_thr", C.c_double),
        ("softmax", C.c_int32), ("resample_mode", C.c_int32), ("floor", C.c_int32), ("eps", C.c_double),
        ("unit_weights", C.c_int32), ("telemetry", C.c_void_p),
        ("score_stamps", C.c_void_p), ("score_epoch", C.c_uint32),
        ("host_mirror", C.c_void_p), ("grid_n", C.c_int64), ("anneal_small", C.c_int32), ("topk_ties", C.c_int32),
        ("anneal_frozen", C.c_int32), ("pad2_", C.c_int32),
    ]


# phases of midas_loop_step and the control-block indices (include/midas_hip.h MIDAS_LOOP_*)
LOOP_FRONT, LOOP_DBSCAN, LOOP_ANNEAL, LOOP_RESAMPLE = 1, 2, 4, 8
MT19937_HIST_WORDS = 20560  # MIDAS_MT19937_HIST_WORDS
TOPK_TIES_INDEX, TOPK_TIES_ATEN_CPU = 0, 1  # whom annealing's torch.topk takes inside a tie (include/midas_hip.h)
LOOP_MAX_CLUSTERS, LOOP_LOG_DOUBLES = 64, 168
(LOOP_I_N, LOOP_I_NSET, LOOP_I_MODE, LOOP_I_K, LOOP_I_INIT, LOOP_I_VARSET, LOOP_I_KEPT, LOOP_I_DRIFT, LOOP_I_STATUS,
 LOOP_I_RAW, LOOP_I_NCL, LOOP_I_NPRES, LOOP_I_FRAME, LOOP_I_NAN, LOOP_I_ERR) = range(15)
(LOOP_D_S, LOOP_D_VARPREV, LOOP_D_VAR, LOOP_D_RMSE_T, LOOP_D_RMSE_R, LOOP_D_XMAX, LOOP_D_XMIN, LOOP_D_TOTAL) = range(8)

# name -> (restype, argtypes); must list every symbol include/midas_hip.h declares
_P, _I32, _I64, _U64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
SIGNATURES = {
    "midas_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "midas_ctx_destroy": (C.c_int, [_P]),
    "midas_ctx_set_stream": (C.c_int, [_P, _P]),
    "midas_sync": (C.c_int, [_P]),
    "midas_selftest_wave_sums": (C.c_int, [_P, _P, _P]),
    "midas_scratch_reserve": (C.c_int, [_P, C.c_int64]),
    "midas_strerror": (C.c_char_p, [C.c_int]),
    "midas_last_error": (C.c_char_p, [_P]),
    "midas_version": (C.c_char_p, []),
    "midas_codebook_create": (C.c_int, [_P, _I64, _I32, _P, _I32, C.POINTER(_P)]),
    "midas_codebook_destroy": (C.c_int, [_P]),
    "midas_score": (C.c_int, [_P, _P, _I32, _P, _P]),
    "midas_score_batch": (C.c_int, [_P, _P, _I32, _P, _P]),
    "midas_se3_feature": (C.c_int, [_P, _I64, _P, _F, _P]),
    "midas_tree_build": (C.c_int, [_P, _I32, _I64, _P, C.POINTER(_P)]),
    "midas_tree_destroy": (C.c_int, [_P]),
    "midas_tree_attach_mesh": (C.c_int, [_P, _P, _P, _P]),
    "midas_nn6": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "midas_knn6": (C.c_int, [_P, _P, _I64, _P, C.c_int32, _P, _P]),
    "midas_tree_export": (C.c_int, [_P, _P, C.c_int32, _P, _I64]),
    "midas_nn6_stats": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "midas_nn3": (C.c_int, [_P, _P, _I64, _P, _P]),
    "midas_propagate": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P, _F, _F, _U64, _U64]),
    "midas_check_poses": (C.c_int, [_P, _I64, _P, _P, _P]),
    "midas_gather_f64": (C.c_int, [_P, _I64, _P, _P, _P]),
    "midas_softmax": (C.c_int, [_P, _I64, _P, _I32, _P]),
    "midas_prune": (C.c_int, [_P, _I64, _P, _P, _D, _P]),
    "midas_cdf": (C.c_int, [_P, _I64, _P, _P, _P]),
    "midas_score_list_seed": (C.c_int, [_P, _I64, _P, C.c_uint32, _P, _I64, _P]),
    "midas_mt19937_seed": (C.c_int, [_P, _U64, _P]),
    "midas_mt19937_rand64": (C.c_int, [_P, _P, _I64, _I64, _P]),
    "midas_mt19937_rand64_chunked": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _I32]),
    "midas_mt19937_normal32": (C.c_int, [_P, _P, _I64, _I64, _F, _F, _P, _P, _P, _P, _P, _P, _I32]),
    "midas_mt19937_draws": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32]),
    "midas_resample_search": (C.c_int, [_P, _I64, _P, _I64, _I32, _P, _F, _U64, _U64, _P]),
    "midas_gather_rows": (C.c_int, [_P, _I64, _P, _P, _P, _I32]),
    "midas_rmse": (C.c_int, [_P, _I64, _P, _P, _P]),
    "midas_topn_pose_error": (C.c_int, [_P, _I32, _I64, _P, _I64, _I32, _P, _I32, _P, _P]),
    "midas_selfsim_topn": (C.c_int, [_P, _P, _I32, _P, _I32, _I64, _P, _P]),
    "midas_selfsim_panel": (C.c_int, [_P, _P, _I64, _I64, _P, _I64]),
    "midas_cluster_centers": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I32, _P, _P, _P, _P]),
    "midas_filter_step": (C.c_int, [_P, _P, _P, _P, C.POINTER(StepArgs)]),
    "midas_filter_step_batch": (C.c_int, [_P, _P, _P, _P, C.POINTER(StepArgs), _I32]),
    "midas_lazy_step": (C.c_int, [_P, _P, _P, _P, C.POINTER(LazyArgs)]),
    "midas_lazy_run": (C.c_int, [_P, _P, _P, _P, C.POINTER(LazyArgs), _I32, _P]),
    "midas_lazy_flush": (C.c_int, [_P, C.POINTER(LazyFlushArgs)]),
    "midas_lazy_tables_doubles": (C.c_int64, [_I64]),
    "midas_lazy_guide_bytes": (C.c_int64, [_I64]),
    "midas_lazy_guide_layout": (C.c_int, [_P, _P, _P]),
    "midas_lazy_step_batch": (C.c_int, [_P, _P, _P, _P, C.POINTER(LazyArgs), _I32]),
    "midas_lazy_flush_batch": (C.c_int, [_P, C.POINTER(LazyFlushArgs), _I32]),
    "midas_loop_step": (C.c_int, [_P, _P, _P, _P, C.POINTER(LoopArgs), _I32]),
    "midas_dbscan": (C.c_int, [_P, _I64, _P, _D, _I64, _P, _P]),
    "midas_dbscan_points": (C.c_int, [_P, _I64, C.c_int32, _P, _D, _I64, _P, _P]),
    "midas_anneal_select": (C.c_int, [_P, _I64, _P, _I32, _I64, _P]),
    "midas_anneal_select_ties": (C.c_int, [_P, _I64, _P, _I32, _I64, _I32, _P, _P]),
    "midas_shard_front": (C.c_int, [_P, _P, _P, _P, C.POINTER(ShardFrontArgs)]),
    "midas_shard_tail_a": (C.c_int, [_P, _I64, _P, _P, _P, _I32, _P, _P, _P]),
    "midas_shard_tail_fin": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I32, _P, _I32, _I64, _I32, _P, _P]),
    "midas_shard_route_count": (C.c_int, [_P, C.POINTER(ShardRouteArgs)]),
    "midas_shard_route_pack": (C.c_int, [_P, C.POINTER(ShardRouteArgs)]),
    "midas_shard_unpack": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P]),
    "midas_shard_unpack_rows": (C.c_int, [_P, _I64, _P, _I32, _P, _P, _P, _P]),
    "midas_shard_unpack_fixed": (C.c_int, [_P, _I64, _P, _I64, _P, _I32, _I64, _P, _P, _P, _P, _P]),
    "midas_shard_unpack_peer": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P]),
    "midas_comm_unique_id": (C.c_int, [_P, C.c_char_p, _P]),
    "midas_comm_create": (C.c_int, [_P, C.c_char_p, _P, _I32, _I32, C.POINTER(C.c_void_p)]),
    "midas_comm_destroy": (C.c_int, [_P]),
    "midas_comm_all_gather": (C.c_int, [_P, _P, _P, _I64]),
    "midas_shard_step": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ShardStepArgs), _I32]),
    "midas_shard_run": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ShardStepArgs), _I32]),
    "midas_peer_alloc": (C.c_int, [_P, _I64, C.POINTER(C.c_void_p), _P]),
    "midas_peer_free": (C.c_int, [_P, _P]),
    "midas_peer_open": (C.c_int, [_P, _P, C.POINTER(C.c_void_p)]),
    "midas_peer_close": (C.c_int, [_P, _P]),
    "midas_peer_probe_write": (C.c_int, [_P, _P, _I32, _I32, _I32]),
    "midas_peer_probe_check": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "midas_tail_resample": (C.c_int, [_P, C.POINTER(TailResampleArgs)]),
    "midas_profile_enable": (C.c_int, [_P, _I32]),
    "midas_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "midas_profile_slot_name": (C.c_char_p, [_I32]),
}


def build(force: bool = False) -> str:
    """Compile libmidas_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "midas_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC, "-j4", "-s"] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def load():
    """Load the library and bind every declared symbol; raises MidasError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MidasError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch first: the library shares the process's HIP runtime with torch (device memory and streams come from
    # torch).  torch ships its own libamdhip64; if this library were loaded before it, the loader would bind
    # /opt/rocm's copy instead and the two runtimes would not see each other's devices and streams.
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise MidasError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MidasError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Context:
    """One midas_ctx bound to a device and to torch's current HIP stream on that device."""

    def __init__(self, device=None):
        import torch

        if not torch.cuda.is_available():
            raise MidasError("no HIP device visible (torch.cuda.is_available() is False); the filter kernels "
                             "run only on an MI355X - there is no CPU fallback")
        self.lib = load()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise MidasError(f"device must be a HIP device, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self._stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = self.lib.midas_ctx_create(self.device.index, C.c_void_p(self._stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise MidasError(f"midas_ctx_create failed: {self.lib.midas_strerror(rc).decode()}")
        self.h = h

    def check(self, rc: int):
        if rc != 0:
            detail = self.lib.midas_last_error(self.h)
            raise MidasError(detail.decode() if detail else self.lib.midas_strerror(rc).decode())

    def bind_current_stream(self):
        """Follow torch's current stream on this device (kernels stay ordered with torch's own work)."""
        import torch

        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle alone: a third of current_stream()'s cost
        if raw is not None:
            if raw(self.device.index) == self._stream.cuda_stream:
                return
        s = torch.cuda.current_stream(self.device)
        if s.cuda_stream != self._stream.cuda_stream:
            self.check(self.lib.midas_ctx_set_stream(self.h, C.c_void_p(s.cuda_stream)))
            self._stream = s

    def call(self, name: str, *args):
        self.bind_current_stream()
        self.check(getattr(self.lib, name)(self.h, *args))

    def sync(self):
        self.call("midas_sync")

    def close(self):
        if getattr(self, "h", None):
            self.lib.midas_ctx_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):  # pragma: no cover  # (bound at import: module globals are gone by then)
        if _finalizing():  # the process is going away: the HIP runtime may be gone already (its calls would abort, not raise)
            return
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context(device=None) -> Context:
    """Process-wide context per device (created on first use)."""
    import torch

    if not torch.cuda.is_available():
        raise MidasError("no HIP device visible (torch.cuda.is_available() is False); the filter kernels "
                         "run only on an MI355X - there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise MidasError(f"tensors must live on a HIP device, got {dev}; there is no CPU fallback")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _contexts:
        _contexts[idx] = Context(torch.device("cuda", idx))
    return _contexts[idx]
