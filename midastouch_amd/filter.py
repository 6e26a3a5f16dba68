"""Filter runner: the reference's per-frame loop (`midastouch/filter/filter.py:42-256`) on the MI355X path.

Two ways to run a sequence:

* `filter(cfg, ...)`  - the reference's loop body call for call (motionModel -> particle_rmse -> SE3_NN ->
  get_similarity -> remove_invalid_particles -> cluster_particles/get_cluster_centers -> annealing ->
  resampler), same order of operations, same RNG draw order, same guards and the same `filter_stats`
  keys (:99-116).  It is the counterpart of row H of SURVEY.md 8(a).
* `step(...) / update_weights(...) / resample(...)` - the north-star aliases.  `step` drives a
  `FilterEngine` (one fused C-ABI call per frame, fixed N); `update_weights` is SE3_NN + get_similarity
  without the (N, D) gather; `resample` is `particle_filter.resampler`.

Upstream perception (TDN/TCN, `filter.py:144-147`) is out of scope: tactile codes are inputs.  The
datasets of the reference are external downloads, so sequences come from `synthetic.py`.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .engine import FilterEngine
from .particle_filter import Particles, particle_filter, particle_rmse
from .synthetic import make_codebook, make_trajectory
from .tactile_tree import tactile_tree


@dataclass
class Sequence:
    """What the reference loads per log: gt / measured poses (`extract_poses_sim`, pose.py:272-300) and,
    instead of tactile images, the tactile codes the TCN would emit for them."""
    gt_p: torch.Tensor       # (T,4,4) f32
    meas_p: torch.Tensor     # (T,4,4) f32
    codes: torch.Tensor      # (T,D)   f64
    codebook: tactile_tree
    mesh_vertices: np.ndarray
    obj_model: str
    mesh_tree: Optional[object] = None  # an ops.Tree over mesh_vertices already on the device (shared with other engines)


def synthetic_sequence(cfg, device, T: int = 100, D: Optional[int] = None, seed: int = 0) -> Sequence:
    obj = cfg.expt.obj_model
    K = int(cfg.expt.codebook_size)
    D = int(D if D is not None else cfg.tcn.model.output_dim)
    cb = make_codebook(obj, K=K, D=D, seed=1000 + seed)
    traj = make_trajectory(cb, T=T, seed=2000 + seed)
    tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
    tree.to_device(device)
    return Sequence(torch.as_tensor(traj.gt_poses).to(device), torch.as_tensor(traj.meas_poses).to(device),
                    torch.as_tensor(traj.codes).to(device), tree, cb.mesh_vertices, obj)


def load_sequence(sequence_npz: str, codebook_path: str, device, obj_model: str = "object", check_logmap: bool = False) -> Sequence:
    """A recorded sequence: `codebook_path` is a `codebook.npz` or a reference `codebook.pkl` (codebook_io.py);
    `sequence_npz` holds `gt_poses (T,4,4)`, `meas_poses (T,4,4)`, `codes (T,D)` (the TCN outputs, upstream of this
    path) and `mesh_vertices (M,3)` (the decimated STL vertices of `particle_filter.__init__`, :108-110)."""
    tree = tactile_tree.load(codebook_path, device=device, check_logmap=check_logmap)
    with np.load(sequence_npz) as z:
        need = ("gt_poses", "meas_poses", "codes", "mesh_vertices")
        missing = [k for k in need if k not in z.files]
        if missing:
            raise ValueError(f"{sequence_npz}: arrays {missing} missing")
        gt, meas, codes, verts = (np.asarray(z[k]) for k in need)
    if codes.shape[1] != tree.embeddings.shape[1]:
        raise ValueError(f"tactile codes are {codes.shape[1]}-d, the codebook's embeddings {tree.embeddings.shape[1]}-d")
    return Sequence(torch.as_tensor(gt, dtype=torch.float32).to(device), torch.as_tensor(meas, dtype=torch.float32).to(device),
                    torch.as_tensor(codes, dtype=torch.float64).to(device), tree, verts, obj_model)


# ---- north-star aliases ---------------------------------------------------------------------------
def update_weights(pf: particle_filter, codebook: tactile_tree, particles: Particles, tactile_code: torch.Tensor,
                   softmax: bool = True) -> Particles:
    """particles.weights <- similarity of the tactile code to each particle's nearest codebook pose
    (filter/filter.py:170-173) - the codebook is scored once, no (N, D) gather."""
    _, _, nn_codes = codebook.SE3_NN(particles.poses)
    particles.weights = pf.get_similarity(tactile_code, nn_codes, softmax=softmax)
    return particles


def resample(pf: particle_filter, particles: Particles, mode: str = "weighted_random") -> Particles:
    return pf.resampler(particles, resample=mode)


def step(engine: FilterEngine, odom: torch.Tensor, tactile_code: torch.Tensor, gt_pose: torch.Tensor = None, **draws):
    """One fused frame on a FilterEngine (propagate -> score/NN -> weights -> prune -> resample)."""
    engine.step(odom, tactile_code, gt=gt_pose, **draws)
    return engine


# ---- the reference loop -----------------------------------------------------------------------------
def filter(cfg, seq: Optional[Sequence] = None, viz=None, device=None, pace: str = "fixed", max_frames: int = None,
           cluster: bool = True, progress: bool = False, softmax: bool = True, update_freq: int = 1, floor: int = 1000,
           results_path: Optional[str] = None, draws: str = "device", seed: int = 4000) -> dict:
    """Run the filter over a sequence; returns the reference's `filter_stats` dict (filter.py:99-116).

    The loop body (filter.py:150-190: motionModel -> particle_rmse -> SE3_NN + get_similarity -> remove_invalid_particles
    [-> re-projection] -> cluster_particles every 50th frame -> get_cluster_centers -> annealing -> resampler) is one
    `LoopEngine.step` per frame: same order of operations, same guards, the particle count on the device, nothing read back
    between frames; rmse, particle counts and cluster centres come out of the engine's frame log after the last frame.

    draws="device" (default): Philox streams on the device, frames are enqueued back to back.  draws="host": the
    reference's own random streams - torch.normal tn, rot and the resampler's torch.rand on the CPU generator, in its
    order - which needs the particle count on the host twice per frame (bit-parity with the reference under
    torch.manual_seed, at the price of those read-backs and of the host generator); annealing's top-k then also resolves its
    ties the way torch.topk does on the CPU (LoopEngine(topk_ties="aten_cpu")), so the particle SET is the reference's too.
    draws="seeded": the SAME numbers as draws="host" - torch's default CPU generator continued on the device (torch_rng.py:
    its mt19937 stream, `torch.normal`'s float32 transform as tables read off torch itself, `torch.rand` float64) - without the
    host generating and uploading 6 N normals and N uniforms a frame; the host-side draws of `init_filter` take the stream
    over and hand it back, and at the end torch's generator stands where draws="host" leaves it.
    pace="fixed" steps one frame per iteration; pace="wallclock" reproduces `idx = int(frame_rate * total_time)`
    (:134-135: slow iterations skip frames, fast ones repeat) and therefore waits for every frame.
    The defaults are `filter/filter.py`'s; `filter_real(...)` presets the real-data script's variations
    (`filter/filter_real.py`: raw scores :208-210, measurement update every `update_freq`-th frame with unit weights in
    between :205-212, annealing floor 10000 :228).  results_path: directory `filter_stats.npy` is written to (:252).
    """
    from . import _lib
    from .loop_engine import LoopEngine

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if draws not in ("device", "host", "seeded"):
        raise ValueError("draws must be 'device', 'host' or 'seeded'")
    expt_cfg = cfg.expt
    init_particles = int(expt_cfg.params.num_particles)
    noise_ratio = expt_cfg.params.noise_ratio
    frame_rate = expt_cfg.frame_rate
    if seq is None:
        seq = synthetic_sequence(cfg, device)
    gt_p, meas_p, codebook = seq.gt_p, seq.meas_p, seq.codebook
    traj_size = gt_p.shape[0] if expt_cfg.max_length in (None, "None") else min(gt_p.shape[0], int(expt_cfg.max_length))
    if max_frames is not None:
        traj_size = min(traj_size, max_frames)
    pf = particle_filter(cfg, seq.mesh_vertices, noise_ratio, downsample=1, device=device)
    if seq.mesh_tree is not None:
        pf._mesh_tree = seq.mesh_tree
    eng = LoopEngine(codebook, None, pf.mesh_kdtree, init_particles, sig_t=pf.motion_noise["sig_t"], sig_r=pf.motion_noise["sig_r"],
                     pen_max=pf.pen_max, seed=seed, softmax=softmax, floor=floor, cluster=cluster, log_frames=max(traj_size + 8, 64),
                     device=device, topk_ties="aten_cpu" if draws in ("host", "seeded") else "index")
    stream = None
    if draws == "seeded":
        from .torch_rng import TorchCpuStream
        stream = TorchCpuStream(0, device)
    on_device = False  # seeded: who holds torch's stream at the moment (False: the host's default generator)

    def stream_to(dev_side: bool):
        nonlocal on_device
        if stream is None or on_device == dev_side:
            return
        stream.from_host() if dev_side else stream.to_host()
        on_device = dev_side

    def draw_normal(std, n):  # add_noise_to_odom's torch.normal(mu, std, (n, 3)) (:326-335)
        if stream is not None and 3 * n >= 16:
            stream_to(True)
            return stream.normal(pf.motion_noise["mu"], std, (n, 3))
        stream_to(False)  # (host mode, or fewer than 16 values: ATen's scalar path, drawn where it is defined)
        return torch.normal(mean=pf.motion_noise["mu"], std=std, size=(n, 3))

    def draw_uniform(n):  # the resampler's torch.multinomial stream (:245)
        if stream is not None:
            stream_to(True)
            return stream.rand64(n)
        return torch.rand(n, dtype=torch.float64)
    # odom = inv(meas[prev]) @ meas[idx] (:154): the inverses in one call, and - frame after frame in fixed pace - the
    # products too (a 4x4 product per frame through the BLAS library costs more device time than the whole frame's kernels)
    inv_meas = torch.linalg.inv(meas_p)
    step_odoms = torch.matmul(inv_meas[:-1], meas_p[1:]).contiguous() if meas_p.shape[0] > 1 else None
    eye = torch.eye(4, device=device)
    heatmap_poses, _ = codebook.get_poses()
    heatmap_embeddings = codebook.get_embeddings()
    every_frame = pace == "wallclock" or viz is not None or progress  # somebody looks at every frame: wait for it

    filter_stats = {
        "rmse_t": [], "rmse_r": [], "time": [], "traj_size": traj_size, "avg_time": None, "total_time": 0,
        "cluster_poses": [], "cluster_stds": [], "obj_name": seq.obj_model, "tree_size": len(codebook),
        "noise_ratio": noise_ratio, "init_noise": pf.init_noise, "init_particles": init_particles,
        "num_particles": [], "log_id": str(expt_cfg.log_id).zfill(2), "trial_id": 0,
    }
    motion_time = []
    events = [torch.cuda.Event(enable_timing=True)]
    events[0].record()
    prev_idx, count, fixed_idx, total_time = 0, 0, 0, 0.0
    frame_idx = []  # the sequence frame each iteration looked at (pace="wallclock" skips and repeats)
    records = []    # frame-log rows read back so far
    import gc
    gc.collect()
    gc.freeze()  # the libraries' ~10^6 long-lived objects out of the collector's way: no 30 ms pass in the middle of a run
    try:
        while True:
            if pace == "wallclock":
                idx = int(frame_rate * total_time)
            else:
                idx = fixed_idx
                fixed_idx += 1
            if idx >= traj_size:
                break
            if count and count % eng.log_frames == 0:
                # wall-clock pace on a fast host repeats frames: more iterations than the engine's frame log (a ring) holds
                records.extend(eng.read_log(count - eng.log_frames, count))
            start = time.time()
            moving = prev_idx > 0
            if not moving:  # (filter.py:152,156-160) - like the reference, frames seen while prev_idx == 0 re-initialise
                stream_to(False)  # init_filter draws on the host generator
                particles = pf.init_filter(gt_p[idx, :], init_particles)
                eng.set_particles(particles.poses, reset_annealing=count == 0)
                eng.project_to_codebook()
                odom = eye
            else:
                odom = step_odoms[prev_idx] if idx == prev_idx + 1 else inv_meas[prev_idx] @ meas_p[idx, :]
            unit = count % max(int(update_freq), 1) != 0  # filter_real.py:205-212: no measurement update on this frame
            kw = dict(gt=gt_p[idx, :], dbscan=cluster and count % 50 == 0, unit_weights=unit, std_override=None if moving else (0.0, 0.0))
            if draws in ("host", "seeded"):
                n = eng.n
                std_t, std_r = (pf.motion_noise["sig_t"], pf.motion_noise["sig_r"]) if moving else (0.0, 0.0)
                if moving:  # add_noise_to_odom's draws, its order (:326-335); the initial frames draw inside init_filter
                    kw["tn"] = draw_normal(std_t, n)
                    kw["rot"] = draw_normal(std_r, n)
                else:
                    kw["tn"], kw["rot"] = torch.zeros((n, 3)), torch.zeros((n, 3))
                eng.step(odom, seq.codes[idx], phases=_lib.LOOP_FRONT | _lib.LOOP_DBSCAN | _lib.LOOP_ANNEAL, **kw)
                n_set = int(eng.ctl_i[_lib.LOOP_I_NSET].item())
                eng.step(None, None, u=draw_uniform(n_set), phases=_lib.LOOP_RESAMPLE)  # multinomial's stream
            else:
                eng.step(odom, seq.codes[idx], **kw)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            events.append(ev)
            if every_frame:
                ev.synchronize()
                total_time += events[-2].elapsed_time(ev) * 1e-3 if pace != "wallclock" else time.time() - start
            if viz is not None:
                heatmap_weights = pf.get_similarity(seq.codes[idx][None], heatmap_embeddings, softmax=False)
                rec = eng.frame_view()
                # the visualiser reads particles.poses asynchronously (viz/visualizer.py:329-361): hand it a snapshot
                snap = Particles(rec["poses"].clone(), rec["weights_res"].clone(), rec["labels"].clone())
                viz.update(snap, torch.as_tensor(rec["cluster_poses"]), torch.as_tensor(rec["cluster_stds"]), gt_p[idx, :], heatmap_poses,
                           heatmap_weights, None, None, None, idx)
            if progress:
                rec = eng.read_log(count, count + 1)[0]
                print(f"[{idx}] RMSE {1000 * rec['rmse_t']:.1f} mm {rec['rmse_r']:.0f} deg P {rec['n_after']} "
                      f"rate {1.0 / max(events[-2].elapsed_time(ev) * 1e-3, 1e-9):.1f} Hz")
            motion_time.append(time.time() - start)
            frame_idx.append(idx)
            prev_idx = idx
            count += 1
        torch.cuda.synchronize(device)
        stream_to(False)  # seeded: torch's generator continues where the run's draws ended
    finally:
        gc.unfreeze()  # also when a frame raises: a frozen collector would outlive the run
    records.extend(eng.read_log(len(records), count))
    for rec, e0, e1 in zip(records, events, events[1:]):
        filter_stats["rmse_t"].append(rec["rmse_t"])
        filter_stats["rmse_r"].append(rec["rmse_r"])
        filter_stats["cluster_poses"].append(torch.as_tensor(rec["cluster_poses"]))
        filter_stats["cluster_stds"].append(torch.as_tensor(rec["cluster_stds"]))
        filter_stats["num_particles"].append(rec["n_after"])
        filter_stats["time"].append(e0.elapsed_time(e1) * 1e-3)  # frame to frame on the device's clock
    filter_stats["total_time"] = sum(filter_stats["time"])
    filter_stats["avg_time"] = filter_stats["total_time"] / max(len(filter_stats["time"]), 1)
    # the reference's three timers: perception is upstream of this path; motion and measurement share one launch here
    filter_stats["avg_timer"] = {"tactile": 0.0, "motion": 0.0, "meas": filter_stats["avg_time"],
                                 "host_enqueue": float(np.average(motion_time)) if motion_time else 0.0}
    filter_stats["frames"] = records
    filter_stats["frame_idx"] = frame_idx
    filter_stats["host_time"] = motion_time  # seconds the host spent on each iteration (enqueueing; waiting too when somebody looks at every frame)
    if results_path is not None:
        save_filter_stats(filter_stats, results_path)
    return filter_stats


def save_filter_stats(filter_stats: dict, results_path: str) -> str:
    """`np.save(results_path/filter_stats.npy, filter_stats)` (filter.py:252) - a pickled dict, read back with
    `np.load(..., allow_pickle=True).item()`; device tensors are moved to the host first."""
    import os

    os.makedirs(results_path, exist_ok=True)
    host = {k: ([t.cpu() if torch.is_tensor(t) else t for t in v] if isinstance(v, list) else
                (v.cpu() if torch.is_tensor(v) else v)) for k, v in filter_stats.items()}
    path = os.path.join(results_path, "filter_stats.npy")
    np.save(path, host)
    return path


def filter_real(cfg, seq: Optional[Sequence] = None, **kw) -> dict:
    """The loop with `filter/filter_real.py`'s settings: raw similarity scores as weights, annealing floor 10000."""
    kw.setdefault("softmax", False)
    kw.setdefault("floor", 10000)
    return filter(cfg, seq, **kw)


def main(argv=None):
    import sys

    from .config import load_config

    cfg = load_config(list(sys.argv[1:] if argv is None else argv))
    stats = filter(cfg, progress=True, results_path=".")
    print(f"Total time: {stats['total_time']:.3f}, Per iteration time: {stats['avg_time']:.4f}")
    t = stats["avg_timer"]
    print(f'Avg time: tactile: {t["tactile"]:.2f}, motion : {t["motion"]:.2f}, meas : {t["meas"]:.2f} ')


if __name__ == "__main__":
    main()
