#!/usr/bin/env python3
"""bench.py - filter steps/s of the MidasTouch particle-filter hot path on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "c2"): 004_sugar_box synthetic trajectory, N = 100 000 particles
per GPU, 50 000 x 512-d codebook.  A step = one frame of the filter loop body
(score codebook -> propagate -> feature -> NN -> softmax weights -> prune -> CDF -> resample -> gather,
rmse epilogue), inputs (codebook, trajectory, particles) resident in HBM, random draws from the
on-device Philox streams (real work inside the timed region).

Protocol: particles start from `init_filter(gt_0, N)` (sigma_t = mesh scale / 3, sigma_r = 60 deg,
particle_filter.py:124-145) projected onto the codebook (filter.py:159-160); the first 20 frames after that wide start are
timed on their own (`config.diffuse_regime`: stale hints, tree-search fallbacks), then W warm-up steps, then exactly K
timed steps enqueued by ONE C-ABI call (midas_lazy_run) between barrier + synchronize; a second pass of K steps with one
HIP event per step gives `config.ms_per_step_median / _p95`.

Single GPU: the pipelined engine - the resample + gather of frame t runs as a prologue of frame t+1's front kernel
(a per-slot dependence), so every timed step performs one resample (the previous frame's), one propagate / NN /
prune, one codebook scoring and one softmax / CDF pass and leaves its rmse: the same work per step, two launches instead
of three, the resampled poses never written to HBM.  The K-th frame's resample (nobody's prologue yet) is materialised
by `eng.flush()` INSIDE the timed region: `value` contains K complete resamples.  Beside the headline (the fixed-N step of SURVEY.md 8(d)) `config` carries the rates a caller sees who
wants more per frame: `steps_per_sec_materialised_every_frame` (the particle set read after every frame, three launches)
and `reference_loop_frames_per_sec` (midastouch_amd.filter.filter: the reference's loop with DBSCAN every 50th frame,
cluster centres and annealing every frame on the device-side particle count, N0 = N).

Multi-GPU (N>1): one filter whose particles are sharded across the ranks (N_total = gpus x 100k,
weak scaling); per frame the ranks all_gather one small record of per-block sums / extrema and exchange the
resampled particles over RCCL (all_to_all of the rows each rank needs, resolved by the owners of the sources).
`value` = frames processed by all ranks / wall time = gpus x steps / t.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec peak (6.3 TB/s achievable)
PER_PARTICLE_UPDATE = 140  # bytes, SURVEY.md section 8(d): propagate 128 + idx w 4 + score gather 4 + dist w 4
PER_PARTICLE_TAIL = 200    # the remaining 200 of the 340 B/particle
# of which the pipelined front performs, for the previous frame: CDF read 8, resample index write 4, pose gather read 64,
# label (hint) gather read 4 - the 64-byte write of the gathered pose and the weight gather are what the fusion removes
PER_PARTICLE_FOLDED = 80
# ... of which the 64-byte pose read is the SAME read as the propagate's (the fused kernel takes a particle's pose once, through
# the resample source): the bytes the fused launch has to move per particle are 140 + 80 - 64
PER_PARTICLE_FUSED_NEEDED = PER_PARTICLE_UPDATE + PER_PARTICLE_FOLDED - 64


def cpu_quota() -> int:
    """CPUs this process may really use: the cgroup's quota (cpu.max) when there is one, else the affinity mask.  The GPU boxes show
    256 hardware threads under a 16-CPU quota: torch's default 128 intra-op threads then exhaust the quota within one scheduling
    period and the kernel parks the whole process - enqueueing thread included - for 20 - 40 ms (tools/diag_loop_stall.py: the
    "one slow frame" of the reference-named loop, once per process, right after init_filter's host-side tensor ops)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def spread(vals, scale=1.0):
    """median / min / max of repeated measurements (one sample cannot tell a regression from a hiccup)"""
    v = sorted(float(x) * scale for x in vals)
    return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "repeats": len(v)}


def algorithmic_bytes(N, K, D, B=1):
    """Bytes = K(4D+24) + B(4D+4K) + B*N*340, split per kernel group."""
    score = K * 4 * D + B * (4 * D + 4 * K)
    update = K * 24 + B * N * PER_PARTICLE_UPDATE
    tail = B * N * PER_PARTICLE_TAIL
    return {"score_codebook": score, "particle_update": update, "tail": tail, "step": score + update + tail}


def cpu_baseline(cb, traj, N, budget_s=12.0, max_steps=40):
    """The reference-shaped CPU path (oracle/ref_shaped.py) on the host cores, bounded sample."""
    from oracle.ref_shaped import RefShapedFilter

    # 32 threads is where this path peaks on the 256-thread host of the GPU box (8/16/32/64/128 threads:
    # 0.36/0.34/0.30/0.40/1.45 s per frame); more threads only add contention
    nthreads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    flt = RefShapedFilter(cb.poses, cb.embeddings, cb.mesh_vertices, workers=nthreads)
    rng = np.random.default_rng(0)
    poses = torch.as_tensor(cb.poses[rng.integers(0, cb.K, N)])
    odoms, codes = torch.as_tensor(traj.odoms), torch.as_tensor(traj.codes)
    torch.manual_seed(0)
    poses, _ = flt.step(poses, odoms[1], codes[1][None])  # warm-up (thread pools, page faults)
    t0 = time.perf_counter()
    done = 0
    while done < max_steps and (time.perf_counter() - t0) < budget_s:
        t = 2 + done % (len(odoms) - 2)
        poses, _ = flt.step(poses, odoms[t], codes[t][None])
        done += 1
    dt = time.perf_counter() - t0
    # "restructured" mode: the same frame with the data movement of this implementation - the
    # codebook scored once per frame (K x D GEMV), one score gathered per particle - on the same threads
    emb64 = torch.as_tensor(cb.embeddings).double()
    nrm = emb64.norm(dim=1).clamp_min(1e-8)
    from scipy.spatial import cKDTree
    from oracle import oracle as orc
    tree, mtree = flt.tree, flt.mesh_tree
    p = poses.numpy() if torch.is_tensor(poses) else np.asarray(poses)
    t1 = time.perf_counter()
    done2 = 0
    while done2 < max_steps and (time.perf_counter() - t1) < budget_s / 2:
        t = 2 + done2 % (len(odoms) - 2)
        tn = torch.normal(0.0, 2e-4, size=(N, 3)).numpy()
        rot = torch.normal(0.0, 0.5, size=(N, 3)).numpy()
        p1 = orc.propagate(p, traj.odoms[t], tn, rot)
        idx = tree.query(orc.R3_SE3(p1), workers=-1)[1]
        c = codes[t].double()
        scores = (emb64 @ c) / (nrm * c.norm().clamp_min(1e-8))
        w = torch.softmax(scores[torch.as_tensor(idx)], dim=0)
        w = w * torch.as_tensor(mtree.query(p1[:, :3, 3].astype(np.float64), workers=-1)[0] <= 0.002)
        p = p1[torch.multinomial(w, N, replacement=True).numpy()]
        done2 += 1
    dt2 = time.perf_counter() - t1
    return {"value": done / dt, "unit": "steps/s", "cores": int(torch.get_num_threads()), "cpu_quota": cpu_quota(), "kind": "port",
            "sample": f"{done} frames of the same workload (N={N}, K={cb.K}, D={cb.D}) through the reference-shaped "
                      f"torch-CPU path (gather (N,D) f64 + cosine + softmax + multinomial; scipy cKDTree workers=-1 "
                      f"stands in for pynanoflann n_jobs=16), {dt:.1f} s",
            "restructured_value": done2 / dt2,
            "restructured_sample": f"{done2} frames with the codebook scored once per frame and one score gathered per "
                                   f"particle (K x D GEMV instead of the (N,D) gather), same threads, {dt2:.1f} s"}


def reference_loop_rate(cb, traj, N, dev, tree, mesh_tree, T=200, floor=1000, repeats=3):
    """midastouch_amd.filter.filter - the reference's loop body with DBSCAN every 50th frame, cluster centres and annealing
    every frame (filter/filter.py:150-190), N0 = N, device draws - over T frames of the same trajectory: frames / s after the
    two initial frames (whose init_filter runs on the host like the reference's)."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import Sequence, filter as run_filter

    cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={cb.K}", f"tcn.model.output_dim={cb.D}"])
    T = min(T, traj.gt_poses.shape[0])
    seq = Sequence(torch.as_tensor(traj.gt_poses[:T]).to(dev), torch.as_tensor(traj.meas_poses[:T]).to(dev),
                   torch.as_tensor(traj.codes[:T]).to(dev), tree, cb.mesh_vertices, "004_sugar_box", mesh_tree=mesh_tree)
    # the interpreter's cyclic collector is parked for the run like around the timed region (a generation-2 pass inside one frame
    # was a 35 ms frame in a 25 ms run: 7.9k -> 3.0k frames/s in one of two otherwise identical runs)
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    runs = []
    try:
        for _ in range(repeats):
            runs.append(run_filter(cfg, seq, device=dev, floor=floor))
    finally:
        if was:
            gc.enable()
    rates = [len(st["time"][2:]) / sum(st["time"][2:]) for st in runs]
    st = runs[int(np.argsort(rates)[len(rates) // 2])]  # the run with the median rate is the one described
    steady = st["time"][2:]
    return {"frames_per_sec": len(steady) / sum(steady), "frames_per_sec_runs": spread(rates), "ms_per_frame": 1e3 * sum(steady) / len(steady), "frames": len(steady),
            "N0": N, "floor": floor, "N_final": st["num_particles"][-1], "N_min": min(st["num_particles"]),
            "ms_frame_median": 1e3 * sorted(steady)[len(steady) // 2],
            "ms_frame_max": 1e3 * max(steady), "ms_frame_max_runs": spread([max(r["time"][2:]) for r in runs], 1e3), "slowest_frame": 2 + int(np.argmax(steady)),
            "slowest_frames": [{"frame": 2 + int(i), "ms": 1e3 * steady[int(i)]} for i in np.argsort(steady)[::-1][:3]],
            "rmse_t_mm_final": 1e3 * st["rmse_t"][-1]}


def config5_rate(dev, frames=60, repeats=3):
    """BASELINE configs[4] (c5): B = 64 trajectories x N = 10 000 particles on the cotter pin's 50k x 512 codebook, pipelined batch
    engine (two launches per batch frame), device draws: us per batch frame from a spread start and from a start near the truth."""
    from midastouch_amd.engine import PipelinedBatchFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory

    cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
    B, N = 64, 10000
    trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
    od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
    co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
    out = {"workload": "c5: cotter-pin, B=64 trajectories x N=10000 particles, K=50000 x D=512, device draws, pipelined batch step"}
    eng = PipelinedBatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
    rng = np.random.default_rng(1)
    for init in ("spread", "near"):
        if init == "spread":
            start = np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)])
        else:
            start = []
            for b in range(B):
                d0 = np.linalg.norm(cb.poses[:, :3, 3] - trs[b % 8].gt_poses[0][:3, 3], axis=1)
                start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
            start = np.stack(start)
        uss = []
        for rep in range(repeats):
            eng.set_particles(torch.as_tensor(start))
            eng.project_to_codebook()
            for i in range(10):
                eng.step(od[1 + i % 38], co[1 + i % 38])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                eng.step(od[1 + (10 + i) % 38], co[1 + (10 + i) % 38])
            torch.cuda.synchronize()
            uss.append((time.perf_counter() - t0) / frames * 1e6)
        us = sorted(uss)[len(uss) // 2]
        out[init] = {"us_per_batch_frame": us, "us_per_batch_frame_runs": spread(uss), "trajectory_steps_per_sec": B * 1e6 / us}
        if init == "near":
            near_start = start
    # the batch frame with ALL K rows scored for all 64 codes on the matrix cores (k_score_mfma: the B x K x D contraction the
    # north-star names, on a side stream beside the particle update; a caller who wants every trajectory's heat-map every frame -
    # filter/live_demo.py:104-109): 3.28 GFLOP and 128 MB per batch frame on top of the particle work.  The eager batch engine
    # (three launches per batch frame: the pipelined one needs the sparse scores)
    try:
        from midastouch_amd.engine import BatchFilterEngine
        was = os.environ.get("MIDAS_DENSE_SCORES")
        os.environ["MIDAS_DENSE_SCORES"] = "1"
        try:
            engd = BatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
        finally:
            if was is None:
                del os.environ["MIDAS_DENSE_SCORES"]
            else:
                os.environ["MIDAS_DENSE_SCORES"] = was
        uss = []
        for rep in range(repeats):
            engd.set_particles(torch.as_tensor(near_start))
            engd.project_to_codebook()
            for i in range(10):
                engd.step(od[1 + i % 38], co[1 + i % 38])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                engd.step(od[1 + (10 + i) % 38], co[1 + (10 + i) % 38])
            torch.cuda.synchronize()
            uss.append((time.perf_counter() - t0) / frames * 1e6)
        us = sorted(uss)[len(uss) // 2]
        out["near_dense_mfma"] = {"us_per_batch_frame": us, "us_per_batch_frame_runs": spread(uss), "trajectory_steps_per_sec": B * 1e6 / us,
                                  "engine": "BatchFilterEngine (eager: the resampled sets are materialised every frame)",
                                  "scoring": "dense: midas_score_batch (k_score_mfma, f32 MFMA 16x16x4) over all K rows for the 64 codes every batch frame, sparse scoring off"}
    except Exception as e:  # noqa: BLE001
        out["near_dense_mfma"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _timed_run(eng, od, co, T, warm, steps, start=None, repeats=1):
    """steps frames by ONE midas_lazy_run call after `warm` frames (the form the headline is timed in), us per frame; with `start`
    the run is repeated from that particle set (projected onto the codebook) and the list of all repeats is returned"""
    uss = []
    for rep in range(repeats):
        if start is not None:
            eng.set_particles(start)
            eng.project_to_codebook()
        eng.run(od[1:1 + warm], co[1:1 + warm])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(od[1 + warm:1 + warm + steps], co[1 + warm:1 + warm + steps])
        torch.cuda.synchronize()
        uss.append((time.perf_counter() - t0) / steps * 1e6)
    return uss if start is not None else uss[0]


def config1_rates(dev, budget_s=4.0):
    """BASELINE configs[0] (c1): 004_sugar_box, N = 1000 particles, ~5k-entry codebook, D = 256 (expt=ycb: config/expt/ycb.yaml:15,18,
    config/tcn/default.yaml:19) - the reference's own CPU-runnable case.  GPU: the pipelined engine, frames by one run() call.
    CPU: the reference-shaped torch-CPU frame (oracle/ref_shaped.py: 6-d tree query, (N,D) float64 gather, cosine, softmax,
    mesh prune, torch.multinomial) on the host cores, a bounded sample - the reference's path at the reference's size."""
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    from oracle.ref_shaped import RefShapedFilter

    N, K, D = 1000, 5000, 256
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1000)
    tr = make_trajectory(cb, T=262, seed=2000)
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    start = cb.poses[np.random.default_rng(0).integers(0, K, N)]
    uss = _timed_run(eng, od, co, 262, 40, 200, start=torch.as_tensor(start), repeats=3)
    us = sorted(uss)[len(uss) // 2]
    out = {"workload": "c1: 004_sugar_box, N=1000, K=5000, D=256", "gpu": {"steps_per_sec": 1e6 / us, "us_per_step": us, "us_per_step_runs": spread(uss), "steps": 200}}
    nthreads = min(8, os.cpu_count() or 1)  # 1000 particles: more threads only add hand-over time
    was = torch.get_num_threads()
    torch.set_num_threads(nthreads)
    try:
        flt = RefShapedFilter(cb.poses, cb.embeddings, cb.mesh_vertices, workers=nthreads)
        poses = torch.as_tensor(start)
        odc, coc = torch.as_tensor(tr.odoms), torch.as_tensor(tr.codes)
        torch.manual_seed(0)
        poses, _ = flt.step(poses, odc[1], coc[1][None])
        t0, done = time.perf_counter(), 0
        while done < 2000 and time.perf_counter() - t0 < budget_s:
            t = 2 + done % 258
            poses, _ = flt.step(poses, odc[t], coc[t][None])
            done += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(was)
    out["cpu_reference_shaped"] = {"steps_per_sec": done / dt, "ms_per_step": 1e3 * dt / done, "cores": nthreads, "kind": "port",
                                   "sample": f"{done} frames, {dt:.1f} s, torch CPU ops + scipy cKDTree (stands in for pynanoflann / sklearn)"}
    out["gpu_over_cpu"] = out["gpu"]["steps_per_sec"] / out["cpu_reference_shaped"]["steps_per_sec"]
    return out


def big_config_rates(dev, which):
    """c3 / c4 at their TOTAL size on this one GPU (the 8-GPU forms shard N resp. K): N = 1 M particles x 50k x 512, and
    N = 100k x the whole 500k x 512 codebook.  us per frame, frames by one run() call; the step's algorithmic bytes."""
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory

    obj, N, K, seed = {"c3": ("035_power_drill", 1_000_000, 50_000, 1003), "c4": ("025_mug", 100_000, 500_000, 1004)}[which]
    D = 512
    t0 = time.perf_counter()
    cb = make_codebook(obj, K=K, D=D, seed=seed)
    tr = make_trajectory(cb, T=72, seed=seed + 1000)
    t1 = time.perf_counter()
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
    near = np.argsort(d0)[: K // 20]
    start = torch.as_tensor(cb.poses[np.random.default_rng(4).choice(near, N)])
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    uss = _timed_run(eng, od, co, 72, 20, 50, start=start, repeats=3)
    us = sorted(uss)[len(uss) // 2]
    ab = algorithmic_bytes(N, K, D)["step"]
    out = {"workload": f"{which} total on one GPU: {obj}, N={N}, K={K}, D={D}", "steps_per_sec": 1e6 / us, "us_per_step": us, "us_per_step_runs": spread(uss), "steps": 50,
           "scoring": "sparse: only the rows some particle's nearest entry points at are read (the codebook stream of SURVEY 8(d)'s byte model does not move: no roofline figure for this form)",
           "algorithmic_MB_per_step_survey_model": ab / 1e6, "host_codebook_s": t1 - t0, "index_build_s": t2 - t1}
    if which == "c4":
        # BASELINE's "bandwidth-bound regime": every one of the 500k rows scored every frame - what a caller with the heat-map on runs
        # (filter/filter.py:213-215, live_demo.py:107-109): the front kernel streams the whole 1.02 GB codebook beside the particle waves
        eng.sparse_scores = False
        usd = _timed_run(eng, od, co, 72, 20, 50, start=start, repeats=3)
        eng.sparse_scores = True
        ud = sorted(usd)[len(usd) // 2]
        out["dense"] = {"us_per_step": ud, "us_per_step_runs": spread(usd), "steps_per_sec": 1e6 / ud, "algorithmic_MB_per_step": ab / 1e6,
                        "achieved_GBs": ab / (ud * 1e-6) / 1e9, "step_frac_of_hbm_peak": ab / (ud * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "note": "all K = 500k rows streamed every frame (K1 GEMV inside the front kernel): SURVEY 8(d) bytes / whole-step time"}
    return out


def parity_mode_rate(cb, traj, N, dev, tree, mesh_tree, steps=100, motion=False):
    """The fixed-seed mode at c2: every frame resamples with the draws torch.multinomial would take from torch's CPU
    generator under torch.manual_seed(3000) (modules/particle_filter.py:245), generated by the device replica of that
    generator (midas_mt19937_rand64) - the mode in which resample indices are bit-exact against the reference's
    (tests/test_torch_stream.py, fixture G2b).  Motion noise stays on the device Philox streams (torch.normal's float32 path
    goes through a vectorised math library whose results are not reproducible).  One step() call per frame."""
    from midastouch_amd.engine import PipelinedFilterEngine

    eng = PipelinedFilterEngine(tree, None, mesh_tree, N, seed=4000, device=dev)
    rng = np.random.default_rng(0)
    eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, cb.K, N)]))
    eng.project_to_codebook()
    eng.seed_torch_stream(3000, motion=motion)
    odoms, codes = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    T = odoms.shape[0]
    for i in range(20):
        eng.step(odoms[1 + i % (T - 1)], codes[1 + i % (T - 1)])
    dts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.step(odoms[1 + (20 + rep * steps + i) % (T - 1)], codes[1 + (20 + rep * steps + i) % (T - 1)])
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    return {"steps_per_sec": steps / dt, "steps_per_sec_runs": spread([steps / d for d in dts]), "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "draws": "resample: device replica of torch's CPU mt19937 under torch.manual_seed(3000) (torch.rand(N, float64) stream, "
                     "generated on the generator's own stream beside each frame's kernels, a frame's 2 N words in six pieces side by side whose start "
                     "states follow from the previous frame's words by GF(2) jump polynomials: midastouch_amd/mt_jump.py); motion noise: "
                     + ("the same stream - torch.normal(0, sig, (N, 3)) twice a frame in front of the uniforms, its float32 transform as tables read "
                        "off torch.normal itself (midastouch_amd/torch_normal.py): every draw of the frame is the reference's" if motion else "device Philox"),
            "status": eng.status.cpu().numpy().tolist()}


def parity_probe(eng, N, seed):
    """How many resample indices of the last frame would differ had the softmax numerators been taken with the C library's exp
    instead of the spec exponential (hard part 2: expected 0 - 2 at N = 10^5)?  Uses the device's own nearest
    entries, scores and prune mask; the CDF (blocked order) and the search are the oracle's.  Also checks that the spec path
    reproduces the device's indices exactly.  Part of the CPU leg: the oracle is the checker here, never the thing measured."""
    from oracle import oracle as orc

    ridx = eng.ridx.cpu().numpy()  # materialises the last frame
    nn = eng.nn_idx.cpu().numpy()
    valid = eng._valid.cpu().numpy().astype(bool)
    x = eng._scores.cpu().numpy()[nn]
    step = eng._draw[2]
    u = orc.philox_uniform64(N, seed, step)
    out = {}
    for tag, e in (("spec_exp", orc.exp_spec(x, 1.0)), ("libm_exp", np.exp(x - 1.0))):
        ref, status = orc.resample_indices(e * valid, "weighted_random", u=u)
        out["index_mismatches_vs_" + tag] = int((ref != ridx).sum()) if status == 0 else None
    out["note"] = ("last frame of the run, N = %d: resample indices of the device against the oracle's search over the blocked CDF of "
                   "exp(x - 1) * mask with the spec exponential (must be 0) and with libm's exp (SURVEY 7 hard part 2)" % N)
    return out


def sharded_one_gpu(steps=20, warmup=5):
    """The C-side sharded frame (midas_shard_run on a library-owned RCCL communicator, world 1: front, tail, record all-gather,
    owner-side routing into the peer-mapped inbox, unpack folded into the next front) on this one GPU, in a process of its own
    (it needs a process group): what `bench.py --sharded` prints, reduced to the figures of the frame."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-extras",
           "--no-loop", "--no-profile", "--no-diffuse"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    d = json.loads(line[-1])
    return {"steps_per_sec": d["value"], "us_per_frame": 1e3 * d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "exchange": d["config"]["exchange"],
            "timed_region": d["config"]["timed_region"], "engine": d["config"]["engine"],
            "note": "particle-sharded engine with one shard: every phase of the multi-GPU frame runs (the all-gather and the inbox stores stay on this GPU)"}


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(gpus: int, launch_check: bool = False) -> int:
    """`python bench.py --gpus N` without a launcher: re-executes this script under torch.distributed.run (--nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1, a free port), one rank per GPU.  Returns the launcher's exit code: non-zero
    when any rank failed (torch.distributed.run tears the others down).  Only rank 0 writes to stdout."""
    import subprocess

    if not launch_check:
        if not torch.cuda.is_available():
            print("bench.py needs MI355X GPUs: no HIP device visible", file=sys.stderr)
            return 2
        if torch.cuda.device_count() < gpus:
            print("bench.py --gpus %d: only %d GPU(s) visible on this node" % (gpus, torch.cuda.device_count()), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    if env.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):  # RCCL's banner goes to stdout, behind the JSON line
        del env["NCCL_DEBUG"]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(rank: int, local_rank: int, world: int) -> int:
    """The launch path on its own: process group over RCCL (one GPU per rank) or gloo (no GPU: the CPU test), one all_reduce,
    one JSON line from rank 0."""
    import torch.distributed as dist

    gpu = torch.cuda.is_available() and torch.cuda.device_count() > local_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())
    if gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, local_rank))
    ok = int(ones.item()) == world and sorted(r for r, _ in ranks) == list(range(world))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "ok": ok, "world": int(ones.item()), "n_gpus": world, "backend": "nccl (RCCL)" if gpu else "gloo",
                          "ranks": ranks}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=100_000, help="particles per GPU")
    ap.add_argument("--codebook", type=int, default=50_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-loop", action="store_true", help="skip the reference-named loop (filter() with clustering + annealing)")
    ap.add_argument("--no-diffuse", action="store_true", help="skip the diffuse-regime figure (profiling runs: keeps the wide-start frames out of the kernel statistics)")
    ap.add_argument("--no-extras", action="store_true", help="skip config.c5, config.parity_mode and roofline.dense (profiling runs)")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, form the process group (RCCL on GPUs, gloo without) and "
                                                                "print one JSON line with the world size: tests the launch path, measures nothing")
    ap.add_argument("--sharded", action="store_true", help="use the particle-sharded engine even on one GPU (smoke test)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer_c", "peer", "a2a", "a2a_fixed", "allgather"], help="sharded engine: form of the resample exchange")
    ap.add_argument("--eager", action="store_true", help="materialise the resampled particles every frame (three launches per frame)")
    ap.add_argument("--resample", default="weighted_random", choices=["weighted_random", "low_var"],
                    help="resampler mode (particle_filter.py:230-307): the reference's default multinomial draws, or systematic")
    args = ap.parse_args()

    def guarded(fn, *a, **kw):
        """the figures beside the headline must not cost the line: a failure is reported in their place"""
        try:
            return fn(*a, **kw)
        except Exception as e:  # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        # started as a plain `python bench.py --gpus N`: launch the N ranks here (one process per GPU, the contract's
        # torch.distributed.run command line on a free port) and pass their exit code on; rank 0 prints the one JSON line
        raise SystemExit(self_launch(args.gpus, launch_check=args.launch_check))
    if world != args.gpus:
        args.gpus = world
    if args.launch_check:
        raise SystemExit(launch_check(rank, local_rank, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # host-side tensor ops (init_filter's draws, the runner's 4x4 products) on as many threads as the cgroup grants, not as the box
    # shows: see cpu_quota()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_quota())))
    dist = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints its version banner on STDOUT (NCCL_DEBUG=VERSION is exported on the GPU boxes) and libc flushes it at
        # exit, i.e. AFTER the JSON line: keep stdout to the one line the driver parses
        if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
            del os.environ["NCCL_DEBUG"]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (--sharded in one process: nobody else needs to know the port)
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # the communicator really spans `world` ranks: every rank contributes 1 (reported as config.exchange.rccl_world; a
        # world that is not the one asked for is an error, not a line)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        rccl_world = int(ones.item())
        if rccl_world != world or dist.get_world_size() != world:
            raise SystemExit("bench.py: the RCCL world has %d ranks, %d were asked for" % (rccl_world, world))

    from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory

    N, K, D = args.particles, args.codebook, args.dim
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    NPROF = 50  # frames per kernel of the per-kernel timing passes; they continue the trajectory
    T = min(max(args.warmup + 3 * args.steps + 4 * NPROF + 2, 202), 1024)
    traj = make_trajectory(cb, T=T, seed=2001)

    sharded = world > 1 or args.sharded
    tree = None
    if not sharded:
        # pipelined: the resample of frame t runs inside the front kernel of frame t+1 (two launches per frame); the
        # particle set is materialised when it is read - here once, by the flush at the end of the timed region
        from midastouch_amd.tactile_tree import tactile_tree
        tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
        tree.to_device(dev)  # one codebook index, shared by the engine and by the reference-named loop below
        cls = FilterEngine if args.eager else PipelinedFilterEngine
        eng = cls(tree, None, cb.mesh_vertices, N, seed=4000, device=dev, resample=args.resample)
    else:
        from midastouch_amd.dist import ShardedFilterEngine
        eng = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev, exchange=args.exchange,
                                  resample=args.resample)
    odoms = torch.as_tensor(traj.odoms).to(dev)
    codes = torch.as_tensor(traj.codes).to(dev)
    gts = torch.as_tensor(traj.gt_poses).to(dev)

    def frame(i):
        t = 1 + i % (T - 1)
        eng.step(odoms[t], codes[t], gt=gts[t])

    def frames(i0, n, prepared=None):
        """n consecutive frames starting at trajectory position i0; one C-ABI call where the engine has one.  prepared: the three
        input views of frames_inputs(i0, n) (taking them is not part of a step: the timed region passes them in)"""
        if n <= 0:  # (--warmup 0)
            return None
        t0_ = 1 + i0 % (T - 1)
        can_run = hasattr(eng, "run") and (not sharded or (eng.exchange == "peer_c" and eng._ccomm is not None))
        if can_run and not args.eager and t0_ + n <= T:
            od_, co_, gt_ = prepared if prepared is not None else (odoms[t0_:t0_ + n], codes[t0_:t0_ + n], gts[t0_:t0_ + n])
            return eng.run(od_, co_, gt_)
        for i in range(n):
            frame(i0 + i)
        return None

    def frames_inputs(i0, n):
        t0_ = 1 + i0 % (T - 1)
        return (odoms[t0_:t0_ + n], codes[t0_:t0_ + n], gts[t0_:t0_ + n]) if n > 0 and t0_ + n <= T else None

    # start: init_filter(gt_0, N) - sigma_t = mesh scale / 3, sigma_r = 60 deg (particle_filter.py:124-145) - projected onto
    # the codebook (filter.py:159-160); sharded runs draw every rank's slice from its own seed
    from midastouch_amd.synthetic import wide_start

    def wide_init(seed):
        eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, seed)))
        eng.project_to_codebook()

    wide_init(100 + rank)
    frames(0, 2)  # library load, allocator, first-touch: not part of any figure
    torch.cuda.synchronize()
    exchange_fallback = None
    if sharded and eng.exchange in ("peer", "peer_c"):
        # the peer-mapped forms wait (bounded) for rows / flags other ranks store into this rank's memory: if that path does not
        # deliver on this node (status bit 16 on any rank after the first frames), every rank falls back to the counted
        # all_to_all over torch.distributed rather than timing a frame that lost its exchange
        bad = torch.tensor([int(eng.status[0].item()) & 16], dtype=torch.int32, device=dev)
        if dist is not None and world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            exchange_fallback = f"{eng.exchange} -> a2a (flags of the peer-mapped exchange did not arrive)"
            eng.close()
            eng = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev, exchange="a2a", resample=args.resample)
            wide_init(100 + rank)
            frames(0, 2)
            torch.cuda.synchronize()
    diffuse = None
    if not sharded and not args.no_diffuse:  # the first frames after a wide start: hints are stale, the cloud covers the whole object
        wide_init(200 + rank)
        tele0 = eng.telemetry.cpu().numpy().copy()
        ND = 20
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(ND + 1)]
        evs[0].record()
        for i in range(ND):
            frame(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        dms = [evs[i].elapsed_time(evs[i + 1]) for i in range(ND)]
        tele1 = eng.telemetry.cpu().numpy()
        diffuse = {"frames": ND, "ms_per_step_mean": float(np.mean(dms)), "ms_first_frame": dms[0], "ms_per_step_max": float(np.max(dms)),
                   "steps_per_sec": 1e3 / float(np.mean(dms)),
                   "tree_search_fallbacks_per_frame": {"nn": float(tele1[0] - tele0[0]) / ND, "prune": float(tele1[1] - tele0[1]) / ND},
                   "note": "per-step HIP events (one step per call), frames 1..20 after init_filter(gt_0, N) + projection"}
        wide_init(100 + rank)
    # the interpreter's cyclic collector walks ~10^6 objects of the imported libraries when a generation-2 pass falls into
    # the timed region (tens of ms against a 1 - 10 ms region: seen in 3 of 12 runs); it is parked for the measurement -
    # BEFORE the warm-up frames: a collection takes a few hundred ms during which the device idles and clocks down, and the
    # first launches after such a pause were slow enough to add 0.2 ms to a 1 ms timed region
    import gc
    gc.collect()
    gc.disable()
    if lazy_warm := (not sharded and not args.eager and args.warmup >= 2 and hasattr(eng, "flush")):
        # the timed region ends with a materialisation (flush): its kernel takes part in the warm-up like every other one
        frames(0, 1)
        eng.flush()
    frames(1 if lazy_warm else 0, args.warmup - (1 if lazy_warm else 0))
    torch.cuda.synchronize()
    tele_before = None if sharded else eng.telemetry.cpu().numpy().copy()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timed_inputs = frames_inputs(args.warmup, args.steps)
    lazy_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if not sharded and not args.eager else None
    t0 = time.perf_counter()
    if lazy_ev:
        lazy_ev[0].record()
    run_log = frames(args.warmup, args.steps, timed_inputs)  # (a view of the engine's log buffer: copied below, outside the timed region)
    if lazy_ev:
        # the pipelined engine folds frame t's resample into frame t + 1's front: the K-th frame's search + gather would be left
        # for whoever reads the set.  It is materialised INSIDE the timed region, so `value` holds K complete resamples - the work
        # the reference's step does (particle_filter.py:245-249); the K frames without it: config.steps_per_sec_last_resample_pending
        lazy_ev[1].record()
        eng.flush()
    t_enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    gc.enable()
    if run_log is not None:
        run_log = run_log.clone()
    rows_timed = None
    if tele_before is not None:  # codebook rows the timed frames scored (sparse scoring): claimed by particle waves + off the list
        td = eng.telemetry.cpu().numpy()[:4].astype(np.int64) - tele_before[:4].astype(np.int64)
        rows_timed = {"by_particle_waves_per_frame": float(td[2]) / args.steps, "off_prediction_list_per_frame": float(td[3]) / args.steps}
    ms_per_step = dt / args.steps * 1e3
    lazy_rate = args.steps / (lazy_ev[0].elapsed_time(lazy_ev[1]) * 1e-3) if lazy_ev else None  # (device time of the K frames, events on the engine's stream)
    if hasattr(eng, "check"):
        eng.check()
    status = eng.status.cpu().numpy().tolist()
    # the same engine with the resampled particle set materialised (read) after every frame: three launches per frame
    eager_rate = None
    if not sharded and not args.eager:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            frame(args.warmup + args.steps + i)
            eng.flush()
        torch.cuda.synchronize()
        eager_rate = args.steps / (time.perf_counter() - t1)
    # per-step distribution inside the timed call, from the device clock each frame leaves in the run's log
    run_stats = None
    if run_log is not None and args.steps > 1:
        ts = run_log[:, 2].cpu().numpy()
        d = np.diff(ts) * 1e-3
        run_stats = {"host_enqueue_ms": 1e3 * t_enqueued, "ms_per_step_median": float(np.median(d)), "ms_per_step_p95": float(np.percentile(d, 95)), "ms_per_step_max": float(d.max()),
                     "slowest_step": int(d.argmax()) + 1, "device_span_ms": float(ts[-1] - ts[0]) * 1e-3,
                     "note": "device wall clock at the end of each frame of the timed midas_lazy_run call"}
    # per-step distribution: the same K steps again, one HIP event after each (one step per call)
    step_stats = None
    if not sharded:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            frame(args.warmup + 2 * args.steps + i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
        step_stats = {"ms_per_step_median": float(np.median(per_step)), "ms_per_step_p95": float(np.percentile(per_step, 95)),
                      "ms_per_step_min": float(per_step.min()), "ms_per_step_max": float(per_step.max())}
    # the reference-named loop (clustering + annealing every frame, device-side particle count) on the same workload
    loop_rate = None
    if not sharded and not args.no_loop:
        loop_rate = reference_loop_rate(cb, traj, N, dev, tree, eng.tree3)
        loop_rate["note"] = ("the reference's annealing floor (1000) lets the set shrink to N_final within about ten frames, so this is "
                             "mostly a small-N figure; floor_N holds all N particles (the rate a caller sees who never lets the set anneal): "
                             "bound by its DBSCAN frames (every 50th, ms_frame_max)")
        if not args.no_extras:
            loop_rate["floor_N"] = guarded(reference_loop_rate, cb, traj, N, dev, tree, eng.tree3, T=200, floor=N)
    exchange_info = None
    if sharded:
        exchange_info = {"rccl_world": rccl_world, "form": eng.exchange, "peer_mapping": "ok" if eng.exchange in ("peer", "peer_c") else (eng.peer_error or "not tried"),
                         "library_owned_rccl_communicator": bool(getattr(eng, "_ccomm", None) is not None), "fallback": exchange_fallback,
                         "status_last_frame": eng.status.cpu().numpy().tolist()}
        if eng.exchange == "a2a_fixed":  # rows beyond the overflow block's capacity would have been lost: must not happen
            ov = eng.backend.overflow_rows(eng.st, world)
            exchange_info.update(segment_rows=eng.seg_cap, overflow_capacity=eng.ovf_cap, overflow_rows_last_frame=ov,
                                 valid=bool(ov <= eng.ovf_cap))
    tele = (eng.st.telemetry if sharded else eng.telemetry).cpu().numpy().tolist()
    frames_run = 2 + (40 if diffuse else 0) + args.warmup + args.steps * (3 if eager_rate else 1)

    ab = algorithmic_bytes(N, K, D)
    out = {
        "metric": "filter_steps_per_sec", "value": world * args.steps / dt, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "c2: 004_sugar_box synthetic trajectory, N=%d particles/GPU x K=%d x D=%d codebook, "
                               "device Philox draws, %s resample" % (N, K, D, "multinomial" if args.resample == "weighted_random" else "systematic"),
                   "particles_per_gpu": N, "particles_total": N * world, "codebook_rows": K, "embedding_dim": D,
                   "parallelism": "particle-sharded x%d" % world if sharded else "single",
                   # weak scaling: `value` counts a frame of the G x N-particle filter as G steps of N particles; the frame
                   # rate of that one global filter is value / G
                   "global_filter_frames_per_sec": args.steps / dt,
                   "engine": ("sharded, exchange=" + eng.exchange) if sharded else ("eager: 3 launches/frame" if args.eager else
                                                        "pipelined: resample of frame t folded into the front kernel of frame t+1, 2 launches/frame"),
                   "arith": "f32 poses/NN, f64 scores/weights/CDF", "last_status": status,
                   "scoring": ("sparse: the particle kernels score only the codebook rows that are some particle's nearest entry "
                               "(same arithmetic, same scores); %d distinct rows in the last frame" % int(torch.unique(eng.nn_idx).numel()))
                   if getattr(eng, "sparse_scores", getattr(getattr(eng, "backend", None), "_sparse", False)) else "dense: all K rows every frame",
                   "init": "init_filter(gt_0, N) (sigma_t = mesh scale / 3, sigma_r = 60 deg) projected onto the codebook",
                   "timed_region": ("K steps by one midas_shard_run call (library-owned RCCL communicator)" if sharded and eng.exchange == "peer_c" and eng._ccomm is not None
                                    else "K steps by one midas_lazy_run call" if hasattr(eng, "run") and not sharded and not args.eager else "K step() calls"),
                   "steps_per_sec_materialised_every_frame": eager_rate,
                   "steps_per_sec_last_resample_pending": lazy_rate,
                   "reference_loop_frames_per_sec": loop_rate,
                   "per_step": step_stats, "per_step_in_timed_call": run_stats, "diffuse_regime": diffuse, "exchange": exchange_info,
                   "rows_scored_in_timed_region": rows_timed,
                   "tree_search_fallbacks_per_frame": {"nn": tele[0] / frames_run, "prune": tele[1] / frames_run}},
    }

    # per-kernel HIP-event timing (separate pass so the events do not perturb the headline)
    if not sharded and not args.no_profile:
        # one kernel bracketed at a time (two events per frame) so the others run back to back
        names = ["score_codebook", "particle_update", "tail_a", "tail_b"]
        fi = args.warmup + 3 * args.steps

        def kernel_pass(slots):
            nonlocal fi
            per_ = {}
            for slot in slots:
                eng.profile(True, only_slot=slot)
                eng.profile_read(reset=True)
                for i in range(NPROF):
                    frame(fi)
                    fi += 1
                ms, calls = eng.profile_read(reset=True)
                # an empty event pair recorded in the same frames measures the bracketing overhead; remove it
                per_[names[slot]] = max(ms[names[slot]] - ms["event_pair_overhead"], 0.0) / calls
                per_.setdefault("event_pair_overhead", ms["event_pair_overhead"] / calls)
            eng.profile(False)
            return per_

        tele_p0 = eng.telemetry.cpu().numpy()[:4].astype(np.int64)
        per = kernel_pass(range(4))
        tele_p1 = eng.telemetry.cpu().numpy()[:4].astype(np.int64)
        rows_prof = float((tele_p1[2] - tele_p0[2]) + (tele_p1[3] - tele_p0[3])) / (4 * NPROF)
        overhead = per.pop("event_pair_overhead")
        fused = per["score_codebook"] == 0.0  # the scoring shares the launch of the particle update (k_frame_front)
        sparse = bool(getattr(eng, "sparse_scores", False))
        row_bytes = 4 * D + 8 + 24  # embedding row + its norm + the entry's 6-d feature (first record of its neighbour list)
        if fused:
            per = {"frame_front": per["particle_update"], "tail_a": per["tail_a"], "tail_b": per["tail_b"]}
            if per["tail_b"] == 0.0:  # pipelined: no separate resample launch
                per.pop("tail_b")
            groups = {"frame_front": per["frame_front"], "tail": per["tail_a"] + per.get("tail_b", 0.0)}
            folded = 0 if "tail_b" in per else N * PER_PARTICLE_FOLDED
            # SURVEY 8(d)'s model charges all K rows to every frame; the kernel needs the rows some particle's nearest entry
            # points at (measured: telemetry [2] + [3]), so its own byte count is per-particle bytes + those rows
            ab["frame_front"] = ab["score_codebook"] + ab["particle_update"] + folded
            per_particle = PER_PARTICLE_FUSED_NEEDED if folded else PER_PARTICLE_UPDATE
            needed = N * per_particle + (rows_prof * row_bytes if sparse else ab["score_codebook"] + K * 24)
            dom = "frame_front"
        else:
            groups = {"score_codebook": per["score_codebook"], "particle_update": per["particle_update"],
                      "tail": per["tail_a"] + per["tail_b"]}
            dom = max(("score_codebook", "particle_update"), key=lambda k: groups[k])
            needed = ab[dom]
        achieved = needed / (groups[dom] * 1e-3) / 1e9
        survey = ab[dom] / (groups[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (PMC counters
        # cannot be read from inside the process): profiles/r04_traffic.json, tools/pmc_traffic.sh
        traffic, traffic_src = None, None
        for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", name)))
                traffic, traffic_src = tj["kernels"][dom]["hbm_bytes"], "profiles/" + name
                break
            except Exception:
                pass
        # the same kernel's mean duration in the committed rocprofv3 --kernel-trace of this command (the steady 200-step form):
        # the HIP-event figure above has an empty event pair's overhead taken off, the trace has nothing taken off
        traced = {}
        for tag, key in (("front", "frame_front_pipelined"),):
            for name in ("r06_front_trace.json", "r05_j_front_trace.json"):
                try:
                    traced = {"ms": json.load(open(os.path.join(REPO, "profiles", name)))[key]["mean_us_all_calls"] * 1e-3, "source": "profiles/" + name}
                    break
                except Exception:
                    pass
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": needed, "needed_bytes": needed,
                           "survey_8d_bytes_of_the_launch": (N * (PER_PARTICLE_UPDATE + PER_PARTICLE_FOLDED) + ab["score_codebook"] + K * 24) if fused else ab[dom],
                           "kernel_ms": groups[dom],
                           "kernel_ms_rocprof": traced.get("ms") if dom == "frame_front" else None,
                           "frac_rocprof": (needed / traced["ms"] / 1e6 / HBM_PEAK_GBS) if (dom == "frame_front" and traced) else None,
                           "kernel_ms_rocprof_source": traced.get("source") if dom == "frame_front" else None,
                           "rows_scored_per_launch": rows_prof if sparse else float(K),
                           "per_kernel_ms": per, "event_pair_overhead_ms": overhead,
                           "frac_survey_model": survey / HBM_PEAK_GBS, "survey_model_bytes_per_launch": ab[dom],
                           "step_bytes_survey_model": ab["step"],
                           "step_frac_survey_model": ab["step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "achieved = bytes this launch has to move / its HIP-event time: N x (140 update + 80 folded resample - 64: the pose is "
                                   "read once, through the resample source) + "
                                   "rows x (4 D + 32), rows = codebook rows actually scored per frame (sparse scoring: the rows some particle's "
                                   "nearest entry points at, counted by the kernels) - the kernel is bound by the dependent-fetch chain of a "
                                   "particle wave, not by bandwidth (DESIGN.md section 4).  frac_survey_model keeps SURVEY.md 8(d)'s byte model, "
                                   "which charges all K rows of the codebook to every frame although the sparse kernel does not read them; "
                                   "roofline.dense is the same frame with every row streamed (the K1 GEMV the north-star names): there SURVEY 8(d)'s "
                                   "bytes really move, and that is the figure to hold against the HBM roofline; traffic = HBM bytes per launch from the "
                                   "rocprofv3 --pmc passes of this command committed under profiles/ (counters cannot be read inside the process).  "
                                   "needed_bytes / rows_scored_per_launch describe the CONVERGED regime of this profiling pass (frames behind the timed "
                                   "window, a few hundred rows in use); the timed window itself, right after the wide start, scored "
                                   "config.rows_scored_in_timed_region rows per frame - two regimes, do not mix them"}
        if fused and sparse and not args.no_extras:
            # the dense K1 beside it: every codebook row streamed by the front kernel (horizontal fusion, MIDAS_DENSE_SCORES=1 form) -
            # what a caller with the heat-map on runs every frame (filter/filter.py:213-215); the survey model's bytes are real here
            eng.sparse_scores = False
            for i in range(10):
                frame(fi)
                fi += 1
            dper = kernel_pass([1, 2])
            # ... and its whole step, K frames by one call like the headline: the rate of a caller who wants all K scores every frame
            torch.cuda.synchronize()
            td0 = time.perf_counter()
            t0_ = 1 + fi % (T - 1)
            nd = min(args.steps, T - t0_)
            eng.run(odoms[t0_:t0_ + nd], codes[t0_:t0_ + nd], gts[t0_:t0_ + nd])
            torch.cuda.synchronize()
            out["config"]["dense_steps_per_sec"] = nd / (time.perf_counter() - td0)
            fi += nd
            eng.sparse_scores = True
            d_ms = dper["particle_update"]
            dtr = {}
            for name in ("r06_dense_front_trace.json", "r05_j_dense_front_trace.json"):
                try:
                    dtr = {"ms": json.load(open(os.path.join(REPO, "profiles", name)))["frame_front_pipelined"]["mean_us_all_calls"] * 1e-3, "source": "profiles/" + name}
                    break
                except Exception:
                    pass
            out["roofline"]["dense"] = {"kernel": "frame_front with the codebook stream (all K rows)", "kernel_ms": d_ms,
                                        "kernel_ms_rocprof": dtr.get("ms"), "kernel_ms_rocprof_source": dtr.get("source"),
                                        "frac_rocprof": (ab["frame_front"] / dtr["ms"] / 1e6 / HBM_PEAK_GBS) if dtr else None,
                                        "algorithmic_bytes_per_launch": ab["frame_front"], "achieved": ab["frame_front"] / (d_ms * 1e-3) / 1e9,
                                        "frac": ab["frame_front"] / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "tail_a_ms": dper["tail_a"],
                                        "score_stream_bytes": ab["score_codebook"]}
    if sharded:
        # no per-kernel event passes in the sharded frame: the step-level figure per GPU (each rank moves the algorithmic
        # bytes of its own N particles and of the whole replicated codebook every frame)
        achieved = ab["step"] / (ms_per_step * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "step (per GPU, sharded frame incl. the exchanges)", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": ab["step"], "kernel_ms": ms_per_step, "step_bytes": ab["step"],
                           "step_frac": achieved / HBM_PEAK_GBS}
    if not sharded and not args.no_extras:
        out["config"]["parity_mode"] = guarded(parity_mode_rate, cb, traj, N, dev, tree, eng.tree3)
        out["config"]["parity_mode_all_draws"] = guarded(parity_mode_rate, cb, traj, N, dev, tree, eng.tree3, 100, True)
        if N == 100_000 and K == 50_000 and D == 512:  # beside the headline workload only
            out["config"]["c5"] = guarded(config5_rate, dev)
            out["config"]["c1"] = guarded(config1_rates, dev)
            out["config"]["c3_single_gpu"] = guarded(big_config_rates, dev, "c3")
            out["config"]["c4_single_gpu"] = guarded(big_config_rates, dev, "c4")
            out["config"]["sharded_one_gpu"] = guarded(sharded_one_gpu)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cb, traj, N)
        if not sharded and not args.eager and args.resample == "weighted_random":
            out["cpu_baseline"]["parity_probe"] = guarded(parity_probe, eng, N, 4000)
    if rank == 0:
        try:  # anything native libraries left in the C stdio buffer goes out first: the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
