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

Single GPU: the pipelined engine - the resample + gather of frame t runs as a prologue of frame t+1's front kernel
(a per-slot dependence), so every timed step performs one resample (the previous frame's), one propagate / NN /
prune, one codebook scoring and one softmax / CDF pass: the same work per step, two launches instead of three, the
resampled poses never written to HBM.  The last frame's particle set is materialised after the timed region
(`eng.status`).  `config.steps_per_sec_materialised_every_frame` is the same engine with the particle set read
after every frame (three launches per frame, what a caller that looks at the particles each frame gets).

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
    return {"value": done / dt, "unit": "steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{done} frames of the same workload (N={N}, K={cb.K}, D={cb.D}) through the reference-shaped "
                      f"torch-CPU path (gather (N,D) f64 + cosine + softmax + multinomial; scipy cKDTree workers=-1 "
                      f"stands in for pynanoflann n_jobs=16), {dt:.1f} s"}


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
    ap.add_argument("--sharded", action="store_true", help="use the particle-sharded engine even on one GPU (smoke test)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "a2a", "allgather"], help="sharded engine: form of the resample exchange")
    ap.add_argument("--eager", action="store_true", help="materialise the resampled particles every frame (three launches per frame)")
    ap.add_argument("--resample", default="weighted_random", choices=["weighted_random", "low_var"],
                    help="resampler mode (particle_filter.py:230-307): the reference's default multinomial draws, or systematic")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints its version banner on STDOUT (NCCL_DEBUG=VERSION is exported on the GPU boxes) and libc flushes it at
        # exit, i.e. AFTER the JSON line: keep stdout to the one line the driver parses
        if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
            del os.environ["NCCL_DEBUG"]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory

    N, K, D = args.particles, args.codebook, args.dim
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    NPROF = 50  # frames per kernel of the per-kernel timing passes; they continue the trajectory
    T = min(args.warmup + 2 * args.steps + 4 * NPROF + 2, 1024)
    traj = make_trajectory(cb, T=T, seed=2001)

    sharded = world > 1 or args.sharded
    if not sharded:
        # pipelined: the resample of frame t runs inside the front kernel of frame t+1 (two launches per frame); the
        # particle set is materialised when it is read - here once, after the timed region (eng.status below)
        cls = FilterEngine if args.eager else PipelinedFilterEngine
        eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev, resample=args.resample)
    else:
        from midastouch_amd.dist import ShardedFilterEngine
        eng = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev, exchange=args.exchange,
                                  resample=args.resample)
    rng = np.random.default_rng(100 + rank)
    # particles start on codebook poses within ~2 cm of the first ground-truth pose
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
    near = np.argsort(d0)[: max(64, K // 20)]
    eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
    eng.project_to_codebook()
    odoms = torch.as_tensor(traj.odoms).to(dev)
    codes = torch.as_tensor(traj.codes).to(dev)
    gts = torch.as_tensor(traj.gt_poses).to(dev)

    def frame(i):
        t = 1 + i % (T - 1)
        eng.step(odoms[t], codes[t], gt=gts[t])

    for i in range(args.warmup):
        frame(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
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
    tele = (eng.st.telemetry if sharded else eng.telemetry).cpu().numpy().tolist()
    frames_run = args.warmup + args.steps * (2 if eager_rate else 1)

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
                   "engine": ("sharded, exchange=" + eng.exchange) if sharded else ("eager: 3 launches/frame" if args.eager else
                                                        "pipelined: resample of frame t folded into the front kernel of frame t+1, 2 launches/frame"),
                   "arith": "f32 poses/NN, f64 scores/weights/CDF", "last_status": status,
                   "steps_per_sec_materialised_every_frame": eager_rate,
                   "tree_search_fallbacks_per_frame": {"nn": tele[0] / frames_run, "prune": tele[1] / frames_run}},
    }

    # per-kernel HIP-event timing (separate pass so the events do not perturb the headline)
    if not sharded and not args.no_profile:
        # one kernel bracketed at a time (two events per frame) so the others run back to back
        names = ["score_codebook", "particle_update", "tail_a", "tail_b"]
        per, fi = {}, args.warmup + 2 * args.steps
        for slot, name in enumerate(names):
            eng.profile(True, only_slot=slot)
            eng.profile_read(reset=True)
            for i in range(NPROF):
                frame(fi)
                fi += 1
            ms, calls = eng.profile_read(reset=True)
            # an empty event pair recorded in the same frames measures the bracketing overhead; remove it
            per[name] = max(ms[name] - ms["event_pair_overhead"], 0.0) / calls
            per.setdefault("event_pair_overhead", ms["event_pair_overhead"] / calls)
        eng.profile(False)
        overhead = per.pop("event_pair_overhead")
        fused = per["score_codebook"] == 0.0  # the scoring shares the launch of the particle update (k_frame_front)
        if fused:
            per = {"frame_front": per["particle_update"], "tail_a": per["tail_a"], "tail_b": per["tail_b"]}
            if per["tail_b"] == 0.0:  # pipelined: no separate resample launch
                per.pop("tail_b")
            groups = {"frame_front": per["frame_front"], "tail": per["tail_a"] + per.get("tail_b", 0.0)}
            # algorithmic bytes of the front: scoring + particle update (+ the folded resample's share when pipelined)
            ab["frame_front"] = ab["score_codebook"] + ab["particle_update"] + (0 if "tail_b" in per else N * PER_PARTICLE_FOLDED)
            dom = "frame_front"
        else:
            groups = {"score_codebook": per["score_codebook"], "particle_update": per["particle_update"],
                      "tail": per["tail_a"] + per["tail_b"]}
            dom = max(("score_codebook", "particle_update"), key=lambda k: groups[k])
        achieved = ab[dom] / (groups[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (PMC counters
        # cannot be read from inside the process): profiles/r01_traffic.json, tools/pmc_traffic.sh
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "r01_traffic.json")))
            traffic, traffic_src = tj["kernels"][dom]["hbm_bytes"], "profiles/r01_traffic.json"
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": ab[dom], "kernel_ms": groups[dom],
                           "per_kernel_ms": per, "event_pair_overhead_ms": overhead,
                           "step_bytes": ab["step"],
                           "step_frac": ab["step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if sharded:
        # no per-kernel event passes in the sharded frame: the step-level figure per GPU (each rank moves the algorithmic
        # bytes of its own N particles and of the whole replicated codebook every frame)
        achieved = ab["step"] / (ms_per_step * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "step (per GPU, sharded frame incl. the exchanges)", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": ab["step"], "kernel_ms": ms_per_step, "step_bytes": ab["step"],
                           "step_frac": achieved / HBM_PEAK_GBS}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cb, traj, N)
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
