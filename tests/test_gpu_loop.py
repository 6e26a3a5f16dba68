"""The loop engine (midas_loop_step: the reference's whole loop body with clustering and annealing on a device-side
particle count), device DBSCAN and the annealing selection against the oracle and the reference's fixtures."""
import copy

import numpy as np
import pytest

from _recipes import recipe_weights, sha

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _poses_of(X, dev):
    P = torch.eye(4)[None].repeat(len(X), 1, 1).clone()
    P[:, :3, 3] = torch.as_tensor(np.asarray(X, dtype=np.float32))
    return P.to(dev)


# ---- DBSCAN ---------------------------------------------------------------------------------------------------------
def test_dbscan_matches_reference_fixture(dev, golden):
    """cluster_particles labels written by the reference (sklearn DBSCAN, min_samples = N // 5): exact."""
    from midastouch_amd import ops
    g = golden("g9_dbscan")
    for tag in ("two", "one", "noise", "three", "lattice", "eps2"):
        X, ref = g[f"{tag}_X"], g[f"{tag}_labels"]
        lab, info = ops.dbscan(_poses_of(X, dev), float(g[f"{tag}_eps"]))
        assert np.array_equal(lab.cpu().numpy(), ref), tag
        assert info.cpu().tolist() == [int(ref.max()) + 1, 0]


@pytest.mark.parametrize("case", ["blobs_small_ms", "many_clusters", "chain", "spread20k", "dense", "single", "converged24k", "two_scales",
                                  "wide_hashed", "wide_sparse_hashed", "clusters150", "clusters150_hashed"])
def test_dbscan_matches_oracle(dev, oracle, case):
    from midastouch_amd import ops
    rng = np.random.default_rng(hash(case) % 1000)
    eps, ms = 1e-2, -1
    if case == "blobs_small_ms":
        X = np.concatenate([rng.normal(c, 0.004, (400, 3)) for c in ([0, 0, 0], [0.03, 0, 0], [0, 0.04, 0.01])] +
                           [rng.uniform(-0.03, 0.07, (300, 3))])
        ms = 25
    elif case == "many_clusters":  # 40 tight clusters + noise: numbering by first core point, border assignment
        cen = rng.uniform(-0.2, 0.2, (40, 3))
        X = np.concatenate([rng.normal(c, 0.002, (60, 3)) for c in cen] + [rng.uniform(-0.25, 0.25, (400, 3))])
        X = X[rng.permutation(len(X))]
        ms = 12
    elif case == "chain":  # a long thin cluster: many cells, deep union chains
        t = rng.uniform(0, 0.5, 6000)
        X = np.stack([t, 0.01 * np.sin(40 * t), rng.normal(0, 0.001, 6000)], axis=1)
        ms = 30
    elif case == "spread20k":
        X = rng.uniform(-0.5, 0.5, (20000, 3)) * np.array([0.038, 0.089, 0.175])
    elif case == "dense":  # everything inside a couple of cells
        X = rng.normal(0, 0.0015, (5000, 3))
    elif case == "converged24k":  # a converged cloud over a few cells, each below N / 5: decided by the cells' tight boxes
        X = rng.normal(0, 0.0022, (24000, 3)) + np.array([0.0029, 0.0029, 0.0029])  # centred on a cell corner
    elif case == "two_scales":  # a tight core, a halo around eps (boxes cut by the ball: exact tests) and far noise
        X = np.concatenate([rng.normal(0, 0.001, (6000, 3)), rng.normal(0, 0.006, (6000, 3)), rng.uniform(-0.05, 0.05, (3000, 3))])
        X = X[rng.permutation(len(X))]
    elif case == "wide_hashed":  # 3 m across at eps = 1e-2: 520 cells per axis - the hash table of occupied cells, not the dense grid
        cen = rng.uniform(-1.5, 1.5, (6, 3))
        X = np.concatenate([rng.normal(c, 0.004, (1500, 3)) for c in cen] + [rng.uniform(-1.5, 1.5, (1000, 3))])
        X = X[rng.permutation(len(X))]
        ms = 200
    elif case == "wide_sparse_hashed":  # the wide init_filter start of a big object: nothing clusters, every point its own cell
        X = rng.normal(0, 0.9, (8000, 3))
    elif case in ("clusters150", "clusters150_hashed"):  # more clusters than the LDS ranking holds (62): the prefix-sum ranking
        span = 0.3 if case == "clusters150" else 2.0
        cen = rng.uniform(-span, span, (150, 3))
        X = np.concatenate([rng.normal(c, 0.0015, (40, 3)) for c in cen] + [rng.uniform(-span, span, (500, 3))])
        X = X[rng.permutation(len(X))]
        ms = 10
    else:
        X = np.zeros((1, 3))
    X = X.astype(np.float32)
    ref, ncl = oracle.dbscan(X, eps, len(X) // 5 if ms < 0 else ms)
    if case.startswith("clusters150"):
        assert ncl > 100
    if case.endswith("hashed"):
        assert np.ptp(X, axis=0).max() > 128 * 0.577 * eps
    lab, info = ops.dbscan(_poses_of(X, dev), eps, ms)
    assert info.cpu().tolist() == [ncl, 0]
    assert np.array_equal(lab.cpu().numpy(), ref), case


def test_cluster_particles_wide_cloud_stays_on_the_device(dev, oracle, recwarn):
    """particle_filter.cluster_particles on a cloud wider than the dense grid (the regime right after a wide init_filter start
    on a large object; round 3 warned and called sklearn on the host there): the reference's labels, no warning, no host path."""
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter
    rng = np.random.default_rng(5)
    X = np.concatenate([rng.normal([0.9, -0.4, 0.2], 0.003, (1200, 3)), rng.normal([-0.8, 0.5, 0.0], 0.003, (1100, 3)),
                        rng.uniform(-1.0, 1.0, (1700, 3))]).astype(np.float32)
    X = X[rng.permutation(len(X))]
    pf = particle_filter(load_config(), np.zeros((8, 3)), 1.0, downsample=1, device=dev)
    out = pf.cluster_particles(Particles(_poses_of(X, dev)))
    ref, ncl = oracle.dbscan(X, 1e-2, len(X) // 5)
    assert ncl == 2 and np.array_equal(out.labels.cpu().numpy(), ref)
    assert not [w for w in recwarn.list if "cluster_particles" in str(w.message)]


@pytest.mark.parametrize("case", ["blobs6", "chain6", "noise6", "border6", "three3", "single"])
def test_dbscan_points_matches_sklearn(dev, case):
    """midas_dbscan_points (cluster_particles(method="logmap"): six dimensions, all pairs) against sklearn's DBSCAN on the same
    float64 points - labels equal element by element (same predicate, clusters numbered by their first core point, border
    points to the smallest adjacent cluster); in three dimensions also against the grid kernel (midas_dbscan)."""
    from sklearn.cluster import DBSCAN
    from midastouch_amd import ops
    rng = np.random.default_rng(len(case) * 7 + ord(case[0]))
    eps, ms = 1e-2, -1
    if case == "blobs6":
        X = np.concatenate([rng.normal(c, 0.0025, (900, 6)) for c in (np.zeros(6), np.r_[0.05, 0, 0, 0.02, 0, 0], np.r_[0, 0.03, 0.03, 0, 0, 0.04])] +
                           [rng.uniform(-0.05, 0.1, (500, 6))])
        X = X[rng.permutation(len(X))]
    elif case == "chain6":  # a long thin cluster: the spread needs many steps (pointer jumping shortens them)
        t = rng.uniform(0, 0.6, 5000)
        X = np.stack([t, 0.01 * np.sin(30 * t), 0.01 * np.cos(30 * t)] + [rng.normal(0, 0.0005, 5000) for _ in range(3)], axis=1)
        ms = 40
    elif case == "noise6":
        X = rng.uniform(-1, 1, (2000, 6))
    elif case == "border6":  # two dense cores a little more than eps apart with a sparse bridge: border points between two clusters
        X = np.concatenate([rng.normal(0, 0.001, (800, 6)), rng.normal(0, 0.001, (800, 6)) + np.r_[0.032, 0, 0, 0, 0, 0],
                            np.stack([rng.uniform(0.003, 0.029, 80)] + [rng.normal(0, 0.0005, 80) for _ in range(5)], axis=1)])
        X = X[rng.permutation(len(X))]
        ms = 300
    elif case == "three3":
        X = np.concatenate([rng.normal(c, 0.003, (700, 3)) for c in ([0, 0, 0], [0.04, 0, 0], [0, 0.05, 0.01])] + [rng.uniform(-0.03, 0.08, (400, 3))])
        X = X.astype(np.float32).astype(np.float64)  # float32-valued: the grid kernel takes float32 translations
        ms = 120
    else:
        X = np.zeros((1, 6))
    n_min = max(len(X) // 5, 1) if ms < 0 else ms
    ref = DBSCAN(eps=eps, min_samples=n_min).fit(X).labels_
    lab, info = ops.dbscan_points(torch.as_tensor(X).to(dev), eps, n_min)
    assert np.array_equal(lab.cpu().numpy(), ref), case
    assert int(info[0].item()) == int(ref.max()) + 1 and int(info[1].item()) >= 1
    if case == "border6":
        assert (ref == -1).sum() < 80 and len(np.unique(ref[ref >= 0])) == 2   # the case holds what it is meant to
    if case == "chain6":
        assert int(info[1].item()) > 2
    if X.shape[1] == 3:
        lab3, _ = ops.dbscan(_poses_of(X.astype(np.float32), dev), eps, ms)
        assert np.array_equal(lab3.cpu().numpy(), ref)


def test_cluster_particles_logmap_on_device(dev):
    """particle_filter.cluster_particles(method="logmap") (modules/particle_filter.py:218-223) against sklearn on the same SE(3)
    logarithms: two pose clusters that differ in ROTATION only (the translations alone would be one cluster) and outliers."""
    from scipy.spatial.transform import Rotation
    from sklearn.cluster import DBSCAN
    from midastouch_amd.particle_filter import Particles, particle_filter
    from midastouch_amd.pose import se3_log
    rng = np.random.default_rng(3)
    N = 3000
    P = np.tile(np.eye(4, dtype=np.float32), (N, 1, 1))
    grp = rng.integers(0, 3, N)
    base = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.05], [0.4, -0.3, 0.2]])
    rv = base[grp] + rng.normal(0, 0.002, (N, 3))
    rv[grp == 2] = rng.uniform(-1, 1, ((grp == 2).sum(), 3))
    P[:, :3, :3] = Rotation.from_rotvec(rv).as_matrix().astype(np.float32)
    P[:, :3, 3] = rng.normal(0, 0.002, (N, 3)).astype(np.float32)
    poses = torch.as_tensor(P).to(dev)
    pf = particle_filter.__new__(particle_filter)
    parts = Particles(poses, torch.ones(N, device=dev), torch.zeros(N, dtype=torch.long, device=dev))
    out = particle_filter.cluster_particles(pf, parts, method="logmap", eps=1e-2)
    ref = DBSCAN(eps=1e-2, min_samples=N // 5).fit(se3_log(poses).cpu().numpy()).labels_
    assert np.array_equal(out.labels.cpu().numpy(), ref)
    assert len(np.unique(ref[ref >= 0])) == 2 and (ref == -1).sum() > 500
    eu = particle_filter.cluster_particles(pf, parts, method="euclidean", eps=1e-2)
    assert len(torch.unique(eu.labels)) == 1   # by translation the same particles are one cluster


# ---- annealing selection ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("small", [0, 1])
@pytest.mark.parametrize("n", [5, 100, 1024, 1025, 4096, 4097, 10000, 16384, 100000])
def test_anneal_select_matches_stable_argsort(dev, n, small, monkeypatch):
    """small = 1: the single-workgroup path of the loop engine (n <= 16384) on the same plans."""
    from midastouch_amd import ops
    if small and n > 16384:
        pytest.skip("single-workgroup path holds 16384 particles")
    monkeypatch.setenv("MIDAS_ANNEAL_SMALL", str(small))
    rng = np.random.default_rng(n)
    sets = {
        "distinct": rng.permutation(n).astype(np.float64) + 1.0,
        "dupes": recipe_weights(n, "dupes", 7),
        "masked": recipe_weights(n, "masked", 9),
        "raw": rng.uniform(-1, 1, n) * (rng.uniform(size=n) > 0.3),   # negative scores, -0.0 among the zeros
    }
    for tag, w in sets.items():
        wd = torch.as_tensor(w).to(dev)
        for k in sorted({0, 1, 2, n // 7, n // 3}):
            if k > n // 3:
                continue
            rem = np.argsort(w, kind="stable")[:k]
            keep = np.ones(n, bool)
            keep[rem] = False
            got = ops.anneal_select(wd, 1, k).cpu().numpy()
            assert np.array_equal(got, np.nonzero(keep)[0]), (tag, k, "remove")
            add = np.argsort(-w, kind="stable")[:k]
            got = ops.anneal_select(wd, 2, k).cpu().numpy()
            assert np.array_equal(got, np.concatenate([np.arange(n), add])), (tag, k, "add")


def test_annealing_api_matches_reference_golden(dev, golden):
    """particle_filter.annealing on the device selection: kept / duplicated ids of the reference's own runs (G5)."""
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter
    g = golden("g5_anneal")
    for tag in ("shrink", "grow", "floor"):
        pf = particle_filter(load_config(), np.zeros((8, 3)), 1.0, downsample=1, device=dev)
        w = torch.as_tensor(g[f"{tag}_w0"]).to(dev)
        n = w.shape[0]
        ids = torch.arange(n, dtype=torch.float32, device=dev)
        P = torch.eye(4, device=dev)[None].repeat(n, 1, 1).contiguous()
        P[:, 0, 3] = ids
        parts = Particles(P, w, ids.clone())
        for i, v in enumerate(g[f"{tag}_vars"]):
            parts = pf.annealing(parts, torch.tensor(float(v)), floor=int(g[f"{tag}_floor"]))  # float32 scalar, as the generator passed it
            assert np.array_equal(parts.labels.cpu().numpy().astype(np.int32), g[f"{tag}_ids_{i}"]), (tag, i)
            assert np.array_equal(parts.poses[:, 0, 3].cpu().numpy().astype(np.int32), g[f"{tag}_ids_{i}"])


# ---- the loop engine --------------------------------------------------------------------------------------------------
def _compare_frame(fv, ref, t, dbscan_frame):
    n = ref["poses_prop"].shape[0]
    assert fv["n"] == n, f"frame {t}: particle count"
    assert np.array_equal(fv["poses_prop"].cpu().numpy(), ref["poses_prop"]), f"frame {t}: propagated poses"
    assert np.array_equal(fv["nn_idx"].cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN"
    assert np.array_equal(fv["valid"].cpu().numpy().astype(bool), ref["mask"]), f"frame {t}: prune mask"
    np.testing.assert_allclose(fv["weights"].cpu().numpy(), ref["weights"], rtol=1e-12, atol=0, err_msg=f"frame {t}")
    assert fv["drifted"] == ref["drifted"]
    if "var" in ref:
        if dbscan_frame:
            assert np.array_equal(fv["labels_frame"].cpu().numpy(), ref["labels_frame"]), f"frame {t}: DBSCAN labels"
        assert fv["clusters"] == len(ref["cluster_labels"]), f"frame {t}: clusters present"
        assert np.float32(fv["var"]) == ref["var"], f"frame {t}: mean cluster spread {fv['var']} vs {ref['var']}"
        np.testing.assert_allclose(fv["cluster_poses"], ref["cluster_poses"][:8], atol=2e-6)
        np.testing.assert_allclose(fv["cluster_stds"], ref["cluster_stds"][:8], rtol=1e-5, atol=1e-9)
    assert fv["n_after"] == ref["N"], f"frame {t}: annealed size {fv['n_after']} vs {ref['N']}"
    assert np.array_equal(fv["src"].cpu().numpy(), ref["keep"]), f"frame {t}: annealed set"
    assert fv["status"] == ref["status"]
    assert np.array_equal(fv["ridx"].cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
    assert np.array_equal(fv["poses"].cpu().numpy(), ref["poses"]), f"frame {t}: resampled poses"
    np.testing.assert_allclose(fv["weights_res"].cpu().numpy(), ref["weights_res"], rtol=1e-12)
    assert np.array_equal(fv["hint"].cpu().numpy(), ref["nn_idx_res"])
    assert np.array_equal(fv["labels"].cpu().numpy(), ref["labels"]), f"frame {t}: labels carried"
    if "rmse" in ref:
        assert fv["rmse_t"] == pytest.approx(ref["rmse"][0], rel=1e-9)
        assert fv["rmse_r"] == pytest.approx(ref["rmse"][1], rel=1e-6, abs=1e-6)


@pytest.mark.parametrize("N0,mode,cluster", [(6000, "weighted_random", True), (9000, "low_var", True), (3000, "weighted_random", False)])
def test_loop_engine_free_running_vs_oracle(dev, oracle, N0, mode, cluster):
    """Device Philox draws, 40 frames, DBSCAN every 5th: N, annealed sets, labels, resample indices and poses identical to
    the oracle's loop body frame after frame (nothing is teacher-forced)."""
    from midastouch_amd.loop_engine import LoopEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale
    K, D, T, seed = 3000, 256, 40, 4100
    cb = make_codebook(K=K, D=D, seed=1013, mesh_points=20000)
    traj = make_trajectory(cb, T=T + 1, seed=2013)
    g = torch.Generator().manual_seed(11)
    sc = mesh_scale(cb.extents)
    tn0 = torch.normal(0.0, sc / 3.0 * 0.15, size=(N0, 3), generator=g).numpy()
    rot0 = torch.normal(0.0, 60.0 * 0.15, size=(N0, 3), generator=g).numpy()
    poses = oracle.init_filter_compose(traj.gt_poses[0], tn0, rot0)
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices, cluster=cluster, cluster_every=5)
    poses = cb.poses[loop.f.SE3_NN_idx(poses)]
    eng = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, seed=seed, resample=mode, cluster=cluster, cluster_every=5, device=dev)
    eng.set_particles(torch.as_tensor(poses))
    labels = np.zeros(N0, dtype=np.int64)
    sizes = []
    for t in range(T):
        n = poses.shape[0]
        tn, rot = oracle.philox_noise(n, seed, t, np.float32(2e-4), np.float32(0.5))
        u32 = oracle.philox_uniform32(seed, t) if mode == "low_var" else None
        ref = loop.step(poses, labels, traj.odoms[t + 1], traj.codes[t + 1], tn, rot, gt=traj.gt_poses[t + 1], mode=mode, u32=u32,
                        draws=(lambda n2: oracle.philox_uniform64(n2, seed, t)) if mode == "weighted_random" else None)
        eng.step(torch.as_tensor(traj.odoms[t + 1]), torch.as_tensor(traj.codes[t + 1]), gt=torch.as_tensor(traj.gt_poses[t + 1]))
        _compare_frame(eng.frame_view(), ref, t, cluster and t % 5 == 0)
        poses, labels = ref["poses"], ref["labels"]
        sizes.append(ref["N"])
    if cluster:
        assert min(sizes) < N0 and any(b > a for a, b in zip(sizes, sizes[1:])), sizes  # removed and duplicated
    log = eng.read_log()
    assert [r["n_after"] for r in log] == sizes and eng.n == sizes[-1]


@pytest.mark.parametrize("N0", [6000, 40000])
def test_loop_engine_frozen_annealing_runs_the_decision_only(dev, oracle, N0):
    """floor == live count == init_particles: the annealing rule (particle_filter.py:421-446) can neither remove nor duplicate, the
    engine says so (midas_loop_args.anneal_frozen) and the ANNEAL phase runs its decision without the selection's launches - frame
    after frame the oracle's loop body (which applies the rule), and identical to an engine that keeps the launches.  A wrong statement
    is found by the device and read_log() raises."""
    from midastouch_amd._lib import MidasError
    from midastouch_amd.loop_engine import LoopEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    K, D, T, seed = 3000, 256, 14, 4300
    cb = make_codebook(K=K, D=D, seed=1014, mesh_points=20000)
    traj = make_trajectory(cb, T=T + 1, seed=2014)
    rng = np.random.default_rng(3)
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices, cluster=True, cluster_every=5, floor=N0)
    poses = cb.poses[rng.integers(0, K, N0)]
    engs = []
    for allow in (True, False):
        e = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, seed=seed, floor=N0, cluster_every=5, device=dev)
        e.allow_frozen = allow
        e.set_particles(torch.as_tensor(poses))
        assert e._frozen() is allow
        engs.append(e)
    labels = np.zeros(N0, dtype=np.int64)
    for t in range(T):
        tn, rot = oracle.philox_noise(N0, seed, t, np.float32(2e-4), np.float32(0.5))
        ref = loop.step(poses, labels, traj.odoms[t + 1], traj.codes[t + 1], tn, rot, gt=traj.gt_poses[t + 1], mode="weighted_random",
                        draws=lambda n2: oracle.philox_uniform64(n2, seed, t))
        assert ref["N"] == N0
        for e in engs:
            e.step(torch.as_tensor(traj.odoms[t + 1]), torch.as_tensor(traj.codes[t + 1]), gt=torch.as_tensor(traj.gt_poses[t + 1]))
            _compare_frame(e.frame_view(), ref, t, t % 5 == 0)
        poses, labels = ref["poses"], ref["labels"]
    logs = [e.read_log() for e in engs]
    for ra, rb in zip(*logs):
        assert ra["n_after"] == rb["n_after"] == N0 and ra["mode"] == rb["mode"] == 0 and ra["err"] == rb["err"] == 0
        assert np.array_equal(ra["cluster_poses"], rb["cluster_poses"]) and np.array_equal(ra["cluster_stds"], rb["cluster_stds"])
    # a wrong statement: the floor below the live count lets the rule remove particles; the device notices
    bad = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, seed=seed, floor=N0 // 2, cluster_every=5, device=dev)
    bad.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N0)]))
    assert not bad._frozen()
    bad._frozen = lambda: True
    for t in range(8):
        bad.step(torch.as_tensor(traj.odoms[t + 1]), torch.as_tensor(traj.codes[t + 1]))
    with pytest.raises(MidasError, match="annealing could not act"):
        bad.read_log()


def test_loop_engine_replays_reference_loop_trace(dev, golden, oracle):
    """G13 (the reference's loop body with DBSCAN + annealing, N0 = 4096, 64 frames) with the reference's host draws in
    its order - tn, rot, then, once the annealed size is known, the resampler's uniforms (the frame is split there) - and
    `topk_ties="aten_cpu"`: annealing's torch.topk decides 60 of the 64 frames inside a tie, and the device keeps the
    particles the reference kept - the kept-set and resample-index digests the REFERENCE wrote hold in ALL 64 frames, the run
    carries on from its own state through the oracle's loop under the same rule (no teacher forcing of the choice)."""
    from midastouch_amd import _lib
    from midastouch_amd.loop_engine import LoopEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g13_loop_trace")
    cb = make_codebook(K=int(g["K"]), D=int(g["D"]), seed=int(g["cb_seed"]), mesh_points=20000)
    T, N0 = int(g["T"]), int(g["N0"])
    traj = make_trajectory(cb, T=T + 1, seed=int(g["traj_seed"]))
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices, ties="aten_cpu")
    eng = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, device=dev, topk_ties="aten_cpu")
    poses, labels = g["poses0"], np.zeros(N0, dtype=np.int64)
    tie_frames = 0
    for t in range(1, T + 1):
        n = poses.shape[0]
        eng.set_particles(torch.as_tensor(poses), torch.as_tensor(labels), reset_annealing=False)
        eng.set_annealing_state(float(loop.annealer.particle_var), loop.annealer.init_particles or 0)
        eng.step_count = t - 1
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(n, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(n, 3))
        dbs = (t - 1) % 50 == 0
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]), gt=torch.as_tensor(traj.gt_poses[t]), tn=tn, rot=rot,
                 dbscan=dbs, phases=_lib.LOOP_FRONT | _lib.LOOP_DBSCAN | _lib.LOOP_ANNEAL)
        n2 = int(eng.ctl_i[_lib.LOOP_I_NSET].item())
        u = torch.rand(n2, dtype=torch.float64)
        eng.step(None, None, u=u, phases=_lib.LOOP_RESAMPLE)
        ref = loop.step(poses, labels, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), gt=traj.gt_poses[t], u=u.numpy())
        assert ref["N"] == n2 == int(g[f"N2_{t}"])
        fv = eng.frame_view()
        _compare_frame(fv, ref, t, dbs)
        # the reference's own digests, tie or not
        assert sha(fv["src"].cpu().numpy().astype(np.int32)) == str(g[f"keep_{t}_sha"]), f"frame {t}: kept set is not the reference's"
        assert sha(fv["ridx"].cpu().numpy().astype(np.int32)) == str(g[f"ridx_{t}_sha"]), f"frame {t}: resample indices are not the reference's"
        if bool(g[f"tie_{t}"]):
            assert np.array_equal(fv["src"].cpu().numpy(), g[f"keep_{t}"])
            tie_frames += 1
        poses, labels = ref["poses"], ref["labels"]
    assert tie_frames == 60  # the fixture's count: the index rule would have left the reference's set in every one of them


def test_loop_engine_index_rule_on_reference_loop_trace(dev, golden, oracle):
    """The default rule (ties by index: torch's CUDA kernel, the radix select) on G13's first tie frames: identical to the
    oracle under that rule, and the kept set differs from the reference's CPU choice only inside the tie."""
    from midastouch_amd import _lib
    from midastouch_amd.loop_engine import LoopEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g13_loop_trace")
    cb = make_codebook(K=int(g["K"]), D=int(g["D"]), seed=int(g["cb_seed"]), mesh_points=20000)
    N0 = int(g["N0"])
    traj = make_trajectory(cb, T=int(g["T"]) + 1, seed=int(g["traj_seed"]))
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, device=dev)
    poses, labels = g["poses0"], np.zeros(N0, dtype=np.int64)
    seen = 0
    for t in range(1, 9):
        n = poses.shape[0]
        eng.set_particles(torch.as_tensor(poses), torch.as_tensor(labels), reset_annealing=False)
        eng.set_annealing_state(float(loop.annealer.particle_var), loop.annealer.init_particles or 0)
        eng.step_count = t - 1
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(n, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(n, 3))
        dbs = (t - 1) % 50 == 0
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]), gt=torch.as_tensor(traj.gt_poses[t]), tn=tn, rot=rot,
                 dbscan=dbs, phases=_lib.LOOP_FRONT | _lib.LOOP_DBSCAN | _lib.LOOP_ANNEAL)
        n2 = int(eng.ctl_i[_lib.LOOP_I_NSET].item())
        u = torch.rand(n2, dtype=torch.float64)
        eng.step(None, None, u=u, phases=_lib.LOOP_RESAMPLE)
        tie = bool(g[f"tie_{t}"])
        by_index = copy.deepcopy(loop)
        ref_idx = by_index.step(poses, labels, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), gt=traj.gt_poses[t], u=u.numpy())
        fv = eng.frame_view()
        _compare_frame(fv, ref_idx, t, dbs)
        if tie:
            w = fv["weights"].cpu().numpy()
            mine, theirs = fv["src"].cpu().numpy(), g[f"keep_{t}"]
            assert not np.array_equal(mine, theirs) and np.array_equal(np.sort(w[mine]), np.sort(w[theirs]))
            seen += 1
        # carry on from the reference's own choice
        ref = loop.step(poses, labels, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), gt=traj.gt_poses[t], u=u.numpy(),
                        keep_override=g[f"keep_{t}"] if tie else None)
        poses, labels = ref["poses"], ref["labels"]
    assert seen >= 3


def test_filter_runner_host_draws_matches_oracle_loop(dev, oracle):
    """filter(draws="host"): the runner's orchestration (initial frames, the re-initialisation quirk of filter.py:152, odometry,
    DBSCAN cadence, draw order on the torch CPU generator) against the oracle's loop body fed with the same draws."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    from midastouch_amd.particle_filter import particle_filter
    N, T = 1500, 14
    cfg = load_config([f"expt.params.num_particles={N}", "expt.codebook_size=2500"])
    seq = synthetic_sequence(cfg, dev, T=T)
    torch.manual_seed(5)
    stats = run_filter(cfg, seq=seq, device=dev, draws="host", floor=300)
    # the same frames on the CPU
    cb_poses, emb = seq.codebook.poses.cpu().numpy(), seq.codebook.embeddings.cpu().numpy()
    gt, meas, codes = seq.gt_p.cpu().numpy(), seq.meas_p.cpu().numpy(), seq.codes.cpu().numpy()
    pf = particle_filter(cfg, seq.mesh_vertices, cfg.expt.params.noise_ratio, downsample=1, device=dev)
    loop = oracle.OracleLoop(cb_poses, emb, seq.mesh_vertices, floor=300, ties="aten_cpu")  # draws="host" replays the CPU run
    inv = torch.linalg.inv(seq.meas_p).cpu().numpy()
    torch.manual_seed(5)
    prev, poses, labels = 0, None, None
    for idx in range(T):
        if prev > 0:
            n = poses.shape[0]
            odom = (torch.as_tensor(inv[prev]).to(dev) @ seq.meas_p[idx]).cpu().numpy()
            tn = torch.normal(mean=0.0, std=pf.motion_noise["sig_t"], size=(n, 3)).numpy()
            rot = torch.normal(mean=0.0, std=pf.motion_noise["sig_r"], size=(n, 3)).numpy()
        else:
            p0 = pf.init_filter(seq.gt_p[idx], N).poses.cpu().numpy()
            poses, labels = cb_poses[loop.f.SE3_NN_idx(p0)], np.zeros(N, dtype=np.int64)
            odom, tn, rot = np.eye(4, dtype=np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
        r = loop.step(poses, labels, odom, codes[idx], tn, rot, gt=gt[idx], draws=lambda n2: torch.rand(n2, dtype=torch.float64).numpy())
        assert stats["num_particles"][idx] == r["N"], idx
        assert stats["rmse_t"][idx] == pytest.approx(r["rmse"][0], rel=1e-9), idx
        assert len(stats["cluster_stds"][idx]) == len(r["cluster_labels"])
        poses, labels, prev = r["poses"], r["labels"], idx
    assert stats["frames"][-1]["n_after"] == poses.shape[0]


def test_filter_runner_seeded_draws_equal_host_draws(dev):
    """filter(draws="seeded"): torch's CPU generator continued on the device - mt19937 words, torch.normal's float32 transform,
    torch.rand float64, handed over to the host for init_filter and back - gives the run of draws="host" number for number: the
    same particle counts, rmse and cluster spreads in every frame, and torch's generator stands at the same place afterwards."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    N, T = 6000, 30
    cfg = load_config([f"expt.params.num_particles={N}", "expt.codebook_size=2500"])
    seq = synthetic_sequence(cfg, dev, T=T)
    out = {}
    for mode in ("host", "seeded"):
        torch.manual_seed(77)
        st = run_filter(cfg, seq=seq, device=dev, draws=mode, floor=500)
        out[mode] = (st, torch.rand(8, dtype=torch.float64))
    a, b = out["host"][0], out["seeded"][0]
    assert a["num_particles"] == b["num_particles"] and min(a["num_particles"]) < N
    assert a["rmse_t"] == b["rmse_t"] and a["rmse_r"] == b["rmse_r"]
    for x, y in zip(a["cluster_stds"], b["cluster_stds"]):
        assert torch.equal(x, y)
    assert [f["kept"] for f in a["frames"]] == [f["kept"] for f in b["frames"]]
    assert torch.equal(out["host"][1], out["seeded"][1]), "torch's generator does not stand where the host-draw run leaves it"


def test_filter_runner_device_draws_tracks_and_anneals(dev):
    """filter() with its default device draws: frames are enqueued back to back, the log is read once at the end."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    cfg = load_config(["expt.params.num_particles=20000", "expt.codebook_size=5000"])
    seq = synthetic_sequence(cfg, dev, T=80)
    stats = run_filter(cfg, seq=seq, device=dev)
    assert len(stats["rmse_t"]) == 80 and np.isfinite(stats["rmse_t"]).all()
    assert stats["rmse_t"][-1] < 0.02
    assert min(stats["num_particles"]) >= 1000 and min(stats["num_particles"]) < 20000  # annealed
    assert all(f["err"] == 0 for f in stats["frames"])
    assert stats["frames"][0]["mode"] == 0 and any(f["mode"] == 1 for f in stats["frames"])


@pytest.mark.parametrize("N0", [1, 7, 63, 300, 1025])
def test_loop_engine_tiny_particle_sets(dev, oracle, N0):
    """Edge sizes: one particle, less than a chunk, less than a wave, a few hundred (floor above the count: annealing can only
    add), just over a tile of the single-workgroup annealing - every frame against the oracle's loop body."""
    from midastouch_amd.loop_engine import LoopEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    K, D, T, seed = 1500, 128, 12, 77
    cb = make_codebook(K=K, D=D, seed=1021, mesh_points=8000)
    traj = make_trajectory(cb, T=T + 1, seed=2021)
    rng = np.random.default_rng(N0)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
    poses = cb.poses[rng.choice(np.argsort(d0)[:200], N0)]
    floor = max(N0 // 2, 1)
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices, floor=floor, cluster_every=3)
    eng = LoopEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N0, seed=seed, floor=floor, cluster_every=3, device=dev)
    eng.set_particles(torch.as_tensor(poses))
    labels = np.zeros(N0, dtype=np.int64)
    for t in range(T):
        n = poses.shape[0]
        tn, rot = oracle.philox_noise(n, seed, t, np.float32(2e-4), np.float32(0.5))
        ref = loop.step(poses, labels, traj.odoms[t + 1], traj.codes[t + 1], tn, rot, gt=traj.gt_poses[t + 1],
                        draws=lambda n2: oracle.philox_uniform64(n2, seed, t))
        eng.step(torch.as_tensor(traj.odoms[t + 1]), torch.as_tensor(traj.codes[t + 1]), gt=torch.as_tensor(traj.gt_poses[t + 1]))
        _compare_frame(eng.frame_view(), ref, t, t % 3 == 0)
        poses, labels = ref["poses"], ref["labels"]
    assert int(eng.ctl_i[14].item()) == 0  # no limit / bound error flagged
