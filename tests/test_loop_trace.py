"""The fused-step specs against traces of the reference's loop body.

* `OracleFilter.step` - the formulation every engine test compares the kernels with (exp(x - 1) shift, CDF from
  e * mask with blocked sums) - against G10 (T = 24, N = 256, the reference's own compose, teacher-forced per frame) and
  G10b (T = 64, N = 4096, free-running): NN exact, resample indices exact, weights 1e-12.
* `OracleLoop.step` - the same plus DBSCAN / cluster centres / annealing / variable particle count - against G13
  (T = 64, N0 = 4096): N per frame, kept lists, DBSCAN labels, resample indices exact.
Fixtures: tools/gen_trace_golden.py, tools/gen_loop_trace.py (the real reference functions)."""
import numpy as np
import pytest
import torch

from _recipes import sha


def _check_digest(g, key, a):
    a = np.ascontiguousarray(a)
    assert np.array_equal(a[:32], g[key + "_head"]) and np.array_equal(a[-32:], g[key + "_tail"]), key
    assert sha(a) == str(g[key + "_sha"]), key


def _close_digest(g, key, a, rtol):
    np.testing.assert_allclose(a[:32], g[key + "_head"], rtol=rtol, atol=0, err_msg=key)
    np.testing.assert_allclose(a[-32:], g[key + "_tail"], rtol=rtol, atol=0, err_msg=key)


def _draws(t, N):
    torch.manual_seed(3000 + t)
    tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3)).numpy()
    rot = torch.normal(mean=0.0, std=0.5, size=(N, 3)).numpy()
    return tn, rot


def test_oracle_filter_step_vs_g10_trace(golden, oracle):
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g10_trace")
    cb = make_codebook(K=1200, D=256, seed=1000, mesh_points=20000)
    traj = make_trajectory(cb, T=25, seed=2000)
    f = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    N, T = int(g["N"]), int(g["T"])
    poses = g["poses0"]
    for t in range(1, T + 1):
        tn, rot = _draws(t, N)
        u = torch.rand(N, dtype=torch.float64).numpy()
        r = f.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u)
        np.testing.assert_allclose(r["poses_prop"], g[f"prop_{t}"], rtol=0, atol=2e-6)
        # from the reference's own propagated poses on (teacher forcing: a float32 ulp must not cascade)
        r = f.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u, prop_override=g[f"prop_{t}"])
        assert np.array_equal(r["nn_idx"], g[f"nn_{t}"]), t
        np.testing.assert_allclose(r["weights_pre"], g[f"wsim_{t}"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(r["weights"], g[f"wprune_{t}"], rtol=1e-12, atol=0)
        assert np.array_equal(r["weights"] == 0, g[f"wprune_{t}"] == 0)
        assert r["drifted"] == bool(g[f"drifted_{t}"])
        assert np.array_equal(r["ridx"], g[f"ridx_{t}"]), f"frame {t}"
        poses = g[f"prop_{t}"][g[f"ridx_{t}"]]


def test_oracle_filter_step_vs_g10b_trace64(golden, oracle):
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g10b_trace64")
    cb = make_codebook(K=int(g["K"]), D=int(g["D"]), seed=int(g["cb_seed"]), mesh_points=20000)
    assert sha(cb.embeddings.astype(np.float32)) == str(g["cb_sha"])
    traj = make_trajectory(cb, T=int(g["T"]) + 1, seed=int(g["traj_seed"]))
    f = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    N = int(g["N0"])
    poses = g["poses0"]
    for t in range(1, int(g["T"]) + 1):
        tn, rot = _draws(t, N)
        u = torch.rand(N, dtype=torch.float64).numpy()
        r = f.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u)
        _check_digest(g, f"nn_{t}", r["nn_idx"])
        _close_digest(g, f"wsim_{t}", r["weights_pre"], 1e-12)
        _close_digest(g, f"wprune_{t}", r["weights"], 1e-12)
        _check_digest(g, f"ridx_{t}", r["ridx"])
        rt, rr = oracle.particle_rmse(r["poses_prop"], traj.gt_poses[t])
        assert rt == pytest.approx(float(g[f"rmse_{t}"][0]), rel=1e-5)
        poses = r["poses"]


def test_oracle_loop_vs_g13_trace(golden, oracle):
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g13_loop_trace")
    cb = make_codebook(K=int(g["K"]), D=int(g["D"]), seed=int(g["cb_seed"]), mesh_points=20000)
    traj = make_trajectory(cb, T=int(g["T"]) + 1, seed=int(g["traj_seed"]))
    loop = oracle.OracleLoop(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses, labels = g["poses0"], np.zeros(int(g["N0"]), dtype=np.int64)
    n_tie = 0
    for t in range(1, int(g["T"]) + 1):
        N = poses.shape[0]
        assert N == int(g[f"N_{t}"]), t
        tn, rot = _draws(t, N)
        tie = bool(g[f"tie_{t}"])
        n_tie += tie
        # on a tie frame (torch.topk's choice among equal weights: implementation-defined) follow the reference's list
        r = loop.step(poses, labels, traj.odoms[t], traj.codes[t], tn, rot, gt=traj.gt_poses[t],
                      draws=lambda n: torch.rand(n, dtype=torch.float64).numpy(),
                      keep_override=g[f"keep_{t}"] if tie else None)
        _check_digest(g, f"nn_{t}", r["nn_idx"])
        _close_digest(g, f"wprune_{t}", r["weights"], 1e-12)
        assert r["drifted"] == bool(g[f"drifted_{t}"])
        if f"dbscan_{t}_sha" in g.files:
            _check_digest(g, f"dbscan_{t}", r["labels_frame"].astype(np.int32))
        assert np.array_equal(r["cluster_labels"], g[f"cl_labels_{t}"])
        assert r["var"] == np.float32(g[f"var_{t}"])
        _check_digest(g, f"keep_{t}", r["keep"])
        assert r["N"] == int(g[f"N2_{t}"])
        _check_digest(g, f"ridx_{t}", r["ridx"])
        assert r["rmse"][0] == pytest.approx(float(g[f"rmse_{t}"][0]), rel=1e-5)
        poses, labels = r["poses"], r["labels"]
    assert n_tie < int(g["T"])  # some frames are decided without ties
