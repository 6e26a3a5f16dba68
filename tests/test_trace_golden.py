"""G10: the reference's own loop body over 24 frames (tools/gen_trace_golden.py drove the real reference
functions), replayed stage by stage with teacher forcing: each stage starts from the reference's own
input for that stage, so a float32 ulp in one stage cannot cascade.  Oracle on CPU; the HIP path on GPU."""
import numpy as np
import pytest
import torch


def _data():
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    cb = make_codebook(K=1200, D=256, seed=1000, mesh_points=20000)
    traj = make_trajectory(cb, T=25, seed=2000)
    return cb, traj


def test_trace_oracle(golden, oracle):
    g = golden("g10_trace")
    cb, traj = _data()
    cb_feat = oracle.R3_SE3(cb.poses)
    N, T = int(g["N"]), int(g["T"])
    poses = g["poses0"]
    for t in range(1, T + 1):
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3)).numpy()
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3)).numpy()
        u = torch.rand(N, dtype=torch.float64).numpy()
        prop = oracle.propagate(poses, traj.odoms[t], tn, rot)
        np.testing.assert_allclose(prop, g[f"prop_{t}"], rtol=0, atol=2e-6)
        ref_prop = g[f"prop_{t}"]                       # teacher forcing from here on
        nn = oracle.nn6(oracle.R3_SE3(ref_prop), cb_feat)[0]
        assert np.array_equal(nn, g[f"nn_{t}"])
        scores = oracle.score_codebook(cb.embeddings, traj.codes[t])
        w, _ = oracle.softmax_weights(scores[nn], True)
        np.testing.assert_allclose(w, g[f"wsim_{t}"], rtol=1e-12)
        dist = oracle.nn3_dist(ref_prop, cb.mesh_vertices)
        wp = g[f"wsim_{t}"] * ~(dist > 0.002)
        assert np.array_equal(wp, g[f"wprune_{t}"])
        ridx, status = oracle.resample_indices(g[f"wprune_{t}"], "weighted_random", u=u)
        if status:  # all particles pruned: the reference returns its input (identity)
            ridx = np.arange(N, dtype=np.int32)
        assert np.array_equal(ridx, g[f"ridx_{t}"]), f"frame {t}"
        rt, rr = oracle.particle_rmse(ref_prop, traj.gt_poses[t])
        assert rt == pytest.approx(float(g[f"rmse_{t}"][0]), rel=1e-5)
        assert rr == pytest.approx(float(g[f"rmse_{t}"][1]), rel=1e-4, abs=0.03)
        poses = ref_prop[g[f"ridx_{t}"]]


@pytest.mark.gpu
def test_trace_hip_api(golden):
    """The reference-named API on the GPU replays the reference's trace: same seeds, same calls."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter, particle_rmse
    from midastouch_amd.tactile_tree import tactile_tree
    dev = torch.device("cuda", 0)
    g = golden("g10_trace")
    cb, traj = _data()
    pf = particle_filter(load_config(), cb.mesh_vertices, 1.0, downsample=1, device=dev)
    tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings).double())
    tree.to_device(dev)
    N, T = int(g["N"]), int(g["T"])
    poses = torch.as_tensor(g["poses0"]).to(dev)
    for t in range(1, T + 1):
        torch.manual_seed(3000 + t)
        parts = pf.motionModel(Particles(poses), torch.as_tensor(traj.odoms[t]).to(dev), multiplier=1.0)
        np.testing.assert_allclose(parts.poses.cpu().numpy(), g[f"prop_{t}"], rtol=0, atol=2e-6)
        parts = Particles(torch.as_tensor(g[f"prop_{t}"]).to(dev))          # teacher forcing
        rt, rr = particle_rmse(parts, torch.as_tensor(traj.gt_poses[t]).to(dev))
        assert float(rt) == pytest.approx(float(g[f"rmse_{t}"][0]), rel=1e-5)
        _, _, codes = tree.SE3_NN(parts.poses)
        assert np.array_equal(codes.idx.cpu().numpy(), g[f"nn_{t}"])
        parts.weights = pf.get_similarity(torch.as_tensor(traj.codes[t])[None].to(dev), codes, softmax=True)
        np.testing.assert_allclose(parts.weights.cpu().numpy(), g[f"wsim_{t}"], rtol=1e-12)
        assert np.max(np.abs(parts.weights.cpu().numpy() - g[f"wsim_{t}"])) < 1e-5
        parts.weights = torch.as_tensor(g[f"wsim_{t}"]).to(dev)
        parts, drifted = pf.remove_invalid_particles(parts)
        assert np.array_equal(parts.weights.cpu().numpy(), g[f"wprune_{t}"]) and bool(drifted) == bool(g[f"drifted_{t}"])
        parts.labels = torch.arange(N, dtype=torch.float32, device=dev)
        res = pf.resampler(parts)                                          # draws torch.rand(N, float64) after the normals
        assert np.array_equal(res.labels.cpu().numpy().astype(np.int32), g[f"ridx_{t}"]), f"frame {t}"
        poses = res.poses
