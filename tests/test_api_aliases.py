"""The north-star aliases `update_weights`, `resample`, `step` (midastouch_amd/filter.py; SURVEY.md 8(b): the thin names
BASELINE.json's north_star uses for filter/filter.py:170-173, :190 and the loop body :150-190) against the reference's golden
vectors G1 / G2 and against the op-by-op calls they delegate to.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _pf(dev):
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import particle_filter
    return particle_filter(load_config(), np.zeros((8, 3)), 1.0, downsample=1, device=dev)


def test_update_weights_matches_reference_golden(dev, golden, oracle):
    """update_weights == SE3_NN + get_similarity (filter/filter.py:170-173).  G1 holds the reference's get_similarity of a code
    against gathered rows C[idx]: a codebook whose K poses are distinct and particles sitting exactly on pose idx[n] make
    the nearest-pose lookup return idx, so the alias must reproduce the reference's weights."""
    from midastouch_amd.filter import update_weights
    from midastouch_amd.particle_filter import Particles
    from midastouch_amd.tactile_tree import tactile_tree
    pf = _pf(dev)
    g = golden("g1_similarity")
    for tag in ("a", "b"):
        C, idx, q = g[f"{tag}_C"], g[f"{tag}_idx"], g[f"{tag}_q"]
        K = C.shape[0]
        poses = torch.eye(4)[None].repeat(K, 1, 1).clone()
        poses[:, 0, 3] = torch.arange(K, dtype=torch.float32) * 1e-2  # distinct entries 1 cm apart
        tree = tactile_tree(poses, poses, torch.as_tensor(C).double())
        tree.to_device(dev)
        parts = Particles(poses[torch.as_tensor(idx).long()].to(dev))
        qt = torch.as_tensor(q).double()[None].to(dev)
        out = update_weights(pf, tree, parts, qt, softmax=True)
        assert out is parts and out.weights.dtype == torch.float64
        np.testing.assert_allclose(out.weights.cpu().numpy(), g[f"{tag}_w_softmax"], rtol=1e-12)
        out = update_weights(pf, tree, parts, qt, softmax=False)
        np.testing.assert_allclose(out.weights.cpu().numpy(), g[f"{tag}_w_raw"], atol=1e-14)
        # and the spec arithmetic, exactly
        assert np.array_equal(update_weights(pf, tree, parts, qt).weights.cpu().numpy(),
                              oracle.get_similarity(q, C[idx], softmax=True))


def test_resample_alias_matches_reference_golden(dev, golden):
    """resample == particle_filter.resampler (filter/filter.py:190) on G2: the reference's indices under its seeds."""
    from midastouch_amd.filter import resample
    from midastouch_amd.particle_filter import Particles
    pf = _pf(dev)
    g = golden("g2_resampler")
    for tag in ("soft4096", "peaky2048", "masked3000", "n1"):
        n = len(g[f"{tag}_w"])
        poses = torch.eye(4)[None].repeat(n, 1, 1).clone()
        poses[:, 0, 3] = torch.arange(n, dtype=torch.float32)
        for mode in ("weighted_random", "low_var"):
            parts = Particles(poses.to(dev), torch.as_tensor(g[f"{tag}_w"]).to(dev), torch.arange(n, dtype=torch.float32).to(dev))
            torch.manual_seed(int(g[f"{tag}_{mode}_seed"]))
            out = resample(pf, parts, mode)
            assert np.array_equal(out.poses[:, 0, 3].cpu().numpy().astype(np.int64), g[f"{tag}_{mode}_idx"]), (tag, mode)
    # default mode = the reference's default ("weighted_random")
    parts = Particles(poses.to(dev), torch.as_tensor(g["n1_w"]).to(dev))
    assert len(resample(pf, parts)) == 1


def test_step_alias_replays_reference_trace(dev, golden):
    """step == one fused frame on a FilterEngine (the loop body filter/filter.py:150-190 at fixed N): the first frames of the
    64-frame trace G10b, written by the reference's own functions under its seeds, replayed through the alias - NN digests,
    weights and resample indices."""
    from _recipes import sha
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.filter import step
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    g = golden("g10b_trace64")
    N, K, D, T = int(g["N0"]), int(g["K"]), int(g["D"]), int(g["T"])
    cb = make_codebook(K=K, D=D, seed=int(g["cb_seed"]), mesh_points=20000)
    assert sha(cb.embeddings.astype(np.float32)) == str(g["cb_sha"])
    traj = make_trajectory(cb, T=T + 1, seed=int(g["traj_seed"]))
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    eng.set_particles(torch.as_tensor(g["poses0"]))
    for t in range(1, 11):
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3))
        u = torch.rand(N, dtype=torch.float64)
        out = step(eng, torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]), torch.as_tensor(traj.gt_poses[t]), tn=tn, rot=rot, u=u)
        assert out is eng
        assert sha(eng.nn_idx.cpu().numpy().astype(np.int32)) == str(g[f"nn_{t}_sha"]), f"frame {t}: NN"
        w = eng.weights.cpu().numpy()
        np.testing.assert_allclose(w[:32], g[f"wprune_{t}_head"], rtol=1e-12, atol=0)
        assert sha(eng.ridx.cpu().numpy().astype(np.int32)) == str(g[f"ridx_{t}_sha"]), f"frame {t}: resample indices"
