"""The visualiser branch of the runner (filter/filter.py:210-228): per frame the heat-map is get_similarity(tactile_code, ALL codebook
embeddings, softmax=False) and Viz.update receives (particles, cluster_poses, cluster_stds, gt_pose, heatmap_points, heatmap_weights,
image, heightmap, mask, frame) - viz/visualizer.py:329-345 - of which the visualiser KEEPS the references and reads them from its own
thread later (:346-361): what it is handed must not alias buffers the next frame rewrites.  A fake Viz records the calls.
Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


class FakeViz:
    """Signature of viz.visualizer.Viz.update; keeps what it is given (as the real one queues it) next to a copy made on arrival."""

    def __init__(self):
        self.calls = []

    def update(self, particles, cluster_poses, cluster_stds, gt_pose, heatmap_points, heatmap_weights, image, heightmap, mask, frame,
               image_savepath=None):
        kept = (particles.poses, particles.weights, particles.labels, cluster_poses, cluster_stds, gt_pose, heatmap_points, heatmap_weights)
        self.calls.append(dict(kept=kept, copies=tuple(t.detach().clone() for t in kept), n=len(particles), image=image, heightmap=heightmap,
                               mask=mask, frame=frame, cls=type(particles).__name__))


def test_filter_viz_branch(oracle):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda", 0)
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    N, T, K = 3000, 12, 2500
    cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}"])
    seq = synthetic_sequence(cfg, dev, T=T)
    viz = FakeViz()
    stats = run_filter(cfg, seq=seq, device=dev, viz=viz, floor=500)
    assert len(viz.calls) == T and [c["frame"] for c in viz.calls] == list(range(T))
    emb = seq.codebook.embeddings.cpu().numpy()
    cb_poses = seq.codebook.poses.cpu().numpy()
    codes, gt = seq.codes.cpu().numpy(), seq.gt_p.cpu().numpy()
    for i, c in enumerate(viz.calls):
        poses, weights, labels, cl_p, cl_s, gt_pose, hm_pts, hm_w = c["kept"]
        assert c["cls"] == "Particles" and c["image"] is None and c["heightmap"] is None and c["mask"] is None
        # the particle set after the frame's resampling, on the device, n = what the run's statistics record
        assert c["n"] == stats["num_particles"][i] and poses.shape == (c["n"], 4, 4) and weights.shape == (c["n"],) and labels.shape == (c["n"],)
        assert poses.dtype == torch.float32 and poses.device.type == "cuda"
        assert cl_p.shape[1:] == (4, 4) and cl_s.shape[1:] == (3,) and cl_p.shape[0] == cl_s.shape[0] >= 1
        assert np.array_equal(cl_p.cpu().numpy(), stats["cluster_poses"][i].cpu().numpy())
        assert np.array_equal(gt_pose.cpu().numpy(), gt[i])
        # heat-map: every codebook pose with its raw cosine score against the frame's code (filter.py:213-215)
        assert np.array_equal(hm_pts.cpu().numpy(), cb_poses)
        want = oracle.score_codebook(emb, codes[i])
        got = hm_w.cpu().numpy()
        assert got.shape == (K,) and got.dtype == np.float64
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
        assert np.array_equal(got, want)  # (spec summation order: exact)
    # nothing the visualiser kept was rewritten by a later frame
    for c in viz.calls:
        for kept, copy in zip(c["kept"], c["copies"]):
            assert torch.equal(kept, copy) or (torch.isnan(kept) == torch.isnan(copy)).all()
    # successive snapshots are different storage (a view of a rotating engine buffer would compare equal by accident only)
    ptrs = [c["kept"][0].data_ptr() for c in viz.calls]
    assert len(set(ptrs)) == len(ptrs)
