"""Guide tables of the folded resample's search (midas_lazy_args.guide_dev, include/midas_hip.h): per summation block, bins over
the block's masked total name the piece of the per-slot prefix table a draw's search starts from.  The tables are a hint - the
indices are decided by the exact comparison on (BP + lp_i) / total either way (torch.multinomial's lower bound,
modules/particle_filter.py:245; the systematic sampler's upper bound, :251-262) - so the tests are (1) the table a frame's tail
writes equals its definition, computed here from the frame's own prefix tables, (2) engines with and without it return the same
indices, particles and statistics bit for bit (softmax / raw scores, both resamplers, ragged and single-block sets, a cloud that
loses most of its particles to the prune), next to the oracle comparisons of test_gpu_pipelined.py / test_gpu_fullsize.py,
which run with the tables on.  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _layout():
    from midastouch_amd import _lib
    lib = _lib.load()
    b, u, s = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.midas_lazy_guide_layout(ctypes.byref(b), ctypes.byref(u), ctypes.byref(s)) == 0
    return b.value, u.value, s.value


def _setup(N, K, D, seed, T=20):
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1000 + seed)
    return cb, make_trajectory(cb, T=T, seed=2000 + seed)


def _tables(eng):
    """The frame's tables as numpy views (layout of midas_lazy_args.tables_dev)."""
    N = eng.N
    Np, ng, nb = -(-N // 16) * 16, -(-N // 16), -(-N // 4096)
    ngp = -(-ng // 16) * 16
    t = eng._tables.cpu().numpy()
    o = 4 * Np + 2 * ngp + 32 * nb
    return dict(lp=t[2 * Np:3 * Np], lp_raw=t[3 * Np:4 * Np], x_raw=t[Np:2 * Np], btot=t[o + nb:o + 2 * nb], btot_raw=t[o + 2 * nb:o + 3 * nb],
                nb=nb, valid=eng._valid.cpu().numpy())


def _expected_guide(lp, W, N, blk, bins, unit):
    lo, hi = blk * 4096, min(N, blk * 4096 + 4096)
    nch = -(-(hi - lo) // 16)
    nunits = nch * (16 // unit)
    # unit ends as the tail has them: the chunk's values past N repeat the last prefix value (masked slots add +0.0)
    v = np.full(nch * 16, lp[hi - 1])
    v[: hi - lo] = lp[lo:hi]
    ends = v[unit - 1::unit]
    q = W * (1.0 / bins)
    edges = np.arange(bins, dtype=np.float64) * q
    cnt = (ends[None, :] < edges[:, None]).sum(axis=1)  # (every unit counts, whether or not the ends rise: raw scores may be negative)
    return np.append(np.minimum(cnt, nunits - 1), nunits - 1)


@pytest.mark.parametrize("N,softmax,sig_t", [(9000, True, 2e-4), (100_000, True, 2e-4), (5000, False, 2e-4), (20_000, True, 3e-3)])
def test_guide_table_matches_its_definition(dev, N, softmax, sig_t):
    from midastouch_amd.engine import PipelinedFilterEngine
    bins, unit, stride = _layout()
    K, D = 3000, 128
    cb, traj = _setup(N, K, D, 21)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=77, softmax=softmax, sig_t=sig_t, device=dev)
    assert eng._guide is not None and eng._guide.numel() == 2 * (-(-N // 4096)) * stride * 2
    eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(3).integers(0, K, N)]))
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    saw_negative = False
    for t in range(1, 7):
        eng.step(od[t], co[t])
        torch.cuda.synchronize()
        tb = _tables(eng)
        g = eng._guide.cpu().numpy().view(np.uint16).reshape(2, tb["nb"], stride)
        kept = int(eng._st[eng._cur].cpu().numpy()[1])
        if sig_t > 1e-3:
            assert kept < N  # the wide motion noise loses particles to the prune: units of zero width
        var, lp, W = (0, tb["lp"], tb["btot"]) if softmax else (1, tb["lp_raw"], tb["btot_raw"])
        for blk in range(tb["nb"]):
            if not (W[blk] > 0 and np.isfinite(W[blk])):
                continue  # (such a block's table is not read)
            want = _expected_guide(lp, W[blk], N, blk, bins, unit)
            if not softmax:  # raw scores: a block with a negative weight has no guide (its prefix values do not rise)
                w = (tb["x_raw"][:N] * tb["valid"])[blk * 4096:(blk + 1) * 4096]
                if (w < 0).any():
                    want = np.full(bins + 1, 0xFFFF)
                    saw_negative = True
            assert np.array_equal(g[var, blk, : bins + 1], want), f"frame {t} block {blk}"
    assert not (softmax and saw_negative)


@pytest.mark.parametrize("N,K,softmax,mode,sig_t", [
    (9000, 3000, True, "weighted_random", 2e-4),    # three blocks, the last one ragged
    (9000, 3000, True, "low_var", 2e-4),
    (1000, 2000, True, "weighted_random", 2e-4),    # one partial block
    (5000, 3000, False, "weighted_random", 2e-4),   # raw scores as weights
    (20_000, 3000, True, "weighted_random", 3e-3),  # most particles pruned: bins that span many units fall back to the table lines
    (100_000, 5000, True, "weighted_random", 2e-4),
    (300_000, 5000, True, "weighted_random", 2e-4),  # more than 64 blocks: the fronts with workgroup-level resample tables
    (300_000, 5000, True, "low_var", 2e-4),
])
def test_same_results_with_and_without_the_guide(dev, monkeypatch, N, K, softmax, mode, sig_t):
    from midastouch_amd.engine import PipelinedFilterEngine
    cb, traj = _setup(N, K, 128, 22)
    start = torch.as_tensor(cb.poses[np.random.default_rng(5).integers(0, K, N)])
    od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIDAS_GUIDE", flag)
        eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=91, softmax=softmax, resample=mode, sig_t=sig_t,
                                    device=dev)
        assert (eng._guide is not None) == (flag == "1")
        eng.set_particles(start)
        rec = []
        for t in range(1, 9):
            eng.step(od[t], co[t], gt=gt[t])
            if t > 1:
                rec.append(eng._ridx.cpu().numpy().copy())  # the folded resample's indices of the frame before
        log = eng.run(od[9:15], co[9:15], gt[9:15])      # one call, six frames (midas_lazy_run)
        rec += [eng.ridx.cpu().numpy().copy(), eng.poses.cpu().numpy().copy(), eng.weights.cpu().numpy().copy(), log.cpu().numpy()[:, :2].copy(),
                eng.status.cpu().numpy().copy()]
        out.append(rec)
    for a, b in zip(*out):
        assert np.array_equal(a, b)
