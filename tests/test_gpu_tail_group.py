"""The step tail with one wave per 256-slot group (csrc/tail_group.hpp) against the form with one workgroup per 4096-slot block
(csrc/tail_block.hpp), which the oracle comparisons pinned: the same additions in the same order, so every table - softmax
numerators, block-local prefix sums, chunk / group / block records, extrema, status, the guide tables - must agree bit for bit
(get_similarity + the CDF of the resampler, modules/particle_filter.py:449-469, :237-252).  Shapes: single chunk .. 74 blocks,
ragged ends, pruned runs, a block whose particles share one score (the isclose guard's raw variant), raw scores of mixed sign,
NaN.  MIDAS_TAIL_GROUPED=0 selects the old form.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _tables_len(N):
    ng, nb = -(-N // 16), -(-N // 4096)
    return 4 * (-(-N // 16) * 16) + 2 * (-(-ng // 16) * 16) + 32 * nb


def _run(dev, monkeypatch, grouped, N, scores, nn, valid, softmax):
    from midastouch_amd import _lib
    monkeypatch.setenv("MIDAS_TAIL_GROUPED", "1" if grouped else "0")
    ctx = _lib.context(dev)
    nb = -(-N // 4096)
    tables = torch.zeros(_tables_len(N), dtype=torch.float64, device=dev)
    r1 = torch.zeros(5 * nb + 4, dtype=torch.float64, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    ctx.call("midas_shard_tail_a", N, _lib._ptr(scores), _lib._ptr(nn), _lib._ptr(valid), int(softmax), _lib._ptr(tables), _lib._ptr(r1),
             _lib._ptr(status))
    torch.cuda.synchronize()
    return tables.cpu().numpy(), r1.cpu().numpy(), status.cpu().numpy()


def _views(t, N, softmax, close_blocks):
    """Parts of the table block that are defined: padding past N and the variant a block did not write are not."""
    Np, ng, nb = -(-N // 16) * 16, -(-N // 16), -(-N // 4096)
    ngp = -(-ng // 16) * 16
    e, x, lp, lpr = (t[i * Np:i * Np + N] for i in range(4))
    o = 4 * Np
    ge, ger = t[o:o + ng], t[o + ngp:o + ngp + ng]
    o += 2 * ngp
    gg, ggr = t[o:o + 16 * nb], t[o + 16 * nb:o + 32 * nb]
    out = {}
    raw_mask = np.zeros(N, bool)
    for b in close_blocks if softmax else range(nb):
        raw_mask[b * 4096:(b + 1) * 4096] = True
    if softmax:
        out.update(e=e, lp=lp, gend=ge, ggend=gg)
    out.update(x_raw=x[raw_mask], lp_raw=lpr[raw_mask], gend_raw=ger[raw_mask[::16][:ng]])
    out["ggend_raw"] = ggr.reshape(nb, 16)[sorted(close_blocks) if softmax else list(range(nb))]
    return out


CASES = [
    # N, softmax, kind
    (16, True, "plain"), (17, True, "plain"), (255, True, "plain"), (256, False, "plain"), (257, True, "plain"), (1000, True, "pruned"),
    (4095, True, "plain"), (4096, True, "plain"), (4097, False, "plain"), (9000, True, "pruned"), (9000, True, "close"),
    (9000, False, "mixed"), (12289, True, "nan"), (20_000, False, "nan"), (100_000, True, "plain"), (100_000, True, "pruned"),
    (100_000, False, "mixed"), (300_000, True, "close"), (8192, True, "allpruned"), (5000, False, "allpruned"),
]


@pytest.mark.parametrize("N,softmax,kind", CASES)
def test_grouped_tail_equals_block_tail(dev, monkeypatch, N, softmax, kind):
    rng = np.random.default_rng(N + 7 * softmax + len(kind))
    K = 3000
    sc = rng.uniform(0.2, 1.0, K)
    if kind == "mixed":
        sc = rng.uniform(-1.0, 1.0, K)
    nn = rng.integers(0, K, N).astype(np.int32)
    valid = np.ones(N, np.uint8)
    close_blocks = set()
    nb = -(-N // 4096)
    if kind == "pruned":  # runs of pruned particles, whole groups and chunks among them
        valid = (rng.uniform(size=N) < 0.3).astype(np.uint8)
        valid[300:1700] = 0
        valid[N // 2:N // 2 + 40] = 0
    if kind == "allpruned":
        valid[:] = 0
        if N > 5000:
            valid[N - 5000:N - 4990] = 1
    if kind == "close":  # every particle of blocks 1 and nb - 1 shares one entry: their own range is within the tolerance
        for b in {1, nb - 1}:
            nn[b * 4096:(b + 1) * 4096] = 17
            close_blocks.add(b)
        valid = (rng.uniform(size=N) < 0.8).astype(np.uint8)
    if kind == "nan":
        sc[5] = np.nan
        nn[min(N - 1, 4500)] = 5
    scores, nn_d, valid_d = (torch.as_tensor(a).to(dev) for a in (sc, nn, valid))
    new = _run(dev, monkeypatch, True, N, scores, nn_d, valid_d, softmax)
    old = _run(dev, monkeypatch, False, N, scores, nn_d, valid_d, softmax)
    assert np.array_equal(new[2], old[2]), (new[2], old[2])
    assert np.array_equal(new[1].view(np.uint64), old[1].view(np.uint64))  # block records + NaN / kept counters
    va, vb = _views(new[0], N, softmax, close_blocks), _views(old[0], N, softmax, close_blocks)
    for k in va:
        assert np.array_equal(va[k].view(np.uint64), vb[k].view(np.uint64)), k
    assert int(new[2][1]) == int(valid.sum())


@pytest.mark.parametrize("N,softmax,mode,sig_t", [(9000, True, "weighted_random", 2e-4), (1000, True, "low_var", 2e-4), (5000, False, "weighted_random", 2e-4),
                                                  (20_000, True, "weighted_random", 3e-3), (100_000, True, "weighted_random", 2e-4)])
def test_engine_tables_and_guide_identical_in_both_forms(dev, monkeypatch, N, softmax, mode, sig_t):
    """The pipelined engine (front with the folded resample + tail) frame by frame: tables, guide tables, indices, log."""
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    K = 3000
    cb = make_codebook("004_sugar_box", K=K, D=128, seed=1031)
    traj = make_trajectory(cb, T=16, seed=2031)
    start = torch.as_tensor(cb.poses[np.random.default_rng(5).integers(0, K, N)])
    od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIDAS_TAIL_GROUPED", flag)
        eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=91, softmax=softmax, resample=mode, sig_t=sig_t, device=dev)
        eng.set_particles(start)
        rec = []
        for t in range(1, 7):
            eng.step(od[t], co[t], gt=gt[t])
            torch.cuda.synchronize()
            Np, nb = -(-N // 16) * 16, -(-N // 4096)
            tb = eng._tables.cpu().numpy()
            # per-slot tables up to N (the variant in use), block records, guide tables of the variant in use
            var = 0 if softmax else 1
            rec.append(tb[(2 + var) * Np:(2 + var) * Np + N].copy())
            g = eng._guide.cpu().numpy().view(np.uint16).reshape(2, nb, -1)
            rec.append(g[var].copy())
            rec.append(eng._st[eng._cur].cpu().numpy().copy())
            if t > 1:
                rec.append(eng._ridx.cpu().numpy().copy())
        log = eng.run(od[7:13], co[7:13], gt[7:13])
        rec += [eng.ridx.cpu().numpy().copy(), eng.poses.cpu().numpy().copy(), eng.weights.cpu().numpy().copy(), log.cpu().numpy()[:, :2].copy(),
                eng.status.cpu().numpy().copy()]
        out.append(rec)
    for i, (a, b) in enumerate(zip(*out)):
        assert np.array_equal(a, b, equal_nan=True), i
