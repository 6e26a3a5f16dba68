"""topk_aten.hip: annealing's top-k with the tie choices of ATen's CPU kernel (midas_anneal_select_ties, MIDAS_TOPK_TIES_ATEN_CPU)
against torch.topk ON THE CPU itself and against the oracle's restatement (oracle/aten_topk.c) - the reference call is
`torch.topk(particles.weights, k, largest=...)`, /root/reference/midastouch/modules/particle_filter.py:433-441."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _expected(w, mode, k):
    """The annealed index list as the reference builds it from torch.topk's CPU answer (Particles.remove keeps the order,
    Particles.add appends in topk's order)."""
    n = w.shape[0]
    idx = torch.topk(torch.from_numpy(w), k, largest=(mode == 2)).indices.numpy()
    if mode == 1:
        mask = np.ones(n, dtype=bool)
        mask[idx] = False
        return np.arange(n)[mask], idx
    return np.concatenate([np.arange(n), idx]), idx


def _weights(rng, n, kind):
    if kind == "random":
        return rng.random(n)
    if kind == "shared":  # particles share codebook entries and with them their weight; a third pruned
        w = rng.random(max(2, n // 30))[rng.integers(0, max(2, n // 30), n)]
        w[rng.random(n) < 0.3] = 0.0
        return w
    if kind == "few":
        return rng.integers(0, 3, n).astype(np.float64)
    if kind == "equal":
        return np.full(n, 0.25)
    if kind == "nan":
        w = rng.integers(0, 4, n).astype(np.float64)
        w[rng.random(n) < 0.1] = np.nan
        w[rng.random(n) < 0.1] = -0.0
        return w
    raise KeyError(kind)


def _check(dev, oracle, w, mode, k, want_fallbacks=False):
    from midastouch_amd import ops
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    src = ops.anneal_select(torch.from_numpy(w).to(dev), mode, k, ties="aten_cpu", info=info).cpu().numpy()
    exp, idx = _expected(w, mode, k)
    assert np.array_equal(oracle.aten_topk(w, k, largest=(mode == 2)), idx)  # the restatement, on this box's torch
    assert np.array_equal(src, exp), (w.shape[0], mode, k)
    if want_fallbacks:
        assert int(info.item()) >= 1


@pytest.mark.parametrize("kind", ["random", "shared", "few", "equal", "nan"])
def test_small_sizes_every_k(dev, oracle, kind):
    """n = 3 .. 200, every admissible k, both modes: insertion-sort ends, 16-element leaves, the k * 64 <= n switch."""
    torch.set_num_threads(1)
    rng = np.random.default_rng(7)
    for n in (3, 4, 5, 7, 16, 17, 18, 48, 49, 51, 64, 65, 100, 129, 200):
        w = _weights(rng, n, kind)
        for k in range(1, n // 3 + 1):
            for mode in (1, 2):
                _check(dev, oracle, w, mode, k)


@pytest.mark.parametrize("kind", ["random", "shared", "few", "nan"])
@pytest.mark.parametrize("n", [1000, 4096, 20000, 100000])
def test_filter_sizes(dev, oracle, kind, n):
    torch.set_num_threads(1)
    rng = np.random.default_rng(n)
    w = _weights(rng, n, kind)
    ks = {1, 2, n // 64 - 1, n // 64, n // 64 + 1, n // 10, n // 3 - 1, n // 3, int(rng.integers(1, n // 3))}
    for k in sorted(x for x in ks if 1 <= x <= n // 3):
        for mode in (1, 2):
            _check(dev, oracle, w, mode, k)


def test_heap_beyond_lds_and_positions_beyond_lds(dev, oracle):
    """k > 3968 on the partial-sort side (heap in global memory) and segments > 7936 on the partition side."""
    torch.set_num_threads(1)
    rng = np.random.default_rng(11)
    n = 400000
    w = _weights(rng, n, "shared")
    for k in (5000, n // 64, n // 64 + 1, 20000):
        for mode in (1, 2):
            _check(dev, oracle, w, mode, k)


@pytest.mark.parametrize("n,k", [(300, 100), (3000, 1000), (15001, 5000), (90001, 30000)])
def test_depth_limit_fallbacks(dev, oracle, n, k):
    """Inputs an adversary built against the median-of-three partition (oracle.aten_topk_killer): nth_element's heap select
    and sort's heap sort run on the device too, and the answer is still torch.topk's."""
    torch.set_num_threads(1)
    v = oracle.aten_topk_killer(n, k - 1, False)
    _check(dev, oracle, v, 1, k, want_fallbacks=True)
    _check(dev, oracle, -v, 2, k, want_fallbacks=True)
    v = np.concatenate([oracle.aten_topk_killer(k - 1, 0, True), np.full(n - k + 1, 1e9)])
    _check(dev, oracle, -v, 2, k, want_fallbacks=True)


def test_index_rule_unchanged_and_class_surface(dev, oracle):
    """ties="index" is midas_anneal_select; particle_filter.annealing follows `topk_ties`."""
    from midastouch_amd import ops
    rng = np.random.default_rng(5)
    n, k = 5000, 700
    w = _weights(rng, n, "shared")
    wd = torch.from_numpy(w).to(dev)
    for mode in (1, 2):
        a = ops.anneal_select(wd, mode, k).cpu().numpy()
        b = ops.anneal_select(wd, mode, k, ties="index").cpu().numpy()
        assert np.array_equal(a, b)
        order = np.argsort(w if mode == 1 else -w, kind="stable")[:k]
        if mode == 1:
            m = np.ones(n, dtype=bool)
            m[order] = False
            assert np.array_equal(a, np.arange(n)[m])
        else:
            assert np.array_equal(a, np.concatenate([np.arange(n), order]))
