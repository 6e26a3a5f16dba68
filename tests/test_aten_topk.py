"""Pins oracle/aten_topk.c - which members of a tie `torch.topk` takes on the CPU, and in which order - against torch.topk
itself (the call of particle_filter.annealing, /root/reference/midastouch/modules/particle_filter.py:433-441).  torch is an
installed library on every box, so the comparison runs wherever the tests run."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc


def _values(rng, n, kind):
    if kind == 0:
        return rng.random(n)
    if kind == 1:
        return rng.integers(0, 3, n).astype(np.float64)
    if kind == 2:
        return rng.integers(0, max(2, n // 8), n).astype(np.float64) / 7
    if kind == 3:  # a pruned particle set: half the weights zero, some of them -0.0
        v = rng.integers(0, 5, n).astype(np.float64)
        v[rng.random(n) < 0.5] = 0.0
        v[rng.random(n) < 0.05] = -0.0
        return v
    v = rng.integers(0, 4, n).astype(np.float64)
    v[rng.random(n) < 0.1] = np.nan
    return v


@pytest.mark.parametrize("seed", range(8))
def test_restatement_equals_torch_topk_on_tie_heavy_inputs(seed):
    """1 500 inputs per seed x {largest, smallest} x {sorted, unsorted}: the index lists are identical, ties included;
    sizes straddle both thresholds (k * 64 <= n: partial sort; 3 / 16: insertion sort)."""
    torch.set_num_threads(1)
    rng = np.random.default_rng(seed)
    for _ in range(1500):
        n = int(rng.choice([1, 2, 3, 4, 5, 16, 17, 18, 33, 64, 100, 129, 1000, 4096, 5000]))
        v = _values(rng, n, int(rng.integers(0, 5)))
        k = int(rng.integers(1, n + 1)) if rng.random() < 0.5 else int(min(n, max(1, rng.integers(1, max(2, n // 3 + 1)))))
        if rng.random() < 0.3:
            k = max(1, min(n, n // 64 + int(rng.integers(-1, 2))))
        for largest in (True, False):
            for srt in (True, False):
                ref = torch.topk(torch.from_numpy(v), k, largest=largest, sorted=srt).indices.numpy()
                assert np.array_equal(ref, orc.aten_topk(v, k, largest, srt)), (n, k, largest, srt)


@pytest.mark.parametrize("n,k", [(100000, 33333), (100000, 1562), (100000, 1563), (262144, 4096), (50000, 1)])
def test_restatement_equals_torch_topk_at_filter_sizes(n, k):
    torch.set_num_threads(1)
    rng = np.random.default_rng(n + k)
    w = rng.random(3000)[rng.integers(0, 3000, n)]  # particles share codebook entries and with them their weight
    w[rng.random(n) < 0.3] = 0.0                     # pruned
    for largest in (True, False):
        ref = torch.topk(torch.from_numpy(w), k, largest=largest).indices.numpy()
        assert np.array_equal(ref, orc.aten_topk(w, k, largest, True))


@pytest.mark.parametrize("n,k,for_sort", [(200, 150, False), (5000, 2000, False), (5000, 4999, True), (70000, 30000, True)])
def test_depth_limit_fallbacks_equal_torch_topk(n, k, for_sort):
    """Inputs an adversary built against the median-of-three partition: nth_element falls back to its heap select, sort to
    its heap sort - and the result is still torch.topk's, so torch went the same way."""
    torch.set_num_threads(1)
    if for_sort:
        v = np.concatenate([orc.aten_topk_killer(k - 1, 0, True), np.full(n - k + 1, 1e9)])
    else:
        v = orc.aten_topk_killer(n, k - 1, False)
    for largest in (False, True):
        vv = -v if largest else v
        ref = torch.topk(torch.from_numpy(vv), k, largest=largest).indices.numpy()
        got, fallbacks = orc.aten_topk(vv, k, largest, True, return_fallbacks=True)
        assert fallbacks >= 1
        assert np.array_equal(ref, got)


def test_annealer_rules_differ_only_inside_ties():
    rng = np.random.default_rng(3)
    w = rng.random(40)[rng.integers(0, 40, 3000)]
    w[rng.random(3000) < 0.4] = 0.0
    for var_seq in ([1.0, 0.7, 0.9, 0.5], [1.0, 1.2, 1.1, 1.3]):
        a, b = orc.Annealer("index"), orc.Annealer("aten_cpu")
        a.step(w, 1.0), b.step(w, 1.0)
        a.init_particles = b.init_particles = 10 ** 6
        for var in var_seq[1:]:
            ka, kb = a.step(w, var, floor=100), b.step(w, var, floor=100)
            assert len(ka) == len(kb) and np.array_equal(np.sort(w[ka]), np.sort(w[kb]))
