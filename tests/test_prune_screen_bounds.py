"""The float32 screen of the prune (csrc/particles.hip mesh_screen_check / screen_thresholds): a restatement of its
thresholds in numpy float32, checked against the exact float64 predicate of the reference's prune
(/root/reference/midastouch/modules/particle_filter.py:332-346: distance to the nearest mesh vertex > threshold) on random and
on adversarial (within rounding of the threshold) pairs.  The claims under test: a "sure hit" is a hit, a "sure miss" is a
miss, and the ambiguous band is narrow.  Runs without a GPU; the kernel itself is compared with the oracle in
tests/test_gpu_step.py::test_prune_screen_threshold_boundary."""
import numpy as np
import pytest

f32 = np.float32


def _thresholds(tqf, thr):
    thr_up = np.nextafter(f32(thr), f32(np.inf)) if float(f32(thr)) < thr else f32(thr)
    thr_dn = np.nextafter(f32(thr), f32(-np.inf)) if float(f32(thr)) > thr else f32(thr)
    E = f32(2.5e-7) * (np.abs(tqf[:, 0]) + np.abs(tqf[:, 1]) + np.abs(tqf[:, 2]))
    lo, hi = thr_dn - E, thr_up + E
    t2lo = np.where(lo > 0, lo * lo * (f32(1.0) - f32(4e-6)), f32(-1.0)).astype(f32)
    t2hi = (hi * hi * (f32(1.0) + f32(4e-6))).astype(f32)
    return t2lo, t2hi


def _d2f(tqf, vf):
    d = tqf - vf
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(f32)


@pytest.mark.parametrize("thr", [2e-3, 1e-4, 5e-2])
@pytest.mark.parametrize("scale", [0.05, 0.3, 3.0])
def test_sure_hits_and_misses_are_exact(thr, scale):
    rng = np.random.default_rng(int(thr * 1e6) + int(scale * 100))
    n = 400_000
    tq = (rng.uniform(-scale, scale, (n, 3))).astype(f32)          # a pose's translation: a float32 value
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    # distances: a third spread widely, two thirds within 1e-4 .. 1e-9 (relative) of the threshold
    rel = np.concatenate([rng.uniform(-0.9, 3.0, n // 3), rng.normal(0, 1, n - n // 3) * 10.0 ** rng.uniform(-9, -4, n - n // 3)])
    v = tq.astype(np.float64) + u * (thr * (1.0 + rel))[:, None]    # mesh vertices: float64
    vf = v.astype(f32)                                              # the screening copy: rounded to nearest
    t2lo, t2hi = _thresholds(tq, thr)
    d2f = _d2f(tq, vf)
    d = tq.astype(np.float64) - v
    exact_hit = np.sqrt((d * d).sum(1)) <= thr                      # the oracle's predicate (nn3_dist > pen_max is a miss)
    sure_hit, sure_miss = d2f <= t2lo, d2f >= t2hi
    assert not (sure_hit & sure_miss).any()
    assert exact_hit[sure_hit].all(), "a sure hit that is not a hit"
    assert (~exact_hit[sure_miss]).all(), "a sure miss that is a hit"
    amb = ~(sure_hit | sure_miss)
    # the band: |d / thr - 1| below ~ (E + rounding) / thr; nothing far from the threshold is ambiguous
    far = np.abs(rel) > 1e-5 + 4.0 * 2.5e-7 * 3 * scale / thr
    assert not (amb & far).any()
    assert (sure_hit | sure_miss)[np.abs(rel) > 0.01].all() or scale / thr > 1e3


def test_nan_and_infinite_are_never_sure():
    tq = np.array([[np.nan, 0, 0], [0.1, 0.2, 0.3]], dtype=f32)
    vf = np.array([[0, 0, 0], [np.inf, np.inf, np.inf]], dtype=f32)   # a NaN pose; the padding record of a short list
    t2lo, t2hi = _thresholds(tq, 2e-3)
    with np.errstate(invalid="ignore"):
        d2f = _d2f(tq, vf)
        assert not (d2f[0] <= t2lo[0]) and not (d2f[0] >= t2hi[0])  # NaN: ambiguous -> the float64 path decides
        assert d2f[1] >= t2hi[1]                                     # padding: a sure miss
