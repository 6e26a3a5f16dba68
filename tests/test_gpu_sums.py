"""The float64 sums over a wave (xor butterfly 32 .. 1) and over a 16-lane row (8 .. 1) are part of the arithmetic spec (the oracle
restates their order: oracle/midas_oracle.c mo_quarter_tree, the blocked scan).  The kernels form them with register moves
(v_permlane32_swap / v_permlane16_swap and row rotations, midas_math.hpp) instead of `__shfl_xor` trips through the LDS crossbar:
this test compares both forms bit for bit on the device (midas_selftest_wave_sums), over magnitudes, signs, subnormals, infinities, NaN."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_register_move_sums_equal_the_shuffle_butterflies():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd import _lib
    ctx = _lib.context(torch.device("cuda", 0))
    rng = np.random.default_rng(0)
    for trial in range(300):
        kind = trial % 5
        if kind == 0:
            x = rng.standard_normal(64)
        elif kind == 1:
            x = rng.standard_normal(64) * 10.0 ** rng.integers(-200, 200, 64)
        elif kind == 2:
            x = rng.uniform(0, 1, 64) * 1e-310  # subnormals
        elif kind == 3:
            x = np.abs(rng.standard_normal(64)) * 1e-3  # the weights' range
        else:
            x = rng.standard_normal(64)
            x[rng.integers(0, 64, 3)] = [np.inf, -0.0, np.nan]
        xin = torch.as_tensor(x).cuda()
        out = torch.zeros(256, dtype=torch.float64, device="cuda")
        ctx.call("midas_selftest_wave_sums", _lib._ptr(xin), _lib._ptr(out))
        o = out.cpu().numpy().reshape(64, 4)
        a, b, c, d = (np.ascontiguousarray(o[:, i]).view(np.uint64) for i in range(4))
        assert np.all((a == b) | (np.isnan(o[:, 0]) & np.isnan(o[:, 1]))), trial
        assert np.all((c == d) | (np.isnan(o[:, 2]) & np.isnan(o[:, 3]))), trial
        if kind != 4:  # and the value is the butterfly's: every lane the same wave sum, every row lane the same row sum
            assert len(set(a.tolist())) == 1 and all(len(set(c[r * 16:(r + 1) * 16].tolist())) == 1 for r in range(4))
