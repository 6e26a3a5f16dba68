"""bench.py: the algorithmic-byte model of SURVEY.md 8(d) and the command line the driver uses."""
import importlib.util
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    b = _bench()
    ab = b.algorithmic_bytes(100_000, 50_000, 512)
    # c2 = 50k * 2072 + (2048 + 200k) + 100k * 340 = 137.8 MB
    assert ab["step"] == 50_000 * (4 * 512 + 24) + (4 * 512 + 4 * 50_000) + 100_000 * 340
    assert abs(ab["step"] / 1e6 - 137.8) < 0.05
    assert ab["score_codebook"] + ab["particle_update"] + ab["tail"] == ab["step"]
    c5 = b.algorithmic_bytes(10_000, 50_000, 512, B=64)
    assert abs(c5["step"] / 1e6 - 334.0) < 1.0
    assert b.PER_PARTICLE_UPDATE + b.PER_PARTICLE_TAIL == 340 and b.HBM_PEAK_GBS == 8000.0


def test_bench_refuses_to_run_without_a_gpu(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert "MI355X" in str(e.value)  # loud failure, no CPU fallback
