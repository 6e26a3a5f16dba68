#!/usr/bin/env python3
"""The reference-named loop at its steady state (N ~ 10^4): what the host spends per iteration against the device's frame period.
usage: tools/loop_host_time.py [N0] [floor]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence
from midastouch_amd import loop_engine
if os.environ.get("MAX_AHEAD"):  # experiment: frames the host may run ahead of the device's last report
    _init = loop_engine.LoopEngine.__init__
    def _patched(self, *a, **k):
        _init(self, *a, **k)
        self.max_ahead = int(os.environ["MAX_AHEAD"])
    loop_engine.LoopEngine.__init__ = _patched
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = load_config([f"expt.params.num_particles={N}", "expt.codebook_size=50000", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=300, D=512)
run_filter(cfg, seq, device=dev, max_frames=20)
for rep in range(3):
    st = run_filter(cfg, seq, device=dev, cluster=True, draws="device", floor=floor, max_frames=300)
    h = 1e6 * np.array(st["host_time"][60:])
    d = 1e6 * np.array(st["time"][60:])
    print("host per iteration: median %.1f us, p10 %.1f, p90 %.1f | device frame to frame: median %.1f us, mean %.1f | N %d .. %d" % (
        np.median(h), np.percentile(h, 10), np.percentile(h, 90), np.median(d), d.mean(), min(st["num_particles"][60:]), max(st["num_particles"][60:])))
