#!/usr/bin/env python3
"""Phase clocks of k_loop_anneal_small (profiling build: tools/variants.sh ack "-DMIDAS_ANNEAL_CLOCKS", MIDAS_HIP_LIB=..., MIDAS_ANNEAL_CLOCKS=1)."""
import os, sys
os.environ["MIDAS_ANNEAL_CLOCKS"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import loop_engine
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence
engines = []
_init = loop_engine.LoopEngine.__init__
def init(self, *a, **k):
    _init(self, *a, **k); engines.append(self)
loop_engine.LoopEngine.__init__ = init
cfg = load_config(["expt.params.num_particles=100000", "expt.codebook_size=50000", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=150, D=512)
run_filter(cfg, seq, device=dev, cluster=True, draws="device", floor=1000, max_frames=150)
c = engines[-1].ctl_d.cpu().numpy()[56:104]
n2 = max(c[6], 1.0)
print("launches with mode 2:", int(c[6]), "mean k", c[8] / n2)
print("cumulative ticks (100 MHz -> us = /100) summed over ALL launches: after decide %.0f, keys %.0f, select %.0f, compaction %.0f | mode-2 only: sort %.0f, write %.0f | rotations block %.0f" % tuple(c[i] / 100 for i in (0, 1, 2, 3, 4, 5, 7)))
nl = c[0] / 146.0  # launches, from the decide phase's ~1.46 us
print("per launch (us): minmax end %.1f | per pass: zero %.2f atomics %.2f scan %.2f pick %.2f ; passes per launch %.2f ; finish %.2f" % (
    c[9] / 100 / nl, c[10] / 100 / max(c[15], 1), c[11] / 100 / max(c[15], 1), c[12] / 100 / max(c[15], 1), c[13] / 100 / max(c[15], 1), c[15] / nl, c[14] / 100 / nl))
# k_loop_resample, workgroup 0 / thread 0, absolute from the kernel's first instruction: prefetch + live count there [16], tables built [17],
# search done [18], rows gathered and stored [19], end [20]; launches [22]
nr = max(c[22], 1.0)
print("k_loop_resample per launch (us): count known %.2f tables %.2f search %.2f rows %.2f end %.2f (launches %d)" % tuple(
    [c[i] / 100 / nr for i in (16, 17, 18, 19, 20)] + [nr]))
# k_loop_weights_moments, workgroup 0 / thread 0: everything it reads there [24], S and the guard [25], weight stored [26], moments of
# every cluster [27], control block + rmse [28]; launches [29]
nw = max(c[29], 1.0)
print("k_loop_weights_moments per launch (us): loads %.2f head %.2f weight %.2f moments %.2f finalise %.2f (launches %d)" % tuple(
    [c[i] / 100 / nw for i in (24, 25, 26, 27, 28)] + [nw]))
# k_loop_anneal_small, thread 0, absolute: decision [0], keys [1], own extrema [30], wave extrema [31], barrier [32], sixteen waves' extrema [33],
# equal count [34], ... [9] as above; selection [2], compaction [3]; launches [35]
na = max(c[35], 1.0)
print("k_loop_anneal_small per launch (us): decide %.2f keys %.2f own %.2f wave %.2f barrier %.2f all %.2f eq %.2f extrema-phase-end %.2f select %.2f compaction %.2f (launches %d; returns before the keys are not in the later stamps)" % tuple(
    [c[i] / 100 / na for i in (0, 1, 30, 31, 32, 33, 34, 9, 2, 3)] + [na]))
# k_loop_xe, workgroup 0 / thread 0: live count + indices there [36], scores there [37], exponentials + stores [38], block total [39], end [40]; launches [41]
nx = max(c[41], 1.0)
print("k_loop_xe per launch (us): indices %.2f scores %.2f exps %.2f total %.2f end %.2f (launches %d)" % tuple([c[i] / 100 / nx for i in (36, 37, 38, 39, 40)] + [nx]))
