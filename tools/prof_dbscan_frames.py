import os, sys, torch
sys.path.insert(0, "/root/repo")
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence
cfg = load_config(["expt.params.num_particles=100000", "expt.codebook_size=50000", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=110, D=512)
run_filter(cfg, seq, device=dev, max_frames=5)
st = run_filter(cfg, seq, device=dev, cluster=True, draws="device", floor=100000, max_frames=110)
print("slowest", sorted(((round(1e3*t,2), i) for i, t in enumerate(st["time"])), reverse=True)[:4])
