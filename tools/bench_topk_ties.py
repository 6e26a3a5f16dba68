import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from midastouch_amd import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for n in (10_000, 100_000, 1_000_000):
    w = rng.random(max(2, n // 30))[rng.integers(0, max(2, n // 30), n)]; w[rng.random(n) < 0.3] = 0.0
    wd = torch.from_numpy(w).to(dev)
    for mode in (1, 2):
        for k in (n // 100, n // 3):
            for ties in ("index", "aten_cpu"):
                ops.anneal_select(wd, mode, k, ties=ties); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): ops.anneal_select(wd, mode, k, ties=ties)
                torch.cuda.synchronize()
                print(f"n={n} mode={mode} k={k} ties={ties}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
