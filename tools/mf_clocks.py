#!/usr/bin/env python3
"""Phase stamps of k_score_mfma's waves (build with tools/ab_score.sh clk "-DMIDAS_MF_DBG=16"; run with MIDAS_HIP_LIB=.../clk.so): per wave the
100 MHz wall clock at kernel entry, after the staging barrier, after the multiplies of its whole group, after its epilogue, at exit."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops, _lib
dev = torch.device("cuda", 0)
K, D, B = 50_000, 512, 64
cb = ops.Codebook(torch.randn((K, D), device=dev))
codes = torch.randn((B, D), dtype=torch.float64, device=dev)
for _ in range(5): cb.score_batch(codes)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
W = int(os.environ.get("MF_WAVES", 12))
buf = (ctypes.c_longlong * (256 * W * 8))()
assert lib.midas_debug_mf_clocks(buf, 256 * W * 8) == 0
c = np.frombuffer(buf, dtype=np.int64).reshape(256, W, 8).astype(np.float64) * 0.01  # us
t0 = c[:, :, 0].min()
slot = np.arange(W) // 4
for s in range(W // 4):
    m = c[:, slot == s, :]
    print(f"slot {s}: entry {np.median(m[..., 0] - t0):6.1f} | staged {np.median(m[..., 1] - t0):6.1f} | multiplies done {np.median(m[..., 2] - t0):6.1f} "
          f"(p10 {np.percentile(m[..., 2] - t0, 10):.1f} p90 {np.percentile(m[..., 2] - t0, 90):.1f}) | epilogue done {np.median(m[..., 3] - t0):6.1f} | exit {np.median(m[..., 4] - t0):6.1f} max {np.max(m[..., 4] - t0):6.1f}")
print("all waves: codes' values there %.1f (p10 %.1f p90 %.1f) | at the staging barrier %.1f (p10 %.1f p90 %.1f, max %.1f)" % (
    np.median(c[..., 5] - t0), np.percentile(c[..., 5] - t0, 10), np.percentile(c[..., 5] - t0, 90),
    np.median(c[..., 6] - t0), np.percentile(c[..., 6] - t0, 10), np.percentile(c[..., 6] - t0, 90), (c[..., 6] - t0).max()))
print("kernel span (first entry -> last exit): %.1f us; entry spread p90 %.1f us" % (c[:, :, 4].max() - t0, np.percentile(c[:, :, 0] - t0, 90)))
