#!/usr/bin/env python3
"""Per-op timing + KD-tree traversal statistics on a realistic particle state (GPU box only)."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops, _lib
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100_000)
    ap.add_argument("--codebook", type=int, default=50_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--frames", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    N, K, D = a.particles, a.codebook, a.dim
    cb = make_codebook(K=K, D=D, seed=1001)
    traj = make_trajectory(cb, T=a.frames + 2, seed=2001)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    rng = np.random.default_rng(100)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
    near = np.argsort(d0)[: max(64, K // 20)]
    eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
    eng.project_to_codebook()
    odoms, codes = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    res = {}
    for t in range(1, a.frames + 1):
        eng.step(odoms[t], codes[t])
        if t in (1, 2, 3, 5, 10, a.frames):
            feat = ops.se3_feature(eng.poses_prop)
            lv, nd = ops.nn6_stats(eng.tree6, feat, None)
            hint_prev = eng.hint_next  # after the swap: the hints this frame's particle update consumed
            lvh, ndh = ops.nn6_stats(eng.tree6, feat, hint_prev)
            cert = ndh < 0
            scanned = (-ndh - 1).clamp(min=0)
            nw = N // 64
            wv_fb = (~cert)[: nw * 64].view(nw, 64).any(dim=1).float().mean()
            wv_sc = scanned[: nw * 64].view(nw, 64).max(dim=1).values.float().mean()
            _, d2 = ops.nn6(eng.tree6, feat, None, want_d2=True)
            dnn = d2.sqrt()
            rh = (feat - eng.cb_feat[hint_prev.long().clamp(min=0)]).norm(dim=1)
            qs = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
            res[f"frame{t}"] = " ".join([
                f"leaves_nohint={float(lv.float().mean()):.2f}", f"nodes_nohint={float(nd.float().mean()):.1f}",
                f"certified={float(cert.float().mean()):.4f}", f"waves_with_fallback={float(wv_fb):.3f}",
                f"scanned_mean={float(scanned[cert].float().mean()):.2f}", f"wave_max_scanned={float(wv_sc):.1f}",
                f"fb_nodes={float(ndh[~cert].float().mean()) if (~cert).any() else 0:.1f}",
                f"fb_leaves={float(lvh[~cert].float().mean()) if (~cert).any() else 0:.1f}",
                f"uniq={int(torch.unique(eng.nn_idx).numel())}",
                "dnn_mm=" + "/".join(f"{float(v)*1e3:.2f}" for v in torch.quantile(dnn, qs)),
                "rhint_mm=" + "/".join(f"{float(v)*1e3:.2f}" for v in torch.quantile(rh, qs))])
            if t == a.frames:
                feat_keep, hint_keep = feat.clone(), hint_prev.clone()
    P = eng.poses_prop.clone()
    feat = ops.se3_feature(P)
    hint = eng.nn_idx.clone()
    od, code = odoms[3], codes[3]
    res["us"] = {
        "score": timeit(lambda: eng.codebook.score(code)),
        "propagate_philox": timeit(lambda: ops.propagate(P, od, None, None, 2e-4, 0.5, 1, 2)),
        "se3_feature": timeit(lambda: ops.se3_feature(P)),
        "nn6_nohint": timeit(lambda: ops.nn6(eng.tree6, feat)),
        "nn6_hint_exact": timeit(lambda: ops.nn6(eng.tree6, feat, hint)),
        "nn6_hint_real": timeit(lambda: ops.nn6(eng.tree6, feat_keep, hint_keep)),
        "nn3_dist": timeit(lambda: ops.nn3_dist(eng.tree3, P)),
        "step": timeit(lambda: eng.step(od, code)),
    }
    x = ops.gather_f64(eng.codebook.score(code)[0], hint)
    w = ops.softmax_weights(x)
    c, _ = ops.cdf(w)
    res["us"].update({
        "softmax": timeit(lambda: ops.softmax_weights(x)),
        "cdf": timeit(lambda: ops.cdf(w)),
        "search": timeit(lambda: ops.resample_search(c, N, 0, seed=1, step=1)),
        "gather_poses": timeit(lambda: ops.gather_rows(P, hint % N)),
        "empty_torch": timeit(lambda: torch.empty(N, device=dev)),
    })
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
