"""Input recipes shared by the CPU and GPU parity tests (pure integer / exact float64 arithmetic, so the fixtures can
store digests instead of megabytes; the digests in the fixtures guard against drift)."""
import hashlib

import numpy as np


def recipe_weights(n: int, kind: str, seed: int) -> np.ndarray:
    """The weight sets of fixture G2b (same construction as tools/gen_goldens_r2.py::recipe_weights)."""
    i = np.arange(n, dtype=np.uint64)
    h = (i + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(32)
    frac = ((h >> np.uint64(11)) & np.uint64((1 << 20) - 1)).astype(np.float64) / float(1 << 20)
    expo = (h & np.uint64(63)).astype(np.int64)
    if kind == "flat":
        w = 1.0 + frac
    elif kind == "peaky":
        w = np.ldexp(1.0 + frac, -(expo % 40))
    elif kind == "masked":
        w = np.ldexp(1.0 + frac, -(expo % 8)) * ((h >> np.uint64(40)) % np.uint64(10) < np.uint64(4))
    elif kind == "dupes":
        w = np.ldexp(1.0, -((expo % 12).astype(np.int64))) * (1.0 + (expo % 3) / 4.0)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(w, dtype=np.float64)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def g2b_cases(g):
    """-> iterator of (case id, weights, mode, torch seed, sha of the reference's indices, head, tail)."""
    n = int(g["N"])
    for ci in range(int(g["ncases"])):
        w = recipe_weights(n, str(g[f"c{ci}_kind"]), int(g[f"c{ci}_wseed"]))
        assert sha(w) == str(g[f"c{ci}_w_sha"]), "weight recipe drifted from the fixture"
        for mode in ("weighted_random", "low_var"):
            yield (ci, w, mode, int(g[f"c{ci}_{mode}_seed"]), str(g[f"c{ci}_{mode}_sha"]), g[f"c{ci}_{mode}_head"],
                   g[f"c{ci}_{mode}_tail"])
