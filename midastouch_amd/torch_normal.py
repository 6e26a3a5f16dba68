"""`torch.normal` of float32 values on the CPU, as tables (host-side set-up of the device replica, csrc/mt19937.hip k_mt_normal).

The reference draws its motion noise with `torch.normal(mean, std, size=(N, 3))` on torch's CPU generator
(/root/reference/midastouch/modules/particle_filter.py:326-335).  ATen fills the output with float32 uniforms (one 32-bit
generator output each, `(w & 0xFFFFFF) * 2^-24`) and transforms them sixteen at a time (`normal_fill_16`, Box-Muller):

    u1 = 1 - data[j], u2 = data[j + 8]            j = 0 .. 7
    radius = sqrt(-2 log(u1)), theta = 2 pi u2
    data[j] = (radius cos(theta)) std + mean,  data[j + 8] = (radius sin(theta)) std + mean

with `log`, `sin`, `cos` from a vectorised math library whose rounding is nobody's specification (round 3 - 5: "stated
unreproducible").  But a float32 uniform takes 2^24 values: `radius` is a function of 2^24 inputs, `cos(theta)` / `sin(theta)` of
2^24 inputs - three tables of 64 MB ARE the functions.  They are read off torch itself: a private `torch.Generator` is handed
crafted mt19937 states whose next outputs are chosen words (tempering is invertible), so that

  * with u2 = 0 (theta = 0: cos = 1, sin = 0 exactly - checked) `torch.normal(0, 1)` returns radius(u1) for chosen u1: table R;
  * with u1 = some u* whose radius is EXACTLY 1.0 (found in R; a power of two would do) it returns cos / sin(theta(u2)): tables C, S.

Whatever instruction set ATen dispatches to on this machine, the tables are what `torch.normal` does here.  On the device a value
is `fma(R[k1] * C[k2], std, mean)` - the product rounded, then one fused multiply-add, as ATen's `_mm256_fmadd_ps` (with mean = 0,
the only use here, the unfused form gives the same float).  tests/test_torch_stream.py holds the device values against
`torch.normal` under `torch.manual_seed`, bit for bit.  ~4 s of set-up per process and machine; cached under ~/.cache.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch

BITS = 24
NVAL = 1 << BITS
_PER_CALL = 608  # crafted outputs per generator state (at most 623 before the generator twists), whole blocks of 16


def _untemper(y: np.ndarray) -> np.ndarray:
    y = y.astype(np.uint64)
    y ^= y >> np.uint64(18)
    y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
    x = y.copy()
    for _ in range(4):
        x = y ^ ((x << np.uint64(7)) & np.uint64(0x9D2C5680))
    y = x & np.uint64(0xFFFFFFFF)
    x = y.copy()
    for _ in range(2):
        x = y ^ (x >> np.uint64(11))
    return x & np.uint64(0xFFFFFFFF)


class _Crafted:
    """A private generator whose next outputs are chosen 32-bit words (the default generator is never touched)."""

    def __init__(self):
        self.g = torch.Generator()
        self.g.manual_seed(0)
        self.buf = self.g.get_state().numpy().copy()
        if self.buf.size != 5056:
            raise RuntimeError("unexpected layout of torch's CPU generator state")
        # legacy pod: uint64 seed | int32 left | int32 seeded | uint64 next | uint64 state[624] | ...
        self.buf[8:12] = np.frombuffer(np.int32(624).tobytes(), dtype=np.uint8)   # 623 outputs before the next twist
        self.buf[16:24] = 0                                                        # next = 0
        self.words = self.buf[24:24 + 624 * 8].view(np.uint64)

    def normal(self, words: np.ndarray) -> np.ndarray:
        n = words.shape[0]
        self.words[:n] = _untemper(words)
        self.g.set_state(torch.from_numpy(self.buf))
        return torch.normal(0.0, 1.0, size=(n,), generator=self.g).numpy()


def _extract(u1_words: np.ndarray | None, u2_words: np.ndarray | None):
    """radius table (u1_words None: all 2^24 of them, u2 = 0) or cos / sin tables (u2_words None: all, u1 = the given word)."""
    cr = _Crafted()
    out1 = np.empty(NVAL, dtype=np.float32)
    out2 = np.empty(NVAL, dtype=np.float32)
    half = _PER_CALL // 2
    blocks = half // 8
    words = np.zeros(_PER_CALL, dtype=np.uint64)
    idx = np.arange(half, dtype=np.int64)
    lo = (idx // 8) * 16 + idx % 8  # positions of the u1 words; the u2 words sit 8 further
    for k0 in range(0, NVAL, half):
        ks = (k0 + idx) & (NVAL - 1)
        if u1_words is None:
            words[lo] = ks
            words[lo + 8] = 0
        else:
            words[lo] = u1_words
            words[lo + 8] = ks
        z = cr.normal(words)
        m = min(half, NVAL - k0)
        out1[k0:k0 + m] = z[lo][:m]
        out2[k0:k0 + m] = z[lo + 8][:m]
    del blocks
    return out1, out2


def _fingerprint() -> str:
    cr = _Crafted()
    probe = cr.normal(np.arange(1, _PER_CALL + 1, dtype=np.uint64) * np.uint64(2654435761) & np.uint64(0xFFFFFFFF))
    return hashlib.sha256(torch.__version__.encode() + probe.tobytes()).hexdigest()[:24]


_tables = None


def host_tables():
    """(R, C, S) float32 arrays of 2^24 entries: R[k] = radius for the uniform k 2^-24 (u1 = 1 - k 2^-24), C / S[k] = cos / sin of
    theta = 2 pi (k 2^-24), all exactly as torch.normal computes them on this machine."""
    global _tables
    if _tables is not None:
        return _tables
    cache = os.path.join(os.path.expanduser(os.environ.get("MIDAS_CACHE", "~/.cache/midastouch_amd")), "torch_normal_" + _fingerprint() + ".npz")
    if os.path.exists(cache):
        try:
            d = np.load(cache)
            _tables = (d["R"], d["C"], d["S"])
            return _tables
        except Exception:
            pass
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        R, zero = _extract(None, None)
        if np.any(zero != 0.0):
            raise RuntimeError("sin(0) is not 0 in this torch build: the radius table cannot be read off torch.normal")
        ones = np.nonzero(R == np.float32(1.0))[0]
        if ones.size == 0:
            raise RuntimeError("no uniform gives a radius of exactly 1.0 in this torch build: the cos / sin tables cannot be read off torch.normal")
        C, S = _extract(np.uint64(ones[0]), None)
        if C[0] != np.float32(1.0) or S[0] != np.float32(0.0):
            raise RuntimeError("cos(0) / sin(0) are not 1 / 0 in this torch build")
    finally:
        torch.set_num_threads(threads)
    _tables = (R, C, S)
    try:
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        np.savez(cache + ".tmp.npz", R=R, C=C, S=S)
        os.replace(cache + ".tmp.npz", cache)
    except Exception:
        pass
    return _tables


def emulate(words: np.ndarray, numel: int, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """torch.normal(mean, std, size=(numel,)) of float32 values from the generator outputs `words` (numel + 16 of them when numel
    is not a multiple of 16, ATen draws the last 16 again) - the tables' own check, in numpy; numel >= 16."""
    R, C, S = host_tables()
    k = (np.asarray(words, dtype=np.uint64) & np.uint64(NVAL - 1)).astype(np.int64)
    out = np.empty(numel, dtype=np.float32)

    def block16(kw):
        r = R[kw[:8]]
        a = (r * C[kw[8:16]]).astype(np.float32)
        b = (r * S[kw[8:16]]).astype(np.float32)
        f = lambda n: (n.astype(np.float64) * np.float64(np.float32(std)) + np.float64(np.float32(mean))).astype(np.float32)  # noqa: E731  (one rounding: fma)
        return np.concatenate([f(a), f(b)])

    full = numel // 16
    for b in range(full):
        out[16 * b:16 * b + 16] = block16(k[16 * b:16 * b + 16])
    if numel % 16:
        out[numel - 16:] = block16(k[numel:numel + 16])
    return out
