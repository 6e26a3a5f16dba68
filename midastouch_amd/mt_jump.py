"""Jump-ahead polynomials of mt19937 (host-side set-up of csrc/mt19937.hip's chunked generator).

The word stream x[k] of mt19937 is linear over GF(2): every bit position obeys the recurrence of the generator's
characteristic polynomial phi(t) (degree 19937, 135 terms), so for any distance J

    x[k + J] = XOR of x[k + i] over the exponents i of  t^J mod phi(t)        (i < 19937, k >= 1)

(Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008 - there
the right-hand side is evaluated by stepping the generator; here the 19937 + 623 words it needs are simply the words the
previous call generated, still on the device).  A frame's 2 N words are then cut into G chunks whose start states come from
that formula - G workgroups run the 624-word block recurrence side by side instead of one walking all of them.

phi is not hard-coded: it is found once per process by Berlekamp-Massey on 2 x 19937 output bits of a reference mt19937
(numpy's bit generator; tempering is linear and bijective, so the tempered bits obey the same recurrence), and every
polynomial handed out is checked against that generator's stream before use.  Replaces nothing in the reference (it draws on
torch's CPU generator, modules/particle_filter.py:245); tests/test_torch_stream.py pins the chunked stream to torch.rand.
"""
from __future__ import annotations

import functools

import numpy as np

DEG = 19937
STATE_WORDS = 624
HIST_WORDS = DEG + STATE_WORDS - 1  # x[k + i], k < 624, i < 19937

_SPREAD = None


def _square(p: int) -> int:
    """p(t)^2 over GF(2): bit i -> bit 2 i."""
    global _SPREAD
    if _SPREAD is None:
        v = np.arange(256, dtype=np.uint16)
        s = np.zeros(256, dtype=np.uint16)
        for b in range(8):
            s |= ((v >> b) & 1) << (2 * b)
        _SPREAD = s
    raw = np.frombuffer(p.to_bytes((p.bit_length() + 7) // 8 or 1, "little"), dtype=np.uint8)
    return int.from_bytes(_SPREAD[raw].astype("<u2").tobytes(), "little")


@functools.lru_cache(maxsize=1)
def char_poly():
    """(phi as an int bitset, exponents of phi below its degree)."""
    bits = (np.random.MT19937(5489).random_raw(2 * DEG + 64) & 1).astype(np.uint8)
    # Berlekamp-Massey over GF(2) on Python ints: C, B connection polynomials (bit i = c_i), R the reversed window (bit i = s[n - i])
    C = B = 1
    L, m, R = 0, -1, 0
    for n in range(bits.shape[0]):
        R = (R << 1) | int(bits[n])
        if (C & R).bit_count() & 1:
            T = C
            C ^= B << (n - m)
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, n
    if L != DEG:
        raise RuntimeError(f"Berlekamp-Massey found degree {L}, not {DEG}")
    # s[n] = sum_{i=1..L} c_i s[n - i]  ->  sum_j phi_j s[k + j] = 0 with phi_j = c_{L - j}
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    low = [j for j in range(DEG) if (phi >> j) & 1]
    return phi, tuple(low)


def _reduce(p: int) -> int:
    _, low = char_poly()
    mask = (1 << DEG) - 1
    while p >> DEG:
        h = p >> DEG
        p &= mask
        for j in low:
            p ^= h << j
    return p


@functools.lru_cache(maxsize=256)
def jump_poly(J: int) -> int:
    """t^J mod phi as an int bitset (bit i = coefficient of t^i)."""
    if J < 0:
        raise ValueError("jump distance must be >= 0")
    p = 1
    for bit in bin(J)[2:]:
        p = _reduce(_square(p))
        if bit == "1":
            p = _reduce(p << 1)
    return p


def jump_words(J: int) -> np.ndarray:
    """The polynomial as 624 uint32 words (bit b of word w = coefficient of t^(32 w + b)) - the layout k_mt_jump reads."""
    p = jump_poly(int(J))
    return np.frombuffer(p.to_bytes(STATE_WORDS * 4, "little"), dtype="<u4").copy()


@functools.lru_cache(maxsize=1)
def _check_stream(log2_words: int):
    return np.random.MT19937(4357).random_raw(DEG + 3 * STATE_WORDS + (1 << log2_words)).astype(np.uint32)


def check(J: int, samples: int = 4) -> bool:
    """x[k + J] == XOR of the taps, on the reference generator's stream (J up to 2^24; longer jumps - more than 8 M values per
    call - are built the same way and not re-checked)."""
    if J > (1 << 24):
        return True
    x = _check_stream(max(21, int(J).bit_length()))
    p = jump_poly(int(J))
    taps = np.array([i for i in range(DEG) if (p >> i) & 1], dtype=np.int64)
    for k in range(1, 1 + samples):
        if np.bitwise_xor.reduce(x[k + taps]) != x[k + J]:
            return False
    return True
