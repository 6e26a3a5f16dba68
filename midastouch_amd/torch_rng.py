"""torch's CPU random stream, reproduced on the device (csrc/mt19937.hip, include/midas_hip.h midas_mt19937_*).

The reference resamples with `WeightedRandomSampler` -> `torch.multinomial(weights.double(), N, True)` on torch's default
CPU generator (modules/particle_filter.py:245): under `torch.manual_seed(s)` the N draws are the stream of
`torch.rand(N, dtype=float64)`.  `TorchCpuStream(seed)` keeps that generator's state in device memory and hands out the
same uniforms without a host generator or a per-frame upload; `skip_normal(numel)` steps over the outputs a
`torch.normal(..., size)` of float32 values would have taken (add_noise_to_odom, :326-335), so a caller that still draws
its motion noise on the host generator stays aligned with the reference's stream.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import _ptr


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def normal_words(numel: int) -> int:
    """32-bit generator outputs one CPU torch.normal of `numel` float32 values consumes (ATen normal_fill: one per value,
    the last 16 drawn again when numel is not a multiple of 16).  Sizes below 16 take ATen's scalar path: not modelled."""
    if numel < 16:
        raise _lib.MidasError("torch.normal of fewer than 16 values uses another code path of ATen: not modelled")
    return numel + (16 if numel % 16 else 0)


class TorchCpuStream:
    """`overlap=True` (default): the generator runs on a stream and a library context of its own - its single-workgroup block
    recurrence (one CU, ~100 us for a frame's 2 N words at N = 100k: 321 blocks of 624 words at 310 ns) then runs BESIDE the frame kernels of the caller's stream
    instead of in front of them.  `rand64()` orders the result behind the caller's stream as before; `rand64_async()` returns
    (tensor, event) and leaves the wait to the consumer (the pipelined engine: the draws of frame t are consumed by frame
    t + 1's launch).  The output buffers rotate (3): a buffer is rewritten two calls later, after the side stream has waited
    for the caller's stream as of THAT call - whoever consumed it was enqueued before.

    `pieces` (default 6; 0 = always the sequential walk): a call's 2 N words are generated in that many pieces side by side once
    the previous call has left MIDAS_MT19937_HIST_WORDS of its words on the device - mt19937 is linear over GF(2), the words
    that start a piece follow from those by one polynomial per piece (mt_jump.py; computed on the host once per distinct
    (previous size, skip, size), ~10 ms each, checked against a reference generator before use).  Same numbers, bit for bit;
    321 sequential blocks (86 us at N = 100k) become 54: 12 us for the jump, 15 for the blocks, 4 for the conversion - the seeded
    step at c2 goes from 10.8k to 22.7k frames/s (4 / 6 / 8 / 12 pieces: 21.7 / 22.7 / 21.7 / 21.8k; tools/bench_parity_mode.py)."""

    def __init__(self, seed: int, device=None, overlap: bool = True, pieces: int = 6):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.device = _lib.context(dev).device
        self.overlap = bool(overlap)
        if self.overlap:
            self.side = torch.cuda.Stream(self.device)
            with torch.cuda.stream(self.side):
                self.ctx = _lib.Context(self.device)  # bound to the side stream for good (never re-bound)
        else:
            self.side = None
            self.ctx = _lib.context(dev)
        self.state = torch.zeros(626, dtype=torch.int32, device=self.device)
        self.pending_skip = 0
        self.pieces = int(pieces)
        self.chain_after = 2  # occurrences of a (previous size, skip, size) pattern before its jump polynomials are computed (1: at once)
        self._hist = torch.zeros(_lib.MT19937_HIST_WORDS, dtype=torch.int32, device=self.device)
        self._hist_words = 0   # words the call that wrote _hist handed out (0: no history)
        self._polys = {}       # (previous words, skip, words) -> device table of the pieces' polynomials
        self._bufs, self._turn = [None, None, None], 0
        self.manual_seed(seed)

    def _enter(self):
        """Orders the generator's stream behind the caller's current stream (state, buffers and seeds are shared with it)."""
        if self.side is None:
            self.ctx.bind_current_stream()
            return
        self.side.wait_stream(torch.cuda.current_stream(self.device))

    def _call(self, name, *args):
        self.ctx.check(getattr(self.ctx.lib, name)(self.ctx.h, *args))

    def manual_seed(self, seed: int):
        """torch.manual_seed(seed) for this stream."""
        self._enter()
        self._call("midas_mt19937_seed", int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(self.state))
        self.pending_skip = 0
        self._hist_words = 0
        self._mark = None
        return self

    # ---- hand-over between torch's host generator and this stream --------------------------------------------------
    # A caller that draws some numbers on the host (init_filter, particle_filter.py:129-145) and the per-frame ones here keeps
    # ONE stream: from_host() continues the host generator where it stands, to_host() gives it back.  Layout of the host
    # state (CPUGeneratorImpl's legacy pod, 5056 bytes): uint64 seed | int32 left | int32 seeded | uint64 next | uint64 state[624] | ..
    def from_host(self, generator: torch.Generator | None = None):
        """Continue `generator` (default: torch's default CPU generator) on the device: the next draw here is the number the host
        generator would have produced next."""
        import numpy as np
        b = (torch.get_rng_state() if generator is None else generator.get_state()).numpy()
        if b.size != 5056:
            raise _lib.MidasError("unexpected layout of torch's CPU generator state")
        left = int(np.frombuffer(b[8:12].tobytes(), dtype=np.int32)[0])
        nxt = int(np.frombuffer(b[16:24].tobytes(), dtype=np.uint64)[0])
        words = np.frombuffer(b[24:24 + 624 * 8].tobytes(), dtype=np.uint64).astype(np.uint32)
        # at::mt19937 twists when --left reaches 0: left == 1 means "a new block is due" whatever next says (fresh seeds: left 1, next 0)
        pos = 624 if left == 1 else nxt
        host = np.concatenate([words, np.array([pos, 0], dtype=np.uint32)]).view(np.int32)
        self._enter()
        with torch.cuda.stream(self.side) if self.side is not None else _null():
            self.state.copy_(torch.from_numpy(host.copy()), non_blocking=False)
        self.pending_skip = 0
        self._hist_words = 0
        self._mark = None
        return self

    def to_host(self, generator: torch.Generator | None = None):
        """Give the stream back: `generator` (default: torch's default CPU generator) continues where this stream stands
        (pending skips applied).  Synchronises with the generator's stream."""
        import numpy as np
        if self.pending_skip:
            self._enter()
            self._call("midas_mt19937_rand64_chunked", _ptr(self.state), self.pending_skip, 0, None, None, None, 0)
            self.pending_skip = 0
            self._hist_words = 0
        if self.side is not None:
            self.side.synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()
        st = self.state.cpu().numpy().view(np.uint32)
        pos = int(st[624])
        g = torch.default_generator if generator is None else generator
        b = g.get_state().numpy().copy()
        if b.size != 5056:
            raise _lib.MidasError("unexpected layout of torch's CPU generator state")
        # pos words of the stored block are consumed: next = pos, left = 624 - pos + 1 (the twist happens when --left hits 0)
        b[8:12] = np.frombuffer(np.int32(624 - pos + 1).tobytes(), dtype=np.uint8)
        b[12:16] = np.frombuffer(np.int32(1).tobytes(), dtype=np.uint8)
        b[16:24] = np.frombuffer(np.uint64(pos if pos < 624 else 0).tobytes(), dtype=np.uint8)
        b[24:24 + 624 * 8] = np.frombuffer(st[:624].astype(np.uint64).tobytes(), dtype=np.uint8)
        g.set_state(torch.from_numpy(b))
        return self

    def skip_words(self, n: int):
        """Step over n 32-bit outputs (applied with the next draw)."""
        self.pending_skip += int(n)
        return self

    def skip_normal(self, numel: int):
        """Step over what torch.normal(mean, std, size) with `numel` float32 values takes."""
        return self.skip_words(normal_words(numel))

    def rand64_async(self, N: int, out: torch.Tensor | None = None):
        """The next N values of torch.rand(N, dtype=torch.float64), enqueued on the generator's stream: (tensor, event or None).
        The consumer waits for the event on its stream before it reads the tensor (None: same stream, already ordered).
        Without `out` the tensor is one of THREE rotating internal buffers: ONE consumer, which has enqueued its use of a buffer
        before the third following call (the engines: one draw per frame, consumed by that frame) - anybody who keeps the
        tensor longer, or shares the stream object between engines, passes `out` (or calls rand64, which allocates)."""
        N = int(N)
        own = out is None
        fresh = False
        if out is None:
            i = self._turn
            self._turn = (i + 1) % 3
            if self._bufs[i] is None or self._bufs[i].numel() != N:
                fresh = True  # (see below: a new block may be memory the caller's stream is still using)
                # (a buffer dropped here may still be being written on the generator's stream: the allocator learns of that stream
                # at allocation, so the block is not handed out again before the write is done)
                self._bufs[i] = torch.empty(N, dtype=torch.float64, device=self.device)
                if self.side is not None:
                    self._bufs[i].record_stream(self.side)
            out = self._bufs[i]
        if own and self.side is not None:
            # A rotating buffer was last handed out three calls ago and, by the contract above, its consumer was enqueued before the
            # call after that one: the generator waits for the caller's stream as it stood at the PREVIOUS call, not as it stands
            # now - it may run a whole frame ahead of the kernels that are being enqueued (state and history are its own).
            cur = torch.cuda.current_stream(self.device)
            mark = torch.cuda.Event()
            mark.record(cur)
            prev, self._mark = getattr(self, "_mark", None), mark
            # (a buffer allocated in THIS call is a block the allocator may have taken back from the caller's stream a moment ago -
            # kernels enqueued there since the previous mark may still read it: the full wait, once per buffer)
            if prev is None or fresh:
                self.side.wait_stream(cur)
            else:
                self.side.wait_event(prev)
        else:
            self._enter()
        words = 2 * N
        polys = self._chain_polys(words)
        self._call("midas_mt19937_rand64_chunked", _ptr(self.state), self.pending_skip, N, _ptr(out), _ptr(self._hist), _ptr(polys),
                   self.pieces if polys is not None else 0)
        return self._drawn(words, out)

    def _chain_polys(self, words: int, pieces: int | None = None):
        """The jump polynomials of a call of `words` words, or None (no history / too short: the sequential walk)."""
        pieces = self.pieces if pieces is None else pieces
        if pieces > 0 and self._hist_words and words >= _lib.MT19937_HIST_WORDS:
            # the polynomials cost the host ~10 ms each: worth it for a pattern of sizes that REPEATS (a fixed-N engine: the same
            # three calls every frame), not for a particle count that changes every frame (the annealing loop: 100 ms a frame of
            # host arithmetic when every call asked for its own set) - a (previous size, skip, size) is served in pieces from its
            # second occurrence on
            key = (self._hist_words, self.pending_skip, words, pieces)
            if key in self._polys:
                return self._polys[key]
            seen = self.__dict__.setdefault("_seen", {})
            if len(seen) > 256:
                seen.clear()
            seen[key] = seen.get(key, 0) + 1
            if seen[key] >= self.chain_after:
                return self._piece_polys(self._hist_words, self.pending_skip, words, pieces)
        return None

    def _drawn(self, words: int, out):
        # (a call too short to leave a history, or a pure skip, breaks the chain: the next call walks sequentially)
        self._hist_words = words if words >= _lib.MT19937_HIST_WORDS else 0
        self.pending_skip = 0
        if self.side is None:
            return out, None
        ev = torch.cuda.Event()
        ev.record(self.side)
        return out, ev

    def _normal_tables(self):
        """The three 2^24-entry tables of torch.normal's float32 path on this machine (torch_normal.py), on the device."""
        tabs = getattr(self, "_ntab", None)
        if tabs is None:
            from . import torch_normal
            with torch.cuda.stream(self.side) if self.side is not None else _null():
                tabs = tuple(torch.from_numpy(a).to(self.device) for a in torch_normal.host_tables())
            self._ntab = tabs
        return tabs

    def normal_async(self, mean: float, std: float, size, out: torch.Tensor | None = None):
        """torch.normal(mean, std, size=size) of float32 values - the reference's motion-noise draws (particle_filter.py:326-335) -
        from this stream, enqueued on the generator's stream: (tensor, event or None) as rand64_async.  The result is a fresh tensor
        unless `out` is given (float32, contiguous, numel elements); at least 16 values (ATen's scalar path below that is not
        modelled)."""
        shape = tuple(size) if hasattr(size, "__len__") else (int(size),)
        numel = 1
        for d in shape:
            numel *= int(d)
        if numel < 16:
            raise _lib.MidasError("torch.normal of fewer than 16 values uses another code path of ATen: not modelled")
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
            if self.side is not None:
                out.record_stream(self.side)
        R, Ct, S = self._normal_tables()
        self._enter()
        words = normal_words(numel)
        polys = self._chain_polys(words) if numel >= _lib.MT19937_HIST_WORDS else None
        self._call("midas_mt19937_normal32", _ptr(self.state), self.pending_skip, numel, float(mean), float(std), _ptr(R), _ptr(Ct), _ptr(S),
                   _ptr(out), _ptr(self._hist), _ptr(polys), self.pieces if polys is not None else 0)
        return self._drawn(words if numel >= _lib.MT19937_HIST_WORDS else 0, out)

    def normal(self, mean: float, std: float, size, out: torch.Tensor | None = None) -> torch.Tensor:
        """torch.normal(mean, std, size=size) (float32), ordered behind the caller's current stream."""
        z, ev = self.normal_async(mean, std, size, out)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return z

    def draws_async(self, spec):
        """Several consecutive draws by ONE walk of the generator (midas_mt19937_draws): spec = sequence of ("rand64", N) and
        ("normal", mean, std, size) in the stream's order - a seeded frame of the reference is normal (N, 3) twice and rand64 N
        (particle_filter.py:326-335, :245).  Returns ([tensors], event or None): the numbers of the separate calls, fresh tensors,
        enqueued on the generator's stream behind the caller's current stream."""
        import ctypes as C
        outs, segs, words, need_tables = [], [], 0, False
        for item in spec:
            if item[0] == "rand64":
                n = int(item[1])
                out = torch.empty(n, dtype=torch.float64, device=self.device)
                segs.append((_lib.MT_SEGMENT_RAND64, n, 0.0, 1.0, out))
                words += 2 * n
            elif item[0] == "normal":
                _, mean, std, size = item
                shape = tuple(size) if hasattr(size, "__len__") else (int(size),)
                numel = 1
                for d in shape:
                    numel *= int(d)
                if numel < 16:
                    raise _lib.MidasError("torch.normal of fewer than 16 values uses another code path of ATen: not modelled")
                out = torch.empty(shape, dtype=torch.float32, device=self.device)
                segs.append((_lib.MT_SEGMENT_NORMAL32, numel, float(mean), float(std), out))
                words += normal_words(numel)
                need_tables = True
            else:
                raise ValueError(f"unknown draw {item[0]!r}")
            if self.side is not None:
                out.record_stream(self.side)
            outs.append(out)
        R, Ct, S = self._normal_tables() if need_tables else (None, None, None)
        self._enter()
        arr = (_lib.MtSegment * len(segs))()
        for a, (kind, count, mean, std, out) in zip(arr, segs):
            a.kind, a.count, a.mean, a.std, a.out_dev = kind, count, mean, std, out.data_ptr()
        pieces = self.pieces_for(words)
        polys = self._chain_polys(words, pieces)
        self._call("midas_mt19937_draws", _ptr(self.state), self.pending_skip, len(segs), C.cast(arr, C.c_void_p), _ptr(R), _ptr(Ct), _ptr(S),
                   _ptr(self._hist), _ptr(polys), pieces if polys is not None else 0)
        _, ev = self._drawn(words, None)
        return outs, ev

    def pieces_for(self, words: int) -> int:
        """Pieces a call of `words` words is cut into: `pieces` for a draw of 2 x 10^5 words (N = 10^5 uniforms), more for longer calls
        so that a piece stays ~54 blocks of 624 words (the walk of a piece is the sequential part; a jump costs ~2 us a piece)."""
        if self.pieces <= 0:
            return 0
        return max(self.pieces, min(48, -(-words // (54 * 624))))

    def _piece_polys(self, prev_words: int, skip: int, words: int, pieces: int | None = None):
        pieces = self.pieces if pieces is None else pieces
        key = (prev_words, skip, words, pieces)
        tab = self._polys.get(key)
        if tab is None:
            import numpy as np

            from . import mt_jump
            nblocks = -(-words // 624)
            bpc = -(-nblocks // pieces)
            rows = []
            for c in range(pieces):
                J = prev_words + skip + c * bpc * 624
                if not mt_jump.check(J):
                    raise _lib.MidasError(f"mt19937 jump polynomial for distance {J} failed its check against the reference generator")
                rows.append(mt_jump.jump_words(J))
            host = torch.from_numpy(np.stack(rows).view(np.int32))
            if len(self._polys) >= 16:
                self._polys.clear()
            # uploaded on the generator's stream (the caller is inside _enter() .. the launch)
            with torch.cuda.stream(self.side) if self.side is not None else _null():
                tab = host.to(self.device)
            self._polys[key] = tab
        return tab

    def rand64(self, N: int, out: torch.Tensor | None = None) -> torch.Tensor:
        """The next N values of torch.rand(N, dtype=torch.float64) (== the draws of torch.multinomial(w64, N, True)), ordered
        behind the caller's current stream; a fresh tensor unless `out` is given."""
        if out is None:
            out = torch.empty(int(N), dtype=torch.float64, device=self.device)
            if self.side is not None:
                out.record_stream(self.side)  # written on the generator's stream: the allocator must know
        u, ev = self.rand64_async(N, out)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return u
