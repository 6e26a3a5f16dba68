"""torch's CPU random stream (at::mt19937) restated: the oracle's generator against torch itself on the CPU (torch is
importable on every box), the device generator (csrc/mt19937.hip) against torch on the GPU box.  This is the stream the
reference's resampler consumes (modules/particle_filter.py:245: WeightedRandomSampler -> torch.multinomial on the default CPU
generator), so with it "bit-exact resample indices under a fixed seed" needs no host generator in the loop."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

SEEDS = (0, 1, 42, 3000, 123456789, 2**32 + 5, 2**63 + 7, 0xFFFFFFFF)
SIZES = (1, 311, 312, 313, 623, 624, 625, 100_000, 1_000_000)


def test_oracle_stream_equals_torch_rand(oracle):
    for seed in SEEDS:
        for N in SIZES:
            if N == 1_000_000 and seed not in (42, 3000):
                continue
            torch.manual_seed(seed)
            ref = torch.rand(N, dtype=torch.float64).numpy()
            assert np.array_equal(oracle.torch_rand64(seed, N), ref), (seed, N)


def test_oracle_stream_alignment_with_the_reference_draw_order(oracle):
    """One frame of the reference's draws (add_noise_to_odom: tn then rot, particle_filter.py:326-335; then the resampler's
    multinomial, :245): the uniforms behind the normals' words are what torch.multinomial consumes."""
    for N in (6, 1000, 4096, 33333):
        torch.manual_seed(11)
        torch.normal(0.0, 2e-4, size=(N, 3))
        torch.normal(0.0, 0.5, size=(N, 3))
        w = torch.rand(N, dtype=torch.float64)
        skip = 2 * oracle.torch_normal_words(3 * N)
        assert np.array_equal(oracle.torch_rand64(11, N, skip), w.numpy()), N
        # and the multinomial itself == inverse-CDF search on that stream
        torch.manual_seed(12)
        idx = torch.multinomial(w, N, True).numpy()
        ridx, status = oracle.resample_indices(w.numpy(), "weighted_random", u=oracle.torch_rand64(12, N))
        assert status == 0 and np.array_equal(ridx, idx), N


@pytest.mark.gpu
def test_device_stream_equals_torch_rand():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.torch_rng import TorchCpuStream
    dev = torch.device("cuda", 0)
    for seed in SEEDS:
        for N in SIZES:
            if N == 1_000_000 and seed not in (42, 3000):
                continue
            torch.manual_seed(seed)
            ref = torch.rand(N, dtype=torch.float64)
            got = TorchCpuStream(seed, dev).rand64(N).cpu()
            assert torch.equal(got, ref), (seed, N)
    # consecutive draws continue the stream (odd and even lengths: values straddling the 624-word blocks), skips included
    torch.manual_seed(5)
    st = TorchCpuStream(5, dev)
    for n in (1, 2, 311, 313, 7, 100_000, 624, 3):
        assert torch.equal(st.rand64(n).cpu(), torch.rand(n, dtype=torch.float64)), n
    torch.manual_seed(6)
    st.manual_seed(6)
    for N in (1000, 33333):
        tn, rot = torch.normal(0.0, 2e-4, size=(N, 3)), torch.normal(0.0, 0.5, size=(N, 3))
        u = torch.rand(N, dtype=torch.float64)
        assert torch.equal(st.skip_normal(3 * N).skip_normal(3 * N).rand64(N).cpu(), u), N


def test_jump_polynomials_against_the_reference_generator():
    """mt_jump: the characteristic polynomial found by Berlekamp-Massey has mt19937's known shape (degree 19937, 135 terms), and
    x[k + J] = XOR of the taps of t^J mod phi on numpy's mt19937 stream for the distances the chunked generator uses."""
    from midastouch_amd import mt_jump
    phi, low = mt_jump.char_poly()
    assert phi.bit_length() - 1 == 19937 and len(low) + 1 == 135
    for J in (0, 1, 623, 624, 19936, 19937, 19938, 200_000, 200_000 + 41 * 624, 2 * 33_333 + 7 * 14 * 624, 2_000_000 + 401 * 624):
        assert mt_jump.check(J, samples=6), J
    w = mt_jump.jump_words(200_000)
    assert w.shape == (624,) and w.dtype == np.uint32 and int(w[-1]) >> 1 == 0  # no term at or beyond t^19937


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [1, 3, 8, 16, 64])
def test_chunked_device_stream_equals_torch_rand(pieces):
    """midas_mt19937_rand64_chunked: from the second draw on a call's words are generated in `pieces` pieces side by side (start
    states by jump polynomials from the previous call's words) - the same stream as torch.rand bit for bit, across changing
    sizes, the reference's draw order (normals skipped between the resampler's draws), calls too short to chain, and re-seeding."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.torch_rng import TorchCpuStream
    dev = torch.device("cuda", 0)
    torch.manual_seed(77)
    st = TorchCpuStream(77, dev, pieces=pieces)
    st.chain_after = 1  # every call that can be served in pieces is (the default waits for a size pattern to repeat)
    chained = 0
    for n in (100_000, 100_000, 100_000, 50_001, 100_000, 10_280, 10_281, 12_000, 5_000, 100_000, 100_000, 311_000):
        had = st._hist_words
        assert torch.equal(st.rand64(n).cpu(), torch.rand(n, dtype=torch.float64)), n
        chained += 1 if (had and 2 * n >= 20560) else 0
    assert chained >= 7
    # the reference's frame: tn, rot (torch.normal, skipped here), then the resampler's N uniforms - three frames
    N = 33_333
    for _ in range(3):
        torch.normal(0.0, 2e-4, size=(N, 3)), torch.normal(0.0, 0.5, size=(N, 3))
        assert torch.equal(st.skip_normal(3 * N).skip_normal(3 * N).rand64(N).cpu(), torch.rand(N, dtype=torch.float64))
    st.manual_seed(5)
    torch.manual_seed(5)
    for n in (20_000, 20_000, 1_000_000, 1_000_000):
        assert torch.equal(st.rand64(n).cpu(), torch.rand(n, dtype=torch.float64)), n
    # the generator's state is torch's: a sequential stream seeded from it continues identically
    seq = TorchCpuStream(0, dev, pieces=0)
    seq.state.copy_(st.state)
    assert torch.equal(seq.rand64(1000).cpu(), torch.rand(1000, dtype=torch.float64))


@pytest.mark.gpu
def test_resampler_on_the_device_stream_matches_reference_digests(golden):
    """G2b (the reference's own resampler at N = 100 000 under torch.manual_seed, tools/gen_goldens_r2.py) through
    particle_filter.resampler with the draws taken from the device replica of torch's generator: the reference's index
    arrays bit for bit, no host generator involved (the default generator is seeded differently on purpose)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from _recipes import g2b_cases, sha
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter
    dev = torch.device("cuda", 0)
    pf = particle_filter(load_config(), np.zeros((8, 3)), 1.0, downsample=1, device=dev)
    g = golden("g2b_resampler_100k")
    n = int(g["N"])
    poses = torch.eye(4, device=dev)[None].repeat(n, 1, 1).contiguous()
    checked = 0
    for ci, w, mode, seed, ref_sha, head, tail in g2b_cases(g):
        if mode != "weighted_random":
            continue
        parts = Particles(poses, torch.as_tensor(w).to(dev), torch.arange(n, dtype=torch.float32, device=dev))
        torch.manual_seed(seed + 1)  # not the reference's seed: the host generator must play no part
        pf.seed_device_stream(seed)
        res = pf.resampler(parts, resample=mode)
        idx = res.labels.cpu().numpy().astype(np.int32)
        assert np.array_equal(idx[:64], head) and np.array_equal(idx[-64:], tail), ci
        assert sha(idx) == ref_sha, ci
        checked += 1
    assert checked == 12


@pytest.mark.gpu
def test_pipelined_engine_on_the_device_stream_equals_host_uniforms():
    """PipelinedFilterEngine.seed_torch_stream(s): the frames resample with torch's stream generated on the device - the
    same particle sets as with u = torch.rand(N, float64) of a host generator seeded s passed in every frame."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    dev = torch.device("cuda", 0)
    N, K, D = 30000, 3000, 256
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    traj = make_trajectory(cb, T=10, seed=2001)
    rng = np.random.default_rng(3)
    start = torch.as_tensor(cb.poses[rng.integers(0, K, N)])
    a = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=77, device=dev)
    b = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=77, device=dev)
    a.set_particles(start)
    b.set_particles(start)
    a.seed_torch_stream(2024)
    torch.manual_seed(2024)
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    for t in range(1, 8):
        a.step(od[t], co[t])
        b.step(od[t], co[t], u=torch.rand(N, dtype=torch.float64))
        if t % 3 == 0:
            assert torch.equal(a.ridx, b.ridx) and torch.equal(a.poses, b.poses), t
    assert torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights)


def test_torch_normal_tables_reproduce_torch_normal():
    """torch_normal.py: the radius / cos / sin tables read off torch.normal (crafted generator states) reproduce torch.normal under a
    seed bit for bit - multiples of 16 and not, the reference's (N, 3) shapes, several scales (CPU: numpy emulation of the device
    kernel's arithmetic, generator words from torch.rand's float32 stream)."""
    from midastouch_amd import torch_normal as tn
    R, C, S = tn.host_tables()
    assert R.shape == C.shape == S.shape == (1 << 24,) and C[0] == 1.0 and S[0] == 0.0 and R[0] == 0.0
    for seed, numel, std in ((5, 16, 1.0), (7, 48, 0.5), (11, 100, 2e-4), (3000, 300_000, 0.5), (42, 3 * 33_333, 2e-4), (9, 17, 60.0)):
        torch.manual_seed(seed)
        ref = torch.normal(0.0, std, size=(numel,)).numpy()
        nw = numel + (16 if numel % 16 else 0)
        torch.manual_seed(seed)
        k = np.round(torch.rand(nw, dtype=torch.float32).numpy().astype(np.float64) * 2.0 ** 24).astype(np.uint64)
        assert np.array_equal(tn.emulate(k, numel, 0.0, std), ref), (seed, numel, std)


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [0, 6])
def test_device_normal_equals_torch_normal(pieces):
    """midas_mt19937_normal32: the reference's motion-noise draws from the device replica of torch's generator - one frame's
    sequence tn, rot (torch.normal, (N, 3)), then the resampler's N float64 uniforms, frame after frame, bit for bit; sizes that are
    and are not multiples of 16, below and above the chaining threshold, sequential and in pieces."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.torch_rng import TorchCpuStream
    dev = torch.device("cuda", 0)
    torch.manual_seed(31)
    st = TorchCpuStream(31, dev, pieces=pieces)
    st.chain_after = 1
    for N in (16, 100, 4096, 33_333, 100_000, 100_000, 100_000, 7_000, 100_000):
        tn_ref = torch.normal(0.0, 2e-4, size=(N, 3))
        rot_ref = torch.normal(0.0, 0.5, size=(N, 3))
        u_ref = torch.rand(N, dtype=torch.float64)
        assert torch.equal(st.normal(0.0, 2e-4, (N, 3)).cpu(), tn_ref), N
        assert torch.equal(st.normal(0.0, 0.5, (N, 3)).cpu(), rot_ref), N
        assert torch.equal(st.rand64(N).cpu(), u_ref), N
    z = torch.normal(1.5, 3.0, size=(1000,))
    assert torch.equal(st.normal(1.5, 3.0, 1000).cpu(), z)  # (a mean that is not zero: the fused multiply-add)


@pytest.mark.gpu
@pytest.mark.parametrize("pieces", [0, 6])
def test_one_walk_for_the_draws_of_a_frame(pieces):
    """midas_mt19937_draws: a frame's draws - the resampler's N float64 uniforms and the next frame's two torch.normal (N, 3) - from ONE
    walk of the generator, in the stream's order; mixed with single calls; sizes that are and are not multiples of 16, below and
    above the chaining threshold, the same pattern repeated (chained from its second occurrence on)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.torch_rng import TorchCpuStream
    dev = torch.device("cuda", 0)
    torch.manual_seed(47)
    st = TorchCpuStream(47, dev, pieces=pieces)
    for N in (100, 33_333, 100_000, 100_000, 100_000, 100_000, 5_000, 100_000, 100_000):
        u_ref = torch.rand(N, dtype=torch.float64)
        a_ref = torch.normal(0.0, 1.0, size=(N, 3))
        b_ref = torch.normal(0.0, 0.5, size=(N, 3))
        (u, a, b), ev = st.draws_async([("rand64", N), ("normal", 0.0, 1.0, (N, 3)), ("normal", 0.0, 0.5, (N, 3))])
        if ev is not None:
            torch.cuda.current_stream(dev).wait_event(ev)
        assert torch.equal(u.cpu(), u_ref), N
        assert torch.equal(a.cpu(), a_ref), N
        assert torch.equal(b.cpu(), b_ref), N
        if N == 5_000:  # a single call in between: the chain goes on from its history or starts again
            assert torch.equal(st.rand64(30_000).cpu(), torch.rand(30_000, dtype=torch.float64))
    # normal first, uniforms last (the reference's order inside one frame), a mean that is not zero
    t_ref, r_ref, u_ref = torch.normal(1.0, 2e-4, size=(40_000, 3)), torch.normal(0.0, 0.5, size=(40_000, 3)), torch.rand(40_000, dtype=torch.float64)
    (t, r, u), ev = st.draws_async([("normal", 1.0, 2e-4, (40_000, 3)), ("normal", 0.0, 0.5, (40_000, 3)), ("rand64", 40_000)])
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)
    assert torch.equal(t.cpu(), t_ref) and torch.equal(r.cpu(), r_ref) and torch.equal(u.cpu(), u_ref)
    st.to_host()
    assert torch.equal(torch.rand(7), torch.rand(7)) is False or True  # (the generator is usable)


@pytest.mark.gpu
def test_stream_hand_over_between_host_and_device():
    """from_host / to_host: one torch stream, drawn alternately on the host generator and on the device replica - as a runner does that
    initialises its particles with host draws (init_filter) and takes the per-frame draws on the device."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from midastouch_amd.torch_rng import TorchCpuStream
    dev = torch.device("cuda", 0)
    torch.manual_seed(123)
    ref = [torch.normal(0.0, 1.0, size=(5000, 3)), torch.rand(777, dtype=torch.float64), torch.normal(0.0, 0.5, size=(100, 3)),
           torch.rand(3), torch.normal(0.0, 2.0, size=(40_000,)), torch.rand(50_000, dtype=torch.float64), torch.rand(5, dtype=torch.float64)]
    torch.manual_seed(123)
    st = TorchCpuStream(0, dev)
    got = [torch.normal(0.0, 1.0, size=(5000, 3))]                 # host
    st.from_host()
    got.append(st.rand64(777).cpu())                               # device
    got.append(st.normal(0.0, 0.5, (100, 3)).cpu())
    st.to_host()
    got.append(torch.rand(3))                                      # host again (float32: one word each)
    st.from_host()
    got.append(st.normal(0.0, 2.0, 40_000).cpu())
    got.append(st.rand64(50_000).cpu())
    st.to_host()
    got.append(torch.rand(5, dtype=torch.float64))
    for i, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), i
    # a freshly seeded host generator (left = 1: the twist is due) hands over as well
    torch.manual_seed(9)
    st.from_host()
    torch.manual_seed(9)
    assert torch.equal(st.rand64(1000).cpu(), torch.rand(1000, dtype=torch.float64))
