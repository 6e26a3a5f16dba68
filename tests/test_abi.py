"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/midas_hip.h
declares (no compute calls are made here); the product fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "midas_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(midas_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from midastouch_amd import _lib
    path = _lib.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/midas_hip.h but not exported"
    # ... and nothing else: every midas_* symbol the library exports is declared (a stray debug entry would be a second, unreviewed ABI)
    import subprocess
    exported = sorted({ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout.splitlines()
                       if ln.split() and ln.split()[-1].startswith("midas_")})
    assert exported == names, sorted(set(exported) ^ set(names))
    # the python binding table covers exactly the declared surface
    assert sorted(_lib.SIGNATURES) == names
    lib.midas_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.midas_version()
    lib.midas_strerror.restype = ctypes.c_char_p
    assert lib.midas_strerror(0) == b"ok" and lib.midas_strerror(-1) == b"invalid argument"


@pytest.mark.parametrize("cname,mirror", [("midas_step_args", "StepArgs"), ("midas_lazy_args", "LazyArgs")])
def test_args_structs_match_header(cname, mirror):
    """Field order of the ctypes mirrors of midas_step_args / midas_lazy_args follows the header."""
    from midastouch_amd import _lib
    text = open(os.path.join(REPO, "include", "midas_hip.h")).read()
    body = text[text.index("typedef struct %s {" % cname):text.index("} %s;" % cname)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split(",")
        first = names[0].split()[-1]
        fields.append(first)
        fields.extend(n.strip() for n in names[1:])
    got = [f[0] for f in getattr(_lib, mirror)._fields_]
    norm = [f.replace("_dev", "") for f in fields]
    assert norm == got


def test_guide_layout_query_needs_no_device():
    """midas_lazy_guide_layout / midas_lazy_guide_bytes are pure size queries (a binding sizes guide_dev with them before any device work)."""
    import ctypes
    from midastouch_amd import _lib
    lib = _lib.load()
    b, u, s = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    lib.midas_lazy_guide_layout.restype = ctypes.c_int
    assert lib.midas_lazy_guide_layout(ctypes.byref(b), ctypes.byref(u), ctypes.byref(s)) == 0
    assert b.value >= 2048 and b.value & (b.value - 1) == 0 and u.value in (4, 8, 16) and s.value >= b.value + 1 and s.value % 8 == 0
    lib.midas_lazy_guide_bytes.restype = ctypes.c_int64
    lib.midas_lazy_guide_bytes.argtypes = [ctypes.c_int64]
    for N in (1, 4096, 4097, 100_000, 1 << 20):
        nb = -(-N // 4096)
        assert lib.midas_lazy_guide_bytes(N) == 2 * nb * s.value * 2  # two variants, 16-bit entries
    assert lib.midas_lazy_guide_bytes(0) == 0


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from midastouch_amd import _lib, ops
    with pytest.raises(_lib.MidasError):
        _lib.context()
    with pytest.raises(_lib.MidasError):
        ops.se3_feature(torch.eye(4)[None])
    with pytest.raises(_lib.MidasError):
        ops.Codebook(torch.zeros(4, 4))


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(REPO, "midastouch_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libmidas_oracle" not in src and "midas_oracle.c" not in src.replace("oracle/midas_oracle.c", ""), f


@pytest.mark.gpu
def test_scratch_reserve_then_no_growth(capfd):
    """midas_scratch_reserve: one chunk that holds what a later call asks for (a DBSCAN call's cell tables: ~90 MB) - the call then
    allocates nothing (MIDAS_SCRATCH_LOG reports every allocation on stderr); reserving less than is held is a no-op."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    import sys
    code = (
        "import torch, numpy as np\n"
        "from midastouch_amd import _lib, ops\n"
        "dev = torch.device('cuda', 0)\n"
        "ctx = _lib.context(dev)\n"
        "ctx.call('midas_scratch_reserve', 160 << 20)\n"
        "print('RESERVED', flush=True)\n"
        "import sys; sys.stderr.write('MARK\\n'); sys.stderr.flush()\n"
        "P = torch.eye(4, device=dev)[None].repeat(5000, 1, 1).contiguous(); P[:, :3, 3] = torch.randn(5000, 3, device=dev) * 0.01\n"
        "lab, info = ops.dbscan(P, 1e-2)\n"
        "ctx.call('midas_scratch_reserve', 1 << 20)\n"
        "torch.cuda.synchronize(); print('DONE', int(info[0]))\n")
    env = dict(os.environ, MIDAS_SCRATCH_LOG="1", PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "DONE" in r.stdout
    before, after = r.stderr.split("MARK")
    assert "reserved one chunk" in before and "[midas] scratch" not in after, r.stderr
