"""Codebook container + converter against G11: pickles written by the reference's own
`tactile_tree` class (tools/gen_codebook_pickle.py) and the arrays that went into them."""
import os
import sys

import numpy as np
import pytest
import torch

from midastouch_amd import codebook_io
from midastouch_amd._lib import MidasError

G = os.path.join(os.path.dirname(__file__), "golden")


def _arrays():
    return np.load(os.path.join(G, "g11_codebook_arrays.npz"))


def test_reads_reference_pickle_without_reference_package():
    assert "midastouch" not in sys.modules and "pynanoflann" not in sys.modules
    cb = codebook_io.read_reference_pickle(os.path.join(G, "g11_codebook_ref.pkl"))
    z = _arrays()
    assert "midastouch" not in sys.modules and "pynanoflann" not in sys.modules  # placeholders, nothing imported
    assert np.array_equal(cb["poses"].numpy(), z["poses"]) and cb["poses"].dtype == torch.float32
    assert np.array_equal(cb["cam_poses"].numpy(), z["cam_poses"])
    assert cb["embeddings"].dtype == torch.float64  # the reference's contract (tcn.py:148)
    assert np.array_equal(cb["embeddings"].numpy(), z["embeddings"].astype(np.float64))
    assert np.array_equal(cb["logmap_pose"].numpy(), z["logmap_pose"])


def test_convert_round_trip(tmp_path):
    dst = str(tmp_path / "codebook.npz")
    info = codebook_io.convert(os.path.join(G, "g11_codebook_ref.pkl"), dst)
    z = _arrays()
    assert info["K"] == z["poses"].shape[0] and info["D"] == z["embeddings"].shape[1]
    assert info["embeddings_dtype"] == "float32" and info["has_reference_logmap"]
    back = codebook_io.load_codebook(dst)
    assert back["embeddings"].dtype == torch.float32  # lossless: float32 casts
    assert np.array_equal(back["embeddings"].numpy(), z["embeddings"])
    assert np.array_equal(back["poses"].numpy(), z["poses"]) and np.array_equal(back["cam_poses"].numpy(), z["cam_poses"])
    via_ext = codebook_io.load_codebook(os.path.join(G, "g11_codebook_ref.pkl"))
    assert torch.equal(via_ext["poses"], back["poses"])


def test_float64_embeddings_stay_float64(tmp_path):
    dst = str(tmp_path / "cb64.npz")
    info = codebook_io.convert(os.path.join(G, "g11_codebook_ref_f64.pkl"), dst)
    assert info["embeddings_dtype"] == "float64" and info["K"] == 40
    assert np.array_equal(codebook_io.load_codebook(dst)["embeddings"].numpy(), _arrays()["embeddings_f64"])


def test_rejects_garbage_and_hostile_pickles(tmp_path):
    bad = tmp_path / "bad.pkl"
    bad.write_bytes(b"not a pickle")
    with pytest.raises(MidasError):
        codebook_io.read_reference_pickle(str(bad))
    import pickle
    hostile = tmp_path / "hostile.pkl"
    hostile.write_bytes(pickle.dumps(eval))  # builtins.eval is not on the allow list
    with pytest.raises(MidasError):
        codebook_io.read_reference_pickle(str(hostile))
    class _OsCall:
        def __reduce__(self):
            return (os.system, ("true",))
    h2 = tmp_path / "h2.pkl"
    h2.write_bytes(pickle.dumps(_OsCall()))
    with pytest.raises(MidasError):  # posix.system becomes an inert placeholder: no poses in the result
        codebook_io.read_reference_pickle(str(h2))
    # re-export gadgets: torch and numpy expose os / sys as attributes, so a prefix allow-list would hand out
    # torch.os.getcwd / numpy.os.system; the loader resolves exact (module, name) pairs only
    for mod, name in (("torch", "os"), ("numpy", "os"), ("torch", "load"), ("numpy", "load"), ("builtins", "getattr"),
                      ("builtins", "type"), ("dill._dill", "_load_type"), ("numpy._core.multiarray", "scalar"),
                      ("torch.serialization", "load"), ("os", "system")):
        g = tmp_path / "gadget.pkl"
        g.write_bytes(b"\x80\x02c" + mod.encode() + b"\n" + name.encode() + b"\n.")
        with pytest.raises(MidasError, match="refusing"):
            codebook_io.read_reference_pickle(str(g))
    # getattr(torch.os, "getcwd")() as a stream: GLOBAL getattr, GLOBAL torch.os, "getcwd", TUPLE2, REDUCE, EMPTY_TUPLE, REDUCE
    g.write_bytes(b"\x80\x02cbuiltins\ngetattr\nctorch\nos\nX\x06\x00\x00\x00getcwd\x86R)R.")
    with pytest.raises(MidasError, match="refusing"):
        codebook_io.read_reference_pickle(str(g))
    # a storage payload is read by torch's weights-only loader, not by the unrestricted torch.load the real
    # torch.storage._load_from_bytes wraps: a hostile nested stream is refused, an honest one still loads
    nested = pickle.dumps(_OsCall())
    g.write_bytes(b"\x80\x02ctorch.storage\n_load_from_bytes\n" + pickle.dumps(nested)[2:-1] + b"\x85R.")
    with pytest.raises(MidasError):
        codebook_io.read_reference_pickle(str(g))
    empty = tmp_path / "e.npz"
    np.savez(str(empty), poses=np.zeros((3, 4, 4), np.float32))
    with pytest.raises(MidasError):
        codebook_io.load_codebook(str(empty))
    with pytest.raises(MidasError):
        codebook_io.save_codebook(str(tmp_path / "x.npz"), np.zeros((3, 4, 4)), np.zeros((2, 4, 4)), np.zeros((3, 8)))


def test_cli(tmp_path, capsys):
    dst = str(tmp_path / "c.npz")
    assert codebook_io.main(["convert", os.path.join(G, "g11_codebook_ref.pkl"), dst]) == 0
    assert codebook_io.main(["info", dst]) == 0
    assert "embeddings" in capsys.readouterr().out
    assert codebook_io.main([]) == 2


@pytest.mark.gpu
def test_tactile_tree_load_checks_reference_logmap(tmp_path):
    from midastouch_amd.tactile_tree import tactile_tree

    dev = torch.device("cuda", 0)
    t = tactile_tree.load(os.path.join(G, "g11_codebook_ref.pkl"), device=dev, check_logmap=True)
    assert len(t) == 300 and t.embeddings.dtype == torch.float32
    assert t.check_reference_logmap() < 1e-6  # kernels' log-map vs the stored (scipy, see the generator) features
    dst = str(tmp_path / "codebook.npz")
    t.save(dst)
    t2 = tactile_tree.load(dst, device=dev, check_logmap=True)
    q = t.poses[:50]
    p1, c1, e1 = t.SE3_NN(q)
    p2, c2, e2 = t2.SE3_NN(q)
    assert torch.equal(p1, p2) and torch.equal(c1, c2) and torch.equal(e1.idx, e2.idx)
    assert torch.equal(e1.idx.cpu(), torch.arange(50, dtype=torch.int32))
    t2.reference_logmap = t2.reference_logmap + 1e-3
    with pytest.raises(MidasError):
        t2.check_reference_logmap()
