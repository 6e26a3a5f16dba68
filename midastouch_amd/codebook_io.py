"""Codebook container and converter.

The reference stores a codebook as a dill pickle of the `tactile_tree` nn.Module, pynanoflann index included
(`tactile_tree/build_codebook.py:130-137`, rewritten by `tactile_tree/process_codebook.py:17-42`, read by
`filter/filter.py:89-93`).  That file cannot be opened without the reference package and pynanoflann importable.
Here the container is a plain `codebook.npz`:

    poses       (K,4,4) float32   sensor poses on the object surface
    cam_poses   (K,4,4) float32   camera poses of the renders
    embeddings  (K,D)   float32 when the float64 values are float32 casts (they are: `tcn.py:148`), else float64
    logmap_pose (K,6)   float32   optional: the reference's own 6-d features (theseus log-map), kept only as a
                                  cross-check for the kernels' log-map (`tactile_tree.load(check_logmap=True)`)

`read_reference_pickle` opens the reference's pickle WITHOUT its classes: an unpickler whose `find_class` resolves an
explicit list of (module, name) pairs (container / ndarray / tensor constructors; `_SAFE_GLOBALS`), refuses every other
name of torch / numpy / dill / os-like modules and hands out inert placeholder classes for the rest, so the module
object and the KD-tree come back as attribute bags and only the tensors are used.

    python -m midastouch_amd.codebook_io convert <codebook.pkl> <codebook.npz>
    python -m midastouch_amd.codebook_io info <codebook.npz | codebook.pkl>
"""
from __future__ import annotations

import io
import os
import pickle
import sys

import numpy as np
import torch

from ._lib import MidasError

FORMAT_VERSION = 1
# The ONLY globals a codebook pickle may resolve to real objects, as exact (module, name) pairs: constructors that
# build containers, arrays and tensors and run nothing else.  Deliberately absent: every module attribute reachable
# through re-exports (torch.os, numpy.os, ...), builtins.getattr / type / eval, numpy's `scalar` (unpickles object
# payloads), torch.load and torch.storage._load_from_bytes (an unrestricted torch.load; replaced below by a
# weights-only one), dill's type / function / code constructors.  Anything else becomes an inert placeholder.
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("copyreg", "_reconstructor"), ("copyreg", "__newobj__"), ("_codecs", "encode"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch", "Size"), ("torch", "device"),
} | {("torch", n + "Storage") for n in ("Float", "Double", "Half", "BFloat16", "Long", "Int", "Short", "Char", "Byte", "Bool")}


def _storage_from_bytes(b):
    """What torch.storage._load_from_bytes does, minus its unrestricted unpickler: the nested stream is read with
    torch's weights-only loader, which builds storages and tensors and nothing else."""
    if not isinstance(b, (bytes, bytearray)):
        raise pickle.UnpicklingError("storage payload is not bytes")
    return torch.load(io.BytesIO(bytes(b)), weights_only=True)


def _create_array(f, args, state, npdict=None):
    """dill._dill._create_array for plain ndarrays: `f` came through find_class, so it is numpy's `_reconstruct`."""
    if getattr(f, "__name__", "") != "_reconstruct" or not getattr(f, "__module__", "").startswith("numpy"):
        raise pickle.UnpicklingError("array reconstructor is not numpy's")
    array = f(*args)
    array.__setstate__(state)
    if array.dtype.hasobject:
        raise pickle.UnpicklingError("object arrays are not codebook data")
    return array


_REPLACED_GLOBALS = {("torch.storage", "_load_from_bytes"): _storage_from_bytes,
                     ("dill._dill", "_create_array"): _create_array}


class _Placeholder:
    """Stands in for a class of the reference (or of pynanoflann) while its pickle is read: keeps whatever state the
    stream hands over, runs nothing."""

    def __init__(self, *args, **kwargs):
        self._pickle_args = (args, kwargs)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and all(isinstance(s, (dict, type(None))) for s in state):
            for s in state:
                self.__dict__.update(s or {})
        else:
            self._pickle_state = state

    def __call__(self, *args, **kwargs):  # placeholder used as a reconstructor function
        return _Placeholder(*args, **kwargs)


def _placeholder_class(module: str, name: str):
    return type(name, (_Placeholder,), {"__module__": module, "_placeholder_for": f"{module}.{name}"})


def _make_unpickler(base):
    class _CodebookUnpickler(base):
        def find_class(self, module, name):
            if (module, name) in _REPLACED_GLOBALS:
                return _REPLACED_GLOBALS[(module, name)]
            if module in ("builtins", "__builtin__"):
                if name in _SAFE_BUILTINS:
                    return getattr(__import__("builtins"), name)
                raise pickle.UnpicklingError(f"refusing builtins.{name} in a codebook pickle")
            if (module, name) in _SAFE_GLOBALS:
                obj = pickle.Unpickler.find_class(self, module, name)  # plain attribute lookup, no dill by-value extras
                if isinstance(obj, type(pickle)):
                    raise pickle.UnpicklingError(f"{module}.{name} is a module")
                return obj
            if module.split(".")[0] in ("torch", "numpy", "dill", "os", "posix", "nt", "subprocess", "sys", "importlib",
                                        "pickle", "shutil", "socket", "ctypes", "runpy", "code", "pty"):
                # a name of these packages that is not on the list is never a codebook's data: refuse loudly
                raise pickle.UnpicklingError(f"refusing {module}.{name} in a codebook pickle")
            return _placeholder_class(module, name)

    return _CodebookUnpickler


def read_reference_pickle(path: str) -> dict:
    """{poses, cam_poses, embeddings[, logmap_pose]} (CPU tensors) from a reference `codebook.pkl`."""
    # the reference writes with dill (build_codebook.py:12): a standard pickle stream whose dill-specific globals
    # (dill._dill._create_array) are answered by find_class above; dill's own Unpickler is not needed and not used
    base = pickle.Unpickler
    with open(path, "rb") as f:
        data = f.read()
    try:
        obj = _make_unpickler(base)(io.BytesIO(data)).load()
    except Exception as e:
        raise MidasError(f"{path}: not a readable codebook pickle ({type(e).__name__}: {e})") from e
    bag = obj if isinstance(obj, dict) else getattr(obj, "__dict__", {})
    out = {}
    for key in ("poses", "cam_poses", "embeddings", "logmap_pose"):
        v = bag.get(key)
        if v is None:
            for store in ("_buffers", "_parameters"):  # had they been registered on the nn.Module
                v = (bag.get(store) or {}).get(key) if isinstance(bag.get(store), dict) else v
                if v is not None:
                    break
        if v is not None:
            out[key] = torch.as_tensor(v).detach().cpu()
    missing = [k for k in ("poses", "cam_poses", "embeddings") if k not in out]
    if missing:
        raise MidasError(f"{path}: no {missing} in the pickled object (attributes found: {sorted(bag)[:12]})")
    return out


def _check_shapes(poses, cam_poses, embeddings):
    K = poses.shape[0]
    if tuple(poses.shape[1:]) != (4, 4) or tuple(cam_poses.shape) != tuple(poses.shape):
        raise MidasError(f"codebook poses must be (K,4,4) twice, got {tuple(poses.shape)} and {tuple(cam_poses.shape)}")
    if embeddings.dim() != 2 or embeddings.shape[0] != K:
        raise MidasError(f"codebook embeddings must be (K,D) with K={K}, got {tuple(embeddings.shape)}")


def compact_embeddings(embeddings: torch.Tensor) -> torch.Tensor:
    """float32 when every value survives the round trip (the reference's float64 codes are float32 network
    outputs), the input dtype otherwise."""
    e = torch.as_tensor(embeddings)
    if e.dtype == torch.float32:
        return e
    e32 = e.to(torch.float32)
    same = torch.equal(e32.to(e.dtype), e) if not torch.isnan(e).any() else False
    return e32 if same else e


def save_codebook(path: str, poses, cam_poses, embeddings, logmap_pose=None):
    poses = torch.as_tensor(poses).detach().cpu().float()
    cam_poses = torch.as_tensor(cam_poses).detach().cpu().float()
    embeddings = compact_embeddings(torch.as_tensor(embeddings).detach().cpu())
    _check_shapes(poses, cam_poses, embeddings)
    arrays = {"format_version": np.int32(FORMAT_VERSION), "poses": poses.numpy(), "cam_poses": cam_poses.numpy(),
              "embeddings": embeddings.numpy()}
    if logmap_pose is not None:
        arrays["logmap_pose"] = torch.as_tensor(logmap_pose).detach().cpu().float().numpy()
    tmp = path + ".tmp.npz"
    np.savez_compressed(tmp, **arrays)
    os.replace(tmp, path)


def load_codebook(path: str) -> dict:
    """Arrays of a `codebook.npz` (or, by extension .pkl, of a reference pickle) as CPU tensors."""
    if path.endswith(".pkl"):
        return read_reference_pickle(path)
    with np.load(path) as z:
        ver = int(z["format_version"]) if "format_version" in z.files else 0
        if ver > FORMAT_VERSION:
            raise MidasError(f"{path}: codebook format {ver} is newer than this build reads ({FORMAT_VERSION})")
        out = {k: torch.as_tensor(z[k]) for k in ("poses", "cam_poses", "embeddings", "logmap_pose") if k in z.files}
    missing = [k for k in ("poses", "cam_poses", "embeddings") if k not in out]
    if missing:
        raise MidasError(f"{path}: arrays {missing} missing")
    _check_shapes(out["poses"], out["cam_poses"], out["embeddings"])
    return out


def convert(src: str, dst: str) -> dict:
    """Reference `codebook.pkl` -> `codebook.npz`; returns a small summary."""
    cb = read_reference_pickle(src)
    save_codebook(dst, cb["poses"], cb["cam_poses"], cb["embeddings"], cb.get("logmap_pose"))
    back = load_codebook(dst)
    return {"K": int(back["poses"].shape[0]), "D": int(back["embeddings"].shape[1]),
            "embeddings_dtype": str(back["embeddings"].dtype).replace("torch.", ""),
            "has_reference_logmap": "logmap_pose" in back, "bytes": os.path.getsize(dst)}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) == 3 and argv[0] == "convert":
        print(convert(argv[1], argv[2]))
        return 0
    if len(argv) == 2 and argv[0] == "info":
        cb = load_codebook(argv[1])
        print({k: (tuple(v.shape), str(v.dtype)) for k, v in cb.items()})
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    raise SystemExit(main())
