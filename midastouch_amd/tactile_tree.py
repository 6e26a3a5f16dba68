"""tactile_tree: the per-object codebook (K surface poses + K embeddings + exact 6-d NN index).

Drop-in counterpart of the reference class `midastouch/tactile_tree/tactile_tree.py:13-70` and of the
free function `R3_SE3` (:73-77), same names, arguments and return shapes.  Differences that are
deliberate and invisible to the runner:

* the KD-tree lives on the GPU (libmidas_hip.so) instead of pynanoflann on the host, so `SE3_NN`
  never leaves the device;
* embeddings are kept as float32 when that is lossless (they are float32 network outputs cast to
  float64, `contrib/tcn_minkloc/tcn.py:148`), float64 otherwise;
* the third element `SE3_NN` returns is an `NNCodes` view (indices into the codebook) instead of a
  materialised (N, D) float64 gather - `particle_filter.get_similarity` consumes it without ever
  building that matrix; `.to_tensor()` materialises it for any other consumer.

There is no CPU fallback: the object can be constructed from CPU tensors (as when the reference
unpickles it) but queries need `to_device(<hip device>)` first.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import MidasError


def R3_SE3(poses: torch.Tensor, w: float = 0.01) -> torch.Tensor:
    """[(1-w) t, w log(R)] of (N,4,4) poses -> (N,6) float32 (reference tactile_tree.py:73-77)."""
    return ops.se3_feature(poses, w)


class EmbeddingMatrix:
    """The whole (K, D) embedding matrix of a codebook, as returned by `get_embeddings()`."""

    def __init__(self, tree: "tactile_tree"):
        self.tree = tree

    @property
    def shape(self):
        return (self.tree.tree_size, self.tree.embeddings.shape[1])

    @property
    def device(self):
        return self.tree.embeddings.device

    dtype = torch.float64

    def __len__(self):
        return self.tree.tree_size

    def to_tensor(self) -> torch.Tensor:
        return self.tree.embeddings.double()

    def __getitem__(self, item):
        return self.tree.embeddings[item].double()

    def cpu(self):
        return self.to_tensor().cpu()


class NNCodes:
    """Rows `idx` of a codebook's embedding matrix without materialising them (SE3_NN's third output)."""

    def __init__(self, tree: "tactile_tree", idx: torch.Tensor):
        self.tree, self.idx = tree, idx

    @property
    def shape(self):
        return (self.idx.shape[0], self.tree.embeddings.shape[1])

    @property
    def device(self):
        return self.idx.device

    dtype = torch.float64

    def __len__(self):
        return self.idx.shape[0]

    def to_tensor(self) -> torch.Tensor:
        """The (N, D) float64 gather the reference performs at tactile_tree.py:57."""
        return ops.gather_rows(self.tree.embeddings, self.idx).double()

    def __getitem__(self, item):
        return self.to_tensor()[item]

    def cpu(self):
        return self.to_tensor().cpu()


class tactile_tree:
    def __init__(self, poses, cam_poses, embeddings):
        self.poses = torch.as_tensor(poses).float()
        self.cam_poses = torch.as_tensor(cam_poses).float()
        self.embeddings = torch.as_tensor(embeddings)
        self.embedding_dtype = self.embeddings.dtype  # what the caller stored: the dtype gathers are returned in (:54-58)
        self.tree_size = self.poses.shape[0]
        self.logmap_pose = None
        self.tree, self._codebook = None, None
        if self.poses.is_cuda:
            self.init_tree()

    def __len__(self):
        return self.tree_size

    def __repr__(self):
        return "tactile Tree of size: {}".format(len(self))

    # -- device state ------------------------------------------------------------------------------
    def to_device(self, device):
        device = torch.device(device)
        self.poses = self.poses.to(device)
        self.cam_poses = self.cam_poses.to(device)
        self.embeddings = self.embeddings.to(device)
        if device.type == "cuda":
            self.init_tree()
        else:
            self.logmap_pose = None if self.logmap_pose is None else self.logmap_pose.to(device)
            self.tree, self._codebook = None, None

    def init_tree(self):
        """6-d features + static KD-tree + embedding norms on the GPU (reference :34-41 builds pynanoflann)."""
        if not self.poses.is_cuda:
            raise MidasError("tactile_tree.init_tree needs the codebook on a HIP device (call to_device first); "
                             "there is no CPU fallback")
        self.poses = self.poses.contiguous()
        self.cam_poses = self.cam_poses.contiguous()
        self.logmap_pose = R3_SE3(self.poses)
        self.tree = ops.Tree(self.logmap_pose)
        self._codebook = ops.Codebook(self.embeddings)
        self.embeddings = self._codebook.emb  # float32 when lossless
        self.tree_size = self.poses.shape[0]

    def _require_tree(self):
        if self.tree is None:
            raise MidasError("tactile_tree is not on a HIP device: call to_device('cuda') before querying; "
                             "there is no CPU fallback")

    @property
    def codebook(self) -> ops.Codebook:
        self._require_tree()
        return self._codebook

    # -- queries -----------------------------------------------------------------------------------
    def SE3_NN_idx(self, _query, hint=None) -> torch.Tensor:
        """Index (int32, on the device) of the nearest codebook pose of each query pose."""
        self._require_tree()
        query = torch.as_tensor(_query).to(self.poses.device)
        query = query[None] if query.dim() == 2 else query
        return ops.nn6(self.tree, R3_SE3(query), hint=hint)

    def SE3_NN(self, _query, nn=1):
        """Best SE(3) matches by R3 + log-map distance (reference :43-58): (poses, cam_poses, embeddings).  nn = 1 (what the
        filter asks for): (N,4,4), (N,4,4) and a lazy (N,D) view; nn > 1: the nn nearest per query in order, as the reference
        returns them - (N,nn,4,4), (N,nn,4,4), (N,nn,D) in the dtype the embeddings were stored with (float64 in the reference's
        codebooks; squeezed for a single query, as the reference's indexing does).  nn <= 64."""
        if nn == 1:
            idx = self.SE3_NN_idx(_query)
            return ops.gather_rows(self.poses, idx), ops.gather_rows(self.cam_poses, idx), NNCodes(self, idx)
        self._require_tree()
        query = torch.as_tensor(_query).to(self.poses.device)
        query = query[None] if query.dim() == 2 else query
        idx = ops.knn6(self.tree, R3_SE3(query), int(nn))  # (N, nn)
        flat = idx.reshape(-1)
        shape = (idx.shape[0], int(nn)) if idx.shape[0] > 1 else (int(nn),)
        return (ops.gather_rows(self.poses, flat).view(*shape, 4, 4), ops.gather_rows(self.cam_poses, flat).view(*shape, 4, 4),
                ops.gather_rows(self.embeddings, flat).to(self.embedding_dtype).view(*shape, -1))

    def get_poses(self):
        return self.poses, self.cam_poses

    def get_pose(self, idx):
        return self.poses[idx, :]

    def get_embeddings(self):
        return EmbeddingMatrix(self)

    def get_embedding(self, idx):
        return self.embeddings[idx, :].to(self.embedding_dtype)

    # -- on-disk container (replaces the dill pickle of build_codebook.py:136-137) --------
    def save(self, path: str):
        """Write `codebook.npz` (midastouch_amd/codebook_io.py)."""
        from . import codebook_io

        codebook_io.save_codebook(path, self.poses, self.cam_poses, self.embeddings, self.reference_logmap)

    @classmethod
    def load(cls, path: str, device=None, check_logmap: bool = False) -> "tactile_tree":
        """Open a `codebook.npz`, or a reference `codebook.pkl` (read without the reference package or pynanoflann).

        check_logmap: after `to_device`, compare the kernels' 6-d features with the ones the reference stored in the
        file (theseus log-map) and raise when they differ by more than 1e-5 - the two log-maps are unpinned against
        each other otherwise."""
        from . import codebook_io

        z = codebook_io.load_codebook(path)
        t = cls(z["poses"], z["cam_poses"], z["embeddings"])
        t.reference_logmap = z.get("logmap_pose")
        if device is not None:
            t.to_device(device)
            if check_logmap:
                t.check_reference_logmap()
        elif check_logmap:
            raise MidasError("check_logmap needs a HIP device: pass device=")
        return t

    reference_logmap = None

    def check_reference_logmap(self, atol: float = 1e-5) -> float:
        """max |R3_SE3(poses) - stored reference features| (rotation part compared modulo the sign at angle pi)."""
        self._require_tree()
        if self.reference_logmap is None:
            raise MidasError("the codebook file holds no reference logmap_pose")
        ref = self.reference_logmap.to(self.logmap_pose.device, torch.float32)
        d = (self.logmap_pose - ref).abs()
        # at |log R| = pi the axis sign is a convention: accept the mirrored rotation vector there
        flip = (self.logmap_pose[:, 3:] + ref[:, 3:]).abs()
        near_pi = self.logmap_pose[:, 3:].norm(dim=1) > 0.01 * 3.13
        d[:, 3:] = torch.where(near_pi[:, None], torch.minimum(d[:, 3:], flip), d[:, 3:])
        err = float(d.max())
        if not err <= atol:
            raise MidasError(f"6-d features differ from the reference's stored logmap_pose by {err:.3e} (> {atol})")
        return err

    def __getstate__(self):  # handles are rebuilt after unpickling + to_device
        st = dict(self.__dict__)
        st["tree"], st["_codebook"] = None, None
        for k in ("poses", "cam_poses", "embeddings", "logmap_pose"):
            if st.get(k) is not None:
                st[k] = st[k].cpu()
        return st
