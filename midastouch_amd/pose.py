"""SE(3) helpers used by the low-rate parts of the filter (cluster centres), torch ops on the device.

Counterparts of `midastouch/modules/pose.py`: `log_map_averaged` (:101-109), `tf_to_xyzquat` (:26-34).
`xyz_quat_averaged` (:112-147), the one the filter loop uses, is the K9 kernel (`ops.cluster_centers`).
"""
from __future__ import annotations

import torch

from . import ops


def rotvec(poses: torch.Tensor) -> torch.Tensor:
    """SO(3) log-map of (N,4,4) poses via the K3 kernel (w = 1 puts log(R) unscaled in columns 3:6)."""
    return ops.se3_feature(poses, 1.0)[:, 3:]


def tf_to_xyzquat(poses: torch.Tensor) -> torch.Tensor:
    """(N,4,4) -> [x, y, z, qw, qx, qy, qz]."""
    poses = poses[None] if poses.dim() == 2 else poses
    w = rotvec(poses).double()
    th = w.norm(dim=1, keepdim=True)
    half = 0.5 * th
    k = torch.where(th > 1e-8, torch.sin(half) / th.clamp_min(1e-30), 0.5 - th * th / 48.0)
    q = torch.cat((torch.cos(half), w * k), dim=1)
    return torch.cat((poses[:, :3, 3].double(), q), dim=1)


def _hat(v: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), dtype=v.dtype, device=v.device)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_log(T: torch.Tensor) -> torch.Tensor:
    """(N,4,4) -> (N,6) float64 tangent vectors [V^-1 t, omega] - theseus `SE3.log_map` (used at pose.py:105-106 and
    by `cluster_particles(method="logmap")`, particle_filter.py:218-223)."""
    T = T[None] if T.dim() == 2 else T
    om = rotvec(T).double()
    th = om.norm(dim=1)
    t = T[:, :3, 3].double()
    # V^-1 t with V^-1 = I - 1/2 [w]x + c [w]x^2
    c = torch.where(th > 1e-6, (1.0 - 0.5 * th * torch.cos(0.5 * th) / torch.sin(0.5 * th).clamp_min(1e-30)) / (th * th).clamp_min(1e-30),
                    torch.full_like(th, 1.0 / 12.0))
    wxt = torch.cross(om, t, dim=1)
    u = t - 0.5 * wxt + c[:, None] * torch.cross(om, wxt, dim=1)
    return torch.cat((u, om), dim=1)


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    """(C,6) float64 tangent vectors [u, omega] -> (C,4,4) float64 poses - theseus `SE3.exp_map` (pose.py:107), batched."""
    xi = xi[None] if xi.dim() == 1 else xi
    u, om = xi[:, :3], xi[:, 3:]
    th = om.norm(dim=1)
    small = th <= 1e-8
    ths = torch.where(small, torch.ones_like(th), th)
    A = torch.where(small, torch.ones_like(th), torch.sin(ths) / ths)
    B = torch.where(small, torch.full_like(th, 0.5), (1.0 - torch.cos(ths)) / (ths * ths))
    Cc = torch.where(small, torch.full_like(th, 1.0 / 6.0), (ths - torch.sin(ths)) / (ths * ths * ths))
    z = torch.zeros_like(th)
    W = torch.stack([torch.stack([z, -om[:, 2], om[:, 1]], 1), torch.stack([om[:, 2], z, -om[:, 0]], 1),
                     torch.stack([-om[:, 1], om[:, 0], z], 1)], 1)  # (C,3,3) hat(omega)
    W2 = W @ W
    eye = torch.eye(3, dtype=xi.dtype, device=xi.device)[None]
    out = torch.zeros((xi.shape[0], 4, 4), dtype=xi.dtype, device=xi.device)
    out[:, :3, :3] = eye + A[:, None, None] * W + B[:, None, None] * W2
    out[:, :3, 3] = ((eye + B[:, None, None] * W + Cc[:, None, None] * W2) @ u[:, :, None])[:, :, 0]
    out[:, 3, 3] = 1.0
    return out


def logmap_average_pose(T: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Weighted mean in the se(3) tangent space, mapped back with exp (pose.py:101-109); one set of poses."""
    wd = w.double()
    xi = (se3_log(T) * wd[:, None]).sum(dim=0) / wd.sum()
    return se3_exp(xi[None])[0].float()


def logmap_cluster_centers(poses: torch.Tensor, weights: torch.Tensor, labels: torch.Tensor):
    """`get_cluster_centers(method="logmap")` (modules/particle_filter.py:153-206 with pose.log_map_averaged, pose.py:101-109)
    for ALL clusters at once: the SE(3) logarithms of the N particles in one pass (rotation vectors from the K3 kernel), every
    cluster's sums - sum w xi, sum w, sum w t, sum w t^2, with the cluster's weights flattened to 1 where isclose(max - min, 0)
    (:178-184) - as ONE float64 matrix product of the (C, N) membership matrix with the (N, 13) per-particle terms (a fixed
    reduction order: the same bits every run, unlike atomics), the C exponentials batched.  The reference walks the clusters
    in a Python loop of ~30 small ops each.  Returns (cluster_poses (C,4,4) f32, cluster_stds (C,3) f32), clusters in the
    order of torch.unique(labels)."""
    N = poses.shape[0]
    uniq, inv = torch.unique(labels, return_inverse=True)
    C = uniq.shape[0]
    w32 = weights.float()  # the reference averages with float32 weights (:161)
    wmax = torch.full((C,), -float("inf"), device=poses.device).scatter_reduce(0, inv, w32, "amax")
    wmin = torch.full((C,), float("inf"), device=poses.device).scatter_reduce(0, inv, w32, "amin")
    flat = torch.isclose(wmax - wmin, torch.zeros_like(wmax))
    w = torch.where(flat[inv], torch.ones_like(w32), w32).double()
    xi = se3_log(poses)
    t = poses[:, :3, 3].double()
    terms = torch.cat((w[:, None] * xi, w[:, None], w[:, None] * t, w[:, None] * t * t), dim=1)  # (N, 13)
    member = torch.zeros((C, N), dtype=torch.float64, device=poses.device)
    member[inv, torch.arange(N, device=poses.device)] = 1.0
    S = member @ terms
    sw = S[:, 6:7]
    centres = se3_exp(S[:, :6] / sw)
    c = centres[:, :3, 3]
    # sum w (t - c)^2 / sum w around the centre's translation (:195-204), from the moments
    var = (S[:, 10:13] - 2.0 * c * S[:, 7:10] + c * c * sw) / sw
    return centres.float(), torch.sqrt(var.clamp_min(0.0)).float()
