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


def logmap_average_pose(T: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Weighted mean in the se(3) tangent space, mapped back with exp (pose.py:101-109)."""
    wd = w.double()
    xi = (se3_log(T) * wd[:, None]).sum(dim=0) / wd.sum()
    u_m, w_m = xi[:3], xi[3:]
    thm = w_m.norm()
    W = _hat(w_m)
    if thm > 1e-8:
        A, B, C = torch.sin(thm) / thm, (1 - torch.cos(thm)) / thm**2, (thm - torch.sin(thm)) / thm**3
    else:
        A, B, C = 1.0, 0.5, 1.0 / 6.0
    eye = torch.eye(3, dtype=torch.float64, device=T.device)
    out = torch.eye(4, dtype=torch.float64, device=T.device)
    out[:3, :3] = eye + A * W + B * (W @ W)
    out[:3, 3] = (eye + B * W + C * (W @ W)) @ u_m
    return out.float()
