"""SE(3) helpers used by the low-rate parts of the filter (cluster centres), torch ops on the device.

Counterparts of `midastouch/modules/pose.py`: `xyz_quat_averaged` (:112-147), `log_map_averaged`
(:101-109), `tf_to_xyzquat` (:26-34).  These run once per frame over a handful of clusters; they are
not on the accelerated per-particle path.
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


def _quat_to_matrix(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = (q / q.norm()).unbind()
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)]),
        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]),
    ])


def quat_average_pose(T: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Markley weighted quaternion mean + weighted mean translation -> (4,4) float32."""
    xq = tf_to_xyzquat(T)
    q = xq[:, 3:]
    q = torch.where(q[:, :1] < 0, -q, q)  # antipodal fix (pose.py:126)
    wd = w.double()
    M = (q[:, :, None] * q[:, None, :] * wd[:, None, None]).sum(dim=0) / wd.sum()
    evals, evecs = torch.linalg.eigh(M.cpu())
    avg_q = evecs[:, -1].to(T.device)
    if avg_q[0] < 0:
        avg_q = -avg_q
    avg_t = (xq[:, :3] * wd[:, None]).sum(dim=0) / wd.sum()
    out = torch.eye(4, dtype=torch.float64, device=T.device)
    out[:3, :3] = _quat_to_matrix(avg_q)
    out[:3, 3] = avg_t
    return out.float()


def _hat(v: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), dtype=v.dtype, device=v.device)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def logmap_average_pose(T: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Weighted mean in the se(3) tangent space, mapped back with exp (pose.py:101-109)."""
    wd = w.double()
    om = rotvec(T).double()
    th = om.norm(dim=1)
    t = T[:, :3, 3].double()
    # V^-1 t with V^-1 = I - 1/2 [w]x + c [w]x^2
    c = torch.where(th > 1e-6, (1.0 - 0.5 * th * torch.cos(0.5 * th) / torch.sin(0.5 * th).clamp_min(1e-30)) / (th * th).clamp_min(1e-30),
                    torch.full_like(th, 1.0 / 12.0))
    wxt = torch.cross(om, t, dim=1)
    u = t - 0.5 * wxt + c[:, None] * torch.cross(om, wxt, dim=1)
    xi = (torch.cat((u, om), dim=1) * wd[:, None]).sum(dim=0) / wd.sum()
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
