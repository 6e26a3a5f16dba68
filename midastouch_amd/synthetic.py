"""Synthetic objects, codebooks and trajectories (host-side, numpy).

No real YCB / McMaster assets ship with the reference checkout (meshes, codebooks and
YCB-Slide logs are external downloads: reference `download_assets.sh:8-17`), so every
configuration of BASELINE.json is driven by data built here.  The shapes and statistics
follow what the reference would feed the filter:

* object = analytic box with the extents of the named YCB model; only `mesh.scale`
  (reference `modules/particle_filter.py:124-127,147-151`) and the surface matter;
* codebook = K sensor poses sampled area-uniformly on the surface with the sensor z-axis
  along the inward normal, a uniform yaw and a <= 5 deg shear cone (what
  `modules/pose.py:375-455` / `modules/mesh.py:126-135` produce), plus unit-norm
  float32 embeddings (the TCN emits float32 codes cast to float64,
  `contrib/tcn_minkloc/tcn.py:148`);
* trajectory = 0.25 mm/step surface walk with the data-gen pose noise
  (`data_gen/config/method/ycb_slide.yaml:10-15`).

Everything is deterministic in the seed.  This module is plain numpy/scipy: it prepares
inputs, it is not on the timed path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.spatial.transform import Rotation

# extents in metres (x, y, z)
OBJECT_EXTENTS = {
    "004_sugar_box": (0.038, 0.089, 0.175),
    "035_power_drill": (0.18, 0.06, 0.19),
    "025_mug": (0.09, 0.09, 0.08),
    "cotter-pin": (0.04, 0.008, 0.008),
}


def mesh_scale(extents) -> float:
    """trimesh `mesh.scale` = length of the bounding-box diagonal."""
    return float(np.linalg.norm(np.asarray(extents, dtype=np.float64)))


def box_surface_samples(extents, n: int, rng: np.random.Generator):
    """Area-uniform points on the surface of an axis-aligned box centred at the origin.

    Returns (points (n,3) f64, outward normals (n,3) f64).
    """
    ex = np.asarray(extents, dtype=np.float64)
    # six faces: axis a, sign s ; area = product of the two other extents
    faces = [(a, s) for a in range(3) for s in (-1.0, 1.0)]
    areas = np.array([ex[(a + 1) % 3] * ex[(a + 2) % 3] for a, _ in faces])
    face_id = rng.choice(6, size=n, p=areas / areas.sum())
    uv = rng.uniform(-0.5, 0.5, size=(n, 2))
    pts = np.zeros((n, 3))
    nrm = np.zeros((n, 3))
    for f, (a, s) in enumerate(faces):
        m = face_id == f
        b, c = (a + 1) % 3, (a + 2) % 3
        pts[m, a] = 0.5 * s * ex[a]
        pts[m, b] = uv[m, 0] * ex[b]
        pts[m, c] = uv[m, 1] * ex[c]
        nrm[m, a] = s
    return pts, nrm


def _rot_z_to(v: np.ndarray) -> np.ndarray:
    """Rotation matrices (n,3,3) taking +z onto the unit vectors v (n,3)."""
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(np.broadcast_to(z, v.shape), v)
    s = np.linalg.norm(axis, axis=1)
    c = v @ z
    rotvec = np.zeros_like(v)
    ok = s > 1e-12
    rotvec[ok] = axis[ok] / s[ok, None] * np.arctan2(s[ok], c[ok])[:, None]
    flip = (~ok) & (c < 0)
    rotvec[flip] = np.array([np.pi, 0.0, 0.0])
    return Rotation.from_rotvec(rotvec).as_matrix()


def poses_from_surface(points, normals, rng: np.random.Generator, shear_deg: float = 5.0):
    """Sensor poses (n,4,4) f32: z-axis = inward normal tilted inside a shear cone, random yaw."""
    n = points.shape[0]
    cos_mag = rng.uniform(np.cos(np.deg2rad(shear_deg)), 1.0, size=n)
    phi = rng.uniform(0.0, 2 * np.pi, size=n)
    sin_mag = np.sqrt(1.0 - cos_mag**2)
    shear = np.stack([sin_mag * np.cos(phi), sin_mag * np.sin(phi), cos_mag], axis=1)
    yaw = rng.uniform(0.0, 2 * np.pi, size=n)
    R_align = _rot_z_to(-normals)  # +z -> inward normal
    R_shear = _rot_z_to(shear)  # tilt inside the cone (sensor frame)
    R_yaw = Rotation.from_euler("z", yaw).as_matrix()
    R = R_align @ R_shear @ R_yaw
    T = np.zeros((n, 4, 4), dtype=np.float64)
    T[:, :3, :3] = R
    T[:, :3, 3] = points
    T[:, 3, 3] = 1.0
    return T.astype(np.float32)


def r3_se3_host(poses: np.ndarray, w: float = 0.01) -> np.ndarray:
    """Host-side 6-d pose feature [ (1-w) t , w log(R) ] via scipy (data generation only)."""
    poses = np.asarray(poses, dtype=np.float64)
    rv = Rotation.from_matrix(poses[:, :3, :3]).as_rotvec()
    return np.concatenate([(1.0 - w) * poses[:, :3, 3], w * rv], axis=1)


@dataclass
class SyntheticCodebook:
    obj_model: str
    extents: tuple
    poses: np.ndarray  # (K,4,4) f32
    cam_poses: np.ndarray  # (K,4,4) f32
    embeddings: np.ndarray  # (K,D) f32, unit rows
    mesh_vertices: np.ndarray  # (M,3) f64 (float32-representable, as an STL's are)

    @property
    def K(self) -> int:
        return self.poses.shape[0]

    @property
    def D(self) -> int:
        return self.embeddings.shape[1]


def make_codebook(
    obj_model: str = "004_sugar_box",
    K: int = 5000,
    D: int = 256,
    seed: int = 1000,
    mode: str = "rff",
    mesh_points: int | None = None,
    cam_dist: float = 0.022,
) -> SyntheticCodebook:
    """Build a synthetic codebook for `obj_model`.

    mode "rff": embeddings are normalised random Fourier features of the 6-d pose feature
    plus 10 % noise, so neighbouring poses have correlated codes (as a trained TCN's do);
    mode "iid": normalised gaussian rows.
    """
    extents = OBJECT_EXTENTS[obj_model]
    rng = np.random.default_rng(seed)
    pts, nrm = box_surface_samples(extents, K, rng)
    poses = poses_from_surface(pts, nrm, rng)
    cam = poses.copy().astype(np.float64)
    # camera sits cam_dist behind the gel along the sensor z axis (tdn/default.yaml:13)
    cam[:, :3, 3] -= cam_dist * cam[:, :3, 2]
    cam_poses = cam.astype(np.float32)
    if mode == "rff":
        feat = r3_se3_host(poses)
        W = rng.standard_normal((6, D)) * 60.0
        b = rng.uniform(0, 2 * np.pi, size=D)
        E = np.cos(feat @ W + b) + 0.1 * rng.standard_normal((K, D))
    elif mode == "iid":
        E = rng.standard_normal((K, D))
    else:
        raise ValueError(f"unknown embedding mode {mode!r}")
    E = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float32)
    M = mesh_points if mesh_points is not None else max(K // 2, 16)
    mv, _ = box_surface_samples(extents, M, np.random.default_rng(seed + 7))
    mesh_vertices = mv.astype(np.float32).astype(np.float64)
    return SyntheticCodebook(obj_model, extents, poses, cam_poses, E, mesh_vertices)


@dataclass
class SyntheticTrajectory:
    gt_poses: np.ndarray  # (T,4,4) f32
    meas_poses: np.ndarray  # (T,4,4) f32
    odoms: np.ndarray  # (T,4,4) f32 ; odoms[0] = identity
    codes: np.ndarray  # (T,D) f64 unit rows (float32 values)
    gt_index: np.ndarray  # (T,) int64 codebook entry under the sensor


def make_trajectory(cb: SyntheticCodebook, T: int = 200, seed: int = 2000,
                    step_m: float = 0.25e-3, code_noise: float = 0.05) -> SyntheticTrajectory:
    """Continuous 0.25 mm/step walk of the sensor over the largest face of the box.

    gt pose: z-axis along the inward normal, slowly drifting yaw; measured pose = gt with the
    data-gen noise; odom_t = inv(meas_{t-1}) meas_t; tactile code at t = embedding of the
    codebook entry nearest (6-d feature) to gt_t plus noise, renormalised, float32 values in f64.
    """
    rng = np.random.default_rng(seed)
    ex = np.asarray(cb.extents, dtype=np.float64)
    a = int(np.argmin(ex))  # the largest face is normal to the thinnest axis
    b, c = (a + 1) % 3, (a + 2) % 3
    pos = np.zeros((T, 3))
    uv = rng.uniform(-0.25, 0.25, size=2) * ex[[b, c]]
    heading = rng.uniform(0, 2 * np.pi)
    yaw = np.empty(T)
    yaw_t = rng.uniform(0, 2 * np.pi)
    for t in range(T):
        pos[t, a], pos[t, b], pos[t, c] = 0.5 * ex[a], uv[0], uv[1]
        yaw[t] = yaw_t
        heading += rng.standard_normal() * 0.05
        uv = uv + step_m * np.array([np.cos(heading), np.sin(heading)])
        for j, ax in enumerate((b, c)):  # reflect at the face boundary
            lim = 0.45 * ex[ax]
            if abs(uv[j]) > lim:
                uv[j] = np.sign(uv[j]) * (2 * lim - abs(uv[j]))
                heading += np.pi / 2
        yaw_t += np.deg2rad(0.5) * rng.standard_normal()
    nrm = np.zeros((T, 3))
    nrm[:, a] = 1.0
    R = _rot_z_to(-nrm) @ Rotation.from_euler("z", yaw).as_matrix()
    gt = np.zeros((T, 4, 4))
    gt[:, :3, :3], gt[:, :3, 3], gt[:, 3, 3] = R, pos, 1.0
    # measured = gt * T(N(0,1deg), N(0,5e-4 m))   (ycb_slide.yaml:10-12)
    rot_n = Rotation.from_euler("zyx", rng.standard_normal((T, 3)) * 1.0, degrees=True).as_matrix()
    tn = rng.standard_normal((T, 3)) * 5e-4
    Tn = np.zeros((T, 4, 4))
    Tn[:, :3, :3], Tn[:, :3, 3], Tn[:, 3, 3] = rot_n, tn, 1.0
    meas = gt @ Tn
    odoms = np.zeros((T, 4, 4))
    odoms[0] = np.eye(4)
    odoms[1:] = np.linalg.inv(meas[:-1]) @ meas[1:]
    feat_cb = r3_se3_host(cb.poses).astype(np.float32)
    feat_gt = r3_se3_host(gt).astype(np.float32)
    gt_idx = np.empty(T, dtype=np.int64)
    for t in range(T):
        gt_idx[t] = int(np.argmin(((feat_cb - feat_gt[t]) ** 2).sum(axis=1)))
    codes = cb.embeddings[gt_idx].astype(np.float64) + code_noise * rng.standard_normal((T, cb.D))
    codes = (codes / np.linalg.norm(codes, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    return SyntheticTrajectory(gt.astype(np.float32), meas.astype(np.float32),
                               odoms.astype(np.float32), codes, gt_idx)


def wide_start(extents, gt0, N: int, seed: int) -> np.ndarray:
    """`particle_filter.init_filter(gt_0, N)` of the reference (modules/particle_filter.py:124-145): gt_0 composed with
    N(0, mesh scale / 3) translations and N(0, 60 deg) "zyx" Euler angles - the wide start SURVEY.md 8(d) prescribes for
    the measurement (the caller projects it onto the codebook, filter/filter.py:159-160).  (N,4,4) float32; draws from a
    torch CPU generator seeded with `seed`, translations first, as the reference draws them."""
    import torch

    g = torch.Generator().manual_seed(int(seed))
    tn0 = torch.normal(0.0, mesh_scale(extents) / 3.0, size=(N, 3), generator=g)
    rn0 = torch.normal(0.0, 60.0, size=(N, 3), generator=g)
    Tn = torch.zeros((N, 4, 4))
    Tn[:, :3, :3] = torch.as_tensor(Rotation.from_euler("zyx", rn0.numpy(), degrees=True).as_matrix()).float()
    Tn[:, :3, 3], Tn[:, 3, 3] = tn0, 1.0
    return (torch.as_tensor(np.asarray(gt0, dtype=np.float32))[None] @ Tn).numpy()
