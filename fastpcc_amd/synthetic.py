"""Seeded synthetic voxel clouds for benchmarks and tests (no datasets are available): SURVEY.md section 8d.

`body_cloud` is the stand-in for an 8iVFB frame (cfg#2): a closed, one-voxel-thick surface made of a torso, a head and
two limbs (ellipsoids), voxelised by dense parametric sampling; `scale` is tuned so that the named resolutions give the
named voxel counts.  `surface_cloud` is the small 'ShapeNet-like' plumbing cloud (cfg#1); `enliven` gives a model the
seeded weights every benchmark and parity test uses (no checkpoints exist here)."""
from typing import Tuple

import numpy as np

# (centre, radii) in units of the resolution
_BODY = (
    ((0.50, 0.50, 0.45), (0.16, 0.10, 0.26)),    # torso
    ((0.50, 0.50, 0.80), (0.075, 0.08, 0.09)),   # head
    ((0.36, 0.50, 0.30), (0.05, 0.055, 0.28)),   # limb
    ((0.64, 0.50, 0.30), (0.05, 0.055, 0.28)),   # limb
)


def _ellipsoid_surface(rng, centre, radii, n):
    # area-uniform enough for voxelisation: sample the sphere, stretch, oversample
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return centre + d * radii


def body_cloud(resolution: int = 1024, scale: float = 1.0, seed: int = 2, samples_per_voxel: float = 7.0) -> np.ndarray:
    """unique int32 voxels [n, 3] of the union surface, inside [0, resolution)^3"""
    rng = np.random.default_rng(seed)
    parts = []
    ells = [(np.array(c) * resolution, np.array(r) * resolution * scale) for c, r in _BODY]
    for centre, radii in ells:
        a, b, c = radii
        area = 4 * np.pi * (((a * b) ** 1.6 + (a * c) ** 1.6 + (b * c) ** 1.6) / 3) ** (1 / 1.6)
        pts = _ellipsoid_surface(rng, centre, radii, int(area * samples_per_voxel))
        keep = np.ones(len(pts), bool)
        for c2, r2 in ells:                       # drop what lies strictly inside another part: union surface
            if c2 is centre:
                continue
            keep &= (((pts - c2) / r2) ** 2).sum(1) >= 1.0
        parts.append(pts[keep])
    p = np.round(np.concatenate(parts)).astype(np.int32)
    p = p[((p >= 0) & (p < resolution)).all(1)]
    # unique via a packed key (faster than np.unique(axis=0))
    key = (p[:, 0].astype(np.int64) << 42) | (p[:, 1].astype(np.int64) << 21) | p[:, 2].astype(np.int64)
    key = np.unique(key)
    return np.stack(((key >> 42), (key >> 21) & 0x1fffff, key & 0x1fffff), 1).astype(np.int32)


# scale factors calibrated so that body_cloud(res, SCALE[res]) has the voxel count BASELINE.json names (+-1 %)
SCALE = {1024: 1.25, 2048: 0.735}      # 997 645 voxels at 10 bit (cfg#2), 1 997 208 at 11 bit (cfg#4)


def batched(xyz: np.ndarray, batch: int = 0) -> np.ndarray:
    return np.concatenate((np.full((len(xyz), 1), batch, dtype=np.int32), xyz.astype(np.int32)), 1)


def lidar_cloud(seed: int = 3, beams: int = 64, azimuths: int = 2048, resolution: int = 65536, extent: float = 400.0,
                drop: float = 0.1) -> np.ndarray:
    """KITTI-like spinning-LiDAR frame (cfg#3): rays cast against a ground plane and random boxes, range noise, dropped
    returns, quantised as the reference's KITTI loader does (round((p - min) * (resolution - 1) / extent),
    lib/datasets/KITTIOdometry/dataset.py:91-102).  Returns unique int32 voxels [n, 3]."""
    rng = np.random.default_rng(seed)
    az = np.linspace(0, 2 * np.pi, azimuths, endpoint=False)
    el = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    a, e = np.meshgrid(az, el)
    d = np.stack((np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)), -1).reshape(-1, 3)
    t = np.full(len(d), np.inf)
    down = d[:, 2] < -1e-3
    t[down] = -1.73 / d[down, 2]                                   # ground plane z = -1.73 m
    for _ in range(40):                                            # axis-aligned boxes
        c = np.array([rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(-1.7, 0.5)])
        h = np.array([rng.uniform(0.5, 6), rng.uniform(0.5, 6), rng.uniform(0.5, 4)])
        with np.errstate(divide='ignore', invalid='ignore'):
            t1, t2 = (c - h) / d, (c + h) / d
        near, far = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
        hit = (near <= far) & (near > 0.5)
        t = np.where(hit & (near < t), near, t)
    ok = np.isfinite(t) & (t < 120)
    ok &= rng.random(len(t)) >= drop
    p = d[ok] * (t[ok] + rng.normal(0, 0.02, ok.sum()))[:, None]
    q = np.round((p - p.min(0)) * ((resolution - 1) / extent)).astype(np.int64)
    key = np.unique((q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2])
    return np.stack(((key >> 42), (key >> 21) & 0x1fffff, key & 0x1fffff), 1).astype(np.int32)


def surface_cloud(seed: int, resolution: int, n_samples: int) -> np.ndarray:
    """Unique int voxels [n, 3] on a union of ellipsoid shells and planes inside [0, resolution)^3 -- the
    'ShapeNet-like' plumbing cloud (cfg#1)."""
    rng = np.random.default_rng(seed)
    pts = []
    per = n_samples // 5
    for _ in range(3):
        centre = rng.uniform(0.3, 0.7, 3) * resolution
        radii = rng.uniform(0.12, 0.3, 3) * resolution
        d = rng.normal(size=(per, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts.append(centre + d * radii)
    for _ in range(2):
        origin = rng.uniform(0.2, 0.8, 3) * resolution
        u, v = rng.normal(size=3), rng.normal(size=3)
        u /= np.linalg.norm(u)
        v -= u * (u @ v)
        v /= np.linalg.norm(v)
        ab = rng.uniform(-0.4, 0.4, (per, 2)) * resolution
        pts.append(origin + ab[:, :1] * u + ab[:, 1:] * v)
    p = np.round(np.concatenate(pts)).astype(np.int64)
    p = p[((p >= 0) & (p < resolution)).all(1)]
    return np.unique(p, axis=0)


def enliven(model: 'torch.nn.Module', seed: int, gain: float = 2.35) -> None:
    """Seeded re-initialisation that keeps activations O(1) through the 12-level pyramid (the default
    U(-1/sqrt(fan), 1/sqrt(fan)) init shrinks them to zero, which would make every parity test trivial)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    g_prior = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if '.prior_' in name:                       # deep-factorised prior: keep make_parameters' init; its biases
                if '.prior_biases.' in name:            # (random there) come from their own seeded stream -- training only
                    p.copy_((torch.rand(p.shape, generator=g_prior) - 0.5).to(p.device))
                continue
            if name.endswith('module.weight'):          # PReLU slope
                p.copy_(0.1 + 0.3 * torch.rand(p.shape, generator=g))
            elif name.endswith('kernel') or name.endswith('linear.weight'):
                fan = p.shape[-2] * (p.shape[0] if p.dim() == 3 else 1) if name.endswith('kernel') else p.shape[1]
                if name.endswith('kernel') and p.dim() == 3 and p.shape[0] == 27:
                    fan = p.shape[1] * 13               # about half of the 27 neighbours exist on a surface
                bound = gain / (3.0 * fan) ** 0.5 * 3.0 ** 0.5
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            else:                                       # biases
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.3)
