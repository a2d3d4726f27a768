"""Shared helpers of the test-suite: seeded synthetic clouds and weight randomisation."""
import numpy as np
import torch


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


def batched(xyz: np.ndarray, batch: int = 0) -> np.ndarray:
    return np.concatenate((np.full((len(xyz), 1), batch, dtype=np.int64), xyz.astype(np.int64)), 1)


def enliven(model: torch.nn.Module, seed: int, gain: float = 2.35) -> None:
    """Seeded re-initialisation that keeps activations O(1) through the 12-level pyramid (the default
    U(-1/sqrt(fan), 1/sqrt(fan)) init shrinks them to zero, which would make every parity test trivial)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if '.prior_' in name:                       # deep-factorised prior: keep make_parameters' init
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
