"""Dataset classes in front of the codecs, with the names, constructor arguments and configuration fields of the reference's
(/root/reference/lib/datasets/KITTIOdometry/dataset.py:15-145, ShapeNetCorev2/dataset.py:15-156,
PlyVoxel/Base/dataset.py:15-265 and their dataset_config.py files), producing `fastpcc_amd.data.PCData` samples and batching
through `pc_data_collate_fn` (kd-tree partitions at test time).

What differs is where the per-sample work runs.  The reference voxelises with NumPy on the loader's host thread (scale,
round, np.unique, Morton argsort).  Here every sample goes through `voxelize` below on `device` -- with a GPU that is
float32 scale / round on the device, a row-unique, and the Morton order through the key kernels of libfpcc_hip -- so a
frame leaves the dataset already resident in HBM (the codec's boundary: SURVEY.md section 8d).  On the host (`device=None`,
CPU tensors) the same tensor operations give the reference's voxels; the tests compare both against a NumPy restatement.

Not carried over: reference frames of the inter-frame codecs (`ref_frames_num > 0`), reflectance, normal estimation for the
pc_error cache files (the distortion is computed on the device, fastpcc_amd/evaluators.py), poisson-disk mesh sampling.
"""
import glob
import hashlib
import logging
import math
import os
import pathlib
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.utils.data

from .data import PCData, kd_tree_partition, pc_data_collate_fn, read_kitti_bin, read_ply_file

_LOG = logging.getLogger('fastpcc_amd.datasets')


# ---- device-side sample pipeline ---------------------------------------------------------------------------------------
def morton_order(xyz: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """argsort of the Morton keys (x on bit 0; `inverse`: z on bit 0), the order `morton_sort` asks for
    (lib/space_filling_curves/__init__.py:65-88).  GPU tensors use fpcc_morton3d_encode, host tensors the magic-bits form."""
    cols = (2, 1, 0) if inverse else (0, 1, 2)
    if xyz.is_cuda:
        from . import hipops
        return torch.argsort(hipops.morton3d_encode(xyz.to(torch.int32).contiguous(), cols))
    v = xyz.to(torch.int64)
    key = torch.zeros(v.shape[0], dtype=torch.int64)
    for bit in range(21):
        for axis, c in enumerate(cols):
            key |= ((v[:, c] >> bit) & 1) << (3 * bit + axis)
    return torch.argsort(key)


def voxelize(xyz, scale: float = 1.0, device=None, origin: bool = True):
    """float points [n, 3] -> (unique int32 voxels sorted like np.unique(axis=0), origin float32 [3]): subtract the per-axis
    minimum (when `origin`), multiply by `scale` in float32, round half to even, drop duplicates -- all on `device`."""
    t = torch.as_tensor(np.ascontiguousarray(xyz) if isinstance(xyz, np.ndarray) else xyz)
    if device is not None:
        t = t.to(device)
    t = t.to(torch.float32)
    org = t.amin(0) if origin else torch.zeros(3, dtype=torch.float32, device=t.device)
    t = t - org
    if scale != 1:
        t = t * torch.tensor(scale, dtype=torch.float32, device=t.device)
    return torch.unique(t.round().to(torch.int32), dim=0), org


def kd_tree_partition_randomly(coord: torch.Tensor, target_num: int, attrs: Sequence[Optional[torch.Tensor]] = (),
                               generator: Optional[torch.Generator] = None):
    """A random slab of about `target_num` points for training crops (lib/data_utils.py:236-283): repeatedly keep, along the
    axis of largest variance, the points between the (s+1)-th and (s+k)-th smallest coordinate, k = max(round(n / 2),
    target_num), s uniform in [0, n - k], until k <= target_num.  Tensor in (any device), tensor out."""
    while coord.shape[0] > target_num:
        n = coord.shape[0]
        axis = int(torch.argmax(torch.var(coord.to(torch.float64), dim=0, unbiased=False)).item())
        k = max(round(n * 0.5), target_num)
        s = int(torch.randint(n - k + 1, (1,), generator=generator).item())
        column = coord[:, axis].contiguous()
        lo, hi = torch.kthvalue(column, s + 1).values, torch.kthvalue(column, s + k).values
        keep = (column >= lo) & (column <= hi)
        coord = coord[keep]
        attrs = tuple(None if a is None else a[keep] for a in attrs)
        if k <= target_num:
            break
    return (coord, attrs) if len(attrs) else coord


def random_flip_xy(xyz: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    for axis in (0, 1):
        if torch.rand(1, generator=generator).item() > 0.5:
            xyz[:, axis] = xyz[:, axis].max() - xyz[:, axis]
    return xyz


def sample_mesh_uniform(obj_path: str, points_num: int, generator: Optional[np.random.Generator] = None) -> np.ndarray:
    """area-uniform points on the triangles of a Wavefront OBJ (the role of open3d's sample_points_uniformly in
    lib/data_utils.py:364-378): triangle ~ area, point = (1 - sqrt(u)) a + sqrt(u) (1 - v) b + sqrt(u) v c."""
    rng = generator or np.random.default_rng()
    verts, faces = [], []
    with open(obj_path) as f:
        for line in f:
            if line.startswith('v '):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith('f '):
                idx = [int(t.split('/')[0]) for t in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                faces.extend((idx[0], idx[j], idx[j + 1]) for j in range(1, len(idx) - 1))          # fan triangulation
    if not faces:
        raise ValueError(f'{obj_path} has no faces')
    v, f = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    pick = rng.choice(len(f), size=points_num, p=area / area.sum())
    su, w = np.sqrt(rng.random((points_num, 1))), rng.random((points_num, 1))
    return (1 - su) * a[pick] + su * (1 - w) * b[pick] + su * w * c[pick]


def normalize_coords(xyz: np.ndarray) -> Tuple[np.ndarray, float]:
    lo = xyz.min(0, keepdims=True)
    scale = float((xyz.max(0, keepdims=True) - lo).max())
    xyz -= lo
    xyz /= scale
    return lo, scale


def _load_or_make_filelist(root: str, list_path: str, make: Callable[[], List[str]], interval: int = 1, log=_LOG) -> List[str]:
    path = os.path.join(root, list_path)
    if not os.path.exists(path):
        log.info('no filelist is given. Trying to generate...')
        with open(path, 'w') as f:
            f.write('\n'.join(make()))
    log.info(f'using filelist: "{path}"')
    with open(path) as f:
        return [os.path.join(root, line.strip()) for line in f.read().splitlines()[::interval] if line.strip()]


class _Base(torch.utils.data.Dataset):
    def __init__(self, cfg, is_training: bool, logger=None, device=None):
        super().__init__()
        self.cfg, self.is_training, self.logger, self.device = cfg, is_training, logger or _LOG, device
        self.file_list: List[str] = []

    def __len__(self):
        return len(self.file_list)

    def _finish(self, xyz: torch.Tensor, color: Optional[torch.Tensor] = None):
        if self.cfg.morton_sort:
            order = morton_order(xyz, self.cfg.morton_sort_inverse)
            xyz = xyz[order]
            color = None if color is None else color[order]
        return xyz, color


# ---- KITTI Odometry (lib/datasets/KITTIOdometry) -------------------------------------------------------------------------
@dataclass
class KITTIOdometryConfig:
    root: str = 'datasets/KITTI/sequences'
    train_filelist_path: str = 'train_list.txt'
    test_filelist_path: str = 'test_list.txt'
    train_subset_index: Tuple[int, ...] = tuple(range(0, 11))
    test_subset_index: Tuple[int, ...] = tuple(range(11, 22))
    list_sampling_interval: int = 1
    random_flip: bool = False
    random_rotation: bool = False
    kd_tree_partition_max_points_num: int = 0
    morton_sort: bool = False
    morton_sort_inverse: bool = False
    resolution: Union[int, float] = 4096
    flag_sparsepcgc: bool = False
    ply_file_root: str = ''
    ply_file_train_filelist_path: str = 'train_list.txt'
    ply_file_test_filelist_path: str = 'test_list.txt'
    ply_list_sampling_interval: int = -1
    ply_file_coord_scaler: float = 1.0
    ply_file_resolution: int = 0


class KITTIOdometry(_Base):
    def __init__(self, cfg: KITTIOdometryConfig, is_training: bool, logger=None, device=None):
        super().__init__(cfg, is_training, logger, device)
        subsets = cfg.train_subset_index if is_training else cfg.test_subset_index
        list_path = cfg.train_filelist_path if is_training else cfg.test_filelist_path

        def make():
            out = []
            for i in subsets:
                out.extend(sorted(str(p.relative_to(cfg.root)) for p in pathlib.Path(cfg.root).glob(f'{i:02d}/velodyne/*.bin')))
            return out
        self.file_list = _load_or_make_filelist(cfg.root, list_path, make, cfg.list_sampling_interval, self.logger)
        ply_list = cfg.ply_file_train_filelist_path if is_training else cfg.ply_file_test_filelist_path
        if cfg.ply_file_root and os.path.exists(os.path.join(cfg.ply_file_root, ply_list)):
            step = cfg.ply_list_sampling_interval if cfg.ply_list_sampling_interval > 0 else cfg.list_sampling_interval
            self.file_list += _load_or_make_filelist(cfg.ply_file_root, ply_list, list, step, self.logger)

    def __getitem__(self, index) -> PCData:
        cfg, path = self.cfg, self.file_list[index]
        sweep = path.endswith('bin')
        xyz = read_kitti_bin(path) if sweep else np.asarray(read_ply_file(path)[0], dtype=np.float32)
        n_org = xyz.shape[0]
        scale = (cfg.resolution - 1) / 400 if sweep else cfg.ply_file_coord_scaler
        if self.is_training and cfg.random_rotation:
            a = float(torch.rand(1).item()) * 2 * math.pi
            rot = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], np.float32)
            xyz = xyz @ rot.T
        vox, org = voxelize(xyz, scale, self.device)
        org = org.cpu()
        if self.is_training:
            par = cfg.kd_tree_partition_max_points_num
            if par and vox.shape[0] > par:
                vox = kd_tree_partition_randomly(vox, par)
                lo = vox.amin(0)
                vox = vox - lo
                org = org + lo.cpu().to(torch.float32)
            if cfg.random_flip:
                vox = random_flip_xy(vox)
        vox, _ = self._finish(vox)
        inv = torch.cat([org.reshape(-1), torch.tensor([1.0 / scale], dtype=torch.float32)])
        if sweep and not cfg.flag_sparsepcgc:
            peak = 59.70 + 1
        elif sweep:
            peak, inv = 30000 + 1, inv * 1000
        else:
            peak = cfg.ply_file_resolution
        return PCData(xyz=vox, file_path=path, org_points_num=n_org, resolution=peak, inv_transform=inv)

    def collate_fn(self, batch):
        if not self.is_training:
            if len(batch) != 1:
                raise ValueError('test batches hold one frame')
            return pc_data_collate_fn(batch, kd_tree_partition_max_points_num=self.cfg.kd_tree_partition_max_points_num)
        return pc_data_collate_fn(batch)


# ---- ShapeNetCore.v2 (lib/datasets/ShapeNetCorev2) -----------------------------------------------------------------------
@dataclass
class ShapeNetCorev2Config:
    root: str = 'datasets/ShapeNet/ShapeNetCore.v2'
    shapenet_all_csv: str = 'all.csv'
    train_filelist_path: str = 'all_list_obj.txt'
    test_filelist_path: str = 'test_list_obj.txt'
    train_divisions: Union[str, Tuple[str, ...]] = 'all'
    test_divisions: Union[str, Tuple[str, ...]] = 'test'
    generate_cache: bool = True
    mesh_sample_points_num: int = 2500000
    mesh_sample_point_method: str = 'uniform'
    mesh_sample_point_resolution: int = 256
    ply_cache_dtype: str = '<u2'
    random_rotation: bool = True
    random_offset: Union[int, Tuple[int, ...]] = 0
    kd_tree_partition_max_points_num: int = 0
    morton_sort: bool = False
    morton_sort_inverse: bool = False
    resolution: int = 128


class ShapeNetCorev2(_Base):
    """meshes sampled to `mesh_sample_points_num` surface points, normalised to the unit cube, scaled to
    `mesh_sample_point_resolution` and cached as unique voxels (.npz); per sample: random rotation, rescale to `resolution`,
    voxelise, optional random crop / offset"""

    def __init__(self, cfg: ShapeNetCorev2Config, is_training: bool, logger=None, device=None):
        super().__init__(cfg, is_training, logger, device)
        if cfg.resolution <= 1:
            raise ValueError('resolution must exceed 1')
        if cfg.mesh_sample_point_method != 'uniform':
            raise NotImplementedError('only uniform mesh sampling')
        list_path = cfg.train_filelist_path if is_training else cfg.test_filelist_path
        divisions = cfg.train_divisions if is_training else cfg.test_divisions
        divisions = (divisions,) if isinstance(divisions, str) else tuple(divisions)

        def make():
            if 'all' in divisions:
                found = [p[len(cfg.root) + 1:] for p in glob.glob(f'{cfg.root}/*/*/*/*.obj')]
            else:
                found = []
                with open(os.path.join(cfg.root, cfg.shapenet_all_csv)) as f:
                    next(f)
                    for line in f:
                        _, synset, _, model, split = line.strip().split(',')
                        rel = os.path.join(synset, model, 'models', 'model_normalized.obj')
                        if split in divisions and os.path.exists(os.path.join(cfg.root, rel)):
                            found.append(rel)
            return sorted(p for p in found if '7edb40d76dff7455c2ff7551a4114669' not in p)     # the mesh the reference skips
        self.file_list = _load_or_make_filelist(cfg.root, list_path, make, 1, self.logger)
        self.cache_root = None
        if cfg.generate_cache:
            tag = f'{os.path.join(cfg.root, list_path)} {cfg.mesh_sample_points_num} {cfg.mesh_sample_point_method} ' \
                  f'{cfg.mesh_sample_point_resolution} {cfg.ply_cache_dtype} '
            self.cache_root = os.path.join(cfg.root, 'cache', hashlib.md5(tag.encode()).hexdigest())

    def _cache_path(self, path: str) -> str:
        return path.replace(self.cfg.root, self.cache_root, 1).replace('.obj', '.npz', 1)

    def _voxels_at_sampling_resolution(self, path: str) -> np.ndarray:
        cache = self._cache_path(path) if self.cache_root else None
        if cache and os.path.isfile(cache):
            return np.load(cache)['xyz'].astype(np.float64)
        xyz = sample_mesh_uniform(path, self.cfg.mesh_sample_points_num)
        normalize_coords(xyz)
        xyz *= self.cfg.mesh_sample_point_resolution
        if cache:
            vox = np.unique(xyz.astype(self.cfg.ply_cache_dtype), axis=0)
            os.makedirs(os.path.dirname(cache), exist_ok=True)
            np.savez_compressed(cache, xyz=vox)
            return vox.astype(np.float64)
        return xyz

    def __getitem__(self, index) -> PCData:
        cfg, path = self.cfg, self.file_list[index]
        xyz = self._voxels_at_sampling_resolution(path)
        if cfg.random_rotation:
            q = torch.randn(4).double()
            q = (q / q.norm()).tolist()                                   # uniform random rotation from a unit quaternion
            w, x, y, z = q
            rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            xyz = xyz @ rot.T
            xyz -= xyz.min(0)
        t = torch.as_tensor(xyz, dtype=torch.float64)
        if self.device is not None:
            t = t.to(self.device)
        if cfg.resolution != cfg.mesh_sample_point_resolution:
            t = t * (cfg.resolution / cfg.mesh_sample_point_resolution)
        vox = torch.unique(t.to(torch.int32), dim=0)                      # truncation, as the reference's astype(np.int32)
        if self.is_training:
            par = cfg.kd_tree_partition_max_points_num
            if par and vox.shape[0] > par:
                vox = kd_tree_partition_randomly(vox, par)
                vox = vox - vox.amin(0)
            if cfg.random_offset != 0:
                vox = vox + torch.randint(0, int(cfg.random_offset), (3,), dtype=torch.int32).to(vox.device)
        vox, _ = self._finish(vox)
        return PCData(xyz=vox, file_path=path)

    def collate_fn(self, batch):
        return pc_data_collate_fn(batch)


# ---- voxelised PLY sets: MVUB, 8iVFB, Owlii (lib/datasets/PlyVoxel) -------------------------------------------------------
@dataclass
class PlyVoxelConfig:
    root: Union[str, Tuple[str, ...]] = ('datasets/MVUB', 'datasets/8iVFBv2', 'datasets/Owlii')
    filelist_path: Union[str, Tuple[str, ...]] = 'list.txt'
    file_path_pattern: Union[str, Tuple[str, ...]] = '**/*.ply'
    list_sampling_interval: int = 1
    ref_frames_num: int = 0
    kd_tree_partition_max_points_num: Union[int, Tuple[int, ...]] = 0
    coord_scaler: Union[float, Tuple[float, ...]] = 1.0
    random_batch_coord_scaler_log2: Tuple[int, ...] = (0,)
    with_color: bool = False
    with_reflectance: bool = False
    random_flip: bool = False
    morton_sort: bool = False
    morton_sort_inverse: bool = False
    resolution: Union[int, Tuple[int, ...]] = (512, 1024, 2048)


class PlyVoxel(_Base):
    """several roots, each with its own file list / resolution / scaler / partition limit (scalar fields are broadcast)"""

    def __init__(self, cfg: PlyVoxelConfig, is_training: bool, logger=None, device=None):
        super().__init__(cfg, is_training, logger, device)
        if cfg.ref_frames_num or cfg.with_reflectance:
            raise NotImplementedError('reference frames / reflectance belong to codecs outside this build')
        fields = [cfg.root, cfg.filelist_path, cfg.file_path_pattern, cfg.resolution, cfg.coord_scaler, cfg.kd_tree_partition_max_points_num]
        n = max((len(v) for v in fields if isinstance(v, (tuple, list))), default=1)
        for v in fields:
            if isinstance(v, (tuple, list)) and len(v) != n:
                raise ValueError(f'dataset config items must have one length ({n}), got {len(v)}')
        cols = [tuple(v) if isinstance(v, (tuple, list)) else (v,) * n for v in fields]
        self.file_resolutions, self.file_scalers, self.file_par_nums = [], [], []
        for root, list_path, pattern, res, scaler, par in zip(*cols):
            files = _load_or_make_filelist(root, list_path, lambda: [str(p.relative_to(root)) for p in sorted(pathlib.Path(root).glob(pattern))],
                                           cfg.list_sampling_interval, self.logger)
            self.file_list += files
            self.file_resolutions += [res] * len(files)
            self.file_scalers += [scaler] * len(files)
            self.file_par_nums += [par] * len(files)
        self.batch_scalers = tuple((e, 2 ** e) for e in cfg.random_batch_coord_scaler_log2)

    def __getitem__(self, index):
        return index                          # samples are built in collate_fn: the whole batch shares one random scaler

    def getitem(self, index: int, batch_coord_scaler: float = 1.0) -> PCData:
        cfg, path = self.cfg, self.file_list[index]
        pts, rgb = read_ply_file(path)
        pts = np.asarray(pts, dtype=np.float32) if pts.dtype.kind != 'f' else pts
        n_org = pts.shape[0]
        scaler = self.file_scalers[index] * batch_coord_scaler
        color = None
        if scaler != 1:
            if cfg.with_color:
                raise ValueError('rescaled clouds carry no attributes')
            vox, org = voxelize(pts, scaler, self.device)
        else:                                   # voxelised input without duplicates: keep the row order, colours stay aligned
            t = torch.as_tensor(np.ascontiguousarray(pts))
            t = t.to(self.device) if self.device is not None else t
            org = t.amin(0).to(torch.float32)
            vox = (t - t.amin(0)).to(torch.int32)
            if cfg.with_color:
                if rgb is None or rgb.shape[0] != vox.shape[0]:
                    raise ValueError(f'{path} has no per-vertex colours')
                color = torch.as_tensor(rgb.astype(np.float32)).to(vox.device)
        org = org.cpu()
        if self.is_training:
            par = self.file_par_nums[index]
            if par and vox.shape[0] > par:
                vox, (color,) = kd_tree_partition_randomly(vox, par, (color,))
                lo = vox.amin(0)
                vox = vox - lo
                org = org + lo.cpu().to(torch.float32)
            if cfg.random_flip:
                vox = random_flip_xy(vox)
        vox, color = self._finish(vox, color)
        inv = torch.cat([org.reshape(-1), torch.tensor([1.0 / scaler], dtype=torch.float32)])
        return PCData(xyz=vox, color=color, file_path=path, resolution=None if self.is_training else self.file_resolutions[index],
                      org_points_num=n_org, inv_transform=inv)

    def collate_fn(self, batch: List[int]):
        log2, scaler = self.batch_scalers[int(torch.randint(len(self.batch_scalers), (1,)).item())]
        samples = [self.getitem(i, scaler) for i in batch]
        if not self.is_training:
            if len(samples) != 1:
                raise ValueError('test batches hold one frame')
            out = pc_data_collate_fn(samples, kd_tree_partition_max_points_num=self.file_par_nums[batch[0]])
        else:
            out = pc_data_collate_fn(samples)
        out.batch_coord_scaler_log2 = log2 or None
        return out
