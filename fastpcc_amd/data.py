"""Data front-end at the codec boundary: the batch container `PCData` and `batched_coordinates`
(/root/reference/lib/data_utils.py:14-93), `pc_data_collate_fn` with kd-tree partitioning of large clouds (:95-234; the
partitions of a frame are coded independently, model.py:247-256), and PLY input / output (:286-351).  The reference
reads PLY through the `plyfile` package; this is a small reader / writer of its own (vertex element, ascii or
binary_little_endian), enough for voxelised clouds with optional colour."""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch


def batched_coordinates(coords: Sequence[torch.Tensor]):
    """[(n_i, 3) int] -> ((sum n_i, 4) int32 with the sample index in column 0, [n_i])"""
    sizes = [len(c) for c in coords]
    out = torch.zeros((sum(sizes), coords[0].shape[1] + 1), dtype=torch.int32, device=coords[0].device)
    at = 0
    for b, c in enumerate(coords):
        out[at: at + len(c), 0] = b
        out[at: at + len(c), 1:] = c
        at += len(c)
    return out, sizes


@dataclass
class PCData:
    xyz: Union[torch.Tensor, List[torch.Tensor]]
    batch_size: int = 1
    color: Optional[torch.Tensor] = None
    org_points_num: Optional[List[int]] = None
    resolution: Optional[List[int]] = None
    file_path: Optional[List[str]] = None
    inv_transform: Optional[List[torch.Tensor]] = None
    results_dir: Optional[str] = None
    training_step: Optional[int] = None

    def to(self, device, non_blocking=False):
        if isinstance(self.xyz, torch.Tensor):
            self.xyz = self.xyz.to(device, non_blocking=non_blocking).contiguous()
        else:
            self.xyz = [t.to(device, non_blocking=non_blocking).contiguous() for t in self.xyz]
        return self


# ---- kd-tree partition (lib/data_utils.py:168-234) ---------------------------------------------------------------------
def kd_tree_partition(coord, max_num: int, attrs: Sequence = ()):
    """Recursive median split along the axis of largest variance until every part has at most `max_num` points.
    Split rule of the reference: value = the (len // 2)-th smallest coordinate (1-based), left part = coord <= value.
    -> (list of coordinate arrays, list (per attribute) of lists of arrays | None).  numpy in, numpy out; tensors in,
    tensors out."""
    is_tensor = isinstance(coord, torch.Tensor)
    if is_tensor and coord.is_cuda:
        return _kd_tree_partition_device(coord, max_num, attrs)
    c = coord.cpu().numpy() if is_tensor else np.asarray(coord)
    a = [None if t is None else (t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)) for t in attrs]
    parts, attr_parts = [], [None if t is None else [] for t in a]

    def split(idx: np.ndarray):
        pts = c[idx]
        if len(pts) <= max_num:
            emit(idx)
            return
        axis = int(np.argmax(np.var(pts, 0)))
        half = len(pts) // 2
        value = np.partition(pts[:, axis], half - 1)[half - 1]
        left = pts[:, axis] <= value
        if half <= max_num:
            emit(idx[left])
            emit(idx[~left])
        else:
            split(idx[left])
            split(idx[~left])

    def emit(idx: np.ndarray):
        parts.append(c[idx])
        for t, out in zip(a, attr_parts):
            if t is not None:
                out.append(t[idx])

    split(np.arange(len(c)))
    if is_tensor:
        parts = [torch.from_numpy(p) for p in parts]
        attr_parts = [None if lst is None else [torch.from_numpy(p) for p in lst] for lst in attr_parts]
    return parts, attr_parts


def _kd_tree_partition_device(coord: torch.Tensor, max_num: int, attrs: Sequence = ()):
    """The same rule evaluated where the cloud lives (a frame that is already in HBM is not brought back to the host to be
    cut): variance per axis in float64 (what np.var computes for an integer array), k-th smallest coordinate by
    torch.kthvalue (as the reference does, on the host), boolean-mask splits that keep the row order.  One 8-byte read-back
    per split (the size of the left part); parts and attributes stay on the device."""
    attrs = [None if t is None else (t if isinstance(t, torch.Tensor) else torch.as_tensor(t)).to(coord.device) for t in attrs]
    parts, attr_parts = [], [None if t is None else [] for t in attrs]

    def emit(idx):
        parts.append(coord[idx])
        for t, out in zip(attrs, attr_parts):
            if t is not None:
                out.append(t[idx])

    def split(idx):
        n = idx.numel()
        if n <= max_num:
            emit(idx)
            return
        pts = coord[idx]
        axis = int(torch.argmax(torch.var(pts.to(torch.float64), dim=0, unbiased=False)).item())
        half = n // 2
        column = pts[:, axis].contiguous()
        value = torch.kthvalue(column, half).values
        left = column <= value
        if half <= max_num:
            emit(idx[left])
            emit(idx[~left])
        else:
            split(idx[left])
            split(idx[~left])

    split(torch.arange(coord.shape[0], device=coord.device))
    return parts, attr_parts


def pc_data_collate_fn(data_list: List[PCData], kd_tree_partition_max_points_num: int = 0) -> PCData:
    """batch of samples -> one PCData.  With a partition limit (test time, batch size 1) a cloud larger than the limit
    becomes `xyz = [whole cloud, part 1, part 2, ...]` (each with a zero batch column), what `compress_partitions` reads."""
    if kd_tree_partition_max_points_num > 0:
        if len(data_list) != 1:
            raise ValueError('kd-tree partition is supported for batch size 1 only')
        big = data_list[0].xyz.shape[0] > kd_tree_partition_max_points_num
    else:
        big = False
    if not big:
        xyz, sizes = batched_coordinates([d.xyz for d in data_list])
        colors = [d.color for d in data_list]
        gather = lambda name: [getattr(d, name)[0] if isinstance(getattr(d, name), list) else getattr(d, name) for d in data_list] \
            if all(getattr(d, name) is not None for d in data_list) else None
        return PCData(xyz=xyz, batch_size=len(data_list), color=torch.cat(colors, 0) if all(c is not None for c in colors) else None,
                      org_points_num=gather('org_points_num') or sizes, resolution=gather('resolution'),
                      file_path=gather('file_path'), inv_transform=gather('inv_transform'),
                      results_dir=data_list[0].results_dir)
    d = data_list[0]
    parts, (color_parts,) = kd_tree_partition(d.xyz, kd_tree_partition_max_points_num, [d.color])
    pad = lambda t: torch.nn.functional.pad(t.to(torch.int32), (1, 0, 0, 0), value=0)
    out = PCData(xyz=[pad(d.xyz)] + [pad(p) for p in parts], batch_size=1,
                 color=None if d.color is None else [d.color] + color_parts)
    for name in ('org_points_num', 'resolution', 'file_path', 'inv_transform'):
        v = getattr(d, name)
        setattr(out, name, None if v is None else (v if isinstance(v, list) else [v]))
    out.results_dir = d.results_dir
    return out


# ---- LiDAR sweeps (lib/datasets/KITTIOdometry/dataset.py:66-130) -----------------------------------------------------------
def read_kitti_bin(file_path: str) -> np.ndarray:
    """KITTI Odometry velodyne sweep: little-endian float32 (x, y, z, reflectance) records -> xyz float32 [n, 3]"""
    return np.fromfile(file_path, '<f4').reshape(-1, 4)[:, :3].copy()


def voxelize_points(xyz, scale: float, device=None):
    """Float points -> unique voxels on a grid of `scale` cells per unit, the reference's test-time voxelisation
    (dataset.py:96-101): subtract the per-axis minimum, multiply by `scale` in float32, round half to even, drop duplicates.
    Runs where the points are (or on `device`): elementwise float32 arithmetic and an integer row-unique give the same voxels
    as the NumPy original on any device.  -> (int32 [m, 3] sorted lexicographically like np.unique(axis=0), origin float32 [3])"""
    t = torch.as_tensor(xyz, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    origin = t.amin(0)
    grid = ((t - origin) * torch.tensor(scale, dtype=torch.float32, device=t.device)).round().to(torch.int32)
    return torch.unique(grid, dim=0), origin


def kitti_odometry_sample(file_path: str, resolution: float = 4096, device=None) -> PCData:
    """one test sample as KITTIOdometry.__getitem__ builds it for a .bin sweep (400 m range mapped onto `resolution` cells,
    inv_transform = (origin, metres per cell), peak value 59.70 + 1 for the distortion metric)"""
    raw = read_kitti_bin(file_path)
    scale, inv_scale = (resolution - 1) / 400, 400 / (resolution - 1)
    vox, origin = voxelize_points(raw, scale, device)
    inv = torch.cat([origin.cpu().reshape(-1), torch.tensor([inv_scale], dtype=torch.float32)]).to(torch.float32)
    return PCData(xyz=vox, file_path=[file_path], org_points_num=[raw.shape[0]], resolution=[59.70 + 1], inv_transform=[inv])


# ---- PLY ----------------------------------------------------------------------------------------------------------------
_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2',
              'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
              'double': 'f8', 'float64': 'f8'}


def write_ply_file(xyz, file_path: str, xyz_dtype: str = '<f4', rgb=None, rgb_dtype: str = 'uint8', write_ascii: bool = False,
                   make_dirs: bool = False) -> None:
    """vertex-only PLY with x, y, z (and red, green, blue)"""
    if make_dirs:
        os.makedirs(os.path.dirname(file_path) or '.', exist_ok=True)
    xyz = xyz.cpu().numpy() if isinstance(xyz, torch.Tensor) else np.asarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError('xyz must be [n, 3]')
    fields = [('x', xyz_dtype), ('y', xyz_dtype), ('z', xyz_dtype)]
    if rgb is not None:
        rgb = rgb.cpu().numpy() if isinstance(rgb, torch.Tensor) else np.asarray(rgb)
        if rgb.shape != xyz.shape:
            raise ValueError('rgb must have the shape of xyz')
        fields += [('red', rgb_dtype), ('green', rgb_dtype), ('blue', rgb_dtype)]
    rec = np.empty(len(xyz), dtype=fields)
    for i, name in enumerate('xyz'):
        rec[name] = xyz[:, i]
    if rgb is not None:
        for i, name in enumerate(('red', 'green', 'blue')):
            rec[name] = np.rint(rgb[:, i]) if rgb.dtype.kind == 'f' else rgb[:, i]
    names = {v: k for k, v in (('float', 'f4'), ('double', 'f8'), ('uchar', 'u1'), ('char', 'i1'), ('short', 'i2'), ('ushort', 'u2'),
                               ('int', 'i4'), ('uint', 'u4'))}
    head = ['ply', 'format ' + ('ascii' if write_ascii else 'binary_little_endian') + ' 1.0', f'element vertex {len(xyz)}']
    head += [f'property {names[rec.dtype[n].newbyteorder("=").str[1:]]} {n}' for n in rec.dtype.names]
    head.append('end_header')
    with open(file_path, 'wb') as f:
        f.write(('\n'.join(head) + '\n').encode('ascii'))
        if write_ascii:
            for row in rec:
                f.write((' '.join(repr(v.item()) if isinstance(v.item(), float) else str(v.item()) for v in row) + '\n').encode('ascii'))
        else:
            f.write(rec.astype(rec.dtype.newbyteorder('<')).tobytes())


def read_ply_file(file_path: str) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """-> (xyz float64/float32/int [n, 3] as stored, rgb uint8 [n, 3] | None) of the first (vertex) element"""
    with open(file_path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{file_path} is not a PLY file')
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError('unterminated PLY header')
            tok = line.decode('ascii', 'replace').split()
            if not tok or tok[0] == 'comment':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex' and count is None
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('list properties on vertices are not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if fmt is None or count is None:
            raise ValueError('PLY header without format / vertex element')
        if fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            cols = {name: rows[:, i] for i, (name, _) in enumerate(props)}
        else:
            order = '<' if fmt == 'binary_little_endian' else '>'
            rec = np.frombuffer(f.read(count * np.dtype([(n, order + t) for n, t in props]).itemsize),
                                dtype=[(n, order + t) for n, t in props], count=count)
            cols = {name: rec[name] for name, _ in props}
    xyz = np.stack([cols['x'], cols['y'], cols['z']], 1)
    rgb = np.stack([cols['red'], cols['green'], cols['blue']], 1).astype(np.uint8) if 'red' in cols else None
    return xyz, rgb
