"""TEST ORACLE -- not product code.

NumPy restatement of the coordinate bookkeeping the reference obtains from MinkowskiEngine's CoordinateManager
(un-vendored dependency; call sites /root/reference/models/convolutional/lossy_coord_v2/model.py:126-153,
/root/reference/models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:249-255,281-283,306-317 and
/root/reference/lib/minkowski_sparse_conv_layers.py:67-80) plus the Morton key of
/root/reference/lib/space_filling_curves/__init__.py:46-88.

Parity: Morton keys are PINNED against the reference's CPU path (tests/golden/morton.json, produced by
tests/golden/make_golden.py).  The coordinate-manager semantics are UNPINNED (MinkowskiEngine cannot run here); they
follow the table in SURVEY.md section 8a "ME semantics":
  (ii)  HYPER_CUBE offsets enumerate the first spatial axis fastest; odd sizes centred, even sizes anchored at 0;
  (iii) a stride-2 output coordinate is floor(c / 2ts) * 2ts;
  (iv)  transposed maps are the forward maps with in/out swapped; generative transposed maps emit all 8 children;
  (viii) rows are in Morton order (x on bit 0).
Kernel maps are returned the way MinkowskiEngine returns them: one (in_rows, out_rows) pair list per kernel offset.
"""
from typing import Dict, List, Tuple

import numpy as np

_AXES = {'x': 0, 'y': 1, 'z': 2}


def _spread21(v: np.ndarray) -> np.ndarray:
    x = v.astype(np.uint64) & np.uint64(0x1fffff)
    for shift, mask in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f),
                        (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
        x = (x | (x << np.uint64(shift))) & np.uint64(mask)
    return x


def morton_encode(xyz: np.ndarray, axis_order: str = 'xyz', inverse: bool = False) -> np.ndarray:
    """int64 Morton key; the first axis of `axis_order` lands on bit 0."""
    order = axis_order.lower()
    if inverse:
        order = order[::-1]
    i0, i1, i2 = (_AXES[a] for a in order)
    xyz = np.asarray(xyz)
    key = _spread21(xyz[:, i0]) | (_spread21(xyz[:, i1]) << np.uint64(1)) | (_spread21(xyz[:, i2]) << np.uint64(2))
    return key.astype(np.int64)


class Level:
    """A coordinate map: unique coordinates [n, 4] = (batch, x, y, z) (multiples of `stride`) in Morton order."""

    def __init__(self, coords: np.ndarray, stride: int):
        self.stride = int(stride)
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
        key = self._key(coords)
        order = np.argsort(key, kind='stable')
        key = key[order]
        keep = np.ones(len(key), bool)
        keep[1:] = key[1:] != key[:-1]
        self.order = order[keep]          # input row of every map row
        self.key = key[keep]
        self.coords = coords[self.order]
        self.n = len(self.key)

    def _key(self, coords: np.ndarray) -> np.ndarray:
        # batch-major, then Morton order of the coordinates in units of the stride
        m = morton_encode(coords[:, 1:] // self.stride)
        return (coords[:, 0].astype(np.int64) << 57) | m        # 19 bits per axis are plenty for the tests

    def rows_of(self, coords: np.ndarray) -> np.ndarray:
        """row of each query coordinate, -1 when absent (negative coordinates are always absent)"""
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
        ok = (coords[:, 1:] >= 0).all(1)
        k = self._key(np.where(ok[:, None], coords, 0))
        pos = np.searchsorted(self.key, k)
        pos_c = np.minimum(pos, max(self.n - 1, 0))
        hit = ok & (self.n > 0) & (self.key[pos_c] == k) if self.n else np.zeros(len(k), bool)
        return np.where(hit, pos_c, -1)


def kernel_offsets(kernel_size: int, tensor_stride: int) -> np.ndarray:
    """[K, 3] offsets, first axis fastest; odd: centred, even: anchored at 0 (semantics (ii))."""
    r = np.arange(kernel_size) - (kernel_size // 2 if kernel_size % 2 else 0)
    z, y, x = np.meshgrid(r, r, r, indexing='ij')
    return np.stack((x.reshape(-1), y.reshape(-1), z.reshape(-1)), 1) * tensor_stride


def strided(level: Level) -> Level:
    """stride-2 coordinate map (semantics (iii))"""
    s2 = level.stride * 2
    c = level.coords.copy()
    c[:, 1:] = c[:, 1:] // s2 * s2
    return Level(c, s2)


def generated(level: Level) -> Level:
    """all 8 children of every voxel (generative transposed convolution, semantics (iv))"""
    half = level.stride // 2
    off = kernel_offsets(2, half)
    c = np.repeat(level.coords, 8, axis=0)
    c[:, 1:] += np.tile(off, (level.n, 1))
    return Level(c, half)


KernelMap = List[Tuple[np.ndarray, np.ndarray]]    # per offset: (in_rows, out_rows), out_rows ascending


def kernel_map(src: Level, dst: Level, kernel_size: int) -> KernelMap:
    """Forward convolution map src -> dst: input voxel = output voxel + offset_k.
    kernel 3 (dst.stride == src.stride), kernel 2 stride 2 (dst.stride == 2 * src.stride) or kernel 1."""
    offs = kernel_offsets(kernel_size, src.stride)
    out = []
    rows = np.arange(dst.n)
    for o in offs:
        q = dst.coords.copy()
        q[:, 1:] += o
        r = src.rows_of(q)
        hit = r >= 0
        out.append((r[hit], rows[hit]))
    return out


def transposed_map(src: Level, dst: Level) -> KernelMap:
    """kernel 2 stride 2 transposed: the forward map dst -> src with in/out swapped (semantics (iv))."""
    fwd = kernel_map(dst, src, 2)
    out = []
    for in_rows, out_rows in fwd:          # forward: in = fine (dst), out = coarse (src)
        order = np.argsort(in_rows, kind='stable')
        out.append((out_rows[order], in_rows[order]))
    return out


def dense_table(kmap: KernelMap, n_out: int) -> np.ndarray:
    """[K, n_out] int32 input row per (offset, output row), -1 where absent"""
    t = np.full((len(kmap), n_out), -1, dtype=np.int32)
    for k, (i, o) in enumerate(kmap):
        t[k, o] = i
    return t
