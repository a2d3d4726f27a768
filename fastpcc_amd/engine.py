"""Sparse-tensor engine with the names the reference's model code uses from MinkowskiEngine
(`ME.SparseTensor`, `ME.CoordinateManager`, `ME.CoordinateMapKey`, `ME.cat`, `ME.Minkowski*` modules; call sites listed in
SURVEY.md section 8b), backed by libfpcc_hip.so.

This is not MinkowskiEngine's design.  A coordinate map is a SORTED array of unique Morton keys (x on bit 0) living in a
pyramid: the stride-2 parent of a key is key >> 3, its octant key & 7.  Consequences used throughout:
  * row order of every map is Morton order -- the invariant the codec's bitstream needs
    (/root/reference/models/convolutional/lossy_coord_v2/model.py:121-123,141-142);
  * stride-2 / transposed / generative maps are one `child_row` table [parents, 8];
  * the 3x3x3 kernel map of a level is derived from its parent's (no hash table).
Convolutions are output-stationary (gather -> MFMA GEMM), with bias, activation and channel concatenation fused.

Modules hold ordinary nn.Parameters (state_dict keys equal the reference's).  With gradients enabled the convolutions
run through fastpcc_amd/autograd.py (same kernels; bias / activation as differentiable tensor ops); under
torch.no_grad() they take the fused inference path.
"""
import functools
import math
import os
from enum import Enum
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import hipops as ops


# ---------------------------------------------------------------------------------------------------------------
# enums kept for source compatibility with the reference's call sites
class RegionType(Enum):
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


class MinkowskiAlgorithm(Enum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


class CoordinateMapType(Enum):
    CPU = 0
    CUDA = 1


class SparseTensorQuantizationMode(Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class SparseTensorOperationMode(Enum):
    SEPARATE_COORDINATE_MANAGER = 0
    SHARE_COORDINATE_MANAGER = 1


# The "global" coordinate manager is global PER THREAD: a serving process keeps several frames in flight, each driven by its own
# thread with its own model context (fastpcc_amd/serving.py); the single-threaded use the reference knows is unchanged.
import threading as _threading
_tls = _threading.local()
_operation_mode = SparseTensorOperationMode.SEPARATE_COORDINATE_MANAGER


def set_sparse_tensor_operation_mode(mode: SparseTensorOperationMode):
    global _operation_mode
    _operation_mode = mode


def set_global_coordinate_manager(cm: 'CoordinateManager'):
    _tls.cm = cm


def clear_global_coordinate_manager():
    """Drops the (calling thread's) global manager.  Its maps are released by reference counting right here (the links that would form
    reference cycles are cut): left to Python's cyclic collector, the previous frame's tables -- gigabytes on a 2 M-voxel frame -- stay
    allocated while the next frame is coded, the caching allocator has to grow (a hipMalloc costs 10-25 ms) and its reserve creeps up."""
    cm = getattr(_tls, 'cm', None)
    if cm is not None:
        cm._break_cycles()
    _tls.cm = None


def global_coordinate_manager() -> Optional['CoordinateManager']:
    return getattr(_tls, 'cm', None)


def _as_stride(s) -> Tuple[int, int, int]:
    if isinstance(s, int):
        return (s, s, s)
    s = tuple(int(v) for v in s)
    if len(s) == 1:
        s = s * 3
    if len(s) != 3 or len(set(s)) != 1 or s[0] < 1 or s[0] & (s[0] - 1):
        raise NotImplementedError(f'only isotropic power-of-two tensor strides are supported, got {s}')
    return s


class CoordinateMapKey:
    def __init__(self, tensor_stride, string_id: str = ''):
        self._stride = _as_stride(tensor_stride)
        self._id = string_id

    def get_tensor_stride(self) -> List[int]:
        return list(self._stride)

    def get_key(self):
        return list(self._stride), self._id

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and self._stride == other._stride and self._id == other._id

    def __hash__(self):
        return hash((self._stride, self._id))

    def __repr__(self):
        return f'CoordinateMapKey(stride={list(self._stride)}, id={self._id!r})'


class KernelGenerator:
    def __init__(self, kernel_size=-1, stride=1, dilation=1, region_type: RegionType = RegionType.HYPER_CUBE,
                 dimension: int = 3, **_):
        if dimension != 3:
            raise NotImplementedError('3-D only')
        if region_type not in (RegionType.HYPER_CUBE, 'HYPER_CUBE'):
            raise NotImplementedError('only HYPER_CUBE kernels')
        as3 = lambda v: [v] * 3 if isinstance(v, int) else list(v)
        self.kernel_size, self.kernel_stride, self.kernel_dilation = as3(kernel_size), as3(stride), as3(dilation)
        self.region_type = region_type
        self.dimension = dimension
        self.kernel_volume = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]


# ---------------------------------------------------------------------------------------------------------------
class _Map:
    """One coordinate map: sorted unique keys at pyramid level `level` (tensor stride 1 << level)."""
    __slots__ = ('level', 'bits', 'n', 'keys', 'parent', 'parent_of', 'child_row', 'generated', 'nbr27', 'coords',
                 'gen_child', 'key', 'row_order', 'mask27', 'nbr27_rows', 'nbr27_pos', 'edges', 'k2_order', 'k2_pos')

    def __init__(self, level: int, bits: int, n: int, keys: Optional[torch.Tensor]):
        self.level, self.bits, self.n, self.keys = level, bits, n, keys
        self.parent: Optional['_Map'] = None
        self.parent_of: Optional[torch.Tensor] = None     # [n] row of the parent
        self.child_row: Optional[torch.Tensor] = None     # [parent.n, 8] row of (parent, octant) in THIS map or -1
        self.generated = False                            # all 8 children of every parent row, row = 8p + octant
        self.nbr27: Optional[torch.Tensor] = None
        self.nbr27_rows: Optional[torch.Tensor] = None    # the same table row-major [n, 32] (what the MFMA kernels' prologue reads)
        self.nbr27_pos: Optional[torch.Tensor] = None     # ... with its rows in row_order: row p = the neighbours of output row row_order[p]
        self.mask27: Optional[torch.Tensor] = None        # [n] 27-bit neighbour presence (first layer on a constant input)
        self.row_order = False                            # False: not decided; None: natural order; tensor: permutation
        self.coords: Optional[torch.Tensor] = None
        self.gen_child: Optional['_Map'] = None
        self.key: Optional[CoordinateMapKey] = None
        self.edges: Optional[List[int]] = None            # host copy of the row ranges of the batch's clouds (rows are cloud-major)
        # stride-2 convolution ONTO this map's parent (child_row read as a 2x2x2 kernel map): the parent's rows grouped by which of
        # their 8 children exist (False: not decided), and child_row with its rows in that order
        self.k2_order = False
        self.k2_pos: Optional[torch.Tensor] = None


class CoordinateManager:
    """Owns the pyramid of coordinate maps of one batch of clouds."""
    # below this many rows the top level's 27-neighbour table is found by binary search instead of climbing further
    ROOT_ROWS = 2048

    def __init__(self, D: int = 3, coordinate_map_type=None, minkowski_algorithm=None, bits: Optional[int] = None,
                 clouds: Optional[int] = None):
        """clouds: the batch holds this many INDEPENDENT clouds that are coded in one traversal (compress_many / compress_partitions of
        the codecs: /root/reference/models/convolutional/lossy_coord_v2/model.py:247-256,277-288).  Rows of different clouds never
        neighbour (the cloud index sits above the Morton bits of a key), so a layer computes for every cloud exactly what it would
        compute for that cloud alone -- PROVIDED the rules that look at a map's row count look at the cloud's own count: that is
        what `independent_clouds` switches on (PAD_MIN_ROWS below; the codecs' top-k thresholds and headers are per cloud anyway).
        A training batch (clouds=None) keeps the whole-map rules."""
        if D != 3:
            raise NotImplementedError('3-D only')
        if clouds is not None and not 1 <= clouds <= 64:
            raise ValueError('1 <= clouds <= 64')
        self._maps: Dict[CoordinateMapKey, _Map] = {}
        self._bits0 = bits            # bits per axis at level 0; fixed on the first insertion
        self._n_batch: Optional[int] = clouds
        self.independent_clouds = clouds is not None and clouds > 1
        self.device = None

    @property
    def _manager(self) -> 'CoordinateManager':
        # the reference reaches into `coordinate_manager._manager` (a property, not an attribute: no self-reference cycle)
        return self

    def _break_cycles(self) -> None:
        """a map and the generated set of its children point at each other; cutting the downward link leaves plain parent
        chains, which reference counting frees as soon as the last tensor on them goes"""
        for m in self._maps.values():
            m.gen_child = None

    # -- key bookkeeping --------------------------------------------------------------------------------------------
    def _register(self, m: _Map, string_id: str = '') -> CoordinateMapKey:
        key = CoordinateMapKey(1 << m.level, string_id)
        n = 0
        while key in self._maps and self._maps[key] is not m:
            n += 1
            key = CoordinateMapKey(1 << m.level, f'{string_id}#{n}')
        self._maps[key] = m
        if m.key is None:
            m.key = key
        return key

    def _map(self, key: CoordinateMapKey) -> _Map:
        try:
            return self._maps[key]
        except KeyError:
            raise KeyError(f'{key} is not in this coordinate manager') from None

    def get_coordinate_map_keys(self, tensor_stride) -> List[CoordinateMapKey]:
        s = _as_stride(tensor_stride)
        return [k for k in self._maps if k._stride == s]

    # -- creation ---------------------------------------------------------------------------------------------------
    def insert_and_map(self, coordinates: torch.Tensor, tensor_stride=1, string_id: str = ''):
        """coordinates int32 [n, 4] = (batch, x, y, z), multiples of the stride.  Returns (key, (perm, count)) where
        perm[i] is the input row that became map row i (duplicates: first occurrence in sorted order)."""
        stride = _as_stride(tensor_stride)
        level = stride[0].bit_length() - 1
        if coordinates.dtype != torch.int32 or coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise TypeError('coordinates must be int32 [n, 4] = (batch, x, y, z)')
        coordinates = coordinates.contiguous()
        self.device = coordinates.device
        if self._bits0 is None:
            nb = self._n_batch                           # known up front (clouds=): no read-back
            if nb is None:
                nb = int(coordinates[:, 0].max().item()) + 1 if coordinates.shape[0] else 1
            self._n_batch = nb
            self._bits0 = min(21, (63 - max(nb - 1, 0).bit_length()) // 3)
        bits = self._bits0 - level
        if bits < 1:
            raise ValueError('tensor stride exceeds the coordinate range')
        keys = ops.keys_from_coords(coordinates, level, bits)
        skeys, perm = ops.sort_keys(keys, 63)
        ukeys, first, count = ops.unique_keys(skeys)
        n = int(count.item())
        m = _Map(level, bits, n, ukeys[:n])
        rows = perm if n == coordinates.shape[0] else perm[first[:n].long()]
        key = self._register(m, string_id)
        return key, (rows, n)

    def _ensure_parent(self, m: _Map, rows: Optional[int] = None, edges: Optional[List[int]] = None) -> _Map:
        """rows / edges: the parent map's row count (and its clouds' row ranges) when the caller already knows them (build_pyramid) --
        no read-back then"""
        if m.parent is None:
            if m.bits <= 1:
                raise ValueError('cannot stride past the coordinate range')
            parent_of, pkeys, child_row, count = ops.coarsen(self._keys(m))
            cnt = int(count.item()) if rows is None else rows
            p = _Map(m.level + 1, m.bits - 1, cnt, pkeys[:cnt])
            p.edges = edges
            m.parent, m.parent_of, m.child_row = p, parent_of, child_row[:cnt]
            self._register(p, '')
        return m.parent

    def build_pyramid(self, key: CoordinateMapKey, levels: int) -> None:
        """Create the `levels` next coarser maps of `key` now, up front, so that no row count is read back in the middle of a long
        enqueue (it would stop the host from running ahead of the GPU).  The row counts of all levels come from one pass over the
        finest keys and ONE blocking read-back (ops.level_counts; a read-back per level -- ~0.1 ms each with the stream nearly
        empty -- before round 3)."""
        m = self._map(key)
        # levels still to build below the deepest existing ancestor: their row counts come from ONE pass over that map's keys
        while m.parent is not None and levels > 0:
            m, levels = m.parent, levels - 1
        levels = min(levels, m.bits - 1)
        if levels <= 0:
            return
        if self.independent_clouds and not m.generated:
            # the same single pass and read-back, per cloud: every level's row ranges are known on the host from here on
            per_level = ops.level_counts_clouds(self._keys(m), levels, 3 * m.bits, self._n_batch)
            m.edges = _edges_of(per_level[0])
            for counts in per_level[1:]:
                m = self._ensure_parent(m, sum(counts), _edges_of(counts))
            return
        rows = ops.level_counts(self._keys(m), levels) if not m.generated else [None] * levels
        for r in rows:
            m = self._ensure_parent(m, r)

    def stride(self, key: CoordinateMapKey, stride) -> CoordinateMapKey:
        """Key of the map `stride` times coarser (ME: cm.stride)."""
        s = _as_stride(stride)[0]
        m = self._map(key)
        while s > 1:
            m = self._ensure_parent(m)
            s >>= 1
        return m.key

    def _generated(self, m: _Map) -> _Map:
        """The set of all 8 children of every row of m (output map of a generative transposed convolution)."""
        if m.gen_child is None:
            if m.level < 1:
                raise ValueError('cannot upsample below tensor stride 1')
            g = _Map(m.level - 1, m.bits + 1, 8 * m.n, None)
            g.parent, g.generated = m, True
            m.gen_child = g
            # MinkowskiEngine gives the output map of a generative transposed convolution the empty string id; the reference
            # relies on it (geo_lossl_em.py:272 builds id + 'pruned', lossy_coord_v2/layers.py:155-157 then looks 'pruned' up
            # when the stride holds more than one map).  An id already taken at this stride gets a '#n' suffix (_register).
            self._register(g, '')
        return m.gen_child

    def _refine(self, parent: _Map, mask: torch.Tensor, string_id: str, count_hint: Optional[int] = None,
                cloud_counts: Optional[Sequence[int]] = None) -> _Map:
        """New map = children of `parent` selected by mask[8 * parent.n] (decoder side / pruning of a generated set).
        count_hint: number of set mask entries when the caller already knows it on the host (saves a blocking read-back);
        cloud_counts: the same per cloud of the batch (the new map's row ranges)."""
        keys, parent_of, child_row, count = ops.refine(parent.keys, mask)
        if cloud_counts is not None:
            count_hint = sum(cloud_counts)
        n = int(count.item()) if count_hint is None else int(count_hint)
        t = _Map(parent.level - 1, parent.bits + 1, n, keys[:n])
        if cloud_counts is not None:
            t.edges = _edges_of(cloud_counts)
        t.parent, t.parent_of, t.child_row = parent, parent_of[:n], child_row
        self._register(t, string_id)
        return t

    # -- queries ----------------------------------------------------------------------------------------------------
    def _keys(self, m: _Map) -> torch.Tensor:
        if m.keys is None:      # generated set: materialise lazily
            p = m.parent.keys
            m.keys = ((p << 3).unsqueeze(1) + torch.arange(8, device=p.device, dtype=torch.int64)).reshape(-1)
        return m.keys

    def _nbr27(self, m: _Map, want_rows: bool = False) -> torch.Tensor:
        """[27, n] int32 input row per (offset, output row) of the 3x3x3 kernel on m, offsets x fastest, -1 = absent.
        want_rows: the caller is going to run an MFMA layer in pattern order on m, which reads the table row-major and sorts the rows
        by their presence masks -- both are then written by the producer's pass (a map whose table only feeds gather_sum or a narrow
        layer does not pay the 132 bytes per row; should rows be wanted later after all, _nbr27_rows transposes the table).  A PARENT's
        table is always asked for with rows: every pyramid level of the codecs hosts 3x3x3 layers, and it is 4-8x smaller than the child's"""
        want_rows = want_rows and self.NBR_ROWS and m.n > self.ROW_ORDER_MIN_ROWS and m.nbr27_rows is None and m.nbr27_pos is None
        if m.nbr27 is None:
            if m.generated:
                if want_rows:
                    m.nbr27, m.nbr27_rows, masks = ops.nbr27_from_parent_ex(None, None, self._nbr27(m.parent, True), None, n=m.n)
                    if m.mask27 is None:
                        m.mask27 = masks
                else:
                    m.nbr27 = ops.nbr27_from_parent(None, None, self._nbr27(m.parent, True), None, n=m.n)
            else:
                if m.parent is None and m.n > self.ROOT_ROWS and m.bits > 1:
                    self._ensure_parent(m)
                if m.parent is not None:
                    if want_rows:
                        # a map this large runs its 3x3x3 layers in pattern order, which reads the table row-major and needs the rows'
                        # presence masks for the order: both come out of the producer's pass (round 6; was a transposition pass
                        # and a pass of 27 reads per row)
                        m.nbr27, m.nbr27_rows, masks = ops.nbr27_from_parent_ex(m.keys, m.parent_of, self._nbr27(m.parent, True), m.child_row)
                        if m.mask27 is None:
                            m.mask27 = masks
                    else:
                        m.nbr27 = ops.nbr27_from_parent(m.keys, m.parent_of, self._nbr27(m.parent, True), m.child_row)
                else:
                    m.nbr27 = ops.nbr27_search(m.keys, m.bits)
        return m.nbr27

    # FPCC_NBR_ROWS=0: the MFMA kernels read the offset-major table (A/B of the prologue; result-neutral)
    NBR_ROWS = os.environ.get('FPCC_NBR_ROWS', '1') != '0'

    def _nbr27_rows(self, m: _Map) -> Optional[torch.Tensor]:
        """the 3x3x3 table row-major, [n, 32] int32 (entries 27 .. 31 = -1): a row's 27 entries are one 128-byte line, which the MFMA
        kernels' prologue fetches as 16-byte pieces -- from the offset-major table it is 27 four-byte requests to 27 lines per row
        as soon as a block's rows are not consecutive (neighbour-pattern row order).  Built once per map, shared by its layers."""
        if not self.NBR_ROWS or not isinstance(m.row_order, torch.Tensor):
            return None                                  # consecutive rows read the offset-major table coalesced as it is
        if m.nbr27_pos is not None:
            return m.nbr27_pos                           # (callers below only test for None)
        if m.nbr27_rows is None:
            m.nbr27_rows = ops.transpose_table(self._nbr27(m), 32)
        return m.nbr27_rows

    def _k3_table(self, m: _Map, mfma: bool, row_order: Optional[torch.Tensor] = None) -> dict:
        """neighbour-table arguments of a 3x3x3 conv_f32 call on map m (beside `row_order`: the row-major table in position order)"""
        rows = self._nbr27_rows(m) if (mfma and row_order is not None) else None
        if rows is None:                                 # consecutive rows read the offset-major table coalesced as it is
            return dict(nbr=self._nbr27(m), n_offsets=27, nbr_ks=m.n, nbr_os=1)
        if row_order is not m.row_order:
            raise ValueError('a foreign row order')
        if m.nbr27_pos is None:
            m.nbr27_pos = ops.gather_table_rows(rows, row_order)              # 128-byte rows: one gather per map
            m.nbr27_rows = None                                              # only the position-ordered copy is read from here on
        return dict(nbr=m.nbr27_pos, n_offsets=27, nbr_ks=1, nbr_os=32)

    def _mask27(self, m: _Map) -> Optional[torch.Tensor]:
        """int32 [n]: which of the 27 neighbours of every row exist, derived from the parent level without building the row table;
        None where the map has no parent to derive it from (the small top of the pyramid)"""
        if m.mask27 is None:
            if m.generated or m.nbr27 is not None:       # the table exists already: the general kernel may as well use it
                return None
            if m.parent is None and m.n > self.ROOT_ROWS and m.bits > 1:
                self._ensure_parent(m)
            if m.parent is None:
                return None
            m.mask27 = ops.mask27_from_parent(m.keys, m.parent_of, self._nbr27(m.parent, True), m.child_row)
        return m.mask27

    # maps with more rows than this run their 3x3x3 convolutions in neighbour-pattern order (fpcc_conv_row_keys)
    ROW_ORDER_MIN_ROWS = int(os.environ.get('FPCC_ROW_ORDER_MIN_ROWS', '8192'))
    ROW_ORDER_WINDOW_LOG2 = int(os.environ.get('FPCC_ROW_WINDOW_LOG2', '19'))

    ROW_ORDER_MIN_ROWS_TRAINING = 512         # the weight gradient skips absent (row block, offset) pairs on any map

    def _row_order(self, m: _Map, training: bool = False) -> Optional[torch.Tensor]:
        """permutation of m's rows that groups like neighbour patterns into the same 32-row MFMA block; cached per map and
        shared by every 3x3x3 layer on it"""
        if m.row_order is False or (m.row_order is None and training and m.n > self.ROW_ORDER_MIN_ROWS_TRAINING):
            m.row_order = None
            if m.n > (self.ROW_ORDER_MIN_ROWS_TRAINING if training else self.ROW_ORDER_MIN_ROWS):
                nbr = self._nbr27(m, want_rows=not training)
                # the masks of a table made by nbr27_from_parent_ex describe exactly that table (a mask derived earlier from the parent
                # level does too: same rule); without them the keys are read off the table
                m.row_order = ops.conv_row_order(nbr, 27, m.n, 1, m.n, self.ROW_ORDER_WINDOW_LOG2,
                                                 masks=m.mask27 if m.nbr27_rows is not None else None)
        return m.row_order

    def _k2_order(self, src: _Map):
        """(row order, table) of the stride-2 2x2x2 convolution from `src` onto its parent on the MFMA path.  A parent has ~3.7 of its 8
        children on a surface, yet in Morton order nearly every 32-row block of parents has all 8 octants between its rows, so the
        kernel executed all 8 offsets for every block (0.33 of the matrix peak on the large maps, rounds 2-5).  Grouping the parents by
        their 8-bit child-presence pattern (the machinery of the 3x3x3 layers: fpcc_conv_row_keys + sort, tiles heaviest first) lets a
        block skip the octants none of its rows has; the table travels in position order like the 3x3x3 one.  Results do not depend on
        the order (every output row is its own chain).  -> (None, child_row) on small maps."""
        m = src.parent.n
        if src.k2_order is False:
            src.k2_order = None
            if self.NBR_ROWS and m > self.ROW_ORDER_MIN_ROWS and os.environ.get('FPCC_K2_ROW_ORDER', '1') != '0':
                src.k2_order = ops.conv_row_order(src.child_row, 8, 1, 8, m, self.ROW_ORDER_WINDOW_LOG2)
                src.k2_pos = ops.gather_table_rows(src.child_row, src.k2_order)
        if src.k2_order is None:
            return None, src.child_row
        return src.k2_order, src.k2_pos

    def get_coordinates(self, key: CoordinateMapKey) -> torch.Tensor:
        m = self._map(key)
        if m.coords is None:
            m.coords = ops.coords_from_keys(self._keys(m), m.level, m.bits)
        return m.coords

    def kernel_map(self, in_key: CoordinateMapKey, out_key: CoordinateMapKey, stride=1, kernel_size=1, **_):
        """{kernel index k: int64 [2, L_k]} with row 0 = input rows, row 1 = output rows (ME: cm.kernel_map), kernel indexes
        enumerated x fastest.  kernel_size 1 (membership of `in` rows in `out`, the form the codecs use:
        geo_lossl_em.py:313, lossy_coord_v2/layers.py:186); 3 on one map (the 27-neighbour table); 2 with stride 2 from a map
        to its stride-2 parent (the child table)."""
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        st = stride if isinstance(stride, int) else stride[0]
        a, b = self._map(in_key), self._map(out_key)
        if ks == 3 and st == 1:
            if a is not b:
                raise NotImplementedError('kernel_map(kernel_size=3) is provided on one coordinate map')
            nbr = self._nbr27(a)
            out = {}
            for k in range(27):
                rows_out = torch.nonzero(nbr[k] >= 0).squeeze(1)
                if rows_out.numel():
                    out[k] = torch.stack((nbr[k][rows_out].long(), rows_out))
            return out
        if ks == 2 and st == 2:
            if a.parent is not b or a.generated:
                raise NotImplementedError('kernel_map(kernel_size=2, stride=2) maps a level onto its stride-2 parent')
            out = {}
            for k in range(8):
                rows_out = torch.nonzero(a.child_row[:, k] >= 0).squeeze(1)
                if rows_out.numel():
                    out[k] = torch.stack((a.child_row[rows_out, k].long(), rows_out))
            return out
        if ks != 1:
            raise NotImplementedError(f'kernel_map(kernel_size={ks}, stride={st})')
        if a.level != b.level:
            raise NotImplementedError('kernel_map(kernel_size=1) needs maps of equal stride')
        ka, kb = self._keys(a), self._keys(b)
        if kb.numel() == 0 or ka.numel() == 0:
            return {}
        pos = torch.searchsorted(kb, ka).clamp_(max=kb.numel() - 1)
        hit = kb[pos] == ka
        rows_in = torch.nonzero(hit).squeeze(1)
        return {0: torch.stack((rows_in, pos[hit]))}

    def origin_map(self, key: CoordinateMapKey):
        """(key of the origin map, [rows of sample b for every b]) -- ME: cm.origin_map, whose second element the colour
        codec's training loss indexes per sample (lossy_coord_lossy_color/layers.py:247).  Rows are batch-major here."""
        m = self._map(key)
        edges = self.batch_offsets(m)
        dev = self.device if m.keys is None else m.keys.device
        return CoordinateMapKey(1, 'origin'), [torch.arange(a, b, device=dev) for a, b in zip(edges[:-1], edges[1:])]

    def _ancestor_rows(self, m: _Map, target: _Map) -> torch.Tensor:
        """row of `target` (a coarser map on m's parent chain) that holds each row of m"""
        if m is target:
            raise ValueError('the maps are the same')
        dev = self.device if m.keys is None and m.parent.keys is None else (m.keys if m.keys is not None else m.parent.keys).device
        idx = torch.arange(m.n, device=dev)
        cur = m
        while cur is not target:
            if cur.parent is None:
                raise ValueError('the target map is not an ancestor of the input map')
            idx = torch.div(idx, 8, rounding_mode='floor') if cur.generated else cur.parent_of.long()[idx]
            cur = cur.parent
        return idx

    def batch_offsets(self, m: _Map) -> List[int]:
        """Row ranges of the samples of a batch (rows are batch-major).  Cached on the map; known without a read-back where the
        pyramid was built for independent clouds (build_pyramid, _refine with cloud_counts, generated sets)."""
        if m.edges is not None:
            return m.edges
        if m.n == 0:
            return [0]
        if self._n_batch == 1:
            m.edges = [0, m.n]
        elif m.generated and m.parent is not None and m.parent.n * 8 == m.n:
            m.edges = [8 * e for e in self.batch_offsets(m.parent)]
        else:
            b = (self._keys(m) >> (3 * m.bits))
            nb = self._n_batch if self.independent_clouds else int(b[-1].item()) + 1
            m.edges = torch.searchsorted(b, torch.arange(nb + 1, device=b.device, dtype=b.dtype)).tolist()
        return m.edges

    def cloud_rows(self, m: _Map) -> List[int]:
        """the row counts a size-dependent rule must look at: the map's own, or -- independent clouds -- every cloud's"""
        if not self.independent_clouds:
            return [m.n]
        e = self.batch_offsets(m)
        return [b - a for a, b in zip(e[:-1], e[1:])]


def _edges_of(counts: Sequence[int]) -> List[int]:
    edges = [0]
    for c in counts:
        edges.append(edges[-1] + int(c))
    return edges


# ---------------------------------------------------------------------------------------------------------------
class SparseTensor:
    """Features [n, C] on a coordinate map.  `features` rows follow the map's (Morton) row order."""

    def __init__(self, features: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], coordinates=None,
                 tensor_stride=1, coordinate_map_key: Optional[CoordinateMapKey] = None,
                 coordinate_manager: Optional[CoordinateManager] = None,
                 quantization_mode: SparseTensorQuantizationMode = SparseTensorQuantizationMode.RANDOM_SUBSAMPLE,
                 **_):
        if coordinate_manager is None:
            # the reference's implicit-manager idiom (ME.SparseTensor(feats, coordinates=...) with no manager): the calling thread's
            # global manager when the operation mode shares one, a fresh manager otherwise (MinkowskiEngine's two operation modes)
            coordinate_manager = global_coordinate_manager() \
                if _operation_mode == SparseTensorOperationMode.SHARE_COORDINATE_MANAGER else None
            if coordinate_manager is None:
                coordinate_manager = CoordinateManager()
                if _operation_mode == SparseTensorOperationMode.SHARE_COORDINATE_MANAGER:
                    set_global_coordinate_manager(coordinate_manager)
        self.coordinate_manager = coordinate_manager
        if coordinates is not None:
            if coordinate_map_key is not None:
                raise ValueError('give coordinates or a coordinate_map_key, not both')
            key, (rows, n) = coordinate_manager.insert_and_map(coordinates, tensor_stride)
            if n != coordinates.shape[0] and quantization_mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE:
                features = _average_duplicates(features, coordinate_manager, key, coordinates, tensor_stride)
            else:
                features = ops.gather_rows(features.contiguous(), rows.contiguous())
            coordinate_map_key = key
        elif coordinate_map_key is None:
            raise ValueError('coordinates or coordinate_map_key required')
        self.coordinate_map_key = coordinate_map_key
        self._parts = features if isinstance(features, tuple) else (features,)
        n_rows = coordinate_manager._map(coordinate_map_key).n
        for f in self._parts:
            if f.shape[0] != n_rows:
                raise ValueError(f'features have {f.shape[0]} rows, the coordinate map {n_rows}')

    # features ------------------------------------------------------------------------------------------------------
    @property
    def F(self) -> torch.Tensor:
        if len(self._parts) > 1:
            self._parts = (torch.cat(self._parts, dim=1),)
        return self._parts[0]

    @property
    def parts(self) -> Tuple[torch.Tensor, ...]:
        return self._parts

    @property
    def C(self) -> torch.Tensor:
        return self.coordinate_manager.get_coordinates(self.coordinate_map_key)

    @property
    def tensor_stride(self) -> List[int]:
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def shape(self):
        return torch.Size((self._parts[0].shape[0], sum(p.shape[1] for p in self._parts)))

    @property
    def device(self):
        return self._parts[0].device

    @property
    def dtype(self):
        return self._parts[0].dtype

    def size(self):
        return self.shape

    # batch decomposition -------------------------------------------------------------------------------------------
    @property
    def _batchwise_row_indices(self):
        edges = self.coordinate_manager.batch_offsets(self.coordinate_manager._map(self.coordinate_map_key))
        return [torch.arange(a, b, device=self.device) for a, b in zip(edges[:-1], edges[1:])]

    @property
    def decomposition_permutations(self):
        return self._batchwise_row_indices

    @property
    def decomposed_coordinates(self):
        c = self.C
        return [c[idx, 1:] for idx in self._batchwise_row_indices]

    def __repr__(self):
        return f'SparseTensor(shape={tuple(self.shape)}, key={self.coordinate_map_key})'


def _average_duplicates(features, cm, key, coordinates, tensor_stride):
    # rare path (inputs of the codec are unique): average the features of equal coordinates with torch ops
    m = cm._map(key)
    level = m.level
    keys = ops.keys_from_coords(coordinates.contiguous(), level, m.bits)
    row = torch.searchsorted(m.keys, keys)
    out = torch.zeros((m.n, features.shape[1]), dtype=features.dtype, device=features.device)
    cnt = torch.zeros((m.n, 1), dtype=features.dtype, device=features.device)
    out.index_add_(0, row, features)
    cnt.index_add_(0, row, torch.ones_like(features[:, :1]))
    return out / cnt


def cat(*tensors) -> SparseTensor:
    """Channel concatenation on one coordinate map (ME.cat).  Lazy: a following convolution reads both sources."""
    if len(tensors) == 1 and isinstance(tensors[0], (tuple, list)):
        tensors = tuple(tensors[0])
    key, cm = tensors[0].coordinate_map_key, tensors[0].coordinate_manager
    parts = []
    for t in tensors:
        if t.coordinate_manager is not cm or cm._map(t.coordinate_map_key) is not cm._map(key):
            raise ValueError('cat needs tensors on the same coordinate map')
        parts.extend(t.parts)
    if len(parts) > 2:
        parts = [torch.cat(parts[:-1], dim=1), parts[-1]]
    return SparseTensor(tuple(parts), coordinate_map_key=key, coordinate_manager=cm)


# ---------------------------------------------------------------------------------------------------------------
# modules
class _Act:
    """What a convolution may fuse after its bias."""
    __slots__ = ('kind', 'slope', 'param')

    def __init__(self, kind=ops.ACT_NONE, slope=None, param=None):
        self.kind, self.slope, self.param = kind, slope, param       # param: the live nn.Parameter (training path)


def _act_of(module: Optional[nn.Module]) -> _Act:
    if module is None:
        return _Act()
    if isinstance(module, MinkowskiPReLU):
        w = module.module.weight
        cached = module.__dict__.get('_fpcc_act')
        if cached is None or cached[0] != w.data_ptr():      # the detached view shares storage: in-place updates show through
            if w.numel() != 1:
                raise NotImplementedError('per-channel PReLU')
            cached = (w.data_ptr(), _Act(ops.ACT_PRELU, w.detach(), w))
            module.__dict__['_fpcc_act'] = cached
        return cached[1]
    if isinstance(module, MinkowskiReLU):
        return _Act(ops.ACT_RELU)
    raise NotImplementedError(f'cannot fuse {type(module).__name__}')


def _packed_gen_ok(c_in: int, c_out: int) -> bool:
    """a generative transposed convolution whose 8 octant kernels fit one (or two) MFMA GEMMs of 8*C_out columns"""
    return c_in % 16 == 0 and 8 * c_out in (32, 64, 128, 256)


def _split_k3_ok(c_in: int, c2: int, c_out: int) -> bool:
    """a 3x3x3 convolution with one output channel: dot products per input row on the MFMA kernel + scalar gather"""
    return c_out == 1 and c2 == 0 and c_in % 16 == 0


PAD_MIN_ROWS = 8192        # FPCC_PAD_MIN_ROWS of include/fpcc_hip.h: part of the stream format, not a tuning knob


@functools.lru_cache(maxsize=4096)
def _pad_plan(c1: int, c2: int, c_out: int, n_out: int):
    """Shapes the MFMA kernel does not take as they are (C_in not a multiple of 16, C_out not 32/64/128) but that are big
    enough to matter are zero-padded to the next MFMA shape: -> (c1p, c2p, c_outp) or None.  Zero channels add exact
    zeros to the FMA chains, so only the (documented) summation order changes.
    INVARIANT (the batch interface rests on it): n_out is the row count of the CLOUD a row belongs to, never of the launch -- encoder
    and decoder may batch different sets of clouds together (compress_partitions groups by voxel counts, decompress_partitions by a
    header proxy), so any rule that selects a summation order from a row count must look at the cloud's own rows
    (tests/test_gpu_codec_many.py::test_encoder_and_decoder_may_group_a_partition_list_differently)."""
    if ops.conv_order(c1, c2, c_out) != 0 or c_out > 128 or c1 + c2 < 4 or (c1 + c2) * c_out < 32 or n_out < PAD_MIN_ROWS:
        return None
    c1p = (c1 + 15) // 16 * 16
    c2p = (c2 + 15) // 16 * 16 if c2 else 0
    c_outp = 32 if c_out <= 32 else (64 if c_out <= 64 else 128)
    return c1p, c2p, c_outp


def summation_order(kind: str, c1: int, c2: int, c_out: int, n_out: int = 0) -> int:
    """Which documented fp32 summation order (include/fpcc_hip.h) a layer of this shape and size is evaluated in:
    0 natural chain, 1 MFMA chain (0,4,1,5,2,6,3,7 inside groups of 8 channels), 2 per-offset chains then offset sum.
    kind: 'k1' | 'k3' | 'k2s2' | 'k2s2T' | 'gen' | 'mlp'; n_out = output rows of the launch.  Tests hand this to the
    oracle to compare bit for bit."""
    if kind == 'gen' and c2 == 0 and _packed_gen_ok(c1, c_out):
        return 1
    if kind == 'k3' and _split_k3_ok(c1, c2, c_out):
        return 2
    n_off = {'k3': 27, 'k2s2': 8}.get(kind, 1)
    groups = 8 if kind in ('gen', 'k2s2T') else 1
    if kind in ('k1', 'k3'):
        plan = _pad_plan(c1, c2, c_out, n_out)
        if plan is not None:
            return ops.conv_order(plan[0], plan[1], plan[2], n_off, groups, n_out)
    return ops.conv_order(c1, c2, c_out, n_off, groups, n_out)


class _ConvBase(nn.Module):
    TRANSPOSED = False
    GENERATIVE = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator: Optional[KernelGenerator] = None, expand_coordinates=False, dimension=3):
        super().__init__()
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size, stride, dilation, dimension=dimension)
        kg = kernel_generator
        if len(set(kg.kernel_size)) != 1 or len(set(kg.kernel_stride)) != 1 or set(kg.kernel_dilation) != {1}:
            raise NotImplementedError('isotropic kernels without dilation only')
        self.kernel_generator = kg
        self.in_channels, self.out_channels = in_channels, out_channels
        self.ks, self.st = kg.kernel_size[0], kg.kernel_stride[0]
        if (self.ks, self.st) not in ((1, 1), (3, 1), (2, 2)):
            raise NotImplementedError(f'kernel {self.ks} stride {self.st}')
        if (self.TRANSPOSED or self.GENERATIVE) and (self.ks, self.st) != (2, 2):
            raise NotImplementedError('transposed convolutions are kernel 2 stride 2')
        volume = kg.kernel_volume
        shape = (in_channels, out_channels) if volume == 1 else (volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self._packed = None           # derived weight layouts, rebuilt when the parameters change
        self._packed_tag = None
        self.reset_parameters()

    def _derived(self):
        """weight layouts derived from `kernel`/`bias` for the fused evaluations below (cached)"""
        tag = (self.kernel._version, self.kernel.data_ptr(), None if self.bias is None else self.bias._version)
        if self._packed_tag != tag:
            k = self.kernel.detach()
            b = None if self.bias is None else self.bias.detach().view(-1)
            d = {'w': k, 'b': b}
            if self.GENERATIVE and _packed_gen_ok(self.in_channels, self.out_channels):
                w_all = k.permute(1, 0, 2).reshape(self.in_channels, 8 * self.out_channels).contiguous()
                d['gen_w'] = [w_all] if w_all.shape[1] <= 128 else [w_all[:, :128].contiguous(), w_all[:, 128:].contiguous()]
                b_all = None if b is None else b.repeat(8)
                d['gen_b'] = [b_all] if (b_all is None or b_all.numel() <= 128) else [b_all[:128].contiguous(), b_all[128:].contiguous()]
            if not (self.GENERATIVE or self.TRANSPOSED) and self.ks == 3 and _split_k3_ok(self.in_channels, 0, self.out_channels):
                wt = torch.zeros((self.in_channels, 32), dtype=k.dtype, device=k.device)
                wt[:, :27] = k[:, :, 0].t()
                d['k3_w'] = wt
            self._packed, self._packed_tag = d, tag
        return self._packed

    def _padded_weights(self, c1: int, c2: int, plan):
        """kernel / bias zero-padded to the MFMA shape `plan` = (c1p, c2p, c_outp) (cached)"""
        d = self._derived()
        key = ('pad', c1, c2) + tuple(plan)
        if key not in d:
            c1p, c2p, cop = plan
            k = self.kernel.detach()
            k = k.reshape(-1, k.shape[-2], k.shape[-1])
            w = torch.zeros((k.shape[0], c1p + c2p, cop), dtype=k.dtype, device=k.device)
            w[:, :c1, :self.out_channels] = k[:, :c1]
            if c2:
                w[:, c1p:c1p + c2, :self.out_channels] = k[:, c1:]
            b = None
            if self.bias is not None:
                b = torch.zeros(cop, dtype=k.dtype, device=k.device)
                b[:self.out_channels] = self.bias.detach().view(-1)
            d[key] = (w, b)
        return d[key]

    @staticmethod
    def _pad_cols(x: Optional[torch.Tensor], width: int) -> Optional[torch.Tensor]:
        if x is None or x.shape[1] == width:
            return x
        return torch.nn.functional.pad(x, (0, width - x.shape[1]))

    def _forward_autograd(self, x: SparseTensor, cm, src: _Map, coordinates, act: _Act, clip: float) -> SparseTensor:
        """training path: the convolution as an autograd function (fastpcc_amd/autograd.py), epilogue as tensor ops"""
        from .autograd import ConvSpec, sparse_conv
        feats = x.F
        if self.GENERATIVE:
            dst = cm._generated(src)
            spec = ConvSpec('gen', src.n, dst.n)
        elif self.TRANSPOSED:
            if coordinates is None:
                raise ValueError('a transposed convolution needs the target coordinate key')
            dst = cm._map(coordinates)
            if dst.parent is not src:
                raise ValueError('target map is not a stride-2 child of the input map')
            spec = ConvSpec('gen', src.n, dst.n) if dst.generated else ConvSpec('k2s2T', src.n, dst.n, dst.child_row)
        elif self.ks == 1:
            dst = src
            spec = ConvSpec('k1', src.n, src.n)
        elif self.ks == 3:
            dst = src
            if coordinates is not None and cm._map(coordinates) is not src:
                raise NotImplementedError('stride-1 convolution onto a different coordinate map')
            spec = ConvSpec('k3', src.n, src.n, cm._nbr27(src), cm._row_order(src, training=True))
        else:
            dst = cm._ensure_parent(src)
            if src.generated:
                raise NotImplementedError('stride-2 convolution of a generated set')
            spec = ConvSpec('k2s2', src.n, dst.n, src.child_row)
        if _fusable_in_training(act):
            from .autograd import sparse_conv_act
            out = sparse_conv_act(feats, self.kernel, self.bias, act.param if act.kind == ops.ACT_PRELU else None, spec, act.kind)
            out = _finish_autograd(out, None, _Act(), clip)
        else:
            out = _finish_autograd(sparse_conv(feats, self.kernel, spec), self.bias, act, clip)
        return SparseTensor(out, coordinate_map_key=dst.key, coordinate_manager=cm)

    def reset_parameters(self):
        # MinkowskiEngine 0.5's default: U(-1/sqrt(n), 1/sqrt(n)), n = (C_out if transposed else C_in) * volume
        fan = (self.out_channels if (self.TRANSPOSED or self.GENERATIVE) else self.in_channels) * \
            self.kernel_generator.kernel_volume
        bound = 1.0 / math.sqrt(fan)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x: SparseTensor, coordinates: Optional[CoordinateMapKey] = None, *, act: Optional[_Act] = None,
                clip: float = 0.0) -> SparseTensor:
        act = act or _Act()
        cm = x.coordinate_manager
        src = cm._map(x.coordinate_map_key)
        if torch.is_grad_enabled() and (self.kernel.requires_grad or any(p.requires_grad for p in x.parts)):
            return self._forward_autograd(x, cm, src, coordinates, act, clip)
        plan_rows = src.n
        if cm.independent_clouds and self.ks in (1, 3) and not (self.TRANSPOSED or self.GENERATIVE):
            # PAD_MIN_ROWS is a rule about ONE cloud's map (it selects the summation order, which is part of the stream format): in a
            # batch of independent clouds every cloud gets the evaluation it would get alone.  Clouds on either side of the threshold
            # (a small level of one cloud beside the same level of a larger one): the layer runs in both forms over the union and every
            # cloud takes its rows from its own -- twice the work on a level that is small by definition.
            c1, c2 = x.parts[0].shape[1], (x.parts[1].shape[1] if len(x.parts) > 1 else 0)
            edges = cm.batch_offsets(src)
            padded = [_pad_plan(c1, c2, self.out_channels, b - a) is not None for a, b in zip(edges[:-1], edges[1:]) if b > a]
            if any(padded) and not all(padded):
                big = self._forward_fused(x, cm, src, coordinates, act, clip, PAD_MIN_ROWS).F
                small = self._forward_fused(x, cm, src, coordinates, act, clip, 0).F
                out = torch.empty_like(small)
                for a, b in zip(edges[:-1], edges[1:]):
                    if b > a:
                        out[a:b] = (big if _pad_plan(c1, c2, self.out_channels, b - a) is not None else small)[a:b]
                return SparseTensor(out, coordinate_map_key=src.key, coordinate_manager=cm)
            plan_rows = PAD_MIN_ROWS if any(padded) else 0
        return self._forward_fused(x, cm, src, coordinates, act, clip, plan_rows)

    def _forward_fused(self, x: SparseTensor, cm: CoordinateManager, src: _Map, coordinates: Optional[CoordinateMapKey], act: _Act,
                       clip: float, plan_rows: int) -> SparseTensor:
        """inference path: one fused launch.  plan_rows: the row count PAD_MIN_ROWS is compared with (the map's, or its clouds')"""
        parts = x.parts
        x1 = parts[0]
        x2 = parts[1] if len(parts) > 1 else None
        derived = self._derived()
        kw = dict(x2=x2, bias=derived['b'], act=act.kind, slope=act.slope, clip=clip, pack=True)
        w = derived['w']
        c_out = self.out_channels
        if self.GENERATIVE:
            dst = cm._generated(src)
            d = self._derived()
            if x2 is None and 'gen_w' in d:
                # all 8 octant kernels side by side: one dense GEMM [n, C_in] @ [C_in, 8*C_out]; its row-major output IS
                # the generated tensor [8n, C_out] (row 8*parent + octant)
                halves = d['gen_w']
                wide = 8 * c_out
                out = torch.empty((src.n, wide), dtype=torch.float32, device=x1.device)
                for h, wh in enumerate(halves):
                    cols = wh.shape[1]
                    bh = d['gen_b'][h] if d['gen_b'][0] is not None else None
                    ops.conv_f32(x1, wh, cols, src.n, bias=bh, act=act.kind, slope=act.slope, clip=clip,
                                 out=out[:, h * 128: h * 128 + cols], pack=True)
                out = out.view(8 * src.n, c_out)
            else:
                out = ops.conv_f32(x1, w, c_out, src.n, groups=8, **kw)
        elif self.TRANSPOSED:
            if coordinates is None:
                raise ValueError('a transposed convolution needs the target coordinate key')
            dst = cm._map(coordinates)
            if dst.parent is not src:
                raise ValueError('target map is not a stride-2 child of the input map')
            if dst.generated:
                out = ops.conv_f32(x1, w, c_out, src.n, groups=8, **kw)
            else:
                out = ops.conv_f32(x1, w, c_out, src.n, groups=8, out_map=dst.child_row, om_os=8, om_gs=1,
                                   out_rows=dst.n, **kw)
        elif self.ks == 1:
            dst = src
            plan = _pad_plan(x1.shape[1], 0 if x2 is None else x2.shape[1], c_out, plan_rows)
            if plan is not None:
                wp, bp = self._padded_weights(x1.shape[1], 0 if x2 is None else x2.shape[1], plan)
                out = ops.conv_f32(self._pad_cols(x1, plan[0]), wp, plan[2], src.n, x2=self._pad_cols(x2, plan[1]), bias=bp,
                                   act=act.kind, slope=act.slope, clip=clip, pack=True)[:, :c_out]
            else:
                out = ops.conv_f32(x1, w, c_out, src.n, **kw)
        elif self.ks == 3:
            dst = src
            if coordinates is not None and cm._map(coordinates) is not src:
                raise NotImplementedError('stride-1 convolution onto a different coordinate map')
            d = self._derived()
            plan = _pad_plan(x1.shape[1], 0 if x2 is None else x2.shape[1], c_out, plan_rows)
            if x2 is None and 'k3_w' in d:
                y = ops.conv_f32(x1, d['k3_w'], 32, src.n, pack=True)       # per input row: its dot product with every offset's kernel
                if src.generated and src.nbr27 is None and src.n == 8 * src.parent.n:
                    # a full generated set (the occupancy predictors' 8 candidates per voxel): the neighbour rows follow from the
                    # PARENT's table in registers -- the candidates' own table (108 bytes per row) is never written or read
                    out = ops.gather_sum_generated(y, cm._nbr27(src.parent, True), bias=kw['bias'], act=act.kind, slope=act.slope, clip=clip)
                else:
                    out = ops.gather_sum(y, cm._nbr27(src), 27, src.n, 1, src.n, bias=kw['bias'], act=act.kind,
                                         slope=act.slope, clip=clip)
            elif plan is not None:
                wp, bp = self._padded_weights(x1.shape[1], 0 if x2 is None else x2.shape[1], plan)
                ro = cm._row_order(src) if plan[0] + plan[1] > 16 else None
                out = ops.conv_f32(self._pad_cols(x1, plan[0]), wp, plan[2], src.n, x2=self._pad_cols(x2, plan[1]), bias=bp,
                                   act=act.kind, slope=act.slope, **cm._k3_table(src, True, ro),
                                   clip=clip, row_order=ro, pack=True)[:, :c_out]
                if c_out < 8:
                    out = out.contiguous()
            elif x2 is None and x1.shape[1] == 1 and getattr(x, '_fpcc_all_ones', False) and 4 <= c_out <= 32 and c_out % 4 == 0 \
                    and ops.conv_order(1, 0, c_out) == 0 and cm._mask27(src) is not None:
                # the codec's first layer: every voxel carries the feature 1, so a row's sum only depends on WHICH neighbours
                # exist -- evaluated from 27-bit masks (same FMA chain, same bits); the 108-byte-per-row neighbour table of the
                # finest level is then never built
                out = ops.conv_ones_k3(cm._mask27(src), w, c_out, bias=kw['bias'], act=act.kind, slope=act.slope, clip=clip)
            else:
                mfma = ops.conv_order(x1.shape[1], 0 if x2 is None else x2.shape[1], c_out) != 0
                # 16 input channels: one 64-byte gather per neighbour -- Morton locality beats block skipping
                ro = cm._row_order(src) if mfma and x1.shape[1] + (0 if x2 is None else x2.shape[1]) > 16 else None
                out = ops.conv_f32(x1, w, c_out, src.n, **cm._k3_table(src, mfma and kw.get('pack', False) is True, ro), row_order=ro, **kw)
        else:   # kernel 2, stride 2
            dst = cm._ensure_parent(src)
            if src.generated:
                raise NotImplementedError('stride-2 convolution of a generated set')
            c_in = x1.shape[1] + (0 if x2 is None else x2.shape[1])
            # (16 input channels: the one such layer -- 16 -> 64 onto the stride-2 map -- gains 0.19 ms per batch in pattern order and its
            # order costs more than that to build: measured, left in Morton order)
            if kw.get('pack', False) is True and c_in > 16 and ops.conv_order(x1.shape[1], 0 if x2 is None else x2.shape[1], c_out, 8, 1, dst.n) != 0:
                ro, table = cm._k2_order(src)              # grouped MFMA shapes: parents in child-pattern order on large maps
            else:
                ro, table = None, src.child_row
            out = ops.conv_f32(x1, w, c_out, dst.n, nbr=table, n_offsets=8, nbr_ks=1, nbr_os=8, row_order=ro, **kw)
        return SparseTensor(out, coordinate_map_key=dst.key, coordinate_manager=cm)


class MinkowskiConvolution(_ConvBase):
    pass


def _fusable_in_training(act: _Act) -> bool:
    """the fused training node recovers the activation's derivative from the layer output, which needs a PReLU slope > 0
    (true of every trained checkpoint; the slope starts at 0.25).  The sign is read from a host-side cache that
    fastpcc_amd.train.Trainer refreshes after every optimiser step; without a cache it is fetched (one sync)."""
    if act.kind != ops.ACT_PRELU:
        return True
    if act.param is None:
        return False
    # (parameter version, value): an optimiser step, load_state_dict() or an EMA copy bumps `_version`, which is a host-side
    # integer -- a stale sign can therefore never select the fused backward
    cached = getattr(act.param, '_fpcc_host_value', None)
    if cached is None or cached[0] != act.param._version:
        cached = (act.param._version, float(act.param.detach().reshape(-1)[0].item()))
        act.param._fpcc_host_value = cached
    return cached[1] > 0


def refresh_prelu_cache(model: nn.Module) -> None:
    """one device->host transfer of all single-slope PReLU parameters (call after the optimiser changed them)"""
    ps = [m.module.weight for m in model.modules() if isinstance(m, MinkowskiPReLU) and m.module.weight.numel() == 1]
    if ps:
        for p, v in zip(ps, torch.cat([p.detach().reshape(1) for p in ps]).tolist()):
            p._fpcc_host_value = (p._version, v)


def _finish_autograd(out: torch.Tensor, bias: Optional[torch.Tensor], act: _Act, clip: float) -> torch.Tensor:
    """bias, activation and clamp of the training path as differentiable tensor ops (inference fuses them into the kernel)"""
    if bias is not None:
        out = out + bias.view(1, -1)
    if act.kind == ops.ACT_PRELU:
        out = torch.nn.functional.prelu(out, act.param if act.param is not None else act.slope)
    elif act.kind == ops.ACT_RELU:
        out = torch.relu(out)
    if clip > 0:
        from .autograd import BoundFunction
        from .entropy_models import scalar_tensor
        out = BoundFunction.apply(out, scalar_tensor(float(clip), torch.float32, out.device).view(()))
    return out


class MinkowskiConvolutionTranspose(_ConvBase):
    TRANSPOSED = True


class MinkowskiGenerativeConvolutionTranspose(_ConvBase):
    GENERATIVE = True


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self._wt = None
        self._wt_version = None

    def _weight_t(self) -> torch.Tensor:
        w = self.linear.weight
        if self._wt is None or self._wt_version != (w._version, w.data_ptr()):
            self._wt = w.detach().t().contiguous()
            self._wt_version = (w._version, w.data_ptr())
        return self._wt

    def forward(self, x: SparseTensor, *, act: Optional[_Act] = None, clip: float = 0.0) -> SparseTensor:
        act = act or _Act()
        if torch.is_grad_enabled() and (self.linear.weight.requires_grad or any(p.requires_grad for p in x.parts)):
            from .autograd import ConvSpec, sparse_conv
            n = x.parts[0].shape[0]
            if _fusable_in_training(act):
                from .autograd import sparse_linear_act
                out = sparse_linear_act(x.F, self.linear.weight, self.linear.bias, act.param if act.kind == ops.ACT_PRELU else None, act.kind)
                out = _finish_autograd(out, None, _Act(), clip)
            else:
                out = _finish_autograd(sparse_conv(x.F, self.linear.weight.t(), ConvSpec('k1', n, n)), self.linear.bias, act, clip)
            return SparseTensor(out, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        parts = x.parts
        b = self.linear.bias
        out = ops.conv_f32(parts[0], self._weight_t(), self.linear.out_features, parts[0].shape[0],
                           x2=parts[1] if len(parts) > 1 else None, bias=None if b is None else b.detach(),
                           act=act.kind, slope=act.slope, clip=clip, pack=True)
        return SparseTensor(out, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


class _Pointwise(nn.Module):
    MODULE = None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.module = self.MODULE(*args, **kwargs)

    def forward(self, x: SparseTensor) -> SparseTensor:
        return SparseTensor(self.module(x.F), coordinate_map_key=x.coordinate_map_key,
                            coordinate_manager=x.coordinate_manager)


class MinkowskiReLU(_Pointwise):
    MODULE = nn.ReLU


class MinkowskiPReLU(_Pointwise):
    MODULE = nn.PReLU


class MinkowskiLeakyReLU(_Pointwise):
    MODULE = nn.LeakyReLU


class MinkowskiSigmoid(_Pointwise):
    MODULE = nn.Sigmoid


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm over the feature rows; `.bn` is the nn.BatchNorm1d, as in MinkowskiEngine (state_dict keys `bn.*`)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)

    def forward(self, x: SparseTensor) -> SparseTensor:
        return SparseTensor(self.bn(x.F), coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


class _StridedPool(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=3, **_):
        super().__init__()
        ks, st = _as_stride(kernel_size)[0], _as_stride(stride)[0]
        if ks != st:
            raise NotImplementedError('pooling over non-overlapping cells only (kernel_size == stride), as the codecs use it')
        self.cell = st


class MinkowskiMaxPooling(_StridedPool):
    """Maximum over the cells of a coarser coordinate map that already exists (`coordinates` = its key): the first half of
    the decoder's local-maximum test (lossy_coord_v2/layers.py:159-161).  The product path does this inside fpcc_topk_keep;
    this module exists so that the reference's own call sequence runs on the engine."""

    def forward(self, x: SparseTensor, coordinates: Optional[CoordinateMapKey] = None) -> SparseTensor:
        cm = x.coordinate_manager
        src = cm._map(x.coordinate_map_key)
        if coordinates is None:
            dst = src
            for _ in range(self.cell.bit_length() - 1):
                dst = cm._ensure_parent(dst)
        else:
            dst = cm._map(coordinates)
        if (1 << dst.level) != (1 << src.level) * self.cell:
            raise ValueError('the target map does not have the pooled tensor stride')
        rows = cm._ancestor_rows(src, dst)
        f = x.F
        out = torch.full((dst.n, f.shape[1]), float('-inf'), dtype=f.dtype, device=f.device)
        out.scatter_reduce_(0, rows.unsqueeze(1).expand(-1, f.shape[1]), f, reduce='amax', include_self=True)
        return SparseTensor(out, coordinate_map_key=dst.key, coordinate_manager=cm)


class MinkowskiPoolingTranspose(_StridedPool):
    """Every row of the finer map `coordinates` receives the feature of the cell it lies in (one contributor per row when
    kernel_size == stride)."""

    def forward(self, x: SparseTensor, coordinates: CoordinateMapKey) -> SparseTensor:
        cm = x.coordinate_manager
        src, dst = cm._map(x.coordinate_map_key), cm._map(coordinates)
        if (1 << src.level) != (1 << dst.level) * self.cell:
            raise ValueError('the target map does not have the un-pooled tensor stride')
        rows = cm._ancestor_rows(dst, src)
        return SparseTensor(x.F[rows], coordinate_map_key=dst.key, coordinate_manager=cm)


class MinkowskiPruning(nn.Module):
    """Keeps the rows where mask is true; the new map keeps Morton order.  Only generated sets (the decoder's
    candidates) and their refinement are supported, which is how the reference uses it."""

    def forward(self, x: SparseTensor, mask: torch.Tensor) -> SparseTensor:
        cm = x.coordinate_manager
        src = cm._map(x.coordinate_map_key)
        if not src.generated:
            raise NotImplementedError('pruning of a non-generated map')
        t = cm._refine(src.parent, mask.contiguous(), 'pruned')
        rows = torch.nonzero(mask.view(-1)).squeeze(1)
        if torch.is_grad_enabled() and x.F.requires_grad:
            kept = x.F[rows]                                        # differentiable row selection (training path)
        else:
            kept = ops.gather_rows(x.F, rows.to(torch.int32))
        return SparseTensor(kept, coordinate_map_key=t.key, coordinate_manager=cm)
