"""Rate / distortion bookkeeping of a test run: `PCCEvaluator` with the interface and metric names of the reference's
(/root/reference/lib/evaluators.py:30-157), computing the point-to-point (D1) distortion ON THE DEVICE instead of
writing a PLY file and spawning the external `pc_error` binary per cloud (lib/metrics/pc_error_wrapper.py:40-107).

Definitions (pc_error's, restated in SURVEY.md section 8d): with A the original and B the reconstruction,
    mse1 = mean over a in A of min_b |a-b|^2,  mse2 = mean over b in B of min_a |a-b|^2,  mseF = max(mse1, mse2),
    PSNR = 10 log10(3 peak^2 / mse),  peak = resolution - 1   (the reference passes --resolution={resolution-1}).
Squared distances are exact integers (fpcc_nn_dist2), sums are 64-bit integers, so the numbers do not depend on any
reduction order.
Point-to-plane (D2) and Hausdorff (`pc_error_metrics`; the reference always hands pc_error normals, so its result files carry
'mseF,PSNR (p2plane)', which scripts/compare_performance.py:25 reads for D2 curves): with n_b the normal pc_error gives voxel b,
    plane1 = mean over a in A of mean over the nearest b (all ties) of ((a - b) . n_b)^2,   h. = max instead of mean over a;
normals of A: the caller's (`org_normals`, a PLY's nx ny nz) or PCA over the 30 nearest voxels (Open3D's estimate_normals default,
lib/metrics/pc_error_wrapper.py:66-68); normals of B: transferred from A as pc_error does (fpcc_transfer_normals).  pc_error is an
external binary that is not in the reference tree: these definitions restate its published algorithm (mpeg-pcc-dmetric), parity with
the binary is unpinned (oracle/metrics.py says the same); PCA normals carry this build's sign rule (include/fpcc_hip.h).
Colour (optional): BT.709 YUV of nearest-neighbour pairs, peak 255; where several neighbours tie in
distance pc_error averages their colours, this build takes the first in Morton order.
"""
import json
import math
import os
import os.path as osp
from collections import defaultdict
from typing import Dict, Optional, Union

import torch

from . import hipops as ops


def _keys_of(xyz: torch.Tensor, bits: int):
    """sorted unique level-0 keys of int32 [n, 3] coordinates (batch 0) + the batched coordinate rows"""
    c = torch.zeros((xyz.shape[0], 4), dtype=torch.int32, device=xyz.device)
    c[:, 1:] = xyz
    keys, perm = ops.sort_keys(ops.keys_from_coords(c, 0, bits), 3 * bits + 1)
    return c, keys, perm


def _psnr(peak_sq_times: float, mse: float) -> float:
    return float('inf') if mse == 0 else 10.0 * math.log10(peak_sq_times / mse)


@torch.no_grad()
def d1_metrics(org_xyz: torch.Tensor, rec_xyz: torch.Tensor, resolution: float, org_color: Optional[torch.Tensor] = None,
               rec_color: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """org_xyz / rec_xyz: integer coordinates [n, 3] on the GPU (any integer dtype).  Keys as printed by pc_error."""
    if org_xyz.shape[0] == 0 or rec_xyz.shape[0] == 0:
        raise ValueError('both clouds need at least one point')
    a = org_xyz.to(torch.int32).contiguous()
    b = rec_xyz.to(torch.int32).contiguous()
    hi = int(max(a.max().item(), b.max().item()))
    if int(min(a.min().item(), b.min().item())) < 0:
        raise ValueError('coordinates must be non-negative')
    bits = max(1, hi.bit_length())
    ca, ka, pa = _keys_of(a, bits)
    cb, kb, pb = _keys_of(b, bits)
    d_ab, row_ab = ops.nn_dist2(kb, bits, ca, want_rows=True)
    d_ba, row_ba = ops.nn_dist2(ka, bits, cb, want_rows=True)
    s1 = int(ops.sum_i64(d_ab).item())
    s2 = int(ops.sum_i64(d_ba).item())
    mse1, mse2 = s1 / a.shape[0], s2 / b.shape[0]
    msef = max(mse1, mse2)
    peak = 3.0 * float(resolution - 1) ** 2
    out = {'mse1      (p2point)': mse1, 'mse1,PSNR (p2point)': _psnr(peak, mse1),
           'mse2      (p2point)': mse2, 'mse2,PSNR (p2point)': _psnr(peak, mse2),
           'mseF      (p2point)': msef, 'mseF,PSNR (p2point)': _psnr(peak, msef),
           'mse1+mse2 (p2point)': mse1 + mse2, 'mse1+mse2/2(p2point)': (mse1 + mse2) / 2}
    if org_color is not None and rec_color is not None:
        m = torch.tensor([[0.2126, 0.7152, 0.0722], [-0.1146, -0.3854, 0.5], [0.5, -0.4542, -0.0458]],
                         dtype=torch.float64, device=a.device)
        yuv_a = org_color.to(torch.float64) @ m.T
        yuv_b = rec_color.to(torch.float64) @ m.T
        # row_* index the SORTED key arrays; perm maps them back to the caller's rows
        e1 = ((yuv_a - yuv_b[pb.long()[row_ab.long()]]) ** 2).mean(0)
        e2 = ((yuv_b - yuv_a[pa.long()[row_ba.long()]]) ** 2).mean(0)
        for ch in range(3):
            c1, c2 = float(e1[ch]), float(e2[ch])
            out[f'c[{ch}],    1'] = c1
            out[f'c[{ch}],    2'] = c2
            out[f'c[{ch}],    F'] = max(c1, c2)
            out[f'c[{ch}],PSNR1'] = _psnr(255.0 ** 2, c1)
            out[f'c[{ch}],PSNR2'] = _psnr(255.0 ** 2, c2)
            out[f'c[{ch}],PSNRF'] = _psnr(255.0 ** 2, max(c1, c2))
        out['c[3],PSNRF'] = out['c[0],PSNRF'] * 0.75 + out['c[1],PSNRF'] / 8 + out['c[2],PSNRF'] / 8
    return out


@torch.no_grad()
def estimate_normals(xyz: torch.Tensor, knn: int = 30) -> torch.Tensor:
    """float64 [n, 3] PCA normals of integer voxel coordinates [n, 3] (unique rows), in the caller's row order: the role of
    Open3D's estimate_normals() in lib/metrics/pc_error_wrapper.py:66-68 (KNN search, 30 neighbours, the point itself included)"""
    a = xyz.to(torch.int32).contiguous()
    bits = max(1, int(a.max().item()).bit_length())
    ca, ka, pa = _keys_of(a, bits)
    order = pa.long()
    rows, _ = ops.knn_voxels(ka, bits, ca[order].contiguous(), min(int(knn), 32))
    normals_sorted = ops.pca_normals(ka, bits, rows)
    out = torch.empty_like(normals_sorted)
    out[order] = normals_sorted
    return out


@torch.no_grad()
def pc_error_metrics(org_xyz: torch.Tensor, rec_xyz: torch.Tensor, resolution: float, org_color: Optional[torch.Tensor] = None,
                     rec_color: Optional[torch.Tensor] = None, org_normals: Optional[torch.Tensor] = None, hausdorff: bool = False,
                     knn: int = 30) -> Dict[str, float]:
    """d1_metrics plus the point-to-plane lines (always, like a pc_error run that is given normals) and, with hausdorff=True, the
    'h.' lines -- keys as pc_error prints them.  Coordinates must be unique rows (voxel sets); org_normals [n, 3] in org_xyz's order."""
    out = d1_metrics(org_xyz, rec_xyz, resolution, org_color, rec_color)
    a = org_xyz.to(torch.int32).contiguous()
    b = rec_xyz.to(torch.int32).contiguous()
    bits = max(1, int(max(a.max().item(), b.max().item())).bit_length())
    ca, ka, pa = _keys_of(a, bits)
    cb, kb, pb = _keys_of(b, bits)
    ca_s, cb_s = ca[pa.long()].contiguous(), cb[pb.long()].contiguous()           # voxels in key order
    if org_normals is None:
        rows, _ = ops.knn_voxels(ka, bits, ca_s, min(int(knn), 32))
        na = ops.pca_normals(ka, bits, rows)
    else:
        if org_normals.shape != (a.shape[0], 3):
            raise ValueError('org_normals must be [n, 3] in the order of org_xyz')
        na = org_normals.to(torch.float64)[pa.long()].contiguous()
    nb = ops.transfer_normals(ka, ca_s, na, kb, cb_s, bits)
    peak = 3.0 * float(resolution - 1) ** 2
    res = {}
    for tag, (keys, normals, query) in (('1', (kb, nb, ca_s)), ('2', (ka, na, cb_s))):
        plane, d, _ = ops.nn_plane_dist2(keys, bits, normals, query)
        s, mx = ops.sum_max_f64(plane).tolist()
        res[tag] = (s / query.shape[0], mx, float(d.max().item()))
    for tag in ('1', '2'):
        mse, h_plane, h_point = res[tag]
        out[f'mse{tag}      (p2plane)'] = mse
        out[f'mse{tag},PSNR (p2plane)'] = _psnr(peak, mse)
        if hausdorff:
            out[f'h.       {tag}(p2point)'] = h_point
            out[f'h.,PSNR  {tag}(p2point)'] = _psnr(peak, h_point)
            out[f'h.       {tag}(p2plane)'] = h_plane
            out[f'h.,PSNR  {tag}(p2plane)'] = _psnr(peak, h_plane)
    msef = max(res['1'][0], res['2'][0])
    out['mseF      (p2plane)'] = msef
    out['mseF,PSNR (p2plane)'] = _psnr(peak, msef)
    if hausdorff:
        hp, hl = max(res['1'][2], res['2'][2]), max(res['1'][1], res['2'][1])
        out['h.        (p2point)'] = hp
        out['h.,PSNR   (p2point)'] = _psnr(peak, hp)
        out['h.        (p2plane)'] = hl
        out['h.,PSNR   (p2plane)'] = _psnr(peak, hl)
    return out


class Evaluator:
    def __init__(self):
        self.reset()

    def reset(self):
        raise NotImplementedError

    def log(self, *args, **kwargs):
        raise NotImplementedError

    def show(self, results_dir: str):
        raise NotImplementedError


class PCCEvaluator(Evaluator):
    """`log` takes the original coordinates as a tensor (`org_xyz`) where the reference takes the path of a PLY file to
    hand to pc_error; everything else (arguments, info keys, '(mean)' aggregation of `show`) follows the reference."""

    def __init__(self, cal_mpeg_pc_error: bool = True, cal_avs_pc_evalue: bool = False, mpeg_pc_error_processes: int = 8,
                 p2plane: bool = True, hausdorff: bool = False):
        super().__init__()
        # the reference's pc_error runs always get normals (the PLY's or Open3D's), so their result files carry the p2plane lines;
        # it asks for --hausdorff=0 (lib/evaluators.py:98-104).  p2plane=False keeps the evaluation at the D1 lines
        self.p2plane, self.hausdorff = p2plane, hausdorff
        if cal_mpeg_pc_error + cal_avs_pc_evalue != 1:
            raise ValueError('choose exactly one distortion definition')
        if cal_avs_pc_evalue:
            raise NotImplementedError('only the MPEG pc_error definitions are computed on the device')
        self.cal_mpeg_pc_error = cal_mpeg_pc_error

    def reset(self):
        self.file_path_to_info: Dict[str, Dict[str, Union[int, float]]] = {}

    @torch.no_grad()
    def log(self, pred: torch.Tensor, org_points_num: int, compressed_bytes: bytes, file_path: str, resolution: float,
            results_dir: Optional[str] = None, pred_color: Optional[torch.Tensor] = None,
            pred_reflectance: Optional[torch.Tensor] = None, extra_info_dict: Optional[Dict] = None,
            org_xyz: Optional[torch.Tensor] = None, org_color: Optional[torch.Tensor] = None,
            org_normals: Optional[torch.Tensor] = None) -> bool:
        if pred.ndim != 2 or pred.shape[1] != 3:
            raise ValueError('pred must be [n, 3]')
        info = {'input_points_num': org_points_num, 'output_points_num': pred.shape[0],
                'compressed_bytes': len(compressed_bytes), 'bpp': len(compressed_bytes) * 8 / org_points_num}
        if extra_info_dict is not None:
            info.update(extra_info_dict)
        if results_dir is not None:
            out_path = osp.join(results_dir, osp.splitext(file_path)[0])
            os.makedirs(osp.dirname(out_path) or '.', exist_ok=True)
            with open(out_path + '.bin', 'wb') as f:
                f.write(compressed_bytes)
        if org_xyz is not None and self.cal_mpeg_pc_error:
            rec_color = pred_color if org_color is not None else None
            if self.p2plane or self.hausdorff:
                info.update(pc_error_metrics(org_xyz, pred, resolution, org_color, rec_color, org_normals, self.hausdorff))
            else:
                info.update(d1_metrics(org_xyz, pred, resolution, org_color, rec_color))
        if file_path in self.file_path_to_info:
            print(f'Warning: Duplicated test sample {file_path}')
        self.file_path_to_info[file_path] = info
        return True

    @torch.no_grad()
    def show(self, results_dir: Optional[str]) -> Dict[str, Union[int, float]]:
        if results_dir is not None:
            with open(osp.join(results_dir, '..', 'metric_dict.json'), 'w') as f:
                f.write(json.dumps(self.file_path_to_info, indent=2, sort_keys=False))
        mean, count = defaultdict(float), defaultdict(int)
        for info in self.file_path_to_info.values():
            for key, value in info.items():
                if key not in ('fea_points_num', 'input_points_num', 'output_points_num'):
                    mean[key + '(mean)'] += value
                    count[key + '(mean)'] += 1
        n = len(self.file_path_to_info)
        out = {k: v / n for k, v in mean.items() if count[k] == n}
        out['samples_num'] = n
        if results_dir is not None:
            with open(osp.join(results_dir, '..', 'mean_metric.json'), 'w') as f:
                f.write(json.dumps(out, indent=2, sort_keys=False))
        self.reset()
        return out
