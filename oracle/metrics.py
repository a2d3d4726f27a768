"""TEST ORACLE -- NOT PRODUCT CODE.  Point-to-point (D1) distortion on the CPU.

Definitions of MPEG pc_error as the reference uses it (/root/reference/lib/metrics/pc_error_wrapper.py:40-107, key
'mseF,PSNR (p2point)' read by scripts/compare_performance.py:25; --resolution = resolution - 1,
pc_error_wrapper.py:50).  pc_error itself is an external binary that is not part of the reference tree, so this
restatement is pinned only by hand-derived known answers (tests/test_oracle_float.py::test_d1_known_answers):
parity unpinned against the binary.  Nearest neighbours come from scipy's k-d tree; the squared distances are then
recomputed in int64 from the returned indices, so they are exact integers.
"""
import math

import numpy as np
from scipy.spatial import cKDTree


def nn_dist2(query: np.ndarray, points: np.ndarray):
    """squared distance (int64) from every query row to its nearest row of `points`, and that row's index"""
    q = np.asarray(query, dtype=np.int64)
    p = np.asarray(points, dtype=np.int64)
    _, idx = cKDTree(p.astype(np.float64)).query(q.astype(np.float64), k=1)
    d = ((q - p[idx]) ** 2).sum(1)
    # the tree works in floating point; a tie broken the other way still has the same integer distance
    return d, idx


def brute_nn_dist2(query: np.ndarray, points: np.ndarray) -> np.ndarray:
    q = np.asarray(query, dtype=np.int64)[:, None, :]
    p = np.asarray(points, dtype=np.int64)[None, :, :]
    return ((q - p) ** 2).sum(2).min(1)


def d1(org: np.ndarray, rec: np.ndarray, resolution: float) -> dict:
    d_ab, _ = nn_dist2(org, rec)
    d_ba, _ = nn_dist2(rec, org)
    mse1, mse2 = int(d_ab.sum()) / len(org), int(d_ba.sum()) / len(rec)
    msef = max(mse1, mse2)
    peak = 3.0 * float(resolution - 1) ** 2
    psnr = lambda m: float('inf') if m == 0 else 10.0 * math.log10(peak / m)
    return {'mse1      (p2point)': mse1, 'mse1,PSNR (p2point)': psnr(mse1), 'mse2      (p2point)': mse2,
            'mse2,PSNR (p2point)': psnr(mse2), 'mseF      (p2point)': msef, 'mseF,PSNR (p2point)': psnr(msef)}
