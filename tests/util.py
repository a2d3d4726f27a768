"""Shared helpers of the test-suite: the seeded synthetic clouds and the weight randomisation live
in the package (fastpcc_amd/synthetic.py) because bench.py and smoke() use them too; re-exported here for the tests."""
import numpy as np

from fastpcc_amd.synthetic import enliven, surface_cloud  # noqa: F401


def batched(xyz: np.ndarray, batch: int = 0) -> np.ndarray:
    return np.concatenate((np.full((len(xyz), 1), batch, dtype=np.int64), xyz.astype(np.int64)), 1)
