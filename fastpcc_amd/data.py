"""Minimal batch container at the codec boundary: the fields of the reference's `PCData`
(/root/reference/lib/data_utils.py:43-93) that the encode/decode path reads, plus `batched_coordinates`
(/root/reference/lib/data_utils.py:14-23).  Dataset readers are outside the hot path."""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import torch


def batched_coordinates(coords: Sequence[torch.Tensor]):
    """[(n_i, 3) int] -> ((sum n_i, 4) int32 with the sample index in column 0, [n_i])"""
    sizes = [len(c) for c in coords]
    out = torch.zeros((sum(sizes), coords[0].shape[1] + 1), dtype=torch.int32)
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
