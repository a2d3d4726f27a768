"""Multi-GPU execution of the codec: replicas only.

Frames are coded independently, so N GPUs run N processes that never exchange data on the coding
path; the only collectives are the barrier around a timed region and the two scalar reductions that turn per-rank
timings into one whole-job figure.  Backend 'nccl' (= RCCL over xGMI) on GPUs, 'gloo' in the CPU tests.
"""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) as torch.distributed.run exports them; (0, 1, 0) when run directly"""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend: str) -> None:
    if env_rank()[1] > 1 and not dist.is_initialized():
        dist.init_process_group(backend, init_method='env://')


def frames_of_rank(frames: Sequence, rank: int, world: int) -> List:
    """round-robin assignment of independent frames (or kd-tree partitions) to ranks; no frame is coded twice"""
    return [f for i, f in enumerate(frames) if i % world == rank]


def barrier(device: torch.device) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device.type == 'cuda':
        torch.cuda.synchronize(device)


def aggregate(elapsed_s: float, units: float, device: torch.device) -> Tuple[float, float]:
    """whole-job view of a timed region: (max over ranks of the elapsed time, sum over ranks of the processed units)"""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_bytes(blob: bytes, device: torch.device) -> List[bytes]:
    """variable-length all-gather of one byte string per rank (partition bitstreams of one frame, a few 100 KB):
    lengths first, then one padded uint8 all_gather."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [blob]
    world = dist.get_world_size()
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    cap = max(int(v.item()) for v in lens)
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=device)
    if blob:
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [bytes(o[:int(v.item())].cpu().numpy()) for o, v in zip(out, lens)]
