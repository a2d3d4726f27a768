"""Multi-GPU execution of the codec: replicas only.

Frames are coded independently, so N GPUs run N processes that never exchange data on the coding
path; the only collectives are the barrier around a timed region and the two scalar reductions that turn per-rank
timings into one whole-job figure.  Backend 'nccl' (= RCCL over xGMI) on GPUs, 'gloo' in the CPU tests.
"""
import os
from typing import Optional, List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) as torch.distributed.run exports them; (0, 1, 0) when run directly"""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend: str, timeout_s: Optional[float] = None) -> None:
    """timeout_s bounds every collective of the group: a rank that died leaves the others with an error after that long instead of
    a process that never ends (the default of the backends is 10 - 30 minutes)"""
    if env_rank()[1] > 1 and not dist.is_initialized():
        kw = {}
        if timeout_s is not None:
            import datetime
            kw['timeout'] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend, init_method='env://', **kw)


def all_ok(ok: bool, device: torch.device) -> bool:
    """collective: True when every rank passed True (one MIN all-reduce); lets the ranks leave a multi-rank phase together"""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_device_numa_node(device_index: int) -> Optional[dict]:
    """Pin every thread of the process (those the HIP runtime has already started included, and with them every thread started
    later: the coder pool, torch's workers) to the CPUs of the NUMA node the GPU hangs off, read from sysfs by its PCI address.
    MI355X hosts here are two-socket machines with four GPUs behind each socket; a process scheduled on the far socket allocates its
    pinned staging buffers there and every device<->host copy, flag write and launch crosses the inter-socket link, and an unbound
    process is moved between the sockets by the scheduler.  What `numactl --cpunodebind` does for a serving process.
    On by default (FPCC_NUMA_BIND=0 switches it off).  Measured, bench.py twice each way on one box (tools/r05/g32.sh): the integer
    codec, whose decoder reads 50 MB of DMA-written CDF rows per frame on one host thread, 10.6 + 15.5 / 10.1 + 15.3 ms bound against
    10.8 + 20.2 / 10.4 + 15.9 ms unbound (tools/r05/numa_probe.py: 15.5 ms from the GPU's node, 17.0 from the other, 17.6-19.8 left
    to the scheduler); the training step 46.9 / 48.0 against 48.5 / 50.1 ms; the colour codec 41.5 / 41.3 against 44.0 / 42.5 ms; the
    headline (batches of eight frames) unchanged at 80.3 / 80.4 against 78.8 / 80.7 Mpoints/s.  (Round 3's A/B on a box whose GPU sat
    on node 1 had seen no difference for the one-frame bench, hence the earlier default of off for a single process.)
    Returns None when nothing was changed (topology unreadable, one node, switched off)."""
    want = os.environ.get('FPCC_NUMA_BIND', '1')
    if want == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        props = torch.cuda.get_device_properties(device_index)
        addr = f'{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0'
        base = f'/sys/bus/pci/devices/{addr}'
        with open(f'{base}/local_cpulist') as f:
            cpus = _parse_cpulist(f.read())
        with open(f'{base}/numa_node') as f:
            node = int(f.read().strip())
        allowed = os.sched_getaffinity(0)
        target = sorted(set(cpus) & allowed)
        if not target or len(target) == len(allowed):
            return None
        # every thread the process already has (reading the device properties started the HIP runtime's), not only the caller:
        # sched_setaffinity(0, ...) binds one thread, and only threads created afterwards inherit it
        n_threads = 0
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), target)
                n_threads += 1
            except (OSError, ValueError):                            # a thread that ended meanwhile
                pass
        os.sched_setaffinity(0, target)
        return {'pci': addr, 'numa_node': node, 'cpus': len(target), 'threads_bound': n_threads}
    except (OSError, AttributeError, ValueError, RuntimeError, AssertionError):      # no device / no sysfs entry: nothing to bind to
        return None


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


def rank_spread(value: float, device: torch.device) -> dict:
    """collective: what lets a reader of the bench line confirm the job ran on as many ranks as it says -- `ranks` is the sum over
    ranks of a device tensor of ones through the data-path backend (RCCL on GPUs), next to the backend's name and the min / max over
    ranks of every rank's own `value`"""
    ones = torch.ones(1, dtype=torch.float64, device=device)
    lo = torch.tensor([value], dtype=torch.float64, device=device)
    hi = lo.clone()
    backend = 'none'
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        backend = str(dist.get_backend())
    return {'ranks': int(round(float(ones.item()))), 'backend': backend, 'min': float(lo.item()), 'max': float(hi.item())}


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
