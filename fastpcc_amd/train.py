"""Data-parallel training of the codecs: one process per GPU, gradients all-reduced over RCCL (torch.distributed backend
'nccl' on ROCm) by DistributedDataParallel.  The role of /root/reference/train.py:195-222,262-420 for this path -- DDP
wrapping (with the non-tensor `_extra_state` entries kept out of DDP's broadcast, train.py:204-214), parameter groups
by `Model.params_divider`, AdamW + StepLR per group, gradient accumulation under `no_sync`, gradient clipping -- without
the reference's dataset / logging / checkpoint shell, which is out of scope.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench_train.py --gpus N --steps K --warmup W

Clouds are independent samples: the global batch (8 in config/convolutional/lossy_coord_v2/baseline_r1.yaml:18) is
split as 8 / N clouds per rank; the only collective is the gradient all-reduce of the 26.3 M fp32 parameters (105 MB)
in DDP buckets, overlapped with the rest of the backward pass.
"""
import contextlib
import time
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from . import replicas
from .engine import refresh_prelu_cache
from .data import PCData

_EXTRA_STATE_SUFFIX = '_extra_state'


@dataclass
class TrainConfig:
    """the `train:` keys of the reference's YAML that this path uses (baseline_r1.yaml:16-30)"""
    batch_size: int = 8                                   # global: clouds per optimisation step over all ranks
    optimizer: Sequence[str] = ('AdamW', 'AdamW')
    momentum: float = 0.9
    weight_decay: Sequence[float] = (0.0001, 0.0)
    max_grad_norm: Sequence[float] = (1.0, 1.0)
    learning_rate: Sequence[float] = (0.0003, 0.0001)
    lr_step_size: int = 20
    lr_step_gamma: float = 0.3
    grad_acc_steps: int = 1
    bucket_cap_mb: Optional[int] = None
    find_unused_parameters: bool = False
    fused_optimizer: bool = True                          # torch.optim's fused (one launch, device-side step counters) Adam / AdamW


def unwrap(model: torch.nn.Module) -> torch.nn.Module:
    return model.module if isinstance(model, DDP) else model


def wrap_ddp(model: torch.nn.Module, cfg: TrainConfig, device: torch.device) -> torch.nn.Module:
    """DDP around the model when a process group with more than one rank exists; `_extra_state` (the cached CDF table of
    the entropy bottleneck, a Python tuple) is excluded from parameter / buffer synchronisation"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return model
    ignore = [k for k in model.state_dict() if k.endswith(_EXTRA_STATE_SUFFIX)]
    DDP._set_params_and_buffers_to_ignore_for_model(model, ignore)
    kw = dict(find_unused_parameters=cfg.find_unused_parameters, broadcast_buffers=False)
    if cfg.bucket_cap_mb:
        kw['bucket_cap_mb'] = cfg.bucket_cap_mb
    if device.type == 'cuda':
        return DDP(model, device_ids=[device.index], output_device=device.index, **kw)
    return DDP(model, **kw)


def build_optimizers(model: torch.nn.Module, cfg: TrainConfig):
    """one optimiser + StepLR per parameter group; groups by `Model.params_divider(name)` (negative: frozen)"""
    base = unwrap(model)
    divider = getattr(base, 'params_divider', lambda name: 0)
    groups: List[List[torch.nn.Parameter]] = [[] for _ in cfg.optimizer]
    for name, p in base.named_parameters():
        g = divider(name)
        if g >= 0:
            groups[g].append(p)
    makers = {'Adam': torch.optim.Adam, 'AdamW': torch.optim.AdamW}
    opts, scheds = [], []
    for i, params in enumerate(groups):
        if not params:
            opts.append(None)
            scheds.append(None)
            continue
        # fused: ONE multi-tensor launch per optimiser and step counters that live on the device.  The default ("foreach") form keeps
        # every parameter's step count in a host tensor and reads each three times per update -- 852 `.item()` calls, 576 host-side
        # `add_` and ~2500 dtype resolutions per step of this model (profiles/r05/train_host_ops.md): milliseconds of host time in a
        # step whose GPU work is 46 ms.  Same update rule (torch.optim's own kernels).
        fused = all(p.is_cuda and p.dtype == torch.float32 for p in params) and cfg.fused_optimizer
        opt = makers[cfg.optimizer[i]](params, lr=cfg.learning_rate[i], betas=(cfg.momentum, 0.999),
                                       weight_decay=cfg.weight_decay[i], **({'fused': True} if fused else {}))
        opts.append(opt)
        scheds.append(torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.lr_step_size, gamma=cfg.lr_step_gamma))
    if all(o is None for o in opts):
        raise ValueError('no trainable parameters')
    return opts, scheds


class Trainer:
    """model(batch) must return a dict with a differentiable 'loss' (PCC.forward in train mode)"""

    def __init__(self, model: torch.nn.Module, cfg: TrainConfig, device: torch.device):
        self.cfg, self.device = cfg, device
        self.model = wrap_ddp(model.to(device).train(), cfg, device)
        self.optimizers, self.schedulers = build_optimizers(self.model, cfg)
        self.micro_step = 0

    @property
    def optimisation_step(self) -> int:
        return self.micro_step // self.cfg.grad_acc_steps

    def step(self, batch: PCData) -> Dict[str, float]:
        """one micro-step: forward + backward; every grad_acc_steps-th call all-reduces (DDP) and updates"""
        acc = self.cfg.grad_acc_steps
        update = (self.micro_step + 1) % acc == 0
        batch.training_step = self.optimisation_step
        sync_off = isinstance(self.model, DDP) and not update
        with self.model.no_sync() if sync_off else contextlib.nullcontext():
            out = self.model(batch)
            (out['loss'] / acc).backward()
        if update:
            for opt, clip in zip(self.optimizers, self.cfg.max_grad_norm):
                if opt is None:
                    continue
                if clip:
                    torch.nn.utils.clip_grad_norm_(opt.param_groups[0]['params'], clip, error_if_nonfinite=True)
                opt.step()
            for opt in self.optimizers:
                if opt is not None:
                    opt.zero_grad(set_to_none=True)
            refresh_prelu_cache(unwrap(self.model))       # host copy of the PReLU slopes' signs for the fused training node
        self.micro_step += 1
        # one read-back for every logged term, after the whole step has been queued
        keys = list(out)
        vals = torch.stack([out[k].detach().reshape(()).float() if isinstance(out[k], torch.Tensor) else torch.tensor(float(out[k]))
                            .to(out['loss'].device) for k in keys]).tolist()
        return dict(zip(keys, vals))

    def end_epoch(self):
        for s in self.schedulers:
            if s is not None:
                s.step()


# ---- synthetic data of cfg#5: ShapeNet-like clouds voxelised at 128^3, random rotation -------------------
def shapenet_like(seed: int, resolution: int = 128, points: int = 200000) -> np.ndarray:
    """union of random ellipsoid shells and planes, randomly rotated, voxelised: ~30-50 k unique voxels at 128^3"""
    rng = np.random.default_rng(seed)
    pts = []
    for _ in range(3):
        c, r = rng.uniform(-0.3, 0.3, 3), rng.uniform(0.15, 0.45, 3)
        d = rng.normal(size=(points // 5, 3))
        pts.append(c + r * d / np.linalg.norm(d, axis=1, keepdims=True))
    for _ in range(2):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        u = np.cross(n, rng.normal(size=3))
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        ab = rng.uniform(-0.5, 0.5, (points // 5, 2))
        pts.append(rng.uniform(-0.2, 0.2) * n + ab[:, :1] * u + ab[:, 1:] * v)
    p = np.concatenate(pts)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))                     # random rotation (dataset option random_rotation)
    p = p @ q.T
    p = p[np.abs(p).max(1) <= 0.5 * np.sqrt(3)]
    p = (p - p.min(0)) / (p.max(0) - p.min(0)).max()
    return np.unique(np.minimum((p * (resolution - 1)).round().astype(np.int32), resolution - 1), axis=0)


def synthetic_batches(rank: int, world: int, cfg: TrainConfig, device: torch.device, resolution: int = 128,
                      first_seed: int = 10, pool: int = 16) -> Iterator[PCData]:
    """endless stream of this rank's share (batch_size / world clouds) of the global batches"""
    if cfg.batch_size % world:
        raise ValueError(f'global batch {cfg.batch_size} does not divide over {world} ranks')
    per_rank = cfg.batch_size // world
    clouds = [shapenet_like(first_seed + i, resolution) for i in range(rank, max(pool, cfg.batch_size), world)]
    at = 0
    while True:
        rows = []
        for b in range(per_rank):
            xyz = clouds[(at + b) % len(clouds)]
            rows.append(np.concatenate((np.full((len(xyz), 1), b, np.int32), xyz), 1))
        at += per_rank
        coords = torch.from_numpy(np.concatenate(rows)).to(device)
        yield PCData(xyz=coords, batch_size=per_rank, resolution=[resolution] * per_rank)


def ddp_bench(trainer: Trainer, data: Iterator[PCData], steps: int, warmup: int, device: torch.device,
              units_per_step: float) -> Tuple[float, float, Optional[float], Optional[Dict[str, float]]]:
    """Times `steps` optimisation steps of an already constructed (DDP-wrapped when world > 1) trainer on EVERY rank of the
    process group: barrier, K steps, barrier, max over ranks.  Then -- when gradients are all-reduced -- the same steps once
    more under `no_sync`, which prices the communication that the overlap with the backward pass does not hide.  Collective:
    all ranks must call it.  Returns (seconds of the timed region (max over ranks), voxels summed over ranks and steps,
    exposed all-reduce ms per step | None, the last step's logged terms)."""
    voxels = 0
    for _ in range(warmup):
        trainer.step(next(data))
    replicas.barrier(device)
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        batch = next(data)
        voxels += batch.xyz.shape[0] if batch.xyz.dim() == 2 else 0
        last = trainer.step(batch)
    replicas.barrier(device)
    elapsed_max, total_voxels = replicas.aggregate(time.perf_counter() - t0, float(voxels), device)
    comm_ms = None
    if isinstance(trainer.model, DDP):
        replicas.barrier(device)
        t1 = time.perf_counter()
        for _ in range(steps):
            batch = next(data)
            batch.training_step = trainer.optimisation_step
            with trainer.model.no_sync():
                (trainer.model(batch)['loss']).backward()
            for opt in trainer.optimizers:
                if opt is not None:
                    opt.zero_grad(set_to_none=True)
        replicas.barrier(device)
        local_only, _ = replicas.aggregate(time.perf_counter() - t1, 0.0, device)
        comm_ms = max(0.0, (elapsed_max - local_only) / steps * 1e3)
    if dist.is_initialized():
        dist.barrier()
    return elapsed_max, total_voxels, comm_ms, last


def ddp_training_record(steps: int, warmup: int, device: torch.device, resolution: int = 128,
                        cfg: Optional[TrainConfig] = None) -> Optional[dict]:
    """cfg#5 on the ranks of the EXISTING process group (bench.py --gpus N calls this after its replica timing, so the
    driver's own scaling command produces the DDP figure too): lossy_coord_v2/baseline_r1, global batch 8 split 8 / N per
    rank, gradients all-reduced over RCCL by DDP.  Collective; rank 0 returns the record, other ranks None."""
    from .codecs.lossy_coord_v2 import Model
    from .codecs.lossy_coord_v2.model_config import baseline_r1
    cfg = cfg or TrainConfig()
    rank, world, _ = replicas.env_rank()
    if cfg.batch_size % world:
        return {'skipped': f'global batch {cfg.batch_size} does not divide over {world} ranks'} if rank == 0 else None
    # Everything a rank can fail at on its own (model construction, its share of the data, one local forward + backward: memory,
    # a kernel error) happens BEFORE the first collective of this leg, and the ranks then agree on one verdict: a rank that failed
    # alone would otherwise leave the others blocked in DDP's parameter broadcast or the first gradient all-reduce.
    failure = None
    model = data = None
    try:
        torch.manual_seed(0)                              # same initial weights on every rank
        model = Model(baseline_r1()).to(device).train()
        data = synthetic_batches(rank, world, cfg, device, resolution)
        probe = next(data)
        probe.training_step = 0
        model(probe)['loss'].backward()
        model.zero_grad(set_to_none=True)
        if device.type == 'cuda':
            torch.cuda.synchronize(device)
    except Exception as e:                                # noqa: BLE001 -- any local failure becomes the job's verdict
        failure = repr(e)[:200]
    if not replicas.all_ok(failure is None, device):
        del model, data
        return {'error': failure or 'another rank failed before the first collective'} if rank == 0 else None
    trainer = Trainer(model, cfg, device)
    torch.manual_seed(1000 + rank)                        # different bottleneck noise per rank
    elapsed_max, total_voxels, comm_ms, last = ddp_bench(trainer, data, steps, warmup, device, cfg.batch_size)
    if rank != 0:
        return None
    n_param = sum(p.numel() for p in unwrap(trainer.model).parameters())
    return {'workload': f'lossy_coord_v2/baseline_r1 optimisation step, global batch {cfg.batch_size} ShapeNet-like clouds at '
                        f'{resolution}^3 (cfg#5), {cfg.batch_size // world} per rank, forward + backward + all-reduce + AdamW',
            'ranks': world, 'parallelism': f'ddp{world}', 'ms_per_step': round(elapsed_max / steps * 1e3, 2),
            'clouds_per_s': round(cfg.batch_size * steps / elapsed_max, 3), 'voxels_per_step': round(total_voxels / steps),
            'exposed_allreduce_ms_per_step': None if comm_ms is None else round(comm_ms, 2),
            'parameters': n_param, 'gradient_bytes': 4 * n_param, 'steps': steps, 'warmup': warmup,
            'last_loss': None if last is None else round(last['loss'], 2)}


def bench(steps: int, warmup: int, gpus: int, resolution: int = 128, cfg: Optional[TrainConfig] = None) -> Optional[dict]:
    """times `steps` optimisation steps of lossy_coord_v2/baseline_r1 on synthetic ShapeNet-like batches; rank 0 returns
    the result record, other ranks None"""
    cfg = cfg or TrainConfig()
    rank, world, local = replicas.env_rank()
    if world != gpus:
        raise SystemExit(f'--gpus {gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    replicas.bind_to_device_numa_node(local)
    replicas.init('nccl')
    rec = ddp_training_record(steps, warmup, device, resolution, cfg)
    if rank != 0:
        return None
    if 'skipped' in rec:
        raise SystemExit(rec['skipped'])
    return {'metric': 'training clouds/sec, lossy_coord_v2 baseline_r1 (DDP)', 'value': rec['clouds_per_s'],
            'unit': 'clouds/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': rec['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'lossy_coord_v2/baseline_r1 training, global batch {cfg.batch_size} ShapeNet-like clouds at '
                                   f'{resolution}^3 (cfg#5), {cfg.batch_size // world} per rank',
                       'parallelism': f'ddp{world}', 'voxels_per_step': rec['voxels_per_step'],
                       'parameters': rec['parameters'], 'gradient_bytes': rec['gradient_bytes'],
                       'exposed_allreduce_ms_per_step': rec['exposed_allreduce_ms_per_step'],
                       'last_loss': rec['last_loss']}}
