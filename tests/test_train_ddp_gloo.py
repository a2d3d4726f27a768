"""The DDP training step on CPU with two gloo ranks: `_extra_state` kept out of DDP's synchronisation, parameter groups by
params_divider, gradient accumulation under no_sync, and -- the point of data parallelism -- after an update every rank
holds the same parameters, equal to a single-process update with the averaged gradient.  The module is the (pure
PyTorch) noisy deep-factorised entropy model; the sparse-convolution path needs a GPU and is covered by
tests/test_gpu_training.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


class Toy(torch.nn.Module):
    """entropy bottleneck + a scale: two parameter groups, a non-tensor _extra_state, a dict with 'loss'"""

    @staticmethod
    def params_divider(name: str) -> int:
        return 1 if 'bottom_fea_entropy_model' in name else 0

    def __init__(self):
        super().__init__()
        from fastpcc_amd.entropy_models import NoisyDeepFactorizedEntropyModel
        self.bottom_fea_entropy_model = NoisyDeepFactorizedEntropyModel(torch.Size([4]), 2, broadcast_shape_bytes=(3,))
        self.scale = torch.nn.Parameter(torch.ones(4))

    def forward(self, batch):
        # like PCC.train_forward (warm-up schedule): every forward needs the optimisation step the trainer stamps on the batch
        assert isinstance(batch.training_step, int) and batch.training_step >= 0
        y, d = self.bottom_fea_entropy_model(batch.xyz * self.scale)
        return {'loss': d['bits_loss'] + 0.01 * (y ** 2).sum(), 'bits': float(d['bits_loss'].detach())}


def _data(rank, step):
    from fastpcc_amd.data import PCData
    g = torch.Generator().manual_seed(100 * rank + step)
    return PCData(xyz=torch.randn((1, 50, 4), generator=g) * 3, batch_size=1)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from fastpcc_amd import replicas
    from fastpcc_amd.train import TrainConfig, Trainer
    replicas.init('gloo')
    torch.manual_seed(0)
    cfg = TrainConfig(batch_size=2, grad_acc_steps=2, max_grad_norm=(0.0, 0.0), learning_rate=(0.01, 0.02))
    tr = Trainer(Toy(), cfg, torch.device('cpu'))
    assert isinstance(tr.model, DDP)
    assert any(k.endswith('_extra_state') for k in tr.model.module.state_dict())
    before = {k: v.detach().clone() for k, v in tr.model.module.named_parameters()}
    torch.manual_seed(7 + rank)                       # different noise per rank
    tr.step(_data(rank, 0))                           # accumulation micro-step: no all-reduce, no update
    assert tr.optimisation_step == 0
    mid = {k: v.detach().clone() for k, v in tr.model.module.named_parameters()}
    assert all(torch.equal(before[k], mid[k]) for k in before)
    local_grad = {k: p.grad.detach().clone() for k, p in tr.model.module.named_parameters()}
    tr.step(_data(rank, 1))                           # update step
    assert tr.optimisation_step == 1
    after = {k: v.detach().clone() for k, v in tr.model.module.named_parameters()}
    q.put((rank, {k: v.numpy() for k, v in after.items()}, {k: v.numpy() for k, v in local_grad.items()},
           [len(o.param_groups[0]['params']) for o in tr.optimizers]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_step():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p0, g0, groups0), (_, p1, g1, groups1) = got
    assert groups0 == groups1 == [1, 14]               # `scale` | 5 weights + 5 biases + 4 factors of the prior
    for k in p0:
        assert (p0[k] == p1[k]).all(), k               # identical parameters on both ranks after the update
    assert any((g0[k] != g1[k]).any() for k in g0)     # ... although their local gradients differed


def _bench_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import itertools
    import torch.distributed as dist
    from fastpcc_amd import replicas
    from fastpcc_amd.train import TrainConfig, Trainer, ddp_bench
    replicas.init('gloo')
    torch.manual_seed(0)
    tr = Trainer(Toy(), TrainConfig(batch_size=2, max_grad_norm=(0.0, 0.0)), torch.device('cpu'))
    data = (_data(rank, i) for i in itertools.count())
    elapsed, units, comm_ms, last = ddp_bench(tr, data, steps=3, warmup=1, device=torch.device('cpu'), units_per_step=2)
    q.put((rank, elapsed, comm_ms, last['loss'], tr.optimisation_step))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_bench_is_one_collective_measurement_over_all_ranks():
    """what `bench.py --gpus N` (N > 1) runs for the cfg#5 figure: every rank times the same K steps between barriers, the
    maximum over ranks is the step time, and the steps are repeated without gradient synchronisation to price the all-reduce"""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, e0, c0, l0, s0), (_, e1, c1, l1, s1) = got
    assert e0 == e1 > 0                                # the max over ranks, identical on both
    assert c0 is not None and c0 == c1 and c0 >= 0     # exposed all-reduce time: measured because the model is DDP-wrapped
    assert s0 == s1 == 4                               # 1 warm-up + 3 timed optimisation steps (the no_sync pass does not update)
    assert isinstance(l0, float) and isinstance(l1, float)


def test_single_process_trainer_updates_and_schedules():
    from fastpcc_amd.train import TrainConfig, Trainer
    torch.manual_seed(0)
    cfg = TrainConfig(batch_size=1, lr_step_size=1, lr_step_gamma=0.5, learning_rate=(0.01, 0.02))
    tr = Trainer(Toy(), cfg, torch.device('cpu'))
    w0 = tr.model.scale.detach().clone()
    out = tr.step(_data(0, 0))
    assert isinstance(out['loss'], float) and not torch.equal(w0, tr.model.scale.detach())
    tr.end_epoch()
    assert tr.optimizers[0].param_groups[0]['lr'] == pytest.approx(0.005)
    assert tr.optimizers[1].param_groups[0]['lr'] == pytest.approx(0.01)
