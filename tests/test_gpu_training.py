"""Training path of lossy_coord_v2: PCC.forward in train mode returns the reference's loss dictionary with a differentiable
'loss'; loss.backward() reaches every parameter; a central finite difference on single weights agrees with autograd."""
import numpy as np
import pytest
import torch

from util import batched, surface_cloud

pytestmark = pytest.mark.gpu


def _batch(seeds, res=64, n=6000):
    rows = np.concatenate([batched(surface_cloud(s, res, n), i) for i, s in enumerate(seeds)])
    return torch.from_numpy(rows).to(torch.int32).cuda()


@pytest.fixture(scope='module')
def setup():
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from util import enliven
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)
    return cfg, model.cuda().train()


def _loss(model, coords, seed=5, step=0, bs=2):
    from fastpcc_amd.data import PCData
    torch.manual_seed(seed)                      # fixes the bottleneck noise
    return model(PCData(xyz=coords, batch_size=bs, training_step=step))


def test_loss_terms_and_gradients(setup):
    cfg, model = setup
    coords = _batch([1, 2])
    model.zero_grad(set_to_none=True)
    out = _loss(model, coords)
    assert isinstance(out['loss'], torch.Tensor) and out['loss'].requires_grad
    keys = set(out) - {'loss'}
    assert 'fea_bottom_bits_loss' in keys and 'coord_0_recon_loss' in keys
    assert sum(k.startswith('coord_') and k.endswith('bits_loss') for k in keys) == 6
    assert sum(k.startswith('fea_') and k.endswith('bits_loss') for k in keys) == 11
    # the other terms: detached 0-dim device tensors (read back by the trainer after the step is queued)
    assert all(isinstance(out[k], torch.Tensor) and not out[k].requires_grad and out[k].dim() == 0 for k in keys)
    assert all(np.isfinite(float(out[k])) for k in keys)
    assert float(out['loss'].detach()) == pytest.approx(sum(float(out[k]) for k in keys), rel=1e-5)
    out['loss'].backward()
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert sum(float(p.grad.abs().sum()) > 0 for p in model.parameters()) > 0.9 * len(list(model.parameters()))
    # warm-up weighting of the feature-rate terms (model.py:169-184)
    late = _loss(model, coords, step=cfg.warmup_fea_loss_steps + 1)
    ratio = cfg.bits_loss_factor / cfg.warmup_fea_loss_factor
    assert float(late['fea_bottom_bits_loss']) == pytest.approx(float(out['fea_bottom_bits_loss']) * ratio, rel=1e-4)
    assert float(late['coord_0_recon_loss']) == pytest.approx(float(out['coord_0_recon_loss']), rel=1e-5)


@pytest.mark.parametrize('name', ['encoder.blocks.1.1.conv.kernel',
                                  'em_lossless_based.hyper_decoder_fea.blocks.2.1.conv.kernel',
                                  'em_lossless_based.residual_block.blocks.4.blocks.0.conv.kernel',
                                  'em_lossless_based.hyper_decoder_coord.blocks.3.0.conv.kernel',
                                  'em_lossless_based.encoder.blocks.5.0.conv.kernel',
                                  'decoder.upsample_blocks.0.1.conv.kernel',
                                  'em_lossless_based.decoder_block.blocks.3.decoder.0.mlp.linear.weight',
                                  'em_lossless_based.encoder.blocks.2.1.conv.bias'])
def test_directional_derivative(setup, name):
    """central finite difference ALONG the gradient of one parameter tensor: (L(p + e g/|g|) - L(p - e g/|g|)) / 2e = |g|.
    (Single-weight differences drown in the fp32 resolution of a loss of magnitude 10^5-10^6.)"""
    cfg, model = setup
    coords = _batch([3])
    p = dict(model.named_parameters())[name]
    model.zero_grad(set_to_none=True)
    base = _loss(model, coords, bs=1)['loss']
    base.backward()
    g = p.grad.detach().clone()
    norm = float(g.norm())
    assert norm > 0
    # step sized for a loss change of ~2e-3 of the loss: far above fp32 noise, small enough to stay near-linear
    eps = 1e-3 * abs(float(base.detach())) / norm
    with torch.no_grad():
        p += eps * g / norm
    up = float(_loss(model, coords, bs=1)['loss'].detach())
    with torch.no_grad():
        p -= 2 * eps * g / norm
    dn = float(_loss(model, coords, bs=1)['loss'].detach())
    with torch.no_grad():
        p += eps * g / norm
    fd = (up - dn) / (2 * eps)
    assert fd == pytest.approx(norm, rel=0.08), (fd, norm, eps)


def test_optimizer_step_lowers_the_loss(setup):
    cfg, model = setup
    coords = _batch([4, 5])
    state = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    first = last = None
    for it in range(6):
        opt.zero_grad(set_to_none=True)
        out = _loss(model, coords)
        out['loss'].backward()
        opt.step()
        first = float(out['loss']) if first is None else first
        last = float(out['loss'])
    assert last < first
    model.load_state_dict(state, strict=False)


def _ddp_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from fastpcc_amd import replicas
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.data import PCData
    from fastpcc_amd.train import TrainConfig, Trainer
    from util import enliven
    # both ranks share the one GPU of the test box, so the collective runs over gloo (on a node with one GPU per rank the
    # same code path uses 'nccl' = RCCL)
    replicas.init('gloo')
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    tr = Trainer(model, TrainConfig(batch_size=2), torch.device('cuda', 0))
    assert isinstance(tr.model, DDP)
    torch.manual_seed(50 + rank)
    out = tr.step(PCData(xyz=_batch([20 + rank], n=4000), batch_size=1))
    probe = {n: p.detach().float().sum().item() for n, p in tr.model.module.named_parameters()}
    digest = torch.cat([p.detach().reshape(-1)[:8].cpu() for p in tr.model.module.parameters()])
    q.put((rank, out['loss'], digest.numpy(), len(probe)))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_keep_parameters_identical():
    """the full model under DistributedDataParallel: two ranks, different clouds and noise, one optimiser step ->
    identical parameters on both ranks (every parameter took part in the backward pass: find_unused_parameters=False)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][1] != got[1][1]                       # different data -> different local losses
    assert (got[0][2] == got[1][2]).all()               # same parameters after the all-reduced update


def _train_goldens():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v2.json')) as f:
        return json.load(f)['train']


@pytest.mark.parametrize('run', _train_goldens(), ids=[r['label'] for r in _train_goldens()])
def test_objective_equals_the_reference(run, monkeypatch):
    """tests/golden/codec_v2.json['train']: the loss dictionary of the REFERENCE's PCC.train_forward (make_golden.py: its own
    model / entropy-model code over the functional MinkowskiEngine stand-in, CPU) on batches of three clouds, with the
    bottleneck's uniform noise replaced by zeros on both sides: rate terms, occupancy cross-entropies, reconstruction losses
    of the lossy part and the warm-up factors (constant and linear) must agree term by term"""
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    from util import enliven
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    model = model.cuda().train()
    monkeypatch.setattr(torch.Tensor, 'uniform_', lambda self, *a, **k: self.zero_())
    out = model.train_forward(torch.tensor(run['xyz'], dtype=torch.int32).cuda(), run['training_step'], run['batch_size'])
    terms = {k: float(v) for k, v in out.items() if k != 'loss'}
    assert set(terms) == set(run['terms'])
    for k, want in run['terms'].items():
        assert terms[k] == pytest.approx(want, rel=3e-3, abs=1e-3), k
    assert float(out['loss']) == pytest.approx(run['loss'], rel=3e-3)
