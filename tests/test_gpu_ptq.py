"""Post-training quantisation end to end on the GPU: the float twin codes losslessly, calibration passes feed the
observers, the converted parameters load into the integer codec, which codes the same clouds losslessly at (nearly) the
float model's rate, and the fixed-point logits track the float logits."""
import copy

import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import batched, lidar_cloud

pytestmark = pytest.mark.gpu


def _float_model(channels, tmp_path, seed=0, more=False):
    from fastpcc_amd.codecs.lossl_coord import Config, Model
    torch.manual_seed(seed)
    cfg = Config(channels=channels, use_more_ch_for_multi_step_pred=more, quantize_param=True,
                 int_param_save_path=str(tmp_path / 'int_param.pt'))
    model = Model(cfg, 'cuda').cuda().eval()
    with torch.no_grad():                       # sharpen the predictions so that quantisation error would show in the rate
        for name, p in model.named_parameters():
            if name.endswith('.weight') and p.dim() == 2 and p.shape[0] == 255:
                p.mul_(4.0)
    return cfg, model


def _cloud(seed, beams=16, az=512):
    xyz = lidar_cloud(seed, beams=beams, azimuths=az)
    return xyz, torch.from_numpy(batched(xyz)).cuda()


def _same_points(rec: torch.Tensor, xyz: np.ndarray):
    return sorted(map(tuple, rec.cpu().numpy().tolist())) == sorted(map(tuple, xyz.tolist()))


@pytest.mark.parametrize('channels,more', [(32, False), (16, True)])
def test_float_twin_is_lossless(channels, more, tmp_path):
    _, model = _float_model(channels, tmp_path, more=more)
    xyz, dev = _cloud(3)
    data = model.compress(dev)
    assert _same_points(model.decompress(data), xyz)
    assert model.compress(dev[torch.randperm(len(dev), device='cuda')]) == data         # input order does not matter


def test_calibrate_convert_and_run_integer_codec(tmp_path):
    from fastpcc_amd.codecs.lossl_coord_int import Config as IntConfig, Model as IntModel
    from fastpcc_amd.data import PCData
    cfg, model = _float_model(32, tmp_path, seed=1)
    reference_float = copy.deepcopy(model)
    clouds = [_cloud(s) for s in (11, 12, 13)]
    float_bytes = [len(model.compress(dev)) for _, dev in clouds]

    model.pre_test_hook()
    for xyz, dev in clouds:                                         # calibration = ordinary test passes
        out = model(PCData(xyz=dev))
        assert _same_points(out['pred'], xyz)                       # observers do not disturb the codec
    model.post_test_hook()

    saved = torch.load(cfg.int_param_save_path)['state_dict']
    integer = IntModel(IntConfig(channels=32), 'cuda').cuda().eval()
    integer.load_state_dict(saved)
    for (xyz, dev), nf in zip(clouds + [_cloud(21)], float_bytes + [None]):
        data = integer.compress(dev)
        assert _same_points(integer.decompress(data), xyz)
        assert model.compress(dev) == data                          # the converted model, evaluated in place, is the same codec
        if nf is not None:
            assert abs(len(data) - nf) <= 0.03 * nf, (len(data), nf)

    # fixed-point logits of the coarsest predictor against the float ones, same input
    xyz, dev = clouds[0]
    logits = {}
    for name, m in (('float', reference_float), ('int', integer)):
        coords = dev - torch.nn.functional.pad(dev.amin(0)[1:], (1, 0))
        from fastpcc_amd import hipops as ops
        from fastpcc_amd.int_sparse_conv import SparseTensor
        _, perm = ops.sort_keys(ops.morton3d_encode(coords[:, 1:], (2, 1, 0)))
        coords = coords[perm.long()].contiguous()
        ones = torch.ones((len(coords), 1), dtype=torch.int8, device='cuda')
        level = SparseTensor(ones, coords, (1, 1, 1))
        for _ in range(m.max_downsample_times):
            level = m.get_bin(level, ones)
        feat = ones[:level.C.shape[0]] if name == 'int' else ones[:level.C.shape[0]].float()
        cur = SparseTensor(feat, level.C, level.stride)
        cur._caches = level._caches
        with torch.no_grad():
            _, out = m.block_dec_recurrent._trunk(cur)
        logits[name] = out.float() / (1 << 23) if name == 'int' else out
    err = (logits['float'] - logits['int']).abs().mean().item()
    spread = logits['float'].std().item()
    assert spread > 0.05 and err < 0.1 * spread, (err, spread)
