"""Post-training quantisation of the float LiDAR codec, host part: observer placement, module replacement and the layout of
the written state dict (it must load into the integer model).  The calibrated end-to-end run is in test_gpu_ptq.py."""
import torch
import torch.nn as nn

from fastpcc_amd import int_sparse_conv as isc
from fastpcc_amd.codecs import lossl_coord as fl
from fastpcc_amd.codecs.lossl_coord import model as flm
from fastpcc_amd.codecs.lossl_coord_int import Config as IntConfig, Model as IntModel


def _feed(model, seed=0):
    """stand-in for the calibration passes: every observer sees some data"""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, isc.SparseTensorHistogramObserver):
            m(torch.randn(4000, 8, generator=g) * (0.5 + torch.rand(1, generator=g).item() * 3) + (0.3 if m.qscheme == torch.per_tensor_affine else 0))


def test_observer_placement():
    seq = flm.SparseSequential(nn.PReLU(), nn.Linear(12, 4), nn.PReLU(), isc.Conv3d(4, 4, 3, 1), nn.PReLU(), nn.Linear(4, 32))
    holder = nn.Module(); holder.seq = seq
    flm.insert_obs_into_seqs(holder)
    kinds = [type(m).__name__ for m in holder.seq]
    assert kinds == ['SparseTensorHistogramObserver', 'PReLU', 'SparseTensorHistogramObserver', 'Linear',
                     'SparseTensorHistogramObserver', 'PReLU', 'SparseTensorHistogramObserver', 'Conv3d',
                     'SparseTensorHistogramObserver', 'PReLU', 'SparseTensorHistogramObserver', 'Linear']
    schemes = [m.qscheme for m in holder.seq if isinstance(m, isc.SparseTensorHistogramObserver)]
    sym, aff = torch.per_tensor_symmetric, torch.per_tensor_affine
    assert schemes == [sym, aff, sym, sym, sym, aff]            # affine exactly in front of the linears


def test_sequence_replacement_patterns():
    seq = flm.SparseSequential(nn.PReLU(), nn.Linear(12, 4), nn.PReLU(), isc.Conv3d(4, 4, 3, 1), nn.PReLU(), nn.Linear(4, 32))
    holder = nn.Module(); holder.seq = seq
    flm.insert_obs_into_seqs(holder)
    _feed(holder)
    flm.replace_seqs_with_int_impl(holder)
    assert [type(m).__name__ for m in holder.seq] == ['PReLUIn32Out32', 'RequantFxpToScaledInt8', 'LinearPReLUIn8W8Out8',
                                                      'SparseConvPReLUIn8W8Out8', 'LinearIn8W8Out32']
    # scale chain: what one operator emits is what the next one reads
    lin, conv, last = holder.seq[2], holder.seq[3], holder.seq[4]
    assert torch.equal(lin.scale_out, conv.scale_in) and torch.equal(conv.scale_out, last.scale_in)
    assert torch.equal(holder.seq[1].scale_out, lin.scale_in)

    up = nn.Module(); up.seq = flm.SparseSequential(nn.Linear(12, 4), nn.PReLU(), flm.Block(4), nn.Linear(4, 32))
    flm.insert_obs_into_resblocks(up); flm.insert_obs_into_seqs(up)
    _feed(up)
    flm.replace_resblocks_with_int_impl(up); flm.replace_seqs_with_int_impl(up)
    assert [type(m).__name__ for m in up.seq] == ['RequantFxpToScaledInt8', 'LinearPReLUIn8W8Out32', 'SparseResBlockIn32W8Out32',
                                                  'RequantFxpToScaledInt8', 'LinearIn8W8Out32']


def test_converted_state_dict_loads_into_the_integer_model(tmp_path):
    for more in (False, True):
        cfg = fl.Config(channels=16, use_more_ch_for_multi_step_pred=more, quantize_param=True,
                        int_param_save_path=str(tmp_path / f'int_{more}.pt'))
        model = fl.Model(cfg, 'cpu')
        float_keys = set(model.state_dict())
        assert 'blocks_dec.0.pred.0.0.kernel' in float_keys and 'block_dec_recurrent.dec_init.kernel' in float_keys
        assert 'blocks_dec.3.pred.2.weight' in float_keys and 'blocks_dec.5.upsample.2.conv2.bias' in float_keys
        model.pre_test_hook()
        _feed(model)
        model.post_test_hook()
        saved = torch.load(cfg.int_param_save_path)['state_dict']
        target = IntModel(IntConfig(channels=16, use_more_ch_for_multi_step_pred=more), 'cpu')
        want = target.state_dict()
        assert set(saved) == set(want)
        for k, v in want.items():
            assert saved[k].shape == v.shape and saved[k].dtype == v.dtype, k
        target.load_state_dict(saved)
        assert target.block_dec_recurrent.dec.conv_prelu.requant_mul.dtype == torch.uint32
        assert int(target.block_dec_recurrent.dec_init.scale_in.item()) == 1
