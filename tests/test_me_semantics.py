"""Engine conventions pinned by what the REFERENCE states about them (tests/golden/me_semantics.json, emitted by
tests/golden/make_golden.py from /root/reference -- nothing here is typed in by hand):

  * child table of `minkowski_expand_coord_2x` (lib/minkowski_sparse_conv_layers.py:403-408) == the oracle's even-kernel offset
    enumeration and the engine's generated-set row order (x fastest, anchored at 0);
  * `unfold_kernel` / `fold2bin` identity of lossl_coord_me (ME layout [K, C_in, C_out]) and lossl_coord_int (z fastest);
  * state_dict key / shape lists of the reference's models, built on a parameter-only stub engine, == the product's.
MinkowskiEngine itself cannot be run here; these are the statements about it that the reference
repository does hold."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import coords as oc

with open(os.path.join(os.path.dirname(__file__), 'golden', 'me_semantics.json')) as f:
    G = json.load(f)


@pytest.mark.parametrize('stride', [2, 4, 16])
def test_even_kernel_offsets_are_the_references_child_table(stride):
    table = np.array(G['expand_coord_2x'][str(stride)])                  # [8, 4] = (batch, x, y, z) offsets of the 8 children
    assert (table[:, 0] == 0).all()
    assert oc.kernel_offsets(2, stride // 2).tolist() == table[:, 1:].tolist()
    lvl = oc.Level(np.array([[0, 3 * stride, 5 * stride, 1 * stride], [0, 0, 0, 2 * stride]]), stride)
    gen = oc.generated(lvl)                                              # row 8p + k = parent + table[k]
    assert (gen.coords.reshape(2, 8, 4) - lvl.coords[:, None] == table[None]).all()


def test_me_twin_codec_states_kernel_layout_and_child_order():
    me = G['lossl_coord_me']
    # the (8, 1, 8) identity assigned to MinkowskiConvolution(1, 8, 2, 2).kernel: layout [K, C_in, C_out], kernel index k
    # writes output channel k -- so "kernel index == child index" with the child table below
    assert me['fold2bin_kernel_shape'] == [8, 1, 8] and me['fold2bin_kernel'] == np.eye(8).tolist()
    assert me['unfold_kernel'] == G['expand_coord_2x']['2']                   # same table: x fastest
    assert oc.kernel_offsets(2, 1).tolist() == [r[1:] for r in me['unfold_kernel']]
    assert me['bin2oct_kernel'] == list(range(7, -1, -1))


def test_int_codec_child_order_is_z_fastest():
    from oracle import codec_int as oi
    it = G['lossl_coord_int']
    table = np.array(it['unfold_kernel'])                                    # (batch, x, y, z), z fastest
    assert table[:, 1:].tolist() == [[k >> 2 & 1, k >> 1 & 1, k & 1] for k in range(8)]
    assert it['fold2bin_kernel_shape'] == [8, 8, 1] and it['fold2bin_kernel'] == np.eye(8).tolist()      # [K, C_out, C_in]
    # the oracle's kernel map of a 2x2x2 / stride-2 kernel enumerates the children in exactly this order
    parent = np.array([[0, 4, 4, 4]])
    kids = np.array([[0, 8, 8, 8]]) + table
    t = oi.kernel_table(kids, parent, (2, 2, 2), (2, 2, 2))
    assert t[:, 0].tolist() == list(range(8))
    # ... and the product's buffers are the reference's
    from fastpcc_amd.codecs.lossl_coord_int.model import Model
    from fastpcc_amd.codecs.lossl_coord_int.model_config import Config
    m = Model(Config(), 'cpu')
    assert m.unfold_kernel[0].tolist() == it['unfold_kernel'] and m.bin2oct_kernel.tolist() == it['bin2oct_kernel']


def _keys(model):
    return [[k, list(v.shape) if isinstance(v, torch.Tensor) else None] for k, v in model.state_dict().items()]


def test_v2_state_dict_is_the_references():
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    assert _keys(Model(baseline_r1())) == G['lossy_coord_v2/baseline_r1']


def test_colour_state_dict_is_the_references():
    from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
    from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
    assert _keys(Model(baseline_r1())) == G['lossy_coord_lossy_color/baseline_r1']


def test_int_state_dict_is_the_references():
    from fastpcc_amd.codecs.lossl_coord_int.model import Model
    from fastpcc_amd.codecs.lossl_coord_int.model_config import Config
    assert _keys(Model(Config(), 'cpu')) == G['lossl_coord_int']['state_dict']
