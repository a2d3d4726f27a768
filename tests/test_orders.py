"""The product's summation-order rule (fastpcc_amd.engine.summation_order -> libfpcc_hip's fpcc_conv_f32_order_ex, host code: no GPU
needed) against the oracle's independent restatement of the specification (oracle/orders.py).  The chain-order fixtures
(codec_v2_chain.json, codec_color_chain.json) are generated with the ORACLE's rule; this test is what ties the product to it."""
import itertools

import pytest

from fastpcc_amd import engine as ME
from fastpcc_amd import hipops
from oracle import orders

KINDS = ('k1', 'k3', 'k2s2', 'k2s2T', 'gen', 'mlp')


def test_constants_agree():
    assert orders.NUMERICS_VERSION == hipops.numerics_version()
    assert orders.PAD_MIN_ROWS == ME.PAD_MIN_ROWS


def test_every_layer_shape_of_the_in_scope_models():
    """all (kind, channels) combinations the v2 / colour / v3 model code can ask for, on both sides of PAD_MIN_ROWS"""
    widths = (1, 2, 3, 4, 8, 16, 18, 32, 34, 48, 64, 96, 128, 192, 256)
    outs = (1, 3, 4, 8, 16, 32, 64, 128, 255)
    rows = (0, 1, orders.PAD_MIN_ROWS - 1, orders.PAD_MIN_ROWS, 10 ** 6)
    for kind, c1, c2, c_out, n in itertools.product(KINDS, widths, (0, 2, 16, 32, 128), outs, rows):
        assert ME.summation_order(kind, c1, c2, c_out, n) == orders.summation_order(kind, c1, c2, c_out, n), (kind, c1, c2, c_out, n)


@pytest.mark.parametrize('kind,c1,c2,c_out,n,want', [
    ('k3', 128, 0, 128, 5, 3), ('k3', 128, 128, 128, 5, 3), ('k3', 64, 0, 64, 10 ** 6, 3),      # grouped
    ('k1', 128, 0, 128, 5, 1), ('k2s2T', 128, 0, 64, 5, 1), ('k3', 16, 0, 32, 5, 1), ('k3', 48, 0, 32, 5, 1),   # MFMA chain
    ('k2s2', 16, 0, 64, 5, 1), ('k2s2', 32, 0, 64, 5, 3),                                  # 8 offsets: grouped only with 32-channel chunks
    ('k3', 32, 0, 1, 5, 2), ('k3', 128, 0, 1, 10 ** 6, 2),                                  # two-phase one-channel
    ('k3', 1, 0, 16, 10 ** 6, 0), ('k1', 16, 0, 8, 8191, 0), ('k1', 16, 0, 8, 8192, 1),       # natural chain; zero-padded from PAD_MIN_ROWS rows
    ('k3', 34, 0, 16, 8191, 0), ('k3', 34, 0, 16, 8192, 1), ('k3', 18, 0, 32, 8192, 3),
    ('gen', 64, 0, 16, 5, 1), ('gen', 128, 0, 32, 5, 1), ('gen', 8, 0, 16, 5, 0)])
def test_documented_examples(kind, c1, c2, c_out, n, want):
    assert orders.summation_order(kind, c1, c2, c_out, n) == want == ME.summation_order(kind, c1, c2, c_out, n)
