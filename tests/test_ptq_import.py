"""float -> fixed-point parameter conversion (`import_parameters` of the integer operators) against values produced by the
reference's own conversion code (tests/golden/ptq_import.json, generator: make_golden.py:make_ptq_import)."""
import json
import os
from types import SimpleNamespace as NS

import pytest
import torch

from fastpcc_amd import int_sparse_conv as isc

with open(os.path.join(os.path.dirname(__file__), 'golden', 'ptq_import.json')) as f:
    GOLD = json.load(f)

f32 = lambda v: torch.tensor([v], dtype=torch.float32)
i64 = lambda v: torch.tensor([v], dtype=torch.int64)
F = lambda lst, *shape: torch.tensor(lst, dtype=torch.float64).to(torch.float32).reshape(*shape)


def _same_state(module, want: dict):
    got = module.state_dict()
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, v in want.items():
        g = got[k].flatten()
        if g.dtype.is_floating_point:
            assert torch.equal(g.double(), torch.tensor(v, dtype=torch.float64)), k      # float32 values, exactly
        else:
            assert g.to(torch.int64).tolist() == v, k


@pytest.mark.parametrize('case', GOLD['conv'], ids=lambda c: f"{c['cin']}x{c['cout']}_{'p' if c['prelu'] else 'n'}{'8' if c['out8'] else '32'}_zp{c['zp_in']}")
def test_sparse_conv(case):
    k = case['ks'][0] ** 3
    conv = NS(kernel=F(case['kernel'], k, case['cin'], case['cout']), bias=F(case['bias'], case['cout']))
    act = NS(weight=torch.tensor([case['slope']])) if case['prelu'] else None
    m = isc.SparseConvIn8Out8(case['cin'], case['cout'], tuple(case['ks']), tuple(case['stride']), case['prelu'], case['out8'])
    m.import_parameters(f32(case['s_in']), i64(case['zp_in']), f32(case['s_out']) if case['out8'] else None,
                        i64(case['zp_out']) if case['out8'] else None, conv, act)
    _same_state(m, case['state'])


@pytest.mark.parametrize('case', GOLD['linear'], ids=lambda c: f"{c['cin']}x{c['cout']}_{'p' if c['prelu'] else 'n'}{'8' if c['out8'] else '32'}_zp{c['zp_in']}")
def test_linear(case):
    lin = NS(weight=F(case['weight'], case['cout'], case['cin']), bias=F(case['bias'], case['cout']))
    act = NS(weight=torch.tensor([case['slope']])) if case['prelu'] else None
    m = isc.LinearIn8W8(case['cin'], case['cout'], case['prelu'], case['out8'])
    m.import_parameters(f32(case['s_in']), i64(case['zp_in']), f32(case['s_out']) if case['out8'] else None,
                        i64(case['zp_out']) if case['out8'] else None, lin, act)
    _same_state(m, case['state'])


@pytest.mark.parametrize('case', GOLD['requant'], ids=lambda c: f"s{c['s_out']}")
def test_requantiser(case):
    m = isc.RequantFxpToScaledInt8()
    m.import_parameters(f32(case['s_out']), i64(case['zp_out']))
    _same_state(m, case['state'])


@pytest.mark.parametrize('case', GOLD['prelu'], ids=lambda c: str(c['slope']))
def test_prelu(case):
    m = isc.PReLUIn32Out32()
    m.import_parameters(NS(weight=torch.tensor([case['slope']])))
    _same_state(m, case['state'])


def test_residual_block():
    case = GOLD['resblock'][0]
    ch = case['ch']
    blk = NS(obs=NS(calculate_qparams=lambda: (f32(case['scale']), i64(0))), obs2=NS(calculate_qparams=lambda: (f32(case['scale2']), i64(0))),
             conv=NS(kernel=F(case['kernel'], 27, ch, ch), bias=F(case['bias'], ch)), act=NS(weight=torch.tensor([case['slope']])),
             conv2=NS(kernel=F(case['kernel2'], 27, ch, ch), bias=F(case['bias2'], ch)), act2=NS(weight=torch.tensor([case['slope2']])))
    m = isc.SparseResBlockIn32W8Out32(ch)
    m.import_parameters(blk)
    _same_state(m, case['state'])
