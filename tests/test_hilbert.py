"""Hilbert keys: the oracle's machine, generated from the curve's geometry, against keys of the reference's own 96-entry
table (tests/golden/hilbert.json)."""
import json
import os

import numpy as np

from oracle.hilbert import hilbert3d_encode, state_table

G = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'hilbert.json')))


def test_oracle_matches_reference_keys():
    for case in G:
        assert hilbert3d_encode(np.array(case['xyz']), case['bits'], case['cols']).tolist() == case['keys']


def test_machine_shape_and_curve_properties():
    table = state_table()
    assert len(table) == 12 and all(sorted(d for _, d in row) == list(range(8)) for row in table)
    pts = np.array([[x, y, z] for x in range(16) for y in range(16) for z in range(16)])
    keys = hilbert3d_encode(pts, 4)
    order = np.argsort(keys)
    assert sorted(keys.tolist()) == list(range(4096))                                 # a bijection onto [0, 16^3)
    assert (np.abs(np.diff(pts[order], axis=0)).sum(1) == 1).all()                    # consecutive keys are face neighbours
