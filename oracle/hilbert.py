"""TEST ORACLE -- not product code.

3-D Hilbert keys as `hilbert3d_encode_lut` of the reference computes them
(/root/reference/lib/space_filling_curves/src/hilbert3d.cu:28-60: per bit, from the most significant down, a 12-state machine
maps the Morton octant x | y << 1 | z << 2 to a 3-bit Hilbert digit and the next state).  The reference holds the machine
as a 96-entry table; this restatement GENERATES it from the curve's geometry -- the order in which the root cube visits its
octants and, for every visited octant, the signed axis permutation that maps the root curve onto the child's curve -- and is
PINNED by keys computed from the reference's own table (tests/golden/hilbert.json, tests/golden/make_golden.py)."""
from typing import List, Tuple

import numpy as np

# order in which the root state visits the octants (x, y, z): the reflected Gray path x, y, x, z, x, y, x
BASE = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 0, 1), (0, 0, 1)]
# child k runs the root curve under p -> (flip_i xor p[perm_i])_i
CHILD = [((2, 0, 1), (0, 0, 0)), ((1, 2, 0), (0, 0, 0)), ((1, 2, 0), (0, 0, 0)), ((0, 1, 2), (1, 1, 0)),
         ((0, 1, 2), (1, 1, 0)), ((1, 2, 0), (0, 1, 1)), ((1, 2, 0), (0, 1, 1)), ((2, 0, 1), (1, 0, 1))]


def _apply(t, p):
    perm, flips = t
    return tuple(flips[i] ^ p[perm[i]] for i in range(3))


def _compose(a, b):
    (pa, fa), (pb, fb) = a, b
    return tuple(pb[pa[i]] for i in range(3)), tuple(fa[i] ^ fb[pa[i]] for i in range(3))


def state_table() -> List[List[Tuple[int, int]]]:
    """table[state][morton octant] = (next state, Hilbert digit); states numbered in breadth-first order from the root"""
    states, table, i = [((0, 1, 2), (0, 0, 0))], [], 0
    while i < len(states):
        row = [None] * 8
        for k in range(8):
            o = _apply(states[i], BASE[k])
            nxt = _compose(states[i], CHILD[k])
            if nxt not in states:
                states.append(nxt)
            row[o[0] | o[1] << 1 | o[2] << 2] = (states.index(nxt), k)
        table.append(row)
        i += 1
    return table


def hilbert3d_encode(xyz: np.ndarray, bits: int, cols=(0, 1, 2)) -> np.ndarray:
    """xyz int [n, >=3]; cols = the columns that play x, y, z (axis_order of the reference); -> int64 keys"""
    table = state_table()
    nxt = np.array([[c[0] for c in row] for row in table], dtype=np.int64)
    dig = np.array([[c[1] for c in row] for row in table], dtype=np.int64)
    v = np.asarray(xyz).astype(np.int64)
    x, y, z = (v[:, c] for c in cols)
    state = np.zeros(len(v), dtype=np.int64)
    key = np.zeros(len(v), dtype=np.int64)
    for b in range(bits - 1, -1, -1):
        o = ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2)
        key = (key << 3) | dig[state, o]
        state = nxt[state, o]
    return key
