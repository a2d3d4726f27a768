"""The two shared libraries load on a machine without a GPU and export every symbol their headers declare."""
import ctypes
import os
import re

from fastpcc_amd import _build, _native, hipops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, 'include', header)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fpcc_[a-z0-9_]+)\s*\(', text)))


def test_host_library_exports_header():
    lib = ctypes.CDLL(_build.HOST_LIB)
    names = _declared('fpcc_host.h')
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_native.HOST_SYMBOLS)


def test_hip_library_exports_header():
    lib = ctypes.CDLL(_build.HIP_LIB)
    names = _declared('fpcc_hip.h')
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(hipops.HIP_SYMBOLS)


def test_device_ops_fail_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises((hipops.FpccError, TypeError)):
        hipops.keys_from_coords(torch.zeros((4, 4), dtype=torch.int32), 0, 21)
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    with pytest.raises(RuntimeError):
        Model(baseline_r1()).compress(torch.zeros((4, 4), dtype=torch.int32))
