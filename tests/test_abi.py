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


def test_struct_layouts_match_the_header(tmp_path):
    """the ctypes mirrors of the header's structs (descriptor tables handed to fpcc_int_level_*, the extra int8 outputs of
    fpcc_conv_i8_also): sizes and field offsets as a C compiler lays out include/fpcc_hip.h"""
    import subprocess
    mirrors = {'fpcc_i8_layer': hipops.I8Layer, 'fpcc_i8_requant': hipops.I8Requant, 'fpcc_int_onescale': hipops.IntOneScale,
               'fpcc_requant8': hipops._Requant8}
    lines = []
    for c_name, cls in mirrors.items():
        lines.append(f'printf("{c_name} size %zu\\n", sizeof({c_name}));')
        for field, _ in cls._fields_:
            lines.append(f'printf("{c_name} {field} %zu\\n", offsetof({c_name}, {field}));')
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fpcc_hip.h"\nint main(void) {\n' + '\n'.join(lines) + '\nreturn 0; }\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    seen = 0
    for line in out.splitlines():
        c_name, what, value = line.split()
        cls = mirrors[c_name]
        assert int(value) == (ctypes.sizeof(cls) if what == 'size' else getattr(cls, what).offset), line
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in mirrors.values())
