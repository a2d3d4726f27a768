"""TEST ORACLE -- not product code.

CPU restatement of the reference's encode/decode hot path (see SURVEY.md section 8c and DESIGN.md).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  ``fastpcc_amd`` never imports it.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile oracle/*.c into oracle/liboracle.so (gcc; a few seconds)."""
    path = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('rans.c', 'sparse_conv.c', 'int_ops.c')]
    stale = force or not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if stale:
        subprocess.run(['make', '-C', _HERE, 'oracle'] + (['-B'] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return path


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def ref_dir() -> str:
    """Directory holding the reference's own coders compiled by ``make -C oracle ref`` (may be empty)."""
    return os.path.join(_HERE, '_ref')
