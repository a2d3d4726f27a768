"""ctypes loaders for the in-tree native libraries.  Fail loudly: there is no fallback implementation."""
import ctypes as C
import os

from . import _build

_host = None
_hip = None
_hip_blocking = None

i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
vp = C.c_void_p


class NativeLibraryMissing(RuntimeError):
    pass


def _load(path, what, keep_gil=False):
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f'{what} not built ({path}); run `python -c "import __graft_entry__ as g; g.build()"` at the repo root')
    return C.PyDLL(path) if keep_gil else C.CDLL(path)


def host():
    """libfpcc_host.so with argtypes set."""
    global _host
    if _host is None:
        # FPCC_HOST_LIB: another build of the same sources (the sanitizer builds of tools/r05/sanitize.sh)
        L = _load(os.environ.get('FPCC_HOST_LIB') or _build.HOST_LIB, 'libfpcc_host.so')
        L.fpcc_host_strerror.restype = C.c_char_p
        L.fpcc_host_strerror.argtypes = [i64]
        sig = {
            'fpcc_pmf_to_quantized_cdf': [vp, i64, i32, vp, vp],
            'fpcc_rans_indexed_encode': [vp, vp, i64, vp, vp, vp, vp, i64, i32, vp, i64],
            'fpcc_rans_indexed_decode': [vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, vp],
            'fpcc_rans_binary_encode': [vp, vp, i64, vp, i64],
            'fpcc_rans_binary_decode': [vp, i64, vp, i64, vp],
            'fpcc_rans_binary_encode_multi': [vp, vp, vp, i64, vp, i64, vp, i32],
            'fpcc_simple_enc_push': [vp, vp, i64, i64, vp, i64],
            'fpcc_simple_enc_push_bin': [vp, vp, i64, vp, i64],
            'fpcc_simple_enc_push_ranges': [vp, vp, vp, i64],
            'fpcc_simple_enc_finish': [vp, vp, i64],
            'fpcc_simple_dec_pop': [vp, vp, i64, i64, vp, i64],
            'fpcc_simple_dec_pop_bin': [vp, vp, i64, vp, i64],
            'fpcc_simple_dec_tell': [vp, vp, vp],
            'fpcc_pool_binary_encode': [vp, vp, C.c_uint32, vp, vp, i64, vp, i64, vp],
            'fpcc_pool_histogram_encode': [vp, vp, C.c_uint32, vp, i64, i32, vp, vp, i64, vp, vp, i64, vp],
            'fpcc_pool_table_decode': [vp, vp, i64, i64, vp, i64, i32, vp, i64, vp],
            'fpcc_pool_binary_decode': [vp, vp, i64, vp, i64, vp, vp],
            'fpcc_progress_wait': [vp, i64],
            'fpcc_pool_wait': [vp],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype = i64
            fn.argtypes = args
        L.fpcc_simple_enc_new.restype = vp
        L.fpcc_simple_enc_new.argtypes = [i64]
        L.fpcc_simple_enc_free.restype = None
        L.fpcc_simple_enc_free.argtypes = [vp]
        L.fpcc_simple_dec_new.restype = vp
        L.fpcc_simple_dec_new.argtypes = [vp, i64]
        L.fpcc_simple_dec_free.restype = None
        L.fpcc_simple_dec_free.argtypes = [vp]
        L.fpcc_pool_new.restype = vp
        L.fpcc_pool_new.argtypes = [i32]
        L.fpcc_pool_free.restype = None
        L.fpcc_pool_free.argtypes = [vp]
        _host = L
    return _host


def host_check(code):
    if code < 0:
        raise RuntimeError('libfpcc_host: ' + host().fpcc_host_strerror(code).decode())
    return code


HOST_SYMBOLS = (
    'fpcc_host_strerror', 'fpcc_pmf_to_quantized_cdf', 'fpcc_rans_indexed_encode', 'fpcc_rans_indexed_decode',
    'fpcc_rans_binary_encode', 'fpcc_rans_binary_decode', 'fpcc_rans_binary_encode_multi', 'fpcc_simple_enc_new',
    'fpcc_simple_enc_free', 'fpcc_simple_enc_push', 'fpcc_simple_enc_push_bin', 'fpcc_simple_enc_push_ranges',
    'fpcc_simple_enc_finish', 'fpcc_simple_dec_new', 'fpcc_simple_dec_free', 'fpcc_simple_dec_pop',
    'fpcc_simple_dec_pop_bin', 'fpcc_simple_dec_tell', 'fpcc_pool_new', 'fpcc_pool_free', 'fpcc_pool_binary_encode',
    'fpcc_pool_histogram_encode', 'fpcc_pool_table_decode', 'fpcc_pool_binary_decode', 'fpcc_progress_wait', 'fpcc_pool_wait')


def hip():
    """libfpcc_hip.so (kernels for gfx950).  Signatures are attached in fastpcc_amd/hipops.py."""
    global _hip
    if _hip is None:
        # The entry points of libfpcc_hip only ENQUEUE work (microseconds; the few that wait for the device are bound through
        # hip_blocking() by hipops.lib()), so its calls keep the interpreter lock (PyDLL): released
        # around each of a frame's ~800 launches (CDLL), two frame threads (fastpcc_amd/serving.py) hand the lock back and forth at
        # every launch and each pays a wake-up to get it back.  libfpcc_host's calls block (coder pool waits, rANS passes) and release it.
        # FPCC_HIP_RELEASE_GIL=1 restores the ctypes default (A/B: tools/r04/s51.sh).
        _hip = _load(_build.HIP_LIB, 'libfpcc_hip.so', keep_gil=os.environ.get('FPCC_HIP_RELEASE_GIL', '0') != '1')
    return _hip


def hip_blocking():
    """the same library through a handle whose calls release the interpreter lock: for the entry points that wait for the device"""
    global _hip_blocking
    if _hip_blocking is None:
        _hip_blocking = _load(_build.HIP_LIB, 'libfpcc_hip.so')
    return _hip_blocking
