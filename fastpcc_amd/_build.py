"""Compiles the two native libraries of the package in-tree.

    libfpcc_host.so   g++      host entropy coders                      (include/fpcc_host.h)
    libfpcc_hip.so    hipcc    gfx950 kernels + their C ABI              (include/fpcc_hip.h)

Both are plain shared objects loaded with ctypes; nothing is JIT-built at import time and nothing is installed outside
the repository, so the artefacts travel with the tree to the GPU box.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
HOST_LIB = os.path.join(CSRC, 'libfpcc_host.so')
HIP_LIB = os.path.join(CSRC, 'libfpcc_hip.so')


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError('build failed: ' + ' '.join(cmd))
    return proc.stdout


def build_host(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, 'host', '*.cpp')))
    deps = srcs + glob.glob(os.path.join(PKG, '..', 'include', '*.h'))
    if force or _stale(HOST_LIB, deps):
        # x86-64-v3 (AVX2): the symbol searches of the coders are short fixed-length compare loops
        out = _run(['g++', '-O3', '-march=x86-64-v3', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-Wextra',
                    '-o', HOST_LIB] + srcs)
        if verbose:
            print(out)
    return HOST_LIB


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, 'hip', '*.hip')))
    deps = srcs + glob.glob(os.path.join(CSRC, 'hip', '*.h')) + glob.glob(os.path.join(PKG, '..', 'include', '*.h'))
    if force or _stale(HIP_LIB, deps):
        objs = []
        for s in srcs:
            o = s[:-4] + '.o'
            if force or _stale(o, [s] + [d for d in deps if d.endswith('.h')]):
                out = _run([hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
                            '-Wall', '-Wno-unused-function', '-o', o, s])
                if verbose:
                    print(out)
            objs.append(o)
        out = _run([hipcc_path(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', HIP_LIB] + objs)
        if verbose:
            print(out)
    return HIP_LIB


def build_all(force=False, verbose=False):
    return build_host(force, verbose), build_hip(force, verbose)


if __name__ == '__main__':
    print(build_all(force='--force' in sys.argv, verbose=True))
