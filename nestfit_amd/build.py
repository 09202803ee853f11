"""Build the HIP engine in-tree for gfx950: nestfit_amd/lib/libnestfit_amd.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the snapshot (it is git-ignored).
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / 'csrc' / 'nfa_engine.hip'
DEPS = sorted((HERE / 'csrc').glob('*.h*')) + [HERE / 'csrc' / 'nfa_ring.cpp'] + [HERE.parent / 'include' / 'nestfit_amd.h']
OUT = HERE / 'lib' / 'libnestfit_amd.so'
# the same engine with the unit-test hooks and the "ablate" timing option compiled in: tests and
# measurement scripts only (include/nestfit_amd_test.h); the product library has neither
OUT_TEST = HERE / 'lib' / 'libnestfit_amd_test.so'
TEST_FLAGS = ['-DNFA_TEST_HOOKS', '-DNFA_ABLATE']
# the shared-memory ring alone (csrc/nfa_ring.h), host code without HIP: what a sampler process loads
OUT_RING = HERE / 'lib' / 'libnestfit_amd_ring.so'
SRC_RING = HERE / 'csrc' / 'nfa_ring.cpp'
# host helper of hdf5.py (batches of HDF5 calls per native call; dlopens the HDF5 library itself)
OUT_H5 = HERE / 'lib' / 'libnestfit_amd_h5.so'
SRC_H5 = HERE / 'csrc' / 'nfa_h5.cpp'

FLAGS = [
    '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared',
    # FMA only where the source asks for it: window / table indices must round
    # like the reference's plain double arithmetic
    '-ffp-contract=off',
    '-fno-fast-math',
    '-Wall', '-Wno-unused-function',
]


def needs_build(out=OUT):
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def _compile(out, flags, verbose):
    out.parent.mkdir(exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + list(flags) + ['-o', str(out), str(SRC)]
    if verbose:
        print(' '.join(cmd))
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def build(force=False, verbose=False, extra=(), test_lib=True):
    """Compile the product library and (test_lib) the test library, side by side."""
    jobs = []
    if force or needs_build(OUT):
        jobs.append((OUT, _compile(OUT, extra, verbose)))
    if test_lib and (force or needs_build(OUT_TEST)):
        jobs.append((OUT_TEST, _compile(OUT_TEST, list(extra) + TEST_FLAGS, verbose)))
    if force or needs_build(OUT_RING):
        cxx = os.environ.get('CXX', 'g++')
        cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-o', str(OUT_RING), str(SRC_RING), '-lrt', '-pthread']
        if verbose:
            print(' '.join(cmd))
        jobs.append((OUT_RING, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    if force or not OUT_H5.exists() or OUT_H5.stat().st_mtime < SRC_H5.stat().st_mtime:
        cxx = os.environ.get('CXX', 'g++')
        cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-o', str(OUT_H5), str(SRC_H5), '-ldl']
        if verbose:
            print(' '.join(cmd))
        jobs.append((OUT_H5, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for out, proc in jobs:
        text, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f'compiling {out.name} failed:\n' + text)
        if verbose and text:
            print(text)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a != '--force'])
    print(OUT)
