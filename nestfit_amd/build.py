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
DEPS = sorted((HERE / 'csrc').glob('*.h*')) + [HERE.parent / 'include' / 'nestfit_amd.h']
OUT = HERE / 'lib' / 'libnestfit_amd.so'

FLAGS = [
    '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared',
    # FMA only where the source asks for it: window / table indices must round
    # like the reference's plain double arithmetic
    '-ffp-contract=off',
    '-fno-fast-math',
    '-Wall', '-Wno-unused-function',
]


def needs_build():
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    OUT.parent.mkdir(exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + list(extra) + ['-o', str(OUT), str(SRC)]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a != '--force'])
    print(OUT)
