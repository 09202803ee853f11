#!/usr/bin/env python3
"""Golden set G1 of SURVEY.md 8(c): FastExp of the reference itself.  Runs the reference's own
nestfit/core/fastexp.c, compiled as it lies under /root/reference by oracle/Makefile into
oracle/_ref/libfastexp_ref.so (gcc, libm only; tables calcExpTableEntries(3, 8) like
includes/model_includes.pxi:12), on 4096 float inputs -- every branch edge of fastexp.c:259-273
(0, denormals, 2^-5 -/+ 1 ulp, 32 -/+ 1 ulp, negative arguments, beyond 32), values uniform in bit pattern
over [2^-8, 64) so that every table family is hit, and values uniform over the 5-sigma window range
[0, 13) -- and stores inputs and outputs as raw bits:

    python tests/golden/make_g1_fastexp.py      ->  tests/golden/g1_fastexp.npz  (x: float32, y: float64)

Needs the reference tree (this container); the fixture it writes does not."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def inputs():
    f32 = np.float32
    xs = [0.0, -0.0, 1e-9, 1e-30, 1e-40, 0.01, 0.1, 1.0, 12.5, 31.9, 32.0, 40.0, 1e30, -1.0, -0.5, -20.0]
    for p in range(-8, 7):
        b = f32(2.0) ** f32(p)
        xs += [b, np.nextafter(b, f32(0)), np.nextafter(b, f32(100))]
    edges = np.array(xs, dtype=np.float32)
    rng = np.random.default_rng(20261004)
    n_rest = 4096 - edges.size
    bits = rng.integers(f32(2.0**-8).view(np.uint32), f32(64.0).view(np.uint32), size=n_rest // 2, dtype=np.uint32)
    window = rng.uniform(0, 13, n_rest - n_rest // 2).astype(np.float32)
    return np.concatenate([edges, bits.view(np.float32), window])


def main():
    from oracle import nfo
    ref = nfo.ref_fastexp_lib()
    if ref is None:
        raise SystemExit('oracle/_ref is not built: the reference tree is needed to make this fixture')
    x = inputs()
    y = np.array([ref.FastExp(C.c_float(float(v))) for v in x], dtype=np.float64)
    out = Path(__file__).with_name('g1_fastexp.npz')
    np.savez_compressed(out, x=x, y=y)
    print(out, x.size, 'inputs;', int(np.sum(y == 0)), 'zeros;', 'max', y.max())


if __name__ == '__main__':
    main()
