import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def kat():
    with open(ROOT / 'tests' / 'golden' / 'survey_kat.json') as f:
        return json.load(f)


@pytest.fixture(scope='session')
def nfo():
    """The CPU oracle (test infrastructure; never imported by the product)."""
    from oracle import nfo as _nfo
    _nfo.lib()
    return _nfo


@pytest.fixture(scope='session')
def engine():
    """The HIP engine on cuda:0; fails (not skips) when it cannot run on a GPU box."""
    import nestfit_amd
    from nestfit_amd import _ffi
    _ffi.engine()
    return nestfit_amd


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.maximum(np.abs(b), 1e-300)
    return np.abs(a - b) / denom
