"""Cross-checks the line-data header against the literals in the reference text
(read as text only; skipped where /root/reference does not exist, e.g. the GPU box)."""
import re
from pathlib import Path

import numpy as np
import pytest

REF = Path('/root/reference/nestfit/models/ammonia.pyx')


def _block(text, start, end):
    """Text between the end of `start` and the next `end`."""
    a = text.index(start) + len(start)
    b = text.index(end, a)
    return text[a:b]


def _floats(txt):
    txt = re.sub(r'#.*', '', txt)
    return [float(v) for v in re.findall(r'[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?', txt)]


@pytest.mark.skipif(not REF.exists(), reason='reference tree not present')
def test_line_data_matches_reference_text(nfo):
    text = REF.read_text()
    lib = nfo.lib()
    nhf = [int(v) for v in _floats(_block(text, 'NHF = [', ']'))]
    assert nhf == [lib.nfo_trans_nhf(t) for t in range(1, 10)]
    nu = _floats(_block(text, 'NU = [', ']'))
    assert nu == [lib.nfo_trans_nu(t) for t in range(1, 10)]
    ea = _floats(_block(text, '4 significant digits.\n    EA = [', ']'))
    assert ea[:9] == [lib.nfo_trans_ea(t) for t in range(1, 10)]
    for t in range(1, 10):
        v = _floats(_block(text, f'VOFF[{t-1}][:NHF[{t-1}]] = [', ']'))
        w = _floats(_block(text, f'TAU_WTS[{t-1}][:NHF[{t-1}]] = [', ']'))
        assert len(v) == len(w) == nhf[t - 1]
        assert v == [lib.nfo_trans_voff(t, i) for i in range(nhf[t - 1])]
        assert w == [lib.nfo_trans_tau_wt(t, i) for i in range(nhf[t - 1])]


def test_weights_are_normalised(nfo):
    lib = nfo.lib()
    for t in range(1, 10):
        n = lib.nfo_trans_nhf(t)
        tot = sum(lib.nfo_trans_tau_wt(t, i) for i in range(n))
        assert tot == pytest.approx(1.0, abs=2e-3)
