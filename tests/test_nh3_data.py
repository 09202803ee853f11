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


def _def_value(text, name, after=None):
    """The literal of `DEF <name> = <literal>` (the first one behind the marker `after`, if given)."""
    start = text.index(after) if after else 0
    m = re.compile(rf'DEF {name}\s*=\s*([-+0-9.eE]+)').search(text, start)
    assert m, name
    return float(m.group(1))


@pytest.mark.skipif(not REF.exists(), reason='reference tree not present')
def test_constants_match_reference_text():
    """The physical and spectroscopic constants of csrc/nh3_data.h (shared by the engine and the oracle: a wrong literal
    would be common to both) against the reference's own text: model_includes.pxi:27-36, ammonia.pyx:14-30,
    hyperfine.pyx:10-16, core.pyx:20."""
    inc = (REF.parents[2] / 'includes' / 'model_includes.pxi').read_text()
    amm = REF.read_text()
    hyp = (REF.parent / 'hyperfine.pyx').read_text()
    core = (REF.parents[1] / 'core' / 'core.pyx').read_text()
    hdr = (Path(__file__).resolve().parent.parent / 'nestfit_amd' / 'csrc' / 'nh3_data.h').read_text()

    def mine(name, new_const=True):
        ms = re.findall(rf'#define {name}\s+([-+0-9.eE]+)', hdr)
        assert ms, name
        return float(ms[-1] if new_const and len(ms) > 1 else ms[0])
    assert 'DEF __NEW_CONST = True' in inc and 'DEF __APPROX = True' in inc       # the constant set the reference ships
    for ref_name, my_name in (('CKMS', 'NFA_CKMS'), ('CCMS', 'NFA_CCMS'), ('H', 'NFA_H'), ('KB', 'NFA_KB')):
        assert mine(my_name) == _def_value(inc, ref_name)
    assert mine('NFA_TCMB') == _def_value(inc, 'TCMB', after='IF __NEW_CONST')
    assert mine('NFA_BROT') == _def_value(amm, 'BROT', after='IF __NEW_CONST')
    assert mine('NFA_CROT') == _def_value(amm, 'CROT', after='IF __NEW_CONST')
    assert mine('NFA_NPART') == _def_value(amm, 'NPART') and mine('NFA_N_LEVELS') == _def_value(amm, 'N_LEVELS')
    assert mine('NFA_N_PARAMS') == _def_value(amm, 'N_PARAMS')
    assert mine('NFA_MAX_HF_N') == _def_value((REF.parents[2] / 'includes' / 'array_sizes.pxi').read_text(), 'MAX_HF_N')
    # the 1/(e^x - 1) table: 1000 points between (h 23 GHz / k) / 8 K and (h 28 GHz / k) / 2.7 K
    assert _def_value(hyp, 'T0_SIZE') == 1000 and 'H * 23.0e9 / KB' in hyp and 'H * 28.0e9 / KB' in hyp
    assert 'T0_LO / 8.0' in hyp and 'T0_HI / 2.7' in hyp
    assert _def_value(core, 'FWHM') == 2.3548200450309493
    # Swift et al. 2005 eq. A6 literals and the (J, K) ladder of the partition sum
    assert '41.18' in amm and '15.7' in amm and '0.6' in amm
