"""The C-ABI library loads without a GPU and exports exactly what the header declares."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(header='nestfit_amd.h'):
    text = (ROOT / 'include' / header).read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nfa_[a-z0-9_]+)\s*\(', text)) - {'nfa_broker_loglike_fn', 'nfa_loglike_callback_fn'})


def test_library_exports_every_declared_symbol():
    from nestfit_amd import _ffi
    lib = _ffi.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/nestfit_amd.h but not exported'
    assert sorted(_ffi.SIGNATURES) == names
    assert lib.nfa_version() >= 100


def test_test_hooks_live_in_the_test_library_only():
    """include/nestfit_amd_test.h is what libnestfit_amd_test.so adds; the product library exports none of
    it (no unit-test hooks, no timing ablation in what ships)."""
    import ctypes as C
    from nestfit_amd import _ffi
    product = _ffi.load()
    hooks = _declared('nestfit_amd_test.h')
    assert sorted(_ffi.TEST_SIGNATURES) == hooks and len(hooks) == 8
    test_lib = C.CDLL(str(_ffi.TEST_LIB_PATH))
    for n in hooks:
        assert hasattr(test_lib, n), f'{n} declared in include/nestfit_amd_test.h but not exported'
        assert not hasattr(product, n), f'{n} must not be in the product library'
    for n in _declared():
        assert hasattr(test_lib, n)                       # the test library is the whole engine plus the hooks
    assert product.nfa_set_option(b'ablate', 1) != 0      # the shipped library rejects the timing experiment key
    assert product.nfa_set_exp_mode(1) != 0               # the polynomial likelihood mode is gone (0 table, 2 fast)
    assert b'table' in product.nfa_last_error()


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a host without a device every compute call raises."""
    import nestfit_amd as na
    if na.device_count() > 0:
        pytest.skip('a GPU is visible; covered by the gpu suite')
    x = np.linspace(23.69e9, 23.70e9, 64)
    with pytest.raises(na.EngineError):
        na.AmmoniaSpectrum(x, np.zeros(64), 0.1, 1)
    ut = na.get_irdc_priors(size=50)
    with pytest.raises(na.EngineError):
        ut.transform(np.full(6, 0.5), 1)


def test_product_never_imports_the_oracle():
    for f in list((ROOT / 'nestfit_amd').rglob('*.py')) + list((ROOT / 'nestfit_amd').rglob('*.hip')):
        assert 'oracle' not in f.read_text().replace('test oracle', '').replace('CPU oracle', ''), f


def test_header_is_plain_c_and_links(tmp_path):
    """include/nestfit_amd.h compiles as C99 (no C++ or torch types at the boundary) and a C program
    that references every declared entry point links against the built library."""
    import re
    import subprocess
    from nestfit_amd import _ffi
    from nestfit_amd.build import OUT, build
    build()
    header = (ROOT / 'include' / 'nestfit_amd.h').read_text()
    names = sorted(set(re.findall(r'\b(nfa_[a-z0-9_]+)\s*\(', header)) & set(_ffi.SIGNATURES))
    assert len(names) == len(_ffi.SIGNATURES)
    src = tmp_path / 'use_abi.c'
    table = ',\n'.join(f'    (fn_t){n}' for n in names)
    src.write_text('#include "nestfit_amd.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\n'
                   'static fn_t table[] = {\n' + table + '\n};\n'
                   'int main(void) {\n    printf("%u\\n", (unsigned)(sizeof table / sizeof table[0]));\n'
                   '    return table[0] == 0;\n}\n')
    exe = tmp_path / 'use_abi'
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', f'-I{ROOT / "include"}', str(src), '-o', str(exe),
           f'-L{OUT.parent}', '-lnestfit_amd', f'-Wl,-rpath,{OUT.parent}', '-Wl,--allow-shlib-undefined']
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
