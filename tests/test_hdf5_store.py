"""The store as real HDF5 (nestfit_amd/hdf5.py: ctypes over the HDF5 C library; reference layout
nestfit/main.py:233-377, docs/store_spec.rst:45-110).  Checked three ways: round trips through the
package's own reader, the HDF5 project's `h5dump` where the tool is installed (an independent reader
of the same files), and a fit on CPU whose store is walked against the names of the specification."""
import os
import re
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from nestfit_amd import hdf5
from nestfit_amd.store import HdfStore, StoreFile, store_format

pytestmark = pytest.mark.skipif(not hdf5.available(), reason='no libhdf5 on this host')


def _h5dump():
    for cand in (shutil.which('h5dump'), '/opt/conda/bin/h5dump', '/usr/bin/h5dump'):
        if cand and Path(cand).exists():
            return cand
    return None


def _dump(*args):
    tool = _h5dump()
    if tool is None:
        pytest.skip('no h5dump on this host')
    out = subprocess.run([tool, *map(str, args)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return out.stdout


def test_library_is_found_and_is_the_default_format(monkeypatch):
    assert hdf5.library_version()[0] == 1
    monkeypatch.delenv('NFA_STORE_FORMAT', raising=False)
    assert store_format() == 'hdf5'
    monkeypatch.setenv('NFA_STORE_FORMAT', 'npz')
    assert store_format() == 'npz'


def test_every_value_kind_round_trips(tmp_path):
    rng = np.random.default_rng(0)
    f = StoreFile(tmp_path / 'a.hdf')
    attrs = {
        'an_int': 7, 'a_float': -2.5, 'a_bool': True, 'a_numpy_bool': np.bool_(False), 'a_str': 'RA---SIN',
        'unicode': 'T$_{ex}$ µK', 'empty_str': '', 'np_int32': np.int32(-3), 'np_float32': np.float32(1.25),
        'str_list': ['voff', 'trot', 'tex'], 'float_list': [0.1, 0.2], 'int_tuple': (1, 2, 3),
        'f8_array': rng.normal(size=5), 'i8_array': np.arange(4), 'u1_array': np.arange(3, dtype=np.uint8),
        'two_d': rng.normal(size=(2, 3)), 'bool_array': np.array([True, False, True]),
        'nothing': None,
    }
    g = f.require_group('/pix/12/7')
    g.attrs.update(attrs)
    for dt in ('f8', 'f4', 'i8', 'i4', 'i2', 'u2'):
        g.create_dataset('d_' + dt, data=(100 * rng.normal(size=(3, 4, 2))).astype(dt))
    g.create_dataset('scalar', data=np.float64(3.5))
    g.create_dataset('empty', data=np.zeros((0, 8)))
    g.create_dataset('flags', data=np.array([[True, False]]))
    g.create_dataset('names', data=np.array(['a', 'bb']))
    f.close()

    r = StoreFile(tmp_path / 'a.hdf', 'r')
    back = r['/pix/12/7']
    assert 'nothing' not in back.attrs                       # like h5py: None has no HDF5 form
    for k, v in attrs.items():
        if v is None:
            continue
        got = back.attrs[k]
        if isinstance(v, (list, tuple)) and isinstance(v[0], str):
            assert got == list(v)
        elif isinstance(v, (list, tuple, np.ndarray)):
            np.testing.assert_array_equal(got, np.asarray(v))
            assert np.asarray(got).dtype == np.asarray(v).dtype
        else:
            assert got == v and type(got) in (int, float, bool, str)
    for name, d in g._datasets.items():
        got = back[name]
        if name == 'names':
            assert list(got) == ['a', 'bb']
            continue
        np.testing.assert_array_equal(got, d)
        assert got.dtype == d.dtype and got.shape == d.shape
    with pytest.raises(TypeError):
        bad = StoreFile(tmp_path / 'bad.hdf')
        bad.attrs['obj'] = {'a': 1}
        bad.close()
    with pytest.raises(hdf5.Hdf5Error):
        (tmp_path / 'not_hdf.hdf').write_bytes(b'junk' * 100)
        StoreFile(tmp_path / 'not_hdf.hdf', 'r')


def test_external_links_between_table_and_chunks(tmp_path):
    store_dir = tmp_path / 'run.store'
    store_dir.mkdir()
    for k in range(2):
        with StoreFile(store_dir / f'chunk{k}.hdf') as ch:
            for lat in range(3):
                run = ch.require_group(f'/pix/{k}/{lat}').create_group('1')
                run.attrs['global_lnZ'] = float(10 * k + lat)
                run.create_dataset('posteriors', data=np.full((4, 8), 10 * k + lat, dtype='f8'))
    table = StoreFile(store_dir / 'table.hdf')
    chunks = [StoreFile(store_dir / f'chunk{k}.hdf', 'r') for k in range(2)]
    for k, ch in enumerate(chunks):
        for lat in range(3):
            table[f'/pix/{k}/{lat}'] = ch[f'/pix/{k}/{lat}']
    table.attrs['nchunks'] = 2
    table.close()
    # the table holds links, not copies: it stays small and follows a chunk that is rewritten
    assert (store_dir / 'table.hdf').stat().st_size < (store_dir / 'chunk0.hdf').stat().st_size
    with StoreFile(store_dir / 'chunk1.hdf') as ch:
        ch['/pix/1/2/1'].attrs['global_lnZ'] = -1.0
    again = StoreFile(store_dir / 'table.hdf', 'r')
    assert again['/pix/1/2/1'].attrs['global_lnZ'] == -1.0 and again['/pix/0/1/1'].attrs['global_lnZ'] == 1.0
    assert again['/pix/1/0/1/posteriors'][0, 0] == 10.0
    # a store that has moved as a whole still resolves (relative file names), one that lost a chunk drops its pixels
    moved = tmp_path / 'elsewhere.store'
    shutil.copytree(store_dir, moved)
    (moved / 'chunk0.hdf').unlink()
    lost = StoreFile(moved / 'table.hdf', 'r')
    assert '/pix/1/1' in lost and '/pix/0/1' not in lost
    # the HDF5 project's own tool sees the same structure
    text = _dump('-H', store_dir / 'table.hdf')
    assert text.count('EXTERNAL_LINK') == 6 and 'TARGETFILE "chunk1.hdf"' in text and 'TARGETPATH "/pix/1/2"' in text
    data = _dump('-d', '/pix/0/2/1/posteriors', store_dir / 'table.hdf')
    assert re.search(r'\(0,0\): 2, 2, 2, 2, 2, 2, 2, 2', data)


def test_a_fitted_store_follows_the_specification(tmp_path, monkeypatch):
    """A cube fitted on CPU (numpy twin of the sampler fed by the oracle) into HDF5 files; every name of
    docs/store_spec.rst:57-96 is looked up with h5dump, values against the package's reader."""
    from test_fitter_cpu import _check_store, _fitter, _stack
    monkeypatch.setenv('NFA_STORE_FORMAT', 'hdf5')
    name = str(tmp_path / 'run')
    _fitter(_stack()).fit_cube(name, nproc=2)
    _check_store(name, 4)
    store_dir = tmp_path / 'run.store'
    assert sorted(p.name for p in store_dir.iterdir()) == ['chunk0.hdf', 'chunk1.hdf', 'table.hdf']
    assert (store_dir / 'table.hdf').read_bytes()[:8] == b'\x89HDF\r\n\x1a\n'
    head = _dump('-H', store_dir / 'table.hdf')
    for attr in ('lnZ_threshold', 'multinest_kwargs', 'n_max_components', 'naxis1', 'naxis2', 'nchunks',
                 'model_name', 'n_params', 'par_names', 'par_names_short', 'tex_labels', 'tex_labels_with_units'):
        assert f'ATTRIBUTE "{attr}"' in head, attr
    for group in ('pix', 'full_header', 'simple_header'):
        assert f'GROUP "{group}"' in head
    assert 'EXTERNAL_LINK' in head and 'TARGETFILE "chunk0.hdf"' in head
    run = _dump('-H', '-g', '/pix/0/0/1', store_dir / 'chunk0.hdf')
    # (the specification also lists par_names at this level; the reference's writer, core.pyx:648-676, does not
    # write it and adds the three null_* criteria: the writer is what is followed)
    for attr in ('AIC', 'AICc', 'BIC', 'global_lnZ', 'global_lnZ_err', 'marg_cols', 'marg_quantiles', 'max_loglike',
                 'n_chan_tot', 'n_live', 'n_params', 'n_samples', 'ncomp', 'null_lnZ', 'null_AIC', 'null_AICc',
                 'null_BIC'):
        assert f'ATTRIBUTE "{attr}"' in run, attr
    for dset in ('bestfit_params', 'map_params', 'marginals', 'posteriors'):
        assert f'DATASET "{dset}"' in run, dset
    with HdfStore(name) as store:
        g = store.hdf['/pix/0/0']
        lnz = g['1'].attrs['global_lnZ']
    text = _dump('-a', '/pix/0/0/1/global_lnZ', store_dir / 'chunk0.hdf')
    assert float(re.search(r'\(0\): (\S+)', text).group(1)) == pytest.approx(lnz, rel=1e-5)


def test_post_processing_products_land_in_the_table(tmp_path, monkeypatch):
    from test_fitter_cpu import _fitter, _stack
    from nestfit_amd import postprocess
    monkeypatch.setenv('NFA_STORE_FORMAT', 'hdf5')
    name = str(tmp_path / 'run')
    _fitter(_stack()).fit_cube(name, nproc=1)
    with HdfStore(name) as store:
        postprocess.aggregate_run_attributes(store)
        postprocess.convolve_evidence(store, postprocess.get_indep_info_kernel(0.8))
        postprocess.aggregate_run_products(store)
    with HdfStore(name) as store:
        nbest = store.hdf['/products/nbest']
        assert nbest.shape == (1, 4) and store.hdf['/products/evidence'].ndim == 3
    head = _dump('-H', '-g', '/products', tmp_path / 'run.store' / 'table.hdf')
    for dset in ('nbest', 'evidence', 'evidence_err', 'AIC', 'AICc', 'BIC', 'conv_evidence', 'conv_nbest',
                 'marg_quantiles', 'nbest_MAP', 'nbest_bestfit', 'nbest_marginals'):
        assert f'DATASET "{dset}"' in head, dset


def test_native_helper_and_plain_ctypes_paths_agree(tmp_path, monkeypatch):
    """csrc/nfa_h5.cpp (the attributes / datasets of an object in one native call) against the call-by-call
    ctypes path: files written by either are read identically by both."""
    if hdf5._fast() is None:
        pytest.skip('helper library not built')
    def tree(path):
        rng = np.random.default_rng(5)
        f = StoreFile(path)
        g = f.require_group('/pix/1/2')
        g.attrs.update(i_lon=1, i_lat=2, nbest=2, ok=True, label='µ', cols=['a', 'b'], q=np.linspace(0, 1, 5),
                       six=rng.normal(size=(2, 1, 3, 1, 2, 2)), empty=np.zeros(0), u=np.uint16(7))
        g.create_dataset('posteriors', data=rng.normal(size=(9, 8)).astype('f4'))
        g.create_dataset('six', data=rng.normal(size=(2, 1, 3, 1, 2, 2)))
        g.create_dataset('none', data=np.zeros((0, 4)))
        return f

    def same(a, b):
        assert set(a.attrs) == set(b.attrs) and set(a._datasets) == set(b._datasets)
        for k, v in a.attrs.items():
            w = b.attrs[k]
            assert type(v) is type(w), k
            assert np.array_equal(v, w) and getattr(v, 'dtype', None) == getattr(w, 'dtype', None), k
        for k, v in a._datasets.items():
            assert np.array_equal(v, b._datasets[k]) and v.dtype == b._datasets[k].dtype
    tree(tmp_path / 'fast.hdf').close()
    fast_by_fast = StoreFile(tmp_path / 'fast.hdf', 'r')
    monkeypatch.setattr(hdf5, '_helper', None)               # the plain path from here on
    assert hdf5._fast() is None
    tree(tmp_path / 'plain.hdf').close()
    fast_by_plain = StoreFile(tmp_path / 'fast.hdf', 'r')
    plain_by_plain = StoreFile(tmp_path / 'plain.hdf', 'r')
    monkeypatch.undo()
    assert hdf5._fast() is not None
    plain_by_fast = StoreFile(tmp_path / 'plain.hdf', 'r')
    for other in (fast_by_plain, plain_by_plain, plain_by_fast):
        same(fast_by_fast['/pix/1/2'], other['/pix/1/2'])
    assert _dump('-H', tmp_path / 'fast.hdf').replace('fast.hdf', 'X') == _dump('-H', tmp_path / 'plain.hdf').replace('plain.hdf', 'X')


def _two_chunk_store(tmp_path, file_format):
    """A store with two chunk files, one pixel group each, linked."""
    store = HdfStore(str(tmp_path / 'broken'), nchunks=2, file_format=file_format)
    for k, path in enumerate(store.chunk_paths):
        with StoreFile(path, 'a') as chunk:
            g = chunk.require_group(f'/pix/{k}/0')
            g.attrs['nbest'] = k + 1
    store.link_files()
    store.close()
    return store.store_dir, [str(p) for p in store.chunk_paths]


@pytest.mark.parametrize('file_format', ['hdf5', 'npz'])
def test_missing_chunks_are_an_error_not_a_hole(tmp_path, file_format):
    """The reference opens every chunk file (`h5py.File(chunk_path, 'r')`, main.py:315) and raises on a dangling
    link (main.py:296-298): a stripe that crashed or has not finished must not turn into a map with silent holes."""
    if file_format == 'hdf5' and not hdf5.available():
        pytest.skip('libhdf5 not found')
    store_dir, chunks = _two_chunk_store(tmp_path, file_format)
    table = store_dir / ('table.hdf' if file_format == 'hdf5' else 'table.npz')
    before = table.stat().st_mtime_ns
    with HdfStore(str(store_dir)) as store:                    # opening a linked store to read it ...
        assert sorted(g.attrs['nbest'] for g in store.iter_pix_groups()) == [1, 2]
        assert store.hdf.attrs['linked'] is True or store.hdf.attrs['linked'] == 1
    os.unlink(chunks[1])
    if file_format == 'hdf5':
        with HdfStore(str(store_dir)) as store:                # the link is still in the table, its target is gone
            with pytest.raises(ValueError, match='Broken external HDF link'):
                list(store.iter_pix_groups())
            with pytest.raises(KeyError, match='broken external link'):
                store.hdf['/pix/1/0']
            assert store.hdf['/pix/0/0'].attrs['nbest'] == 1
        with HdfStore(str(store_dir)) as store:                # ... and closing it did not drop the dangling link
            with pytest.raises(ValueError, match='Broken external HDF link'):
                list(store.iter_pix_groups())
    else:
        with pytest.raises(FileNotFoundError, match='chunk1'):
            HdfStore(str(store_dir))
    # linking with a chunk missing links nothing and says which file
    fresh = HdfStore(str(tmp_path / 'fresh'), nchunks=2, file_format=file_format)
    with StoreFile(fresh.chunk_paths[0], 'a') as chunk:
        chunk.require_group('/pix/0/0')
    with pytest.raises(FileNotFoundError, match='chunk1'):
        fresh.link_files()
    assert not fresh.hdf.attrs.get('linked', False) and '/pix' not in fresh.hdf
    fresh.close()
    assert before > 0
