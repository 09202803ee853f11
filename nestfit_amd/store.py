"""Per-pixel result store in the reference's layout (reference:
nestfit/main.py:233-377 ``HdfStore``, docs/store_spec.rst:45-110, writer ``mn_dump``
nestfit/core/core.pyx:627-687):

    <name>.store/table.*            attrs: nchunks, model metadata, fitter parameters, naxis1/2,
                                    groups simple_header, full_header, products, links to /pix
    <name>.store/chunk<i>.*         /pix/<i_lon>/<i_lat>          attrs i_lon, i_lat, nbest
                                    /pix/<i_lon>/<i_lat>/<ncomp>  attrs + datasets of one run
                                        (posteriors, marginals, bestfit_params, map_params)

h5py is not available in this image, so the container is a loss-free twin: a tree of groups
(attrs + datasets, the slice of the h5py API the reference uses) saved as one ``.npz`` per file
(arrays under their full path, attributes as one JSON document).  With h5py importable the same
tree can be exported with `Group.to_hdf5`.
"""
import inspect
import json
import warnings
from pathlib import Path

import numpy as np


class Group:
    """h5py.Group look-alike: attrs, create_group / require_group, create_dataset, item access by
    relative or absolute path, iteration over child names."""

    class _File:
        def __init__(self, root):
            self._root = root

        def flush(self):
            self._root._flush()

    def __init__(self, name='/', parent=None, root=None):
        self.name = name
        self.attrs = {}
        self._children = {}
        self._datasets = {}
        self._parent = parent
        self._root = self if root is None else root
        self.file = Group._File(self._root)

    # ---- tree navigation ------------------------------------------------------------------
    def _walk(self, path, create):
        node = self._root if path.startswith('/') else self
        parts = [p for p in path.split('/') if p]
        for k, part in enumerate(parts):
            if part in node._children:
                node = node._children[part]
            elif k == len(parts) - 1 and part in node._datasets:
                return node._datasets[part]
            elif create:
                child = Group((node.name.rstrip('/') + '/' + part), node, self._root)
                node._children[part] = child
                node = child
            else:
                raise KeyError(path)
        return node

    def create_group(self, name):
        try:
            self._walk(name, create=False)
        except KeyError:
            return self._walk(name, create=True)
        raise ValueError(f'Unable to create group (name already exists): {name}')

    def require_group(self, name):
        return self._walk(name, create=True) if name else self

    def create_dataset(self, name, data=None):
        parts = [p for p in name.split('/') if p]
        node = self.require_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        if parts[-1] in node._datasets or parts[-1] in node._children:
            raise ValueError(f'Unable to create dataset (name already exists): {name}')
        node._datasets[parts[-1]] = np.array(data)
        return node._datasets[parts[-1]]

    def __getitem__(self, path):
        return self._walk(path, create=False)

    def __setitem__(self, path, value):
        """Assigning a Group links it (the store's stand-in for h5py.ExternalLink)."""
        parts = [p for p in path.split('/') if p]
        node = (self._root if path.startswith('/') else self).require_group('/'.join(parts[:-1]))
        if isinstance(value, Group):
            node._children[parts[-1]] = value
        else:
            node._datasets[parts[-1]] = np.array(value)

    def __delitem__(self, path):
        parts = [p for p in path.split('/') if p]
        node = (self._root if path.startswith('/') else self)._walk('/'.join(parts[:-1]), create=False) \
            if len(parts) > 1 else (self._root if path.startswith('/') else self)
        if parts[-1] in node._children:
            del node._children[parts[-1]]
        elif parts[-1] in node._datasets:
            del node._datasets[parts[-1]]
        else:
            raise KeyError(path)

    def __contains__(self, path):
        try:
            self._walk(path, create=False)
            return True
        except KeyError:
            return False

    def __iter__(self):
        yield from list(self._children) + list(self._datasets)

    def keys(self):
        return list(self)

    # ---- persistence ----------------------------------------------------------------------
    def _flush(self):
        pass

    def _collect(self, arrays, attrs):
        if self.attrs:
            attrs[self.name] = {k: _jsonable(v) for k, v in self.attrs.items()}
        else:
            attrs.setdefault(self.name, {})
        for k, d in self._datasets.items():
            arrays[self.name.rstrip('/') + '/' + k] = d
        for c in self._children.values():
            if c._root is self._root:                  # links into other files are not copied
                c._collect(arrays, attrs)

    def to_hdf5(self, h5group):                         # pragma: no cover (needs h5py)
        for k, v in self.attrs.items():
            h5group.attrs[k] = v
        for k, d in self._datasets.items():
            h5group.create_dataset(k, data=d)
        for k, c in self._children.items():
            c.to_hdf5(h5group.create_group(k))


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return {'__ndarray__': v.tolist(), 'dtype': str(v.dtype)}
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    return v


def _unjson(v):
    if isinstance(v, dict) and '__ndarray__' in v:
        return np.array(v['__ndarray__'], dtype=v['dtype'])
    return v


class StoreFile(Group):
    """One file of the store (the table or a chunk): a root group that saves itself as .npz."""

    def __init__(self, path, mode='a'):
        super().__init__('/')
        self.path = Path(path)
        self._open = True
        self.mode = mode
        if self.path.exists() and mode in ('a', 'r'):
            with np.load(self.path, allow_pickle=False) as z:
                attrs = json.loads(str(z['__attrs__']))
                for name in attrs:
                    node = self.require_group(name) if name != '/' else self
                    node.attrs.update({k: _unjson(v) for k, v in attrs[name].items()})
                for key in z.files:
                    if key != '__attrs__':
                        self.create_dataset(key, data=z[key])
        elif mode == 'r':
            raise FileNotFoundError(str(self.path))

    def _flush(self):
        if self.mode == 'r':
            return
        arrays, attrs = {}, {}
        self._collect(arrays, attrs)
        tmp = self.path.with_suffix('.tmp.npz')
        np.savez(tmp, __attrs__=np.array(json.dumps(attrs)), **arrays)
        tmp.replace(self.path)

    def flush(self):
        if not self._open:
            raise ValueError('Not a file (not a file)')
        self._flush()

    def close(self):
        if not self._open:
            raise ValueError('Not a file (not a file)')
        self._flush()
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._open:
            self.close()


def check_ext(store_name, ext='hdf'):
    if store_name.endswith(f'.{ext}'):
        return store_name
    return f'{store_name}.{ext}'


class HdfStore:
    """Same names and behaviour as the reference's ``HdfStore`` (main.py:233-377)."""
    linked_table = Path('table.npz')
    chunk_prefix = 'chunk'
    dpath = '/products'

    def __init__(self, store_name, nchunks=1):
        from . import MODELS
        self.store_name = str(store_name)
        self.store_dir = Path(check_ext(self.store_name, ext='store'))
        self.store_dir.mkdir(parents=True, exist_ok=True)
        self.hdf = StoreFile(self.store_dir / self.linked_table, 'a')
        try:
            self.nchunks = self.hdf.attrs['nchunks']
        except KeyError:
            self.hdf.attrs['nchunks'] = nchunks
            self.nchunks = nchunks
        try:
            self.model = MODELS[self.hdf.attrs['model_name']]
        except KeyError:
            self.model = None
        if self.hdf.attrs.get('linked', False):      # links are not stored: rebuild them
            self.link_files()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    @property
    def chunk_paths(self):
        return [self.store_dir / Path(f'{self.chunk_prefix}{i}.npz') for i in range(self.nchunks)]

    @property
    def is_open(self):
        return self.hdf._open

    def close(self):
        try:
            self.hdf.flush()
            self.hdf.close()
        except ValueError:
            print('Store HDF already closed.')

    def iter_pix_groups(self):
        """Every /pix/<i_lon>/<i_lat> group of the (linked) table."""
        assert self.is_open
        pix = self.hdf['/pix']
        for lon_name in pix:
            for lat_name in pix[lon_name]:
                node = pix[lon_name][lat_name]
                if isinstance(node, Group):
                    yield node

    def find_first_valid_group(self):
        """The one-component run of the first pixel that has one (main.py:297-304)."""
        assert self.is_open
        for group in self.iter_pix_groups():
            if '1' in group:
                return group['1']
        raise ValueError('No valid pix groups found.')

    def link_files(self):
        """Make every pixel group of every chunk file reachable from the table (the reference
        inserts h5py.ExternalLink objects, main.py:306-316)."""
        assert self.is_open
        for chunk_path in self.chunk_paths:
            chunk = StoreFile(chunk_path, 'r')
            if '/pix' not in chunk:
                continue
            for lon_name in chunk['/pix']:
                for lat_name in chunk[f'/pix/{lon_name}']:
                    name = f'/pix/{lon_name}/{lat_name}'
                    self.hdf[name] = chunk[name]
        self.hdf.attrs['linked'] = True
        self.hdf.flush()

    def reset_pix_links(self):
        assert self.is_open
        if '/pix' in self.hdf:
            del self.hdf['/pix']

    def insert_header(self, stack):
        """Cube headers as attributes of the groups simple_header / full_header, map size as
        naxis1 / naxis2 (main.py:323-339)."""
        if not self.is_open:
            warnings.warn('Could not insert header: the HDF5 file is closed.', category=RuntimeWarning)
            return
        for name, header in (('simple_header', stack.simple_header), ('full_header', stack.full_header)):
            self.hdf.create_group(name).attrs.update(header)
        self.hdf.attrs['naxis1'], self.hdf.attrs['naxis2'] = stack.shape[0], stack.shape[1]

    def read_header(self, full=True):
        assert self.is_open
        return dict(self.hdf['full_header' if full else 'simple_header'].attrs)

    def create_dataset(self, dset_name, data, group='', clobber=True):
        """Dataset `group`/`dset_name`; an existing one is replaced (with a warning) when clobber."""
        assert len(dset_name) > 0
        parent = self.hdf.require_group(group)
        path = f'{group.rstrip("/")}/{dset_name}'
        if clobber and path in self.hdf:
            warnings.warn(f'Deleting dataset "{path}"', RuntimeWarning)
            del self.hdf[path]
        return parent.create_dataset(dset_name, data=data)

    def insert_fitter_pars(self, fitter):
        assert self.is_open
        self.hdf.attrs['lnZ_threshold'] = fitter.lnZ_thresh
        self.hdf.attrs['n_max_components'] = fitter.ncomp_max
        self.hdf.attrs['multinest_kwargs'] = str(fitter.mn_kwargs)

    def insert_model_metadata(self, runner_cls):
        module = inspect.getmodule(runner_cls)
        assert self.is_open
        self.hdf.attrs['n_params'] = module.N
        self.hdf.attrs['model_name'] = module.NAME
        self.hdf.attrs['par_names'] = module.PAR_NAMES
        self.hdf.attrs['par_names_short'] = module.PAR_NAMES_SHORT
        self.hdf.attrs['tex_labels'] = module.TEX_LABELS
        self.hdf.attrs['tex_labels_with_units'] = module.TEX_LABELS_WITH_UNITS
