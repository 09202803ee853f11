"""Per-pixel result store in the reference's layout (reference:
nestfit/main.py:233-377 ``HdfStore``, docs/store_spec.rst:45-110, writer ``mn_dump``
nestfit/core/core.pyx:627-687):

    <name>.store/table.*            attrs: nchunks, model metadata, fitter parameters, naxis1/2,
                                    groups simple_header, full_header, products, links to /pix
    <name>.store/chunk<i>.*         /pix/<i_lon>/<i_lat>          attrs i_lon, i_lat, nbest
                                    /pix/<i_lon>/<i_lat>/<ncomp>  attrs + datasets of one run
                                        (posteriors, marginals, bestfit_params, map_params)

Every file is a tree of groups (attrs + datasets, the slice of the h5py API the reference uses)
held in memory and saved whole.  Two file formats, chosen by the suffix: real HDF5 (``.hdf``) written
and read through the HDF5 C library (`nestfit_amd/hdf5.py`: h5py is not in this image, libhdf5 is) --
the table's pixel groups are then external links into the chunk files, like the reference's -- and a
loss-free ``.npz`` twin (arrays under their full path, attributes as one JSON document) for hosts
without libhdf5.  `store_format()` says which one a new store gets.
"""
import inspect
import json
import os
import warnings
from pathlib import Path

import numpy as np

from . import hdf5

HDF5_SUFFIXES = ('.hdf', '.h5', '.hdf5')


def store_format():
    """'hdf5' or 'npz' for a new store: `NFA_STORE_FORMAT` when set, else HDF5 wherever libhdf5 loads."""
    want = os.environ.get('NFA_STORE_FORMAT', '').lower()
    if want in ('hdf5', 'npz'):
        return want
    return 'hdf5' if hdf5.available() else 'npz'


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
                if isinstance(node, BrokenLink):       # what h5py answers for a dangling external link
                    raise KeyError(f'{path}: broken external link ({node.file_name}:{node.obj_path})')
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
            if isinstance(c, Group) and c._root is self._root:      # links into other files are not copied
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


class BrokenLink:
    """An external link whose target file or object could not be opened."""

    def __init__(self, file_name, obj_path):
        self.file_name, self.obj_path = file_name, obj_path

    def __repr__(self):
        return f'<broken external link {self.file_name}:{self.obj_path}>'


class StoreFile(Group):
    """One file of the store (the table or a chunk): a root group that saves itself, as HDF5 when the
    suffix is one of `HDF5_SUFFIXES`, else as .npz."""

    def __init__(self, path, mode='a'):
        super().__init__('/')
        self.path = Path(path)
        self._open = True
        self.mode = mode
        self.is_hdf5 = self.path.suffix in HDF5_SUFFIXES
        self._linked_files = {}                         # chunk files behind this file's external links
        self._link_names = {}                           # file name to store in a link, per linked file
        if self.path.exists() and mode in ('a', 'r') and self.is_hdf5:
            hdf5.read_tree(self.path, self, self._resolve_external)
        elif self.path.exists() and mode in ('a', 'r'):
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

    def _resolve_external(self, parent, name, file_name, obj_path):
        """An external link read from the file: the object of the linked file takes the link's place.  A link
        whose file or object is gone stays in the tree as a `BrokenLink` -- it is written back as the link it was,
        and whoever walks the pixel groups is told (`HdfStore.iter_pix_groups` raises like the reference's,
        main.py:296-298); it is never dropped silently."""
        target = Path(file_name)
        if not target.is_absolute():
            target = self.path.parent / target
        try:
            if target not in self._linked_files:
                self._linked_files[target] = StoreFile(target, 'r')
            parent._children[name] = self._linked_files[target][obj_path]
        except (FileNotFoundError, KeyError, hdf5.Hdf5Error):
            parent._children[name] = BrokenLink(file_name, obj_path)

    def _external_of(self, child):
        """(file name, object path) of a child that lives in another file, None for this file's own."""
        if isinstance(child, BrokenLink):
            return child.file_name, child.obj_path
        if child._root is self._root:
            return None
        other = child._root.path
        known = self._link_names.get(other)             # (one answer per linked file, not per pixel)
        if known is None:
            same_dir = other.parent.resolve() == self.path.parent.resolve()
            known = self._link_names[other] = other.name if same_dir else str(other.resolve())
        return known, child.name

    def _flush(self):
        if self.mode == 'r':
            return
        if self.is_hdf5:
            tmp = self.path.with_name(self.path.name + '.tmp')
            hdf5.write_tree(tmp, self, self._external_of)
            tmp.replace(self.path)
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


# ---------------------------------------------------------------------------------------------
#  The store proper, written against the store specification (docs/store_spec.rst:12-110).
#  What goes where is data, kept in the tables below; the class only walks them.
# ---------------------------------------------------------------------------------------------
STORE_SUFFIX = '.store'
TABLE_NAME = 'table'
CHUNK_STEM = 'chunk'
FILE_SUFFIXES = {'hdf5': '.hdf', 'npz': '.npz'}       # '.hdf' is the reference's (main.py:236)
PRODUCTS_GROUP = '/products'

# root attribute of the table file  <-  attribute of the CubeFitter          (store_spec.rst:60-63)
FITTER_ATTRS = (
    ('lnZ_threshold', lambda f: f.lnZ_thresh),
    ('n_max_components', lambda f: f.ncomp_max),
    ('multinest_kwargs', lambda f: str(f.mn_kwargs)),
)
# root attribute of the table file  <-  module-level name of the model module (store_spec.rst:67-72)
MODEL_ATTRS = (
    ('n_params', 'N'),
    ('model_name', 'NAME'),
    ('par_names', 'PAR_NAMES'),
    ('par_names_short', 'PAR_NAMES_SHORT'),
    ('tex_labels', 'TEX_LABELS'),
    ('tex_labels_with_units', 'TEX_LABELS_WITH_UNITS'),
)
# header groups of the table file  <-  property of the CubeStack              (store_spec.rst:109-110)
HEADER_GROUPS = (('simple_header', 'simple_header'), ('full_header', 'full_header'))


def check_ext(store_name, ext='hdf'):
    """`store_name` with the extension `ext` (added unless it is there already)."""
    name, suffix = str(store_name), '.' + ext
    return name if name.endswith(suffix) else name + suffix


class HdfStore:
    """A `<name>.store` directory: the table file plus one chunk file per stripe, with the reference's
    entry points (nestfit/main.py:233-377).  `hdf` is the open table (a `StoreFile`)."""
    dpath = PRODUCTS_GROUP
    chunk_prefix = CHUNK_STEM

    def __init__(self, store_name, nchunks=1, file_format=None):
        from . import MODELS
        self.store_name = str(store_name)
        self.store_dir = Path(check_ext(self.store_name, ext=STORE_SUFFIX.lstrip('.')))
        self.store_dir.mkdir(parents=True, exist_ok=True)
        # an existing store keeps the format it was written in; a new one takes `file_format` ('hdf5' / 'npz')
        # or what `store_format()` finds
        found = [f for f, sfx in FILE_SUFFIXES.items() if (self.store_dir / (TABLE_NAME + sfx)).exists()]
        self.file_format = found[0] if found else (file_format or store_format())
        self.file_suffix = FILE_SUFFIXES[self.file_format]
        self.linked_table = Path(TABLE_NAME + self.file_suffix)
        self.hdf = StoreFile(self.store_dir / self.linked_table, 'a')
        root = self.hdf.attrs
        root.setdefault('nchunks', nchunks)          # an existing store keeps its own number of chunks
        self.nchunks = root['nchunks']
        self.model = MODELS.get(root.get('model_name'))
        # An HDF5 table keeps its links in the file (external links, resolved while reading it); the .npz twin has no
        # link objects, so there the links of a linked store are rebuilt on opening -- in memory only: opening a
        # store to read it never rewrites it.
        if root.get('linked', False) and self.file_format != 'hdf5':
            self.link_files(flush=False)

    # context manager: `with HdfStore(name) as store`
    def __enter__(self):
        return self

    def __exit__(self, *exc_info):
        self.close()

    # ---- files ---------------------------------------------------------------------------------
    @property
    def is_open(self):
        return self.hdf._open

    @property
    def chunk_paths(self):
        return [self.store_dir / f'{self.chunk_prefix}{k}{self.file_suffix}' for k in range(self.nchunks)]

    def close(self):
        if not self.is_open:
            print(f'{self.store_dir}: already closed')
            return
        self.hdf.close()                             # saves, then marks the file closed

    # ---- pixel groups -------------------------------------------------------------------------
    def iter_pix_groups(self):
        """Every /pix/<i_lon>/<i_lat> group reachable from the table (i.e. after `link_files`)."""
        assert self.is_open
        if '/pix' not in self.hdf:
            return
        lon_level = self.hdf['/pix']
        for lon_name in lon_level:
            for lat_name in lon_level[lon_name]:
                node = lon_level[lon_name]._children.get(lat_name)
                if isinstance(node, BrokenLink):     # the chunk file (or the group in it) is gone: main.py:296-298
                    raise ValueError(f'Broken external HDF link: /pix/{lon_name}/{lat_name} -> {node.file_name}:{node.obj_path}')
                if isinstance(node, Group):
                    yield node

    def find_first_valid_group(self):
        """The one-component run of the first pixel that has one."""
        for group in self.iter_pix_groups():
            if '1' in group:
                return group['1']
        raise ValueError('No valid pix groups found.')

    def link_files(self, loaded=None, flush=True):
        """Hang the pixel groups of every chunk file under the table's /pix: object references in memory,
        external links (`chunk<i>.hdf:/pix/<i_lon>/<i_lat>`, main.py:286-296) in the table's HDF5 file.
        Every one of the `nchunks` chunk files has to be there, like in the reference (its `h5py.File(chunk_path,
        'r')` raises on a missing one, main.py:315): a stripe whose process has crashed or not finished must not
        turn into a map with silent holes.  A chunk without pixels (an empty stripe) is fine."""
        assert self.is_open
        missing = [str(p) for p in self.chunk_paths if p not in (loaded or {}) and not p.exists()]
        if missing:
            raise FileNotFoundError(f'chunk file(s) of the store are missing, nothing linked: {", ".join(missing)}')
        for path in self.chunk_paths:
            chunk = (loaded or {}).get(path) or StoreFile(path, 'r')     # `loaded`: {path: StoreFile} already in memory
            if '/pix' not in chunk:
                continue
            for lon_name in chunk['/pix']:
                for lat_name in chunk['/pix'][lon_name]:
                    target = f'/pix/{lon_name}/{lat_name}'
                    self.hdf[target] = chunk[target]
        self.hdf.attrs['linked'] = True              # only now: all nchunks files were linked
        if flush:
            self.hdf.flush()

    def reset_pix_links(self):
        assert self.is_open
        if '/pix' in self.hdf:
            del self.hdf['/pix']

    # ---- metadata -----------------------------------------------------------------------------
    def insert_header(self, stack):
        """The cube's celestial and full headers as attribute groups, the map size as naxis1 / naxis2."""
        if not self.is_open:
            warnings.warn('Could not insert header: the HDF5 file is closed.', category=RuntimeWarning)
            return
        for group_name, prop in HEADER_GROUPS:
            self.hdf.create_group(group_name).attrs.update(getattr(stack, prop))
        n_lon, n_lat = stack.shape[:2]
        self.hdf.attrs.update(naxis1=n_lon, naxis2=n_lat)

    def read_header(self, full=True):
        assert self.is_open
        return dict(self.hdf[HEADER_GROUPS[1 if full else 0][0]].attrs)

    def insert_fitter_pars(self, fitter):
        assert self.is_open
        self.hdf.attrs.update({name: get(fitter) for name, get in FITTER_ATTRS})
        # the built-in sampler's named setting (sampler.PRECISION: margins of its bound's free rejections), where the results are
        self.hdf.attrs['sampler_precision'] = str(getattr(fitter, 'mn_kwargs', {}).get('precision') or 'default')
        quantum = getattr(fitter, 'nlive_quantum', 1)
        if quantum != 1:                 # a deviation from main.py:445-447 is written down where the results are
            self.hdf.attrs['nlive_quantum'] = int(quantum)

    def insert_model_metadata(self, runner_cls):
        assert self.is_open
        module = inspect.getmodule(runner_cls)
        self.hdf.attrs.update({name: getattr(module, attr) for name, attr in MODEL_ATTRS})

    # ---- products -----------------------------------------------------------------------------
    def create_dataset(self, dset_name, data, group='', clobber=True):
        """Dataset `group`/`dset_name`; with `clobber` an existing one is replaced (and a warning issued)."""
        if not dset_name:
            raise ValueError('a dataset needs a name')
        parent = self.hdf.require_group(group)
        path = '/'.join((group.rstrip('/'), dset_name))
        if clobber and path in self.hdf:
            warnings.warn(f'Deleting dataset "{path}"', RuntimeWarning)
            del self.hdf[path]
        return parent.create_dataset(dset_name, data=data)
