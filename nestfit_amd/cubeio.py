"""Cube ingestion without astropy / spectral_cube: a minimal FITS reader and the
reference's ``NoiseMap`` / ``NoiseMapUniform`` / ``DataCube`` / ``CubeStack`` (reference:
nestfit/main.py:39-223) on top of it, plus the step the GPU adds: all NaN-free pixels of a stack
as ONE device-resident spectra set (``CubeStack.to_device``) instead of a host-to-device copy per
pixel (nestfit/main.py:437-441, 456).

Supported FITS subset (enough for the reference's own test cubes, nestfit/test/data/*.fits, and
for what CASA / spectral_cube write): primary HDU image, BITPIX 8/16/32/64/-32/-64 with
BSCALE/BZERO, 3 axes (or 4 with a degenerate Stokes axis), brightness unit K or Jy/beam (with
BMAJ/BMIN), spectral axis
VRAD / VELO (radio convention) in m/s or km/s, or FREQ in Hz..GHz, with RESTFRQ / RESTFREQ.
"""
from collections.abc import Iterable

import numpy as np

CKMS = 299792.458
_BLOCK = 2880
_DTYPES = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}


def _parse_value(txt):
    txt = txt.strip()
    if not txt:
        return None
    if txt.startswith("'"):
        end = 1
        out = []
        while end < len(txt):                       # '' inside a string is an escaped quote
            if txt[end] == "'":
                if end + 1 < len(txt) and txt[end + 1] == "'":
                    out.append("'")
                    end += 2
                    continue
                break
            out.append(txt[end])
            end += 1
        return ''.join(out).rstrip()
    val = txt.split('/', 1)[0].strip()
    if val in ('T', 'F'):
        return val == 'T'
    try:
        return int(val)
    except ValueError:
        pass
    try:
        return float(val.replace('D', 'E'))
    except ValueError:
        return val


def read_fits(path):
    """(header dict in card order, data array in FITS axis order [NAXISn, ..., NAXIS1]) of the
    primary HDU.  COMMENT / HISTORY / CONTINUE cards are skipped."""
    with open(path, 'rb') as f:
        raw = f.read()
    header = {}
    pos = 0
    done = False
    while not done:
        block = raw[pos:pos + _BLOCK]
        if len(block) < _BLOCK:
            raise ValueError(f'{path}: truncated FITS header')
        pos += _BLOCK
        for i in range(0, _BLOCK, 80):
            card = block[i:i + 80].decode('ascii', 'replace')
            key = card[:8].strip()
            if key == 'END':
                done = True
                break
            if not key or key in ('COMMENT', 'HISTORY', 'CONTINUE') or card[8:10] != '= ':
                continue
            header[key] = _parse_value(card[10:])
    if not header.get('SIMPLE', False):
        raise ValueError(f'{path}: not a standard FITS file')
    naxis = int(header.get('NAXIS', 0))
    shape = [int(header[f'NAXIS{k}']) for k in range(naxis, 0, -1)]
    bitpix = int(header['BITPIX'])
    if bitpix not in _DTYPES:
        raise ValueError(f'{path}: unsupported BITPIX {bitpix}')
    count = int(np.prod(shape)) if shape else 0
    data = np.frombuffer(raw, dtype=_DTYPES[bitpix], count=count, offset=pos).reshape(shape)
    data = data.astype(np.float64)
    bscale, bzero = header.get('BSCALE', 1.0), header.get('BZERO', 0.0)
    if bscale != 1.0 or bzero != 0.0:
        data = data * float(bscale) + float(bzero)
    return header, data


def _format_card(key, value):
    """One 80-character header card `KEYWORD = value` (fixed format: logicals and numbers right-justified
    to column 30, strings quoted from column 11)."""
    if isinstance(value, (bool, np.bool_)):
        txt = f'{"T" if value else "F":>20}'
    elif isinstance(value, (int, np.integer)):
        txt = f'{int(value):>20d}'
    elif isinstance(value, (float, np.floating)):
        txt = f'{float(value):>20.13E}' if np.isfinite(value) else f"{'':>20}"
    else:
        txt = "'" + f'{str(value).replace(chr(39), chr(39) * 2):<8}' + "'"
    return f'{key[:8].upper():<8}= {txt}'[:80].ljust(80)


def write_fits(path, header, data):
    """A primary HDU: `data` (numpy array in FITS axis order [NAXISn, ..., NAXIS1], written as float32 or
    float64 big-endian) under `header` (a dict; SIMPLE / BITPIX / NAXIS* are set from the array, everything
    else is written as given).  The counterpart of `read_fits` for the map products of a fit."""
    data = np.asarray(data)
    bitpix = -32 if data.dtype == np.float32 else -64
    cards = [_format_card('SIMPLE', True), _format_card('BITPIX', bitpix), _format_card('NAXIS', data.ndim)]
    cards += [_format_card(f'NAXIS{k + 1}', n) for k, n in enumerate(data.shape[::-1])]
    skip = {'SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'END'} | {f'NAXIS{k}' for k in range(1, 10)}
    cards += [_format_card(k, v) for k, v in header.items() if k.upper() not in skip and v is not None]
    cards.append('END'.ljust(80))
    head = ''.join(cards).encode('ascii', 'replace')
    head += b' ' * (-len(head) % _BLOCK)
    body = np.ascontiguousarray(data, dtype=_DTYPES[bitpix]).tobytes()
    body += b'\0' * (-len(body) % _BLOCK)
    with open(path, 'wb') as f:
        f.write(head + body)


class SimpleCube:
    """The slice of ``spectral_cube.SpectralCube`` that ``DataCube`` needs: header, data in
    (spectral, lat, lon) order, brightness unit, spectral axis in Hz and in km/s (radio)."""

    def __init__(self, header, data):
        data = np.asarray(data, dtype=np.float64)
        while data.ndim > 3 and data.shape[0] == 1:        # degenerate Stokes axis
            data = data[0]
        if data.ndim != 3:
            raise ValueError(f'Cannot parse shape : {data.shape}')
        self.header = dict(header)
        self._data = data
        self.unit = str(header.get('BUNIT', '')).strip()
        ctype = str(header.get('CTYPE3', '')).strip().upper()
        n = data.shape[0]
        crval, cdelt, crpix = float(header['CRVAL3']), float(header['CDELT3']), float(header.get('CRPIX3', 1.0))
        world = crval + (np.arange(n) + 1.0 - crpix) * cdelt
        cunit = str(header.get('CUNIT3', '')).strip().lower()
        self.rest_freq = header.get('RESTFRQ', header.get('RESTFREQ'))
        if ctype.startswith('FREQ'):
            scale = {'hz': 1.0, 'khz': 1e3, 'mhz': 1e6, 'ghz': 1e9, '': 1.0}[cunit]
            self._freq = world * scale
        elif ctype.startswith('VRAD') or ctype.startswith('VELO'):
            if ctype.startswith('VELO') and int(header.get('VELREF', 257)) < 256:
                raise ValueError('optical-convention velocity axes are not supported')
            if self.rest_freq is None:
                raise ValueError('a velocity axis needs RESTFRQ')
            scale = {'m s-1': 1e-3, 'm/s': 1e-3, 'km s-1': 1.0, 'km/s': 1.0, '': 1e-3}[cunit]
            self._freq = float(self.rest_freq) * (1.0 - world * scale / CKMS)     # radio convention
        else:
            raise ValueError(f'unsupported spectral axis type {ctype!r}')

    @classmethod
    def read(cls, path):
        return cls(*read_fits(str(path)))

    @property
    def shape(self):
        return self._data.shape

    def __getitem__(self, key):
        """Spectral slicing only (``cube[:-1]``, ``cube[::-1]``)."""
        if not isinstance(key, slice):
            raise TypeError('only slices along the spectral axis are supported')
        new = object.__new__(SimpleCube)
        new.header, new.unit, new.rest_freq = dict(self.header), self.unit, self.rest_freq
        new._data = self._data[key]
        new._freq = self._freq[key]
        return new

    def spectral_axis_hz(self):
        return self._freq.copy()

    def spectral_axis_kms(self):
        if self.rest_freq is None:
            raise ValueError('a velocity axis needs RESTFRQ')
        return CKMS * (1.0 - self._freq / float(self.rest_freq))


def read_spectrum(path):
    """A one-dimensional FITS spectrum (NAXIS = 1, FREQ axis: what the CASA spectral profiler writes,
    e.g. the reference's nestfit/test/data/test_spectrum_11.fits) as the model wants it:
    (frequency axis in Hz ascending, intensities, header)."""
    header, data = read_fits(str(path))
    data = np.asarray(data, dtype=np.float64).ravel()
    if int(header.get('NAXIS', 0)) != 1 or not str(header.get('CTYPE1', '')).upper().startswith('FREQ'):
        raise ValueError('expected a one-dimensional spectrum on a frequency axis')
    scale = {'hz': 1.0, 'khz': 1e3, 'mhz': 1e6, 'ghz': 1e9, '': 1.0}[str(header.get('CUNIT1', '')).strip().lower()]
    freq = (float(header['CRVAL1']) + (np.arange(data.size) + 1.0 - float(header.get('CRPIX1', 1.0)))
            * float(header['CDELT1'])) * scale
    if freq[1] < freq[0]:
        freq, data = freq[::-1], data[::-1]
    return freq.copy(), data.copy(), header


def jy_per_beam_to_kelvin(freq_hz, header):
    """K per (Jy/beam) at each frequency for the elliptical Gaussian beam BMAJ x BMIN (degrees, FWHM)
    of the header: T = S c^2 / (2 k nu^2 Omega), Omega = pi BMAJ BMIN / (4 ln 2)."""
    try:
        bmaj, bmin = float(header['BMAJ']), float(header['BMIN'])
    except KeyError:
        raise ValueError('a Jy/beam cube needs BMAJ and BMIN in its header') from None
    omega = np.pi * np.radians(bmaj) * np.radians(bmin) / (4.0 * np.log(2.0))       # sr
    c, kb = 299792458.0, 1.380649e-23
    return 1e-26 * c ** 2 / (2.0 * kb * np.asarray(freq_hz, dtype=np.float64) ** 2 * omega)


class _Noise:
    """RMS per map pixel.  `get_noise(i_lon, i_lat)` is the reference's accessor (main.py:39-72);
    `values_at` is its array form (what the device upload and the pixel screening use)."""
    shape = None

    def get_noise(self, i_lon, i_lat):
        return float(self.values_at(np.asarray(i_lon), np.asarray(i_lat)))

    def values_at(self, lon, lat):
        raise NotImplementedError


class NoiseMap(_Noise):
    """A noise image, given in FITS order (lat, lon) and indexed (i_lon, i_lat) like the transposed cube."""

    def __init__(self, data):
        self.data = np.asarray(data).T
        self.shape = self.data.shape

    @classmethod
    def from_pbimg(cls, rms, pb_img):
        """A flat `rms` divided by the primary-beam response `pb_img` (2-D, or 3-D / 4-D whose leading axes
        are channel / Stokes: the first plane is taken).  Where the response is masked or zero the noise is
        infinite."""
        pb = np.asarray(pb_img, dtype=np.float64)
        if not 2 <= pb.ndim <= 4:
            raise ValueError(f'Cannot parse shape : {pb.shape}')
        plane = pb.reshape((-1,) + pb.shape[-2:])[0]
        with np.errstate(divide='ignore', invalid='ignore'):
            noise = rms / plane
        noise[~np.isfinite(noise)] = np.inf
        return cls(noise)

    def get_noise(self, i_lon, i_lat):
        return self.data[i_lon, i_lat]

    def values_at(self, lon, lat):
        return np.asarray(self.data[lon, lat], dtype=np.float64)


class NoiseMapUniform(_Noise):
    """One RMS for every pixel (`shape` stays None: there is no image to compare with the cube's)."""

    def __init__(self, rms):
        self.rms = rms

    def get_noise(self, i_lon, i_lat):
        return self.rms

    def values_at(self, lon, lat):
        return np.full(np.shape(lon), self.rms, dtype=np.float64)


# celestial keywords a two-dimensional map product inherits from the cube
_MAP_KEYWORDS = ('SIMPLE', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'WCSAXES') + tuple(
    f'{stem}{axis}' for stem in ('CRPIX', 'CDELT', 'CUNIT', 'CTYPE', 'CRVAL') for axis in (1, 2)) + ('RADESYS', 'EQUINOX')
_SKY_AXES = ('ra', 'dec', 'lon', 'lat')
_JY_PER_BEAM = ('jy/beam', 'jybeam-1', 'beam-1jy')


class DataCube:
    """One transition's cube in the layout the fitter wants (the reference's DataCube, main.py:77-172):
    `data[i_lon, i_lat, chan]` in K with the frequency axis `xarr` ascending in Hz, `varr` the matching
    (descending) radio velocities in km/s, `dv` the channel width in km/s.  `cube` is a `SimpleCube` (or
    anything with its attributes); `noise_map` a number or a noise-map object."""

    def __init__(self, cube, noise_map, trans_id=None):
        self.trans_id = trans_id
        self.noise_map = noise_map if hasattr(noise_map, 'get_noise') else NoiseMapUniform(noise_map)
        self._header = dict(cube.header)
        kelvin, freq, velo = self._ingest(cube)
        self.data, self.xarr, self.varr = kelvin, freq, velo
        self.dv = self.get_chan_width(cube)
        self.shape = kelvin.shape                          # (lon, lat, chan)
        self.spatial_shape, self.nchan = kelvin.shape[:2], kelvin.shape[2]
        if self.noise_map.shape is not None and tuple(self.noise_map.shape) != tuple(self.spatial_shape):
            raise AssertionError(f'noise map {self.noise_map.shape} does not match the cube {self.spatial_shape}')

    def _ingest(self, cube):
        """(data in K as [lon, lat, chan], ascending Hz axis, matching km/s axis)."""
        data, freq = self.data_from_cube(cube)
        return data, freq, self.velo_axis_from_cube(cube)

    @property
    def full_header(self):
        return self._header

    @property
    def simple_header(self):
        """The celestial (2-D) part of the header, for map products."""
        hdict = {key: self._header[key] for key in _MAP_KEYWORDS if key in self._header}
        hdict.update(NAXIS=2, WCSAXES=2)
        for key in ('CTYPE1', 'CTYPE2'):                   # of the form "RA---SIN"
            if hdict[key].split('-')[0].lower() not in _SKY_AXES:
                raise AssertionError(f'{key} = {hdict[key]!r} is not a celestial axis')
        return hdict

    def get_chan_width(self, cube):
        v = cube.spectral_axis_kms()
        return abs(v[1] - v[0])

    def data_from_cube(self, cube):
        """Brightness in K on an ascending frequency axis, one pixel's spectrum contiguous.  Jy/beam is
        converted like spectral_cube's `to('K')`: Rayleigh-Jeans brightness temperature of the header's
        Gaussian beam, channel by channel; a cube without a unit is taken to be in K."""
        unit = cube.unit.replace(' ', '').lower()
        data, axis = cube._data, cube.spectral_axis_hz()
        if unit in _JY_PER_BEAM:
            data = data * jy_per_beam_to_kelvin(axis, cube.header)[:, None, None]
        elif unit == '':
            print('-- Assuming cube intensity units of K')
        elif unit != 'k':
            raise ValueError(f'cube intensity unit {cube.unit!r}: only K and Jy/beam are supported')
        if axis[0] > axis[-1]:
            data, axis = data[::-1], axis[::-1]
        return np.ascontiguousarray(data.transpose()), np.array(axis)      # (chan, lat, lon) -> (lon, lat, chan)

    def velo_axis_from_cube(self, cube):
        v = cube.spectral_axis_kms()
        return np.array(v if v[0] > v[-1] else v[::-1])    # descending: the element-wise partner of xarr

    def get_spec_data(self, i_lon, i_lat):
        spec = self.data[i_lon, i_lat, :]
        noise = self.noise_map.get_noise(i_lon, i_lat)
        return self.xarr, spec, noise, self.trans_id, bool(np.isnan(spec).any() or np.isnan(noise))


class CubeStack:
    """The cubes of all transitions of one field, same sky grid (reference: nestfit/main.py:175-223)."""

    def __init__(self, cubes):
        assert isinstance(cubes, Iterable)
        self.cubes = cubes
        self.n_cubes = len(cubes)

    def __iter__(self):
        return iter(self.cubes)

    full_header = property(lambda self: self.cubes[0].full_header)
    simple_header = property(lambda self: self.cubes[0].simple_header)
    shape = property(lambda self: self.cubes[0].shape)
    spatial_shape = property(lambda self: self.cubes[0].spatial_shape)

    def get_arrays(self, i_lon, i_lat):
        return [dc.get_spec_data(i_lon, i_lat)[1] for dc in self.cubes]

    def get_spec_data(self, i_lon, i_lat):
        """([[xarr, spectrum, noise, trans_id] per cube], any NaN in the pixel)."""
        rows = [dc.get_spec_data(i_lon, i_lat) for dc in self.cubes]
        return [list(r[:4]) for r in rows], any(r[4] for r in rows)

    def get_max_snr(self, i_lon, i_lat):
        snr = [np.max(spec) / noise for _, spec, noise, _, _ in (dc.get_spec_data(i_lon, i_lat) for dc in self.cubes)]
        return max([0.0] + [v for v in snr if v > 0.0])

    # ---- what the GPU adds ----------------------------------------------------------------
    def good_pixels(self, lon=None, lat=None):
        """(i_lon, i_lat) of the pixels without NaNs in any cube or noise value: the ones the
        reference's fit loop does not skip (main.py:438-441).  `lon`, `lat` restrict the search to
        given index arrays (e.g. one rank's stripe)."""
        if lon is None:
            lon, lat = (a.ravel() for a in np.indices(self.spatial_shape))
        lon, lat = np.asarray(lon), np.asarray(lat)
        bad = np.zeros(lon.shape, dtype=bool)
        for dcube in self.cubes:
            bad |= np.isnan(dcube.data[lon, lat, :]).any(axis=1)
            noise = dcube.noise_map.values_at(lon, lat)
            bad |= ~(noise > 0) | ~np.isfinite(noise)
        return lon[~bad], lat[~bad]

    def masked_beam_pixels(self, lon=None, lat=None):
        """(i_lon, i_lat) of the pixels whose data are NaN-free but whose noise is infinite in some cube
        (masked primary beam, `NoiseMap.from_pbimg`): the reference does not skip them (only NaNs,
        main.py:437-441); their likelihood is flat and their fit ends with nbest = 0."""
        if lon is None:
            lon, lat = (a.ravel() for a in np.indices(self.spatial_shape))
        lon, lat = np.asarray(lon), np.asarray(lat)
        nan = np.zeros(lon.shape, dtype=bool)
        inf = np.zeros(lon.shape, dtype=bool)
        for dcube in self.cubes:
            noise = dcube.noise_map.values_at(lon, lat)
            nan |= np.isnan(dcube.data[lon, lat, :]).any(axis=1) | np.isnan(noise)
            inf |= np.isinf(noise)
        keep = inf & ~nan
        return lon[keep], lat[keep]

    def to_device(self, utrans, ncomp=1, lon=None, lat=None, model=0, **runner_kwargs):
        """All good pixels as one device-resident spectra set: returns (CubeRunner, i_lon, i_lat);
        row k of the runner is pixel (i_lon[k], i_lat[k])."""
        from .cube import CubeRunner
        lon, lat = self.good_pixels(lon, lat)
        data = np.concatenate([dc.data[lon, lat, :] for dc in self.cubes], axis=1)
        noise = np.stack([dc.noise_map.values_at(lon, lat) for dc in self.cubes], axis=1)
        runner = CubeRunner([dc.xarr for dc in self.cubes], [dc.trans_id for dc in self.cubes], data, noise,
                            utrans, ncomp=ncomp, model=model, **runner_kwargs)
        return runner, lon, lat
