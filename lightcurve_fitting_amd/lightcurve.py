"""Minimal light-curve container and the magnitude -> luminosity preparation that precedes the fit.

Host-side, one-off work: the step *before* the hot path.  ``LC`` is a plain column
store (dict of NumPy arrays + ``meta``) exposing the part of the reference's ``LC(astropy.table.Table)`` API that
``lightcurve_mcmc`` and the models touch (``lightcurve.py:62-360, 677-681, 912-941`` in the reference):
``LC.read``, ``lc[col]``, ``lc.colnames``, ``lc.meta``, ``where``, ``filters_to_objects``, ``zp``, ``calcAbsMag``,
``calcLum``, ``calcFlux`` and the free functions ``mag2flux`` / ``flux2mag``.

Not reproduced: plotting, binning and the Planck18 distance modulus (pass ``dm``).  E(B-V)-based extinction uses
``extinction.py`` (the Fitzpatrick 1999 law restated; third-party arithmetic for the reference).
"""
import numpy as np

from .filters import Filter, filtdict

#: standard column names and the aliases recognised for them (subset of the reference's table)
column_names = {
    'MJD': ['MJD', 'mjd', 'Mjd', 'time', 'Time', 'epoch'],
    'mag': ['mag', 'Mag', 'magnitude', 'Magnitude'],
    'dmag': ['dmag', 'Dmag', 'magerr', 'MagErr', 'mag_err', 'e_mag', 'Error', 'error', 'err'],
    'filter': ['filter', 'Filter', 'filt', 'Filt', 'band', 'Band'],
    'nondet': ['nondet', 'Nondet', 'upperlimit', 'UL', 'l_mag'],
    'source': ['source', 'Source'],
    'telescope': ['telescope', 'Telescope', 'Tel', 'tel', 'tel+inst'],
}


def mag2flux(mag, dmag=np.nan, zp=0., nondet=None, nondetSigmas=3.):
    """Magnitude (+ uncertainty) to flux (+ uncertainty); nondetections imply zero flux (lightcurve.py:912-941)."""
    flux = 10 ** ((zp - np.asarray(mag, dtype=float)) / 2.5)
    dflux = np.log(10) / 2.5 * flux * dmag
    if nondet is not None:
        nondet = np.asarray(nondet)
        if nondet.dtype == bool and nondet.shape == () and not nondet:
            return flux, dflux
        dflux = np.array(dflux, dtype=float)
        flux = np.array(flux, dtype=float)
        dflux[nondet] = flux[nondet] / nondetSigmas
        flux[nondet] = 0
    return flux, dflux


def flux2mag(flux, dflux=np.array(np.nan), zp=0., nondet=None, nondetSigmas=3.):
    """Flux (+ uncertainty) to magnitude (+ uncertainty) (lightcurve.py:878-909)."""
    flux = np.array(flux, dtype=float)
    dflux = np.array(dflux, dtype=float)
    if nondet is not None:
        flux[nondet] = nondetSigmas * dflux[nondet]
        dflux[nondet] = np.nan
    with np.errstate(divide='ignore', invalid='ignore'):
        mag = -2.5 * np.log10(flux, out=np.full_like(flux, -np.inf), where=flux > 0.) + zp
        dmag = 2.5 * dflux / (flux * np.log(10))
    return mag, dmag


def _convert(tokens):
    """Column of strings -> float, bool or str array."""
    try:
        return np.array([float(t) for t in tokens])
    except ValueError:
        pass
    if set(tokens) <= {'True', 'False'}:
        return np.array([t == 'True' for t in tokens])
    return np.array(tokens, dtype=object)


class LC:
    """A broadband light curve: named columns of equal length plus a ``meta`` dictionary."""

    def __init__(self, data=None, meta=None):
        self.columns = {}
        self.meta = dict(meta or {})
        self.nondetSigmas = 3.
        if isinstance(data, LC):
            self.meta = dict(data.meta)
            self.nondetSigmas = data.nondetSigmas
            data = data.columns
        for k, v in (data or {}).items():
            self[k] = v
        self.normalize_column_names()
        if 'filter' in self.columns and not all(isinstance(f, Filter) for f in self.columns['filter']):
            self.filters_to_objects()

    # --- reading ----------------------------------------------------------------------------------------------
    @classmethod
    def read(cls, filepath, format='ascii', **kwargs):
        """Read a whitespace-separated ASCII table with a header row (an optional dashed ruler line is skipped, as in
        astropy's ``fixed_width_two_line``; ``--`` and empty cells read as 0 like the reference's ``fill_values``)."""
        with open(filepath) as fh:
            lines = [ln.rstrip('\n') for ln in fh if ln.strip() and not ln.lstrip().startswith('#')]
        header = lines[0].split()
        body = [ln for ln in lines[1:] if set(ln.strip()) - set('- ')]
        rows = [ln.split() for ln in body]
        bad = [r for r in rows if len(r) != len(header)]
        if bad:
            raise ValueError(f'{filepath}: row with {len(bad[0])} cells, header has {len(header)}')
        cols = {name: _convert(['0' if r[i] == '--' else r[i] for r in rows]) for i, name in enumerate(header)}
        return cls(cols)

    # --- container protocol -----------------------------------------------------------------------------------
    @property
    def colnames(self):
        return list(self.columns)

    def keys(self):
        return self.colnames

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def __contains__(self, name):
        return name in self.columns

    def __getitem__(self, item):
        if isinstance(item, str):
            return self.columns[item]
        out = LC.__new__(LC)
        out.meta = dict(self.meta)
        out.nondetSigmas = self.nondetSigmas
        out.columns = {k: v[item] for k, v in self.columns.items()}
        return out

    def __setitem__(self, name, value):
        arr = value if isinstance(value, np.ndarray) else np.array(value, dtype=object if _has_objects(value) else None)
        if self.columns and arr.shape[:1] != (len(self),):
            arr = np.broadcast_to(arr, (len(self),)).copy()
        self.columns[name] = arr

    def copy(self):
        out = self[slice(None)]
        out.columns = {k: v.copy() for k, v in out.columns.items()}
        return out

    # --- normalisation (lightcurve.py:144-179) ----------------------------------------------------------------
    def normalize_column_names(self):
        for good, aliases in column_names.items():
            if good not in self.columns:
                for bad in aliases[1:]:
                    if bad in self.columns:
                        self.columns[good] = self.columns.pop(bad)
                        break
        if 'MJD' not in self.columns and 'JD' in self.columns:
            self.columns['MJD'] = np.asarray(self.columns.pop('JD'), dtype=float) - 2400000.5
        if 'nondet' in self.columns and self.columns['nondet'].dtype != bool:
            nd = self.columns['nondet']
            self.columns['nondet'] = np.array([str(v) in ('True', 'T', '>', '1.0', '1') for v in nd])

    def filters_to_objects(self):
        """Parse the ``'filter'`` column into :class:`Filter` objects; Swift U/B/V become the UVOT filters."""
        names = [str(f) for f in self.columns['filter']]
        filters = np.array([filtdict.get(n, filtdict['?']) for n in names], dtype=object)
        is_swift = np.zeros(len(names), bool)
        if 'telescope' in self.columns:
            is_swift |= np.isin(self.columns['telescope'].astype(str), ['Swift', 'UVOT', 'Swift/UVOT', 'Swift+UVOT'])
        if 'source' in self.columns:
            is_swift |= self.columns['source'].astype(str) == 'SOUSA'
        if is_swift.any():
            for filt, swiftfilt in zip('UBV', 'sbv'):
                filters[is_swift & (np.array(names) == filt)] = filtdict[swiftfilt]
        self.columns['filter'] = filters

    # --- selection (lightcurve.py:87-134) ------------------------------------------------------------------------
    #: keyword suffix -> (row test against ONE value, how the tests of a list of values combine)
    _CRITERIA = {
        '': (lambda column, v: _same(column, v), np.logical_or),
        '_not': (lambda column, v: ~_same(column, v), np.logical_and),
        '_min': (lambda column, v: column >= v, np.logical_or),
        '_max': (lambda column, v: column <= v, np.logical_or),
    }

    def where(self, **kwargs):
        """Rows matching every criterion: ``col=value``, ``col_not=``, ``col_min=``, ``col_max=``.  A list of values
        matches any of them (``_not``: none of them); filters may be given by name."""
        keep = np.ones(len(self), dtype=bool)
        for key, wanted in kwargs.items():
            suffix = next((s for s in ('_not', '_min', '_max') if s in key), '')
            test, combine = self._CRITERIA[suffix]
            column = self[key.replace(suffix, '') if suffix else key]
            values = wanted if isinstance(wanted, list) else [wanted]
            if key.startswith('filter'):
                values = [filtdict[v] if isinstance(v, str) else v for v in values]
            hits = [np.asarray(test(column, v), dtype=bool) for v in values]
            keep &= combine.reduce(hits) if hits else np.full(len(self), suffix == '_not')
        return self[keep]

    # --- photometric conversions -------------------------------------------------------------------------------
    @property
    def zp(self):
        """Zero point of every row's filter (``Filter.m0``)."""
        return np.array([f.m0 for f in self['filter']])

    def calcFlux(self, nondetSigmas=None, zp=None):
        """``'flux'`` / ``'dflux'`` from ``'mag'`` / ``'dmag'`` (lightcurve.py:188-204)."""
        if nondetSigmas is not None:
            self.nondetSigmas = nondetSigmas
        if zp is None:
            zp = self.zp
        nondet = self['nondet'] if 'nondet' in self else None
        self['flux'], self['dflux'] = mag2flux(self['mag'], self['dmag'], zp, nondet, self.nondetSigmas)

    def calcAbsMag(self, dm=None, extinction=None, hostext=None, ebv=None, rv=None, host_ebv=None, host_rv=None,
                   redshift=None):
        """``'absmag'`` from ``'mag'``: distance modulus and per-filter extinction coefficients, given or computed
        from E(B-V) and R_V at each filter's effective wavelength (lightcurve.py:271-345).  Arguments override what
        ``meta`` holds; what is computed here is remembered in ``meta`` like the reference does."""
        meta = self.meta
        meta['redshift'] = redshift if redshift is not None else meta.get('redshift', 0.)
        if dm is not None:
            meta['dm'] = dm
        elif 'dm' not in meta:
            if meta['redshift']:
                raise NotImplementedError('no cosmology on this host: pass the distance modulus dm')
            meta['dm'] = 0.

        filters = set(self['filter'])
        # (meta key, coefficients given, E(B-V), R_V, redshift of the dust); the reference reads the host's redshift
        # from meta['z'] (lightcurve.py:330)
        screens = (('extinction', extinction, _first(ebv, meta.get('ebv')), _first(rv, meta.get('rv'), 3.1), 0.),
                   ('hostext', hostext, _first(host_ebv, meta.get('host_ebv')),
                    _first(host_rv, meta.get('host_rv'), 3.1), meta.get('z', 0.)))
        absmag = np.asarray(self['mag'], dtype=float) - meta['dm']
        for key, given, colour_excess, r_v, z_dust in screens:
            if given is not None:
                meta[key] = given
            elif key not in meta:
                meta[key] = {} if colour_excess is None else {
                    f.name: f.extinction(colour_excess, r_v, z_dust) for f in filters if f.wl_eff is not None}
            for filtobj in filters:       # the first of the filter's names with a coefficient wins
                a_lambda = next((meta[key][n] for n in filtobj.names if n in meta[key]), None)
                if a_lambda is not None:
                    absmag[_same(self['filter'], filtobj)] -= a_lambda
        self['absmag'] = absmag

    def calcLum(self, nondetSigmas=None):
        """``'lum'`` / ``'dlum'`` [W/Hz] from ``'absmag'`` / ``'dmag'`` (lightcurve.py:347-359)."""
        if nondetSigmas is not None:
            self.nondetSigmas = nondetSigmas
        nondet = self['nondet'] if 'nondet' in self else None
        self['lum'], self['dlum'] = mag2flux(self['absmag'], np.asarray(self['dmag'], dtype=float), self.zp + 90.19,
                                             nondet, self.nondetSigmas)


def _same(column, value):
    """Element-wise ``column == value`` as a boolean array (object columns compare entry by entry; ``None`` by
    identity, as the reference's ``where`` does)."""
    if value is None:
        return np.array([v is None for v in column], dtype=bool)
    if getattr(column, 'dtype', None) == object:
        return np.array([v == value for v in column], dtype=bool)
    return np.asarray(column == value, dtype=bool)


def _first(*candidates):
    """The first argument that is not None (None if there is none)."""
    return next((c for c in candidates if c is not None), None)


def _has_objects(value):
    try:
        return any(isinstance(v, Filter) for v in value)
    except TypeError:
        return False
