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
    def where(self, **kwargs):
        """Rows matching every criterion: ``col=value``, ``col_not=``, ``col_min=``, ``col_max=``; lists match any."""
        use = np.ones(len(self), bool)
        for col, val in kwargs.items():
            if col.startswith('filter'):
                if isinstance(val, str):
                    val = filtdict[val]
                elif isinstance(val, list):
                    val = [filtdict[v] if isinstance(v, str) else v for v in val]
            if isinstance(val, list):
                if '_not' in col:
                    use1 = np.ones(len(self), bool)
                    for v in val:
                        use1 &= self[col.replace('_not', '')] != v
                else:
                    use1 = np.zeros(len(self), bool)
                    for v in val:
                        use1 |= self[col] == v
            elif '_min' in col:
                use1 = self[col.replace('_min', '')] >= val
            elif '_max' in col:
                use1 = self[col.replace('_max', '')] <= val
            elif '_not' in col:
                use1 = self[col.replace('_not', '')] != val
            else:
                use1 = self[col] == val
            use &= np.asarray(use1, dtype=bool)
        return self[use]

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
        from E(B-V) and R_V at each filter's effective wavelength (lightcurve.py:271-345)."""
        if redshift is not None:
            self.meta['redshift'] = redshift
        elif 'redshift' not in self.meta:
            self.meta['redshift'] = 0.
        if dm is not None:
            self.meta['dm'] = dm
        elif 'dm' not in self.meta and self.meta.get('redshift'):
            raise NotImplementedError('no cosmology on this host: pass the distance modulus dm')
        elif 'dm' not in self.meta:
            self.meta['dm'] = 0.
        if ebv is None:
            ebv = self.meta.get('ebv')
        if host_ebv is None:
            host_ebv = self.meta.get('host_ebv')
        if rv is None:
            rv = self.meta.get('rv', 3.1)
        if host_rv is None:
            host_rv = self.meta.get('host_rv', 3.1)
        with_table = [f for f in set(self['filter']) if f.wl_eff is not None]
        if extinction is not None:
            self.meta['extinction'] = extinction
        elif 'extinction' not in self.meta:
            self.meta['extinction'] = {} if ebv is None else {f.name: f.extinction(ebv, rv) for f in with_table}
        if hostext is not None:
            self.meta['hostext'] = hostext
        elif 'hostext' not in self.meta:  # the reference reads the host redshift from meta['z'] here (:330)
            self.meta['hostext'] = {} if host_ebv is None else {
                f.name: f.extinction(host_ebv, host_rv, self.meta.get('z', 0.)) for f in with_table}
        absmag = np.asarray(self['mag'], dtype=float) - self.meta['dm']
        filt_col = self['filter']
        for filtobj in set(filt_col):
            rows = np.array([f == filtobj for f in filt_col])
            for key in ('extinction', 'hostext'):
                for name in filtobj.names:
                    if name in self.meta[key]:
                        absmag[rows] -= self.meta[key][name]
                        break
        self['absmag'] = absmag

    def calcLum(self, nondetSigmas=None):
        """``'lum'`` / ``'dlum'`` [W/Hz] from ``'absmag'`` / ``'dmag'`` (lightcurve.py:347-359)."""
        if nondetSigmas is not None:
            self.nondetSigmas = nondetSigmas
        nondet = self['nondet'] if 'nondet' in self else None
        self['lum'], self['dlum'] = mag2flux(self['absmag'], np.asarray(self['dmag'], dtype=float), self.zp + 90.19,
                                             nondet, self.nondetSigmas)


def _has_objects(value):
    try:
        return any(isinstance(v, Filter) for v in value)
    except TypeError:
        return False
