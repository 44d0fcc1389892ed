"""Minimal stand-in for ``astropy.units`` (build-host only, never shipped to the GPU box).

Purpose: let the reference's ``models.py`` / ``filters.py`` be imported in a container
without astropy so that golden vectors can be generated from the reference's own
arithmetic.  Supplies unit algebra and conversion factors
only -- no model mathematics lives here.
"""
import math
import re

import numpy as np

# base dimensions: metre, kilogram, second, kelvin, magnitude, radian
_NDIM = 6


def _dims(**kw):
    order = ('m', 'kg', 's', 'K', 'mag', 'rad')
    return tuple(kw.get(k, 0) for k in order)


class UnitConversionError(Exception):
    pass


class Unit:
    """A physical unit = SI scale factor x integer/fractional exponents of the base dimensions."""
    __array_priority__ = 1000

    def __init__(self, scale, dims, name=''):
        self.scale = float(scale)
        self.dims = tuple(dims)
        self.name = name

    # --- algebra -------------------------------------------------------
    def _combine(self, other, sign):
        return Unit(self.scale * other.scale ** sign,
                    tuple(a + sign * b for a, b in zip(self.dims, other.dims)))

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        if isinstance(other, Quantity):
            return Quantity(other.value, self * other.unit)
        return Quantity(other, self)

    def __rmul__(self, other):
        if isinstance(other, Column):
            return Quantity(np.asarray(other), self)
        return Quantity(other, self)

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, -1)
        if isinstance(other, Quantity):
            return Quantity(1. / other.value, self / other.unit)
        return Quantity(1. / other, self)

    def __rtruediv__(self, other):
        return Quantity(other, self ** -1)

    def __pow__(self, p):
        return Unit(self.scale ** p, tuple(a * p for a in self.dims))

    def __eq__(self, other):
        return isinstance(other, Unit) and self.dims == other.dims and math.isclose(self.scale, other.scale,
                                                                                    rel_tol=1e-15)

    def __hash__(self):
        return hash((self.scale, self.dims))

    def __repr__(self):
        return f'Unit({self.name or self.scale!r}, {self.dims})'

    def __format__(self, spec):
        return self.name or repr(self)

    # --- conversion ----------------------------------------------------
    def to(self, other, value=1.):
        scale, dims = _target(other)
        if tuple(dims) != self.dims:
            raise UnitConversionError(f'{self} -> {other}')
        return value * self.scale / scale


def _target(other):
    """(scale, dims) of a conversion target given as Unit, Quantity or string."""
    if isinstance(other, str):
        other = _parse(other)
    if isinstance(other, Quantity):
        return float(other.value) * other.unit.scale, other.unit.dims
    return other.scale, other.dims


class Quantity:
    """value (scalar or ndarray) tagged with a Unit.  Deliberately not an ndarray subclass."""
    __array_priority__ = 2000

    def __init__(self, value, unit=None):
        if isinstance(value, Quantity):
            unit = value.unit if unit is None else unit
            value = value.value
        elif isinstance(value, (list, tuple)) and len(value) and isinstance(value[0], Quantity):
            unit = value[0].unit
            value = np.array([v.to(unit).value for v in value])
        if isinstance(value, Column):
            value = np.asarray(value)
        self.value = value if np.isscalar(value) else np.asarray(value)
        self.unit = dimensionless_unscaled if unit is None else unit

    def to(self, other):
        scale, dims = _target(other)
        if tuple(dims) != self.unit.dims:
            raise UnitConversionError(f'{self.unit} -> {other}')
        tgt = other if isinstance(other, Unit) else Unit(scale, dims)
        return Quantity(self.value * (self.unit.scale / scale), tgt)

    def _coerce(self, other):
        if isinstance(other, Quantity):
            return other.value, other.unit
        if isinstance(other, Unit):
            return 1., other
        if isinstance(other, Column):
            return np.asarray(other), (other.unit or dimensionless_unscaled)
        return other, dimensionless_unscaled

    def __mul__(self, other):
        v, un = self._coerce(other)
        return Quantity(self.value * v, self.unit * un)

    __rmul__ = __mul__

    def __truediv__(self, other):
        v, un = self._coerce(other)
        return Quantity(self.value / v, self.unit / un)

    def __rtruediv__(self, other):
        v, un = self._coerce(other)
        return Quantity(v / self.value, un / self.unit)

    def __pow__(self, p):
        return Quantity(self.value ** p, self.unit ** p)

    def __neg__(self):
        return Quantity(-self.value, self.unit)

    def _same(self, other):
        v, un = self._coerce(other)
        if un.dims != self.unit.dims:
            raise UnitConversionError(f'{self.unit} vs {un}')
        return v * (un.scale / self.unit.scale)

    def __add__(self, other):
        return Quantity(self.value + self._same(other), self.unit)

    def __sub__(self, other):
        return Quantity(self.value - self._same(other), self.unit)

    def __getitem__(self, item):
        return Quantity(self.value[item], self.unit)

    def __len__(self):
        return len(self.value)

    def __float__(self):
        return float(self.value)

    def __repr__(self):
        return f'<Quantity {self.value!r} {self.unit!r}>'

    def __array_function__(self, func, types, args, kwargs):
        if func.__name__ in ('trapz', 'trapezoid'):
            y = args[0]
            x = args[1] if len(args) > 1 else kwargs.get('x')
            yv, yu = self._coerce(y)
            if x is None:
                return Quantity(func(yv), yu)
            xv, xu = self._coerce(x)
            return Quantity(func(yv, xv), yu * xu)
        raise NotImplementedError(func.__name__)


quantity = type('quantity', (), {'Quantity': Quantity})  # ``u.quantity.Quantity`` is referenced by the reference


class Column(np.ndarray):
    """ndarray with a settable ``unit``, as far as the reference's filter loader needs."""

    def __new__(cls, data, unit=None, name=None):
        obj = np.asarray(data).view(cls)
        obj.unit = unit
        obj.name = name
        return obj

    def __array_finalize__(self, obj):
        self.unit = getattr(obj, 'unit', None)
        self.name = getattr(obj, 'name', None)

    @property
    def quantity(self):
        return Quantity(np.asarray(self), self.unit or dimensionless_unscaled)

    @property
    def value(self):
        return np.asarray(self)

    @property
    def data(self):
        return np.asarray(self)


# --- unit registry (SI scales: CODATA 2018 / IAU 2015) ---------------------
dimensionless_unscaled = Unit(1., _dims(), '')
m = Unit(1., _dims(m=1), 'm')
cm = Unit(1e-2, _dims(m=1), 'cm')
nm = Unit(1e-9, _dims(m=1), 'nm')
angstrom = AA = Unit(1e-10, _dims(m=1), 'Angstrom')
s = Unit(1., _dims(s=1), 's')
d = day = Unit(86400., _dims(s=1), 'd')
Hz = Unit(1., _dims(s=-1), 'Hz')
THz = Unit(1e12, _dims(s=-1), 'THz')
K = Unit(1., _dims(K=1), 'K')
kK = Unit(1e3, _dims(K=1), 'kK')
kg = Unit(1., _dims(kg=1), 'kg')
J = Unit(1., _dims(kg=1, m=2, s=-2), 'J')
W = Unit(1., _dims(kg=1, m=2, s=-3), 'W')
erg = Unit(1e-7, _dims(kg=1, m=2, s=-2), 'erg')
eV = Unit(1.602176634e-19, _dims(kg=1, m=2, s=-2), 'eV')
Rsun = Unit(6.957e8, _dims(m=1), 'Rsun')
Msun = Unit(1.988409870698051e30, _dims(kg=1), 'Msun')
au = Unit(1.495978707e11, _dims(m=1), 'au')
pc = Unit(au.scale / math.radians(1. / 3600.), _dims(m=1), 'pc')
Mpc = Unit(1e6 * pc.scale, _dims(m=1), 'Mpc')
mag = Unit(1., _dims(mag=1), 'mag')
rad = Unit(1., _dims(rad=1), 'rad')
deg = Unit(math.pi / 180., _dims(rad=1), 'deg')


def def_unit(name, represents, format=None, **kwargs):
    scale, dims = _target(represents)
    return Unit(scale, dims, name)


_TOKEN = re.compile(r'^([A-Za-z]+)(-?\d+)?$')


def _parse(text):
    """Parse strings like ``"eV / kK"`` or ``"erg s-1 Rsun-2 kK-4"``."""
    out = dimensionless_unscaled
    sign = +1
    for tok in text.split():
        if tok == '/':
            sign = -1
            continue
        mt = _TOKEN.match(tok)
        if not mt:
            raise ValueError(f'cannot parse unit token {tok!r}')
        base = globals()[mt.group(1)]
        power = int(mt.group(2)) if mt.group(2) else 1
        out = out * base ** (sign * power)
    return out
