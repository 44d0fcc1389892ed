"""Stand-in for ``astropy.table``: a dict of equal-length Columns with ASCII reading.  I/O only."""
import numpy as np
from .units import Column, Quantity


def _to_float(tok):
    try:
        return float(tok)
    except ValueError:
        return None


class Table:
    def __init__(self, data=None, names=None, meta=None, masked=False, **kwargs):
        self.columns = {}
        self.meta = dict(meta or {})
        if isinstance(data, dict):
            for k, v in data.items():
                self[k] = v
        elif data is not None and names is not None:
            for k, v in zip(names, data):
                self[k] = v

    # --- reading ---------------------------------------------------------
    @classmethod
    def read(cls, filename, format='ascii', names=None, **kwargs):
        header = None
        rows = []
        with open(filename) as fh:
            for line in fh:
                line = line.strip()
                if not line:
                    continue
                if line.startswith('#'):
                    if header is None and not rows:
                        header = line.lstrip('#').split()
                    continue
                toks = line.replace(',', ' ').split()
                vals = [_to_float(tk) for tk in toks]
                if any(v is None for v in vals):
                    if not rows and header is None:
                        header = toks  # un-commented header row (CSV tables)
                    continue
                rows.append(vals)
        arr = np.array(rows, dtype=float)
        if names is None:
            names = header if header is not None and len(header) == arr.shape[1] \
                else [f'col{i + 1}' for i in range(arr.shape[1])]
        out = cls()
        for i, name in enumerate(names):
            out[name] = arr[:, i]
        return out

    # --- container protocol ----------------------------------------------
    @property
    def colnames(self):
        return list(self.columns)

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def __contains__(self, item):
        return item in self.columns

    def __getitem__(self, item):
        if isinstance(item, str):
            return self.columns[item]
        out = type(self)()
        out.meta = dict(self.meta)
        idx = np.asarray(item) if not isinstance(item, slice) else item
        for k, v in self.columns.items():
            out.columns[k] = Column(np.asarray(v)[idx], unit=v.unit, name=k)
        return out

    def __setitem__(self, key, value):
        if isinstance(value, Quantity):
            col = Column(value.value, unit=value.unit, name=key)
        elif isinstance(value, Column):
            col = Column(np.asarray(value), unit=value.unit, name=key)
        else:
            col = Column(np.asarray(value), name=key)
        self.columns[key] = col

    def sort(self, key):
        order = np.argsort(np.asarray(self.columns[key]), kind='stable')
        for k, v in self.columns.items():
            self.columns[k] = Column(np.asarray(v)[order], unit=v.unit, name=k)


class MaskedColumn(Column):
    pass


def vstack(tables, **kwargs):
    out = type(tables[0])()
    for k in tables[0].colnames:
        out[k] = np.concatenate([np.asarray(t[k]) for t in tables])
    return out
