import numpy as np


class Row(dict):
    @property
    def colnames(self):
        return list(self.keys())


class Table:
    """Minimal row container with the calls jolideco/loss.py:192-250 and core.py:249-267 make."""

    def __init__(self, data=None, names=None, dtype=None, meta=None):
        rows = data if isinstance(data, list) else None  # Table(rows): list of dicts
        if rows:
            names = list(rows[0].keys())
        self.colnames = list(names) if names is not None else []
        self._dtype = list(dtype) if dtype is not None else [float] * len(self.colnames)
        self._rows = [Row(r) for r in rows] if rows else []
        self.meta = meta or {}

    def write(self, filename, overwrite=False, format=None):
        """Table.write(format="fits") = an empty primary HDU + the table (recorded, see io/fits.py)."""
        from astropy.io import fits

        if format != "fits":
            raise NotImplementedError(format)
        fits.HDUList([fits.PrimaryHDU(), fits.BinTableHDU(self)]).writeto(filename, overwrite=overwrite)

    def add_row(self, row):
        self._rows.append(Row({name: row[name] for name in self.colnames}))

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, item):
        if isinstance(item, str):
            return np.array([r[item] for r in self._rows])
        if isinstance(item, slice):
            out = Table(names=self.colnames, dtype=self._dtype)
            out._rows = self._rows[item]
            return out
        return self._rows[item]

    def __setitem__(self, key, value):
        if key not in self.colnames:
            self.colnames.append(key)
            self._dtype.append(type(value[0]))
            if not self._rows:
                self._rows = [Row() for _ in value]
        for r, v in zip(self._rows, value):
            r[key] = v

    def copy(self):
        out = Table(names=self.colnames, dtype=self._dtype)
        out._rows = [Row(r) for r in self._rows]
        return out
