"""Minimal trace table with the access pattern of the astropy Table the reference uses for the
loss trace (jolideco/loss.py:192-250, core.py:249-267): named columns, `add_row(dict)`,
`table["col"]` -> numpy array, `table[-1]` -> row mapping, `len(table)`, `colnames`."""
import numpy as np

__all__ = ["TraceTable"]


class TraceRow(dict):
    @property
    def colnames(self):
        return list(self.keys())


class TraceTable:
    def __init__(self, names):
        self.colnames = list(names)
        self._rows = []
        self.meta = {}

    def add_row(self, row):
        missing = [n for n in self.colnames if n not in row]
        if missing:
            raise ValueError(f"row is missing columns {missing}")
        self._rows.append(TraceRow({n: row[n] for n in self.colnames}))

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, item):
        if isinstance(item, str):
            if item not in self.colnames:
                raise KeyError(item)
            return np.array([r[item] for r in self._rows])
        if isinstance(item, slice):
            out = TraceTable(self.colnames)
            out._rows = self._rows[item]
            return out
        return self._rows[item]

    def __iter__(self):
        return iter(self._rows)

    def copy(self):
        out = TraceTable(self.colnames)
        out._rows = [TraceRow(r) for r in self._rows]
        return out

    def to_dict(self):
        return {n: self[n] for n in self.colnames}

    def __repr__(self):
        return f"TraceTable(rows={len(self)}, columns={self.colnames})"
