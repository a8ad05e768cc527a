"""Self-contained FITS reader / writer (numpy only) for the subset of the standard that Jolideco's
files use: a primary HDU, IMAGE extensions (the flux components) and BINTABLE extensions (loss
trace, configuration, calibrations, sparse components, GMM libraries).

The reference goes through astropy (jolideco/utils/io/fits.py:1-6), which is not installed next to
PyTorch-ROCm in this image; this module writes the same bytes-on-disk conventions (FITS 4.0: 2880-byte
blocks, 80-character cards, big-endian data, ``TFORMn`` / ``TDIMn`` columns, ``CONTINUE`` long
strings) so that files written here open in astropy and files written by the reference open here.
Interoperability is checked in tests/test_io_formats.py against astropy itself where an interpreter
that has it exists (``/opt/conda/bin/python3.9`` in the build image) and against
tests/golden/io/result.fits (and components / component / calibrations .fits), files real astropy wrote
from the HDUs the reference's writer produced (oracle/refload/make_golden_fits.py, hdus_to_fits.py).

Not supported (raises): variable-length array columns (P/Q), bit and complex columns, tile-compressed
images, random groups.
"""
import re
from pathlib import Path

import numpy as np

__all__ = ["Header", "HDU", "FitsTable", "read_fits", "write_fits"]

BLOCK = 2880
CARD = 80

_BITPIX_DTYPE = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}
_DTYPE_BITPIX = {"u1": 8, "i2": 16, "i4": 32, "i8": 64, "f4": -32, "f8": -64}
# TFORM letter -> (big-endian numpy dtype, bytes per element)
_TFORM_DTYPE = {"L": ("S1", 1), "B": (">u1", 1), "I": (">i2", 2), "J": (">i4", 4), "K": (">i8", 8),
                "E": (">f4", 4), "D": (">f8", 8), "A": ("S", 1)}
_TFORM_RE = re.compile(r"^\s*(\d*)([A-Z])(.*)$")
_COMMENTARY = ("COMMENT", "HISTORY", "")


class Header:
    """Ordered FITS header: ``header[key]``, ``get``, ``in``, ``items()`` over value cards; commentary
    cards (COMMENT / HISTORY) are kept in order and round-trip, but are not addressable by key."""

    def __init__(self, cards=None):
        self._cards = []  # [key, value, comment]
        for card in cards or []:
            self.append(*card)

    @staticmethod
    def _norm(key):
        return str(key).upper().strip()

    def append(self, key, value, comment=""):
        self._cards.append([self._norm(key), value, comment or ""])

    def _find(self, key):
        key = self._norm(key)
        if key in _COMMENTARY:
            return None
        for i, card in enumerate(self._cards):
            if card[0] == key:
                return i
        return None

    def __contains__(self, key):
        return self._find(key) is not None

    def __getitem__(self, key):
        i = self._find(key)
        if i is None:
            raise KeyError(f"Keyword {key!r} not found.")
        return self._cards[i][1]

    def get(self, key, default=None):
        i = self._find(key)
        return default if i is None else self._cards[i][1]

    def comment(self, key):
        i = self._find(key)
        return "" if i is None else self._cards[i][2]

    def __setitem__(self, key, value):
        comment = None
        if isinstance(value, tuple):
            value, comment = value
        i = self._find(key)
        if i is None:
            self.append(key, value, comment)
        else:
            self._cards[i][1] = value
            if comment is not None:
                self._cards[i][2] = comment

    def __delitem__(self, key):
        i = self._find(key)
        if i is None:
            raise KeyError(key)
        del self._cards[i]

    def pop(self, key, default=None):
        i = self._find(key)
        if i is None:
            return default
        return self._cards.pop(i)[1]

    def keys(self):
        return [c[0] for c in self._cards if c[0] not in _COMMENTARY]

    def items(self):
        return [(c[0], c[1]) for c in self._cards if c[0] not in _COMMENTARY]

    def cards(self):
        return [tuple(c) for c in self._cards]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._cards)

    def copy(self):
        return Header(self.cards())

    def update(self, other):
        for key, value in (other.items() if hasattr(other, "items") else other):
            self[key] = value

    def __repr__(self):
        return "\n".join(_format_card(*c)[0].rstrip() for c in self._cards)


class FitsTable:
    """Column store of a binary table: ``table[name]`` -> numpy array with the row axis first,
    ``colnames``, ``len``, row iteration as dicts, ``meta`` = the non-structural header keywords
    (what ``astropy.table.Table.read`` puts in ``table.meta``)."""

    def __init__(self, columns=None, meta=None):
        self._columns = {}
        self.meta = dict(meta or {})
        for name, values in (columns or {}).items():
            self[name] = values

    @classmethod
    def from_rows(cls, rows):
        """Table from a list of row dicts (``astropy.table.Table(rows)``)."""
        names = list(rows[0].keys()) if rows else []
        return cls({name: [row[name] for row in rows] for name in names})

    @property
    def colnames(self):
        return list(self._columns)

    def __setitem__(self, name, values):
        values = np.asarray(values)
        if values.ndim == 0:
            values = values[None]
        if self._columns and len(values) != len(self):
            raise ValueError(f"column {name!r} has {len(values)} rows, table has {len(self)}")
        self._columns[str(name)] = values

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._columns[item]
        if isinstance(item, (int, np.integer)):
            return {name: _scalar(col[item]) for name, col in self._columns.items()}
        raise TypeError(f"unsupported table index {item!r}")

    def __len__(self):
        for col in self._columns.values():
            return len(col)
        return 0

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __contains__(self, name):
        return name in self._columns

    def __repr__(self):
        return f"FitsTable(rows={len(self)}, columns={self.colnames})"


def _scalar(value):
    if isinstance(value, np.ndarray) and value.ndim > 0:
        return value
    if isinstance(value, (bytes, np.bytes_)):
        return value.decode("ascii")
    return value.item() if isinstance(value, np.generic) else value


class HDU:
    """One header-data unit.  ``kind`` is "primary", "image" or "bintable"; ``data`` is a numpy array,
    a `FitsTable` or None.  ``name`` is EXTNAME ("PRIMARY" for the first HDU when it has none)."""

    def __init__(self, data=None, header=None, name=None, kind=None):
        self.header = header.copy() if header is not None else Header()
        if isinstance(data, dict):
            data = FitsTable(data)
        self.data = data
        if kind is None:
            kind = "bintable" if isinstance(data, FitsTable) else "image"
        if kind not in ("primary", "image", "bintable"):
            raise ValueError(f"unknown HDU kind {kind!r}")
        self.kind = kind
        if name is not None:
            self.header["EXTNAME"] = str(name).upper()

    @property
    def name(self):
        return str(self.header.get("EXTNAME", "PRIMARY" if self.kind == "primary" else "")).strip()

    @property
    def is_image(self):
        return self.kind in ("primary", "image")

    def __repr__(self):
        shape = getattr(self.data, "shape", None) if self.is_image else (len(self.data), len(self.data.colnames))
        return f"HDU({self.name!r}, {self.kind}, {shape})"


# ---------------------------------------------------------------------------------------------- cards
def _format_value(value):
    """FITS fixed-format value field (columns 11-30 for numbers and logicals)."""
    if isinstance(value, (bool, np.bool_)):
        return f"{'T' if value else 'F':>20}"
    if isinstance(value, (int, np.integer)):
        return f"{int(value):>20d}"
    if isinstance(value, (float, np.floating)):
        value = float(value)
        if not np.isfinite(value):
            raise ValueError(f"FITS headers cannot hold {value}")
        text = f"{value:.16G}"
        if "." not in text and "E" not in text:
            text += ".0"
        elif "E" in text and "." not in text.split("E")[0]:
            mantissa, exponent = text.split("E")
            text = f"{mantissa}.0E{exponent}"
        return f"{text:>20}"
    raise TypeError(f"unsupported FITS header value {value!r} ({type(value).__name__})")


def _format_card(key, value, comment=""):
    """One header entry -> list of 80-character cards (several with the CONTINUE convention)."""
    key = str(key).upper()
    if key in _COMMENTARY:
        text = "" if value is None else str(value)
        chunks = [text[i:i + 72] for i in range(0, len(text), 72)] or [""]
        return [f"{key:<8}{chunk}".ljust(CARD) for chunk in chunks]
    if len(key) > 8 or not re.fullmatch(r"[A-Z0-9_-]*", key):
        raise ValueError(f"invalid FITS keyword {key!r} (at most 8 characters of A-Z 0-9 _ -)")
    if value is None:
        body = " " * 20  # undefined value
        card = f"{key:<8}= {body}"
    elif isinstance(value, (str, np.str_)):
        text = str(value).rstrip().replace("'", "''")
        if len(text) <= 68:
            card = f"{key:<8}= '{text:<8}'"
        else:
            # long string: 67 characters + '&' per card, CONTINUE cards for the rest
            cards, pieces = [], []
            while len(text) > 68:
                cut = 67
                if text[cut - 1] == "'" and (len(text[:cut]) - len(text[:cut].rstrip("'"))) % 2 == 1:
                    cut -= 1  # never split an escaped quote pair
                pieces.append(text[:cut])
                text = text[cut:]
            pieces.append(text)
            for i, piece in enumerate(pieces):
                amp = "&" if i < len(pieces) - 1 else ""
                head = f"{key:<8}= " if i == 0 else "CONTINUE  "
                cards.append(f"{head}'{piece}{amp}'".ljust(CARD))
            if comment:
                cards[-1] = (cards[-1].rstrip() + f" / {comment}")[:CARD].ljust(CARD)
            return cards
    else:
        card = f"{key:<8}= {_format_value(value)}"
    if comment:
        card = f"{card} / {comment}"
    return [card[:CARD].ljust(CARD)]


def _parse_value(field):
    """Value field (after '= ') -> (python value, comment, is_continued_string)."""
    text = field.strip()
    if not text:
        return None, "", False
    if text[0] == "'":
        i, chars = 1, []
        while i < len(text):
            if text[i] == "'":
                if i + 1 < len(text) and text[i + 1] == "'":
                    chars.append("'")
                    i += 2
                    continue
                break
            chars.append(text[i])
            i += 1
        value = "".join(chars).rstrip()
        rest = text[i + 1:]
        comment = rest.split("/", 1)[1].strip() if "/" in rest else ""
        return value, comment, value.endswith("&")
    value_text, _, comment = text.partition("/")
    value_text, comment = value_text.strip(), comment.strip()
    if value_text == "T":
        return True, comment, False
    if value_text == "F":
        return False, comment, False
    if not value_text:
        return None, comment, False
    try:
        return int(value_text), comment, False
    except ValueError:
        pass
    try:
        return float(value_text.replace("D", "E").replace("d", "e")), comment, False
    except ValueError:
        return value_text, comment, False


def _parse_header(raw):
    """Bytes of the header blocks -> Header (stops at END)."""
    header = Header()
    continued = False
    for offset in range(0, len(raw), CARD):
        card = raw[offset:offset + CARD].decode("ascii", errors="replace")
        key = card[:8].strip().upper()
        if key == "END":
            return header, True
        if key == "CONTINUE" and continued and header._cards:
            value, comment, continued = _parse_value(card[8:])
            last = header._cards[-1]
            last[1] = last[1][:-1] + (value if isinstance(value, str) else "")
            if comment:
                last[2] = (last[2] + " " + comment).strip()
            continue
        if key in _COMMENTARY:
            if card.strip():
                header.append(key, card[8:].rstrip())
            continued = False
            continue
        if key == "HIERARCH":
            name, _, field = card[8:].partition("=")
            value, comment, continued = _parse_value(field)
            header._cards.append([name.strip().upper(), value, comment])
            continue
        if card[8:10] != "= ":
            if key:  # keyword without a value indicator: keep its text
                header._cards.append([key, card[8:].strip(), ""])
            continued = False
            continue
        value, comment, continued = _parse_value(card[10:])
        header._cards.append([key, value, comment])
    return header, False


def _header_bytes(header):
    cards = []
    for key, value, comment in header.cards():
        cards.extend(_format_card(key, value, comment))
    cards.append("END".ljust(CARD))
    raw = "".join(cards).encode("ascii")
    return raw + b" " * (-len(raw) % BLOCK)


# ---------------------------------------------------------------------------------------------- read
def _tform(text):
    match = _TFORM_RE.match(str(text))
    if not match:
        raise ValueError(f"cannot parse TFORM {text!r}")
    repeat, letter, _rest = match.groups()
    repeat = int(repeat) if repeat else 1
    if letter in ("P", "Q"):
        raise NotImplementedError("variable-length array columns (TFORM P/Q) are not supported")
    if letter not in _TFORM_DTYPE:
        raise NotImplementedError(f"FITS column type {letter!r} (TFORM {text!r}) is not supported")
    return repeat, letter


def _tdim(text):
    dims = [int(v) for v in str(text).strip().strip("()").split(",") if v.strip()]
    return tuple(reversed(dims))  # FITS lists the fastest axis first


_STRUCTURAL = re.compile(
    r"^(XTENSION|SIMPLE|BITPIX|NAXIS\d*|PCOUNT|GCOUNT|TFIELDS|EXTEND|"
    r"(TTYPE|TFORM|TUNIT|TDIM|TNULL|TSCAL|TZERO|TDISP)\d+)$"
)


def _read_table(header, raw):
    n_rows, row_bytes, n_fields = header["NAXIS2"], header["NAXIS1"], header["TFIELDS"]
    if header.get("PCOUNT", 0):
        raise NotImplementedError("binary tables with a heap (variable-length arrays) are not supported")
    fields, offset = [], 0
    for i in range(1, n_fields + 1):
        repeat, letter = _tform(header[f"TFORM{i}"])
        base, size = _TFORM_DTYPE[letter]
        name = str(header.get(f"TTYPE{i}", f"col{i}")).strip()
        fields.append((i, name, repeat, letter, base, offset))
        offset += repeat * size
    if offset != row_bytes:
        raise ValueError(f"TFORM widths add up to {offset} bytes, NAXIS1 says {row_bytes}")
    rows = np.frombuffer(raw, dtype=np.uint8, count=n_rows * row_bytes).reshape(n_rows, row_bytes)
    meta = {key: value for key, value in header.items() if not _STRUCTURAL.match(key)}
    table = FitsTable(meta=meta)
    for i, name, repeat, letter, base, start in fields:
        size = _TFORM_DTYPE[letter][1]
        chunk = np.ascontiguousarray(rows[:, start:start + repeat * size])
        if letter == "A":
            raw_strings = chunk.view(f"S{repeat}").reshape(n_rows) if repeat else np.zeros(n_rows, "S1")
            strings = [s.split(b"\x00", 1)[0].decode("ascii", errors="replace").rstrip() for s in raw_strings]
            column = np.array(strings, dtype=str) if strings else np.zeros(0, dtype="U1")
        elif letter == "L":
            column = chunk.reshape(n_rows, repeat) == ord("T")
        else:
            column = chunk.view(base).reshape(n_rows, repeat)
            column = column.astype(column.dtype.newbyteorder("="))
            scale, zero = header.get(f"TSCAL{i}", 1), header.get(f"TZERO{i}", 0)
            if scale != 1 or zero != 0:
                if scale == 1 and isinstance(zero, int) and letter in "IJK" and zero == 1 << (8 * size - 1):
                    column = (column.astype(np.int64) + zero).astype(f"u{size}")  # unsigned convention
                else:
                    column = column * scale + zero
        if letter != "A":
            if f"TDIM{i}" in header:
                column = column.reshape((n_rows,) + _tdim(header[f"TDIM{i}"]))
            elif repeat == 1:
                column = column.reshape(n_rows)
        table[name] = column
    return table


def read_fits(filename):
    """Read every HDU of a FITS file -> list of `HDU`."""
    raw = Path(filename).read_bytes()
    hdus, pos = [], 0
    while pos < len(raw):
        if not raw[pos:pos + BLOCK].strip(b"\x00 "):
            pos += BLOCK  # trailing padding blocks
            continue
        start = pos
        header, done = Header(), False
        while not done:
            if pos >= len(raw):
                raise ValueError(f"{filename}: header without END card at byte {start}")
            block, done = _parse_header(raw[pos:pos + BLOCK])
            header._cards.extend(block._cards)
            pos += BLOCK
        first = header.keys()[0] if header.keys() else ""
        if not hdus and first != "SIMPLE":
            raise ValueError(f"{filename} is not a FITS file (first keyword {first!r})")
        naxis = int(header.get("NAXIS", 0))
        shape = tuple(int(header[f"NAXIS{i}"]) for i in range(naxis, 0, -1))
        bitpix = int(header.get("BITPIX", 8))
        n_bytes = (abs(bitpix) // 8) * int(np.prod(shape)) if naxis else 0
        n_bytes = (n_bytes + int(header.get("PCOUNT", 0))) * int(header.get("GCOUNT", 1))
        payload = raw[pos:pos + n_bytes]
        if len(payload) < n_bytes:
            raise ValueError(f"{filename}: truncated data in HDU {len(hdus)}")
        pos += n_bytes + (-n_bytes % BLOCK)
        xtension = str(header.get("XTENSION", "")).strip().upper()
        if first == "SIMPLE" or xtension == "IMAGE":
            data = None
            if naxis and n_bytes:
                data = np.frombuffer(payload, dtype=_BITPIX_DTYPE[bitpix]).reshape(shape)
                data = data.astype(data.dtype.newbyteorder("="))
                scale, zero = header.get("BSCALE", 1), header.get("BZERO", 0)
                if scale != 1 or zero != 0:
                    if scale == 1 and bitpix in (16, 32, 64) and zero == 1 << (bitpix - 1):
                        data = (data.astype(np.int64) + zero).astype(f"u{bitpix // 8}")
                    else:
                        data = data * scale + zero
            hdu = HDU(data=data, header=header, kind="primary" if first == "SIMPLE" else "image")
        elif xtension == "BINTABLE":
            hdu = HDU(data=_read_table(header, payload), header=header, kind="bintable")
        else:
            raise NotImplementedError(f"{filename}: XTENSION {xtension!r} is not supported")
        hdus.append(hdu)
    if not hdus:
        raise ValueError(f"{filename} holds no HDU")
    return hdus


# --------------------------------------------------------------------------------------------- write
def _user_cards(header, skip):
    return [(k, v, c) for k, v, c in header.cards() if not (k in skip or _STRUCTURAL.match(k))]


def _image_bytes(hdu, primary):
    data = hdu.data
    header = Header()
    if primary:
        header.append("SIMPLE", True, "conforms to FITS standard")
    else:
        header.append("XTENSION", "IMAGE", "Image extension")
    if data is None:
        header.append("BITPIX", 8, "array data type")
        header.append("NAXIS", 0, "number of array dimensions")
        payload = b""
    else:
        data = np.asarray(data)
        if data.dtype == np.bool_:
            data = data.astype(np.uint8)
        code = f"{data.dtype.kind}{data.dtype.itemsize}"
        if code not in _DTYPE_BITPIX:
            raise TypeError(f"cannot store dtype {data.dtype} in a FITS image")
        header.append("BITPIX", _DTYPE_BITPIX[code], "array data type")
        header.append("NAXIS", data.ndim, "number of array dimensions")
        for i, n in enumerate(reversed(data.shape), start=1):
            header.append(f"NAXIS{i}", int(n))
        payload = np.ascontiguousarray(data, dtype=data.dtype.newbyteorder(">")).tobytes()
    if primary:
        header.append("EXTEND", True)
    else:
        header.append("PCOUNT", 0, "number of parameters")
        header.append("GCOUNT", 1, "number of groups")
    for card in _user_cards(hdu.header, skip=()):
        header.append(*card)
    return _header_bytes(header) + payload + b"\x00" * (-len(payload) % BLOCK)


def _column_spec(name, values):
    """numpy column -> (TFORM, TDIM or None, big-endian bytes per row as a 2-d uint8 array)."""
    values = np.asarray(values)
    n_rows = len(values)
    kind = values.dtype.kind
    if kind in ("U", "S", "O"):
        strings = [v.decode("ascii") if isinstance(v, bytes) else str(v) for v in values.reshape(-1)]
        if values.ndim != 1:
            raise NotImplementedError(f"column {name!r}: arrays of strings are not supported")
        width = max([len(s) for s in strings] + [1])
        raw = np.array([s.encode("ascii") for s in strings], dtype=f"S{width}").view(np.uint8).reshape(n_rows, width)
        return f"{width}A", None, raw
    repeat = int(np.prod(values.shape[1:])) if values.ndim > 1 else 1
    if kind == "b":
        raw = np.where(values.reshape(n_rows, repeat), ord("T"), ord("F")).astype(np.uint8)
        letter, size = "L", 1
    else:
        letter = {"u1": "B", "i2": "I", "i4": "J", "i8": "K", "f4": "E", "f8": "D"}.get(f"{kind}{values.dtype.itemsize}")
        if letter is None:
            raise TypeError(f"column {name!r}: cannot store dtype {values.dtype} in a FITS table")
        size = values.dtype.itemsize
        flat = np.ascontiguousarray(values.reshape(n_rows, repeat), dtype=values.dtype.newbyteorder(">"))
        raw = flat.view(np.uint8).reshape(n_rows, repeat * size)
    tdim = None
    if values.ndim > 2:
        tdim = "(" + ",".join(str(int(n)) for n in reversed(values.shape[1:])) + ")"
    return (f"{repeat}{letter}" if repeat != 1 else letter), tdim, raw.reshape(n_rows, repeat * size)


def _table_bytes(hdu):
    table = hdu.data
    specs = [(name,) + _column_spec(name, table[name]) for name in table.colnames]
    n_rows = len(table)
    row_bytes = sum(raw.shape[1] for *_x, raw in specs)
    header = Header()
    header.append("XTENSION", "BINTABLE", "binary table extension")
    header.append("BITPIX", 8, "array data type")
    header.append("NAXIS", 2, "number of array dimensions")
    header.append("NAXIS1", row_bytes, "length of dimension 1")
    header.append("NAXIS2", n_rows, "length of dimension 2")
    header.append("PCOUNT", 0, "number of group parameters")
    header.append("GCOUNT", 1, "number of groups")
    header.append("TFIELDS", len(specs), "number of table fields")
    for i, (name, tform, tdim, _raw) in enumerate(specs, start=1):
        header.append(f"TTYPE{i}", name)
        header.append(f"TFORM{i}", tform)
        if tdim:
            header.append(f"TDIM{i}", tdim)
    for key, value in table.meta.items():
        if not _STRUCTURAL.match(str(key).upper()) and str(key).upper() != "EXTNAME":
            header.append(key, value)
    for card in _user_cards(hdu.header, skip=set(k.upper() for k in table.meta)):
        header.append(*card)
    payload = np.concatenate([raw for *_x, raw in specs], axis=1).tobytes() if specs and n_rows else b""
    return _header_bytes(header) + payload + b"\x00" * (-len(payload) % BLOCK)


def write_fits(filename, hdus, overwrite=False):
    """Write a list of `HDU` to ``filename``.  The first must be an image (it becomes the primary HDU);
    if it is not, an empty primary HDU is inserted, as astropy does."""
    path = Path(filename)
    if path.exists() and not overwrite:
        raise OSError(f"File {str(path)!r} already exists. Use overwrite=True to replace it.")
    hdus = list(hdus)
    if not hdus or not hdus[0].is_image:
        hdus.insert(0, HDU(kind="primary"))
    chunks = []
    for i, hdu in enumerate(hdus):
        if hdu.is_image:
            chunks.append(_image_bytes(hdu, primary=(i == 0)))
        else:
            chunks.append(_table_bytes(hdu))
    path.write_bytes(b"".join(chunks))
