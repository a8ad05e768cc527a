"""A self-contained codec for the subset of the ASDF 1.x file format Jolideco's result files use.

The reference writes its ASDF files with the ``asdf`` package (jolideco/utils/io/asdf.py); that package is not
available here, so -- as for FITS (`_fitsfile`) -- the format itself is implemented, from the published ASDF standard
(1.5.0, "File layout"):

    #ASDF 1.0.0                                  file format version
    #ASDF_STANDARD 1.5.0
    %YAML 1.1
    %TAG ! tag:stsci.edu:asdf/
    --- !core/asdf-1.1.0                          the tree: one YAML 1.1 document
    ...
    \\xd3BLK <header> <data>                       binary blocks, referred to from the tree by index
    #ASDF BLOCK INDEX                             optional: a YAML list of the block offsets
    %YAML 1.1
    ---
    ...

Block header (big-endian): magic ``\\xd3BLK``, header size uint16 (48), flags uint32, compression 4 bytes (zeros = none),
allocated size, used size, data size (uint64 each), MD5 checksum of the used data (16 bytes).

Tree nodes: numpy arrays are ``!core/ndarray-1.0.0`` mappings ``{source: <block>, datatype, byteorder, shape}`` (small
arrays inline: ``{data: [...], datatype, shape}``); a table (the loss trace) is written the way ``asdf-astropy`` writes an
astropy ``Table`` -- ``!<tag:astropy.org:astropy/table/table-1.0.0> {colnames, columns: [!core/column-1.0.0 {data,
name}], meta, qtable}`` -- and read from that tag or from the standard's own ``!core/table-1.0.0``.

Reading covers what the ``asdf`` package produces for such trees: internal blocks (optionally zlib / bzip2 compressed),
inline arrays, big- or little-endian data, ``offset``; external (``source: <uri>``) and strided arrays raise.
"""
import bz2
import hashlib
import io
import struct
import zlib
from pathlib import Path

import numpy as np
import yaml

__all__ = ["write_asdf", "read_asdf", "Table", "BLOCK_MAGIC"]

ASDF_MAGIC = b"#ASDF"
FILE_FORMAT_VERSION = "1.0.0"
STANDARD_VERSION = "1.5.0"
BLOCK_MAGIC = b"\xd3BLK"
BLOCK_HEADER = struct.Struct(">I4sQQQ16s")  # flags, compression, allocated, used, data size, checksum (after the uint16 size)
INDEX_HEADER = b"#ASDF BLOCK INDEX"
TAG_PREFIX = "tag:stsci.edu:asdf/"
TAG_TREE = TAG_PREFIX + "core/asdf-1.1.0"
TAG_SOFTWARE = TAG_PREFIX + "core/software-1.0.0"
TAG_NDARRAY = TAG_PREFIX + "core/ndarray-1.0.0"
TAG_COLUMN = TAG_PREFIX + "core/column-1.0.0"
TAG_TABLE_ASTROPY = "tag:astropy.org:astropy/table/table-1.0.0"
INLINE_THRESHOLD = 0  # arrays are written to binary blocks (asdf inlines nothing for float data by default either)

_DATATYPES = {
    "int8": "i1", "int16": "i2", "int32": "i4", "int64": "i8", "uint8": "u1", "uint16": "u2", "uint32": "u4", "uint64": "u8",
    "float16": "f2", "float32": "f4", "float64": "f8", "complex64": "c8", "complex128": "c16", "bool8": "b1",
}
_DATATYPE_NAMES = {np.dtype(v).newbyteorder("=").str[1:]: k for k, v in _DATATYPES.items()}


class Table:
    """Columns by name, in order (what the codec returns for a table node; `write_asdf` takes any object with
    ``colnames`` and ``__getitem__(name) -> array`` -- the fit's `TraceTable`, an astropy Table)."""

    def __init__(self, columns, meta=None):
        self.columns = dict(columns)
        self.colnames = list(self.columns)
        self.meta = dict(meta or {})

    def __getitem__(self, name):
        return self.columns[name]

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0


class _Tagged:
    """A YAML node with an explicit tag (value: dict, list or scalar)."""

    def __init__(self, tag, value):
        self.tag, self.value = tag, value


class _Dumper(yaml.SafeDumper):
    def ignore_aliases(self, data):
        return True


def _represent_tagged(dumper, node):
    if isinstance(node.value, dict):
        return dumper.represent_mapping(node.tag, node.value, flow_style=None)
    if isinstance(node.value, (list, tuple)):
        return dumper.represent_sequence(node.tag, node.value)
    return dumper.represent_scalar(node.tag, str(node.value))


_Dumper.add_representer(_Tagged, _represent_tagged)


def _datatype_of(dtype):
    """ASDF ``datatype`` of a numpy dtype: a name, or [ascii | ucs4, length] for strings."""
    dtype = np.dtype(dtype)
    if dtype.kind == "U":
        return ["ucs4", dtype.itemsize // 4]
    if dtype.kind == "S":
        return ["ascii", dtype.itemsize]
    key = dtype.newbyteorder("=").str[1:]
    if key not in _DATATYPE_NAMES:
        raise TypeError(f"no ASDF datatype for numpy dtype {dtype}")
    return _DATATYPE_NAMES[key]


def _numpy_dtype(datatype, byteorder):
    order = ">" if byteorder == "big" else "<"
    if isinstance(datatype, (list, tuple)):
        kind, length = datatype
        if kind == "ucs4":
            return np.dtype(f"{order}U{int(length)}")
        if kind == "ascii":
            return np.dtype(f"S{int(length)}")
        raise ValueError(f"unsupported ASDF datatype {datatype}")
    if datatype not in _DATATYPES:
        raise ValueError(f"unsupported ASDF datatype {datatype!r}")
    return np.dtype(_DATATYPES[datatype]).newbyteorder(order)


class _Writer:
    def __init__(self):
        self.blocks = []

    def ndarray(self, array):
        array = np.asarray(array)
        if array.dtype == object:
            raise TypeError("object arrays cannot be written to ASDF")
        if array.dtype.kind == "U" and array.dtype.itemsize == 0:
            array = array.astype("<U1")
        datatype = _datatype_of(array.dtype)
        little = array.astype(array.dtype.newbyteorder("<"), copy=False)
        if array.size <= INLINE_THRESHOLD:
            return _Tagged(TAG_NDARRAY, {"data": little.tolist(), "datatype": datatype, "shape": list(array.shape)})
        self.blocks.append(np.ascontiguousarray(little).tobytes())
        return _Tagged(TAG_NDARRAY, {"source": len(self.blocks) - 1, "datatype": datatype, "byteorder": "little",
                                     "shape": list(array.shape)})

    def table(self, table):
        columns = []
        for name in table.colnames:
            column = np.asarray(table[name])
            if column.dtype == object:
                column = column.astype(str)
            columns.append(_Tagged(TAG_COLUMN, {"data": self.ndarray(column), "name": str(name)}))
        meta = self.node(dict(getattr(table, "meta", None) or {}))
        return _Tagged(TAG_TABLE_ASTROPY, {"colnames": [str(n) for n in table.colnames], "columns": columns, "meta": meta,
                                           "qtable": False})

    def node(self, value):
        if isinstance(value, _Tagged):
            return value
        if isinstance(value, dict):
            return {str(k): self.node(v) for k, v in value.items()}
        if isinstance(value, np.ndarray):
            return self.ndarray(value)
        if hasattr(value, "colnames") and hasattr(value, "__getitem__"):
            return self.table(value)
        if isinstance(value, (list, tuple)):
            return [self.node(v) for v in value]
        if isinstance(value, np.generic):
            return value.item()
        if isinstance(value, Path):
            return str(value)
        if value is None or isinstance(value, (bool, int, float, str)):
            return value
        if hasattr(value, "detach") and hasattr(value, "cpu"):  # a torch tensor
            return self.ndarray(value.detach().cpu().numpy())
        raise TypeError(f"cannot write an object of type {type(value).__name__} to ASDF")


def _block_bytes(data):
    header = BLOCK_HEADER.pack(0, b"\0\0\0\0", len(data), len(data), len(data), hashlib.md5(data).digest())
    return BLOCK_MAGIC + struct.pack(">H", len(header)) + header + data


def write_asdf(filename, tree, overwrite=False, library=None):
    """Write ``tree`` (nested dicts / lists of scalars, numpy arrays, tables) as an ASDF file."""
    path = Path(filename)
    if path.exists() and not overwrite:
        raise OSError(f"{path} already exists!")
    writer = _Writer()
    body = {"asdf_library": _Tagged(TAG_SOFTWARE, dict(library or {"author": "jolideco_amd", "name": "jolideco_amd",
                                                                    "homepage": "https://github.com/jolideco/jolideco",
                                                                    "version": "0.1"}))}
    body.update(writer.node(tree))
    text = yaml.dump(_Tagged(TAG_TREE, body), Dumper=_Dumper, explicit_start=True, explicit_end=True, version=(1, 1),
                     tags={"!": TAG_PREFIX}, default_flow_style=False, sort_keys=False, allow_unicode=True, width=100)
    out = io.BytesIO()
    out.write(f"#ASDF {FILE_FORMAT_VERSION}\n#ASDF_STANDARD {STANDARD_VERSION}\n".encode("ascii"))
    out.write(text.encode("utf-8"))
    offsets = []
    for data in writer.blocks:
        offsets.append(out.tell())
        out.write(_block_bytes(data))
    if offsets:
        out.write(INDEX_HEADER + b"\n%YAML 1.1\n---\n" + "".join(f"- {o}\n" for o in offsets).encode("ascii") + b"...\n")
    path.write_bytes(out.getvalue())
    return path


# ---- reading ---------------------------------------------------------------------------------------------------------
class _NdarrayNode:
    def __init__(self, mapping):
        self.mapping = mapping


class _TableNode:
    def __init__(self, mapping):
        self.mapping = mapping


class _ColumnNode:
    def __init__(self, mapping):
        self.mapping = mapping


class _Loader(yaml.SafeLoader):
    pass


def _construct_any(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        mapping = loader.construct_mapping(node, deep=True)
        if suffix.startswith("core/ndarray-"):
            return _NdarrayNode(mapping)
        if suffix.startswith("core/column-"):
            return _ColumnNode(mapping)
        if suffix.startswith("core/table-") or suffix.startswith("astropy/table/table-"):
            return _TableNode(mapping)
        return mapping
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    return loader.construct_scalar(node)


_Loader.add_multi_constructor(TAG_PREFIX, _construct_any)
_Loader.add_multi_constructor("tag:astropy.org:", _construct_any)
_Loader.add_multi_constructor("tag:yaml.org,2002:python/", _construct_any)
_Loader.add_multi_constructor("!", _construct_any)


def _read_blocks(raw, start):
    """Blocks (used bytes, decompressed) in file order from offset ``start`` on."""
    blocks, pos = [], start
    while raw[pos : pos + 4] == BLOCK_MAGIC:
        (header_size,) = struct.unpack(">H", raw[pos + 4 : pos + 6])
        if header_size < BLOCK_HEADER.size:
            raise ValueError(f"ASDF block header of {header_size} bytes at offset {pos} is too short")
        flags, compression, allocated, used, data_size, checksum = BLOCK_HEADER.unpack(raw[pos + 6 : pos + 6 + BLOCK_HEADER.size])
        if flags & 1:
            raise NotImplementedError("streamed ASDF blocks are not supported")
        begin = pos + 6 + header_size
        data = raw[begin : begin + used]
        if len(data) != used:
            raise ValueError(f"ASDF block at offset {pos} is truncated")
        if checksum != b"\0" * 16 and hashlib.md5(data).digest() != checksum:
            raise ValueError(f"checksum mismatch in the ASDF block at offset {pos}")
        compression = compression.rstrip(b"\0")
        if compression == b"zlib":
            data = zlib.decompress(data)
        elif compression == b"bzp2":
            data = bz2.decompress(data)
        elif compression:
            raise NotImplementedError(f"ASDF block compression {compression!r} is not supported")
        if len(data) != data_size:
            raise ValueError(f"ASDF block at offset {pos} holds {len(data)} bytes, its header says {data_size}")
        blocks.append(data)
        pos = begin + allocated
    return blocks


def _resolve(node, blocks):
    if isinstance(node, _NdarrayNode):
        m = node.mapping
        if any(k in m for k in ("strides", "mask")):
            raise NotImplementedError("strided / masked ASDF arrays are not supported")
        dtype = _numpy_dtype(m.get("datatype", "float64"), m.get("byteorder", "little"))
        if "data" in m:
            array = np.array(_resolve(m["data"], blocks), dtype=dtype.newbyteorder("="))
            return array.reshape(m["shape"]) if "shape" in m else array
        source = m.get("source")
        if not isinstance(source, int):
            raise NotImplementedError(f"external ASDF array source {source!r} is not supported")
        if not 0 <= source < len(blocks):
            raise ValueError(f"the ASDF tree refers to block {source}, the file has {len(blocks)}")
        shape = [int(s) for s in m["shape"]]
        count = int(np.prod(shape)) if shape else 1
        array = np.frombuffer(blocks[source], dtype=dtype, count=count, offset=int(m.get("offset", 0)))
        return array.reshape(shape).astype(dtype.newbyteorder("="))
    if isinstance(node, _ColumnNode):
        return {"name": node.mapping["name"], "data": _resolve(node.mapping["data"], blocks)}
    if isinstance(node, _TableNode):
        columns = [_resolve(c, blocks) for c in node.mapping.get("columns", [])]
        by_name = {c["name"]: c["data"] for c in columns}
        names = node.mapping.get("colnames") or [c["name"] for c in columns]
        return Table([(n, by_name[n]) for n in names], meta=_resolve(node.mapping.get("meta") or {}, blocks))
    if isinstance(node, dict):
        return {k: _resolve(v, blocks) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, blocks) for v in node]
    return node


def read_asdf(filename):
    """The tree of an ASDF file as nested dicts / lists with numpy arrays and `Table`s (``asdf_library`` and
    ``history`` are dropped)."""
    raw = Path(filename).read_bytes()
    if not raw.startswith(ASDF_MAGIC):
        raise ValueError(f"{filename} is not an ASDF file (no #ASDF header)")
    end = raw.find(b"\n...\n")
    first_block = raw.find(BLOCK_MAGIC)
    if end < 0 or (0 <= first_block < end):
        raise ValueError(f"{filename}: the YAML tree of the ASDF file does not end with '...'")
    text = raw[: end + 5].decode("utf-8")
    tree = yaml.load(text, Loader=_Loader)  # noqa: S506 (a SafeLoader subclass)
    pos = end + 5
    while raw[pos : pos + 1] in (b"\n", b"\r", b" "):  # (padding between the tree and the first block is allowed)
        pos += 1
    blocks = _read_blocks(raw, pos) if raw[pos : pos + 4] == BLOCK_MAGIC else []
    tree = _resolve(tree or {}, blocks)
    for key in ("asdf_library", "history"):
        tree.pop(key, None)
    return tree
