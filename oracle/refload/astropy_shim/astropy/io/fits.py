"""Recording stand-in for astropy.io.fits (oracle loader only, TEST INFRASTRUCTURE).

python3.10 in the build image has PyTorch but no astropy, /opt/conda's python3.9 has astropy but no
PyTorch.  To capture what the reference's FITS writers (jolideco/utils/io/fits.py) ask astropy to
write, these classes only RECORD the HDUs -- kind, name, header cards in order, image array or table
columns -- and `HDUList.writeto` dumps that description as an ``.npz``.  oracle/refload/hdus_to_fits.py
then replays the description through the real astropy under python3.9 to produce the golden FITS files.
Reading is not provided.
"""
import builtins
import json

import numpy as np


class Header(dict):
    """Ordered keyword -> value mapping."""


class _HDU:
    kind = None

    def __init__(self, data=None, header=None, name=None):
        self.data = data
        self.header = Header(header or {})
        self.name = name or ""


class PrimaryHDU(_HDU):
    kind = "primary"


class ImageHDU(_HDU):
    kind = "image"


class BinTableHDU(_HDU):
    kind = "bintable"


def _table_columns(table):
    return {name: np.asarray(table[name]) for name in table.colnames}


class HDUList(list):
    def writeto(self, filename, overwrite=False):
        arrays, layout = {}, []
        for i, hdu in enumerate(self):
            entry = {"kind": hdu.kind, "name": hdu.name, "header": [[k, _plain(v)] for k, v in hdu.header.items()]}
            if hdu.kind == "bintable":
                entry["columns"] = []
                for name, values in _table_columns(hdu.data).items():
                    arrays[f"hdu{i}/{name}"] = values
                    entry["columns"].append(name)
            elif hdu.data is not None:
                arrays[f"hdu{i}/data"] = np.asarray(hdu.data)
                entry["data"] = True
            layout.append(entry)
        arrays["layout"] = np.array(json.dumps(layout))
        with builtins.open(filename, "wb") as f:
            np.savez(f, **arrays)


def _plain(value):
    return value.item() if isinstance(value, np.generic) else value


def open(*args, **kwargs):  # noqa: A001
    raise NotImplementedError("the recording stand-in cannot read FITS files")


def getdata(*args, **kwargs):
    raise NotImplementedError("the recording stand-in cannot read FITS files")
