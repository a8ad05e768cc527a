"""Replay recorded HDU descriptions through the REAL astropy (run with an interpreter that has it:
``/opt/conda/bin/python3.9 oracle/refload/hdus_to_fits.py``).  TEST INFRASTRUCTURE, step 2 of
make_golden_fits.py: the objects are rebuilt with the same astropy calls the reference makes
(`fits.PrimaryHDU()`, `fits.ImageHDU(header=, data=, name=)`, `fits.BinTableHDU(Table, name=)`,
`HDUList.writeto`).

With ``--check FILE.fits`` it instead opens a FITS file with astropy, verifies it against the standard
and prints its structure as JSON (used by tests/test_io_fits.py to check files written by
jolideco_amd)."""
import json
import sys
import warnings
from pathlib import Path

warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402

# astropy 4.3 predates numpy 1.24: give it back the aliases it still imports
for _name, _value in {"asscalar": lambda a: a.item(), "alen": len, "float": float, "int": int, "bool": bool,
                      "object": object, "complex": complex, "str": str}.items():
    if _name not in np.__dict__:
        setattr(np, _name, _value)

from astropy.io import fits  # noqa: E402
from astropy.table import Table  # noqa: E402


def replay(path):
    data = np.load(path)
    layout = json.loads(str(data["layout"]))
    hdulist = fits.HDUList()
    for i, entry in enumerate(layout):
        header = fits.Header()
        for key, value in entry["header"]:
            header[key] = value
        name = entry["name"] or None
        if entry["kind"] == "primary":
            hdu = fits.PrimaryHDU()
        elif entry["kind"] == "image":
            hdu = fits.ImageHDU(header=header, data=data[f"hdu{i}/data"], name=name)
        else:
            table = Table()
            for column in entry["columns"]:
                table[column] = data[f"hdu{i}/{column}"]
            hdu = fits.BinTableHDU(table, header=header if len(header) else None, name=name)
        hdulist.append(hdu)
    target = path.with_name(path.name.replace(".hdus.npz", ".fits"))
    hdulist.writeto(target, overwrite=True)
    print("wrote", target)


def check(path):
    report = []
    with fits.open(path) as hdulist:
        hdulist.verify("exception")
        for hdu in hdulist:
            entry = {"name": hdu.name, "type": type(hdu).__name__,
                     "header": {k: v for k, v in hdu.header.items() if k not in ("COMMENT", "HISTORY", "")}}
            if isinstance(hdu, fits.BinTableHDU):
                table = Table.read(hdu)
                entry["columns"] = {name: [str(table[name].dtype), list(table[name].shape)] for name in table.colnames}
                entry["rows"] = [[_plain(v) for v in row] for row in table]
            elif hdu.data is not None:
                entry["shape"] = list(hdu.data.shape)
                entry["dtype"] = str(hdu.data.dtype)
                entry["sum"] = float(hdu.data.astype(np.float64).sum())
            report.append(entry)
    print(json.dumps(report))


def _plain(value):
    if isinstance(value, bytes):
        return value.decode()
    if isinstance(value, np.ndarray):
        return value.tolist()
    return value.item() if isinstance(value, np.generic) else value


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--check":
        check(sys.argv[2])
    else:
        for hdus in sorted((Path(__file__).resolve().parent.parent.parent / "tests" / "golden" / "io").glob("*.hdus.npz")):
            replay(hdus)
