class _Unavailable:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("astropy.io.fits is not available in the oracle shim")


Header = HDUList = PrimaryHDU = ImageHDU = BinTableHDU = _Unavailable


def open(*args, **kwargs):  # noqa: A001
    raise NotImplementedError


def getdata(*args, **kwargs):
    raise NotImplementedError
