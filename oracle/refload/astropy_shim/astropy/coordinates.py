class SkyCoord:
    def __init__(self, *args, **kwargs):
        pass
