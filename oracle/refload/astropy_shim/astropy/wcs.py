class WCS:
    def __init__(self, *args, **kwargs):
        pass
