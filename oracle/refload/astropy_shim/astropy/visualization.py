def simple_norm(*args, **kwargs):
    return None
