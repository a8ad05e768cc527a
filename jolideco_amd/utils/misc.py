"""Small host helpers (reference: jolideco/utils/misc.py:9-41)."""
from collections.abc import Mapping

__all__ = ["flatten_dict", "unflatten_dict"]


def flatten_dict(data, parent_key="", sep="."):
    """{"a": {"b": 1}} -> {"a.b": 1}"""
    flat = {}
    for key, value in data.items():
        full = f"{parent_key}{sep}{key}" if parent_key else key
        if isinstance(value, Mapping):
            flat.update(flatten_dict(value, full, sep=sep))
        else:
            flat[full] = value
    return flat


def unflatten_dict(data, sep="."):
    """{"a.b": 1} -> {"a": {"b": 1}}"""
    nested = {}
    for key, value in data.items():
        *parents, leaf = key.split(sep)
        node = nested
        for part in parents:
            node = node.setdefault(part, {})
        node[leaf] = value
    return nested
