from functools import cached_property as lazyproperty  # noqa: F401
