"""pytest plugin (-p conftest_plugin) that loads the reference through load_ref before collection."""
from load_ref import load_reference

load_reference()
