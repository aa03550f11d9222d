"""CPU-side checks: the shared library builds for gfx950, loads, and exports every symbol include/vca_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_declared_symbols():
    import __graft_entry__ as ge
    ge.build()
    from vcagan_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'vca_hip.h')).read()
    declared = set(re.findall(r'\b(?:int|long long)\s+(vca_\w+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.exported_symbols())
    assert L.vca_abi_version() >= 1
