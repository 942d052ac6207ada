"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/star_hip.h declares (no compute call is made without a GPU)."""
import ctypes
import os
import re

from star_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "star_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(star_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libstar_hip.so does not export " + n


def test_python_binding_covers_the_header():
    assert sorted(_lib.EXPORTS) == _declared()
    lib = _lib.lib()
    assert lib.star_version() >= 100
    assert isinstance(lib.star_last_error(), bytes)


def test_argument_errors_are_reported_not_raised_across_the_abi():
    lib = _lib.lib()
    # null pointers are rejected before any launch, with a message
    rc = lib.star_mask_sort(None, None, None, 1, 16, None)
    assert rc != 0 and b"star_mask_sort" in lib.star_last_error()
    g = _lib.StarGemm()
    rc = lib.star_gemm(ctypes.byref(g), None)
    assert rc != 0 and b"star_gemm" in lib.star_last_error()
