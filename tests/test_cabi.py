"""CPU: libgigapose_hip.so loads (without a GPU) and exports every symbol include/*.h declares."""
import ctypes
import os
import re

from gigapose_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert lib.gp_abi_version() >= 1


def test_argument_validation_needs_no_gpu():
    lib = _lib.lib()
    lib.gp_last_error.restype = ctypes.c_char_p
    rc = lib.gp_topk(None, 1, 3, 5, None, None, None)
    assert rc == -1 and b"gp_topk" in lib.gp_last_error()
