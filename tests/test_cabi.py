"""CPU: the two libraries load (without a GPU) and export exactly what their headers declare.

include/gigapose_hip.h        <->  gigapose_amd/libgigapose_hip.so         (the product: at most 45 entry points, no A/B switches)
include/gigapose_hip_probe.h  <->  gigapose_amd/libgigapose_hip_probe.so   (the same sources with -DGP_PROBES: + hooks / traced builds)"""
import ctypes
import os
import re
import subprocess

from gigapose_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", src)))


def exported_symbols(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if " T gp_" in ln})


def test_product_library_exports_exactly_the_product_header():
    lib = _lib.lib()
    names = declared_symbols("gigapose_hip.h")
    assert 7 <= len(names) <= 45, f"{len(names)} entry points in the product header (the bar is 45)"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gigapose_hip.h but not exported"
    assert exported_symbols(_lib.LIB_PATH) == names, "the product library exports something its header does not declare (or the reverse)"
    assert not [n for n in names if "_set_" in n and n != "gp_set_status_buffer"], "an A/B switch in the product interface"
    assert lib.gp_abi_version() >= 2


def test_probe_library_adds_the_probe_header_and_nothing_undeclared():
    product, probe = declared_symbols("gigapose_hip.h"), declared_symbols("gigapose_hip_probe.h")
    assert not set(product) & set(probe)
    assert exported_symbols(_lib.PROBE_LIB_PATH) == sorted(product + probe)
    with _lib.probe_library() as lib:
        for n in probe:
            assert hasattr(lib, n)
        assert _lib.lib() is lib
    assert _lib.lib() is not lib


def test_argument_validation_needs_no_gpu():
    lib = _lib.lib()
    lib.gp_last_error.restype = ctypes.c_char_p
    rc = lib.gp_topk(None, 1, 3, 5, None, None, None)
    assert rc == -1 and b"gp_topk" in lib.gp_last_error()
