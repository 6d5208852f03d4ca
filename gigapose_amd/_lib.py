"""ctypes binding of libgigapose_hip.so (C-ABI: include/gigapose_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails this module
raises.  torch must be imported first so the library binds to the HIP runtime torch already
loaded (same SONAME libamdhip64.so.7) and shares its streams and allocations.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL so both use one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgigapose_hip.so")
_lib = None


class GigaPoseHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess

    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout, r.stderr)
    if r.returncode != 0:
        raise GigaPoseHipError("building libgigapose_hip.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GigaPoseHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is deliberately no CPU / PyTorch fallback for the hot path)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.gp_last_error.restype = ctypes.c_char_p
    return _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise GigaPoseHipError("gigapose_amd kernels need tensors on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise GigaPoseHipError("non-contiguous tensor passed to a HIP kernel")
    return ctypes.c_void_p(t.data_ptr())


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise GigaPoseHipError(f"{name} failed (rc={rc}): {lib().gp_last_error().decode()}")


def i(v):
    return ctypes.c_int(int(v))


def f(v):
    return ctypes.c_float(float(v))
