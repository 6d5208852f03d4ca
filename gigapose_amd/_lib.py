"""ctypes binding of libgigapose_hip.so (C-ABI: include/gigapose_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails this module
raises.  torch must be imported first so the library binds to the HIP runtime torch already
loaded (same SONAME libamdhip64.so.7) and shares its streams and allocations.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL so both use one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GIGAPOSE_LIB") or os.path.join(_HERE, "libgigapose_hip.so")   # GIGAPOSE_LIB: another build of the same sources (A/B of two commits, tools/)
PROBE_LIB_PATH = os.path.join(_HERE, "libgigapose_hip_probe.so")   # the same sources with -DGP_PROBES (include/gigapose_hip_probe.h)
_lib = None          # the library calls go to: the product library unless probe_library() is active
_product = None
_probe = None


class GigaPoseHipError(RuntimeError):
    pass


NUMERICS = ("split", "chain")


def default_numerics():
    """Numerics of the contraction kernels when nothing else is said: "split" -- every f32 operand as two
    f16 planes, 3 x f16 MFMA per k-block, f32 accumulate; f32-class accuracy (error vs float64 at or below a sequential f32
    chain's) at ~3x the speed -- is what the drop-in selects and what bench.py measures.  "chain" (env GIGAPOSE_NUMERICS=chain,
    `numerics: chain` in the model YAML, or model.set_numerics("chain")) is the verification mode: f32-input MFMA, every dot
    product the k-ordered fmaf chain the CPU oracle restates, bit-exact against it."""
    mode = os.environ.get("GIGAPOSE_NUMERICS", "split")
    if mode not in NUMERICS:
        raise ValueError(f"GIGAPOSE_NUMERICS must be one of {NUMERICS}, got {mode!r}")
    return mode


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess

    r = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout, r.stderr)
    if r.returncode != 0:
        raise GigaPoseHipError("building libgigapose_hip.so failed")
    return LIB_PATH


def _load(path):
    if not os.path.exists(path):
        raise GigaPoseHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is deliberately no CPU / PyTorch fallback for the hot path)")
    h = ctypes.CDLL(path)
    h.gp_last_error.restype = ctypes.c_char_p
    return h


def lib():
    global _lib, _product
    if _lib is None:
        if _product is None:
            _product = _load(LIB_PATH)
        _lib = _product
    return _lib


class probe_library:
    """`with _lib.probe_library():` -- the calls of the block go to libgigapose_hip_probe.so: the same kernels built with -DGP_PROBES,
    which adds the A/B switches, traced builds, test-only epilogues and error-word readers of include/gigapose_hip_probe.h.  For tests
    that compare a kernel variant with its default bit for bit, and for tools/.  The product library carries none of those hooks;
    nothing in gigapose_amd/ or bench.py's timed region uses this."""

    def __enter__(self):
        global _lib, _probe
        lib()
        if _probe is None:
            _probe = _load(PROBE_LIB_PATH)
        self._prev, _lib = _lib, _probe
        _reregister()
        return _probe

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        _reregister()
        return False


def use_probe_library():
    """tools/: route every call of this process to libgigapose_hip_probe.so from here on (see probe_library)."""
    probe_library().__enter__()


STATUS_BITS = {1: "a stream-K accumulator hand-over of the split GEMM timed out (features are garbage)",
               2: "a stream-K accumulator hand-over of the f32 GEMM timed out (features are garbage)",
               4: "an activation is not finite, or (split numerics) left the range of the f16 planes (|x| >= 8190 at the default plane scale; GigaPose "
                  "re-calibrates the plane scales / falls back to Dinov2ViT.set_split_gemm('128') by itself, a bare ViT call does not)",
               8: "a detection label / template id lies outside the onboarded bank (the reference raises IndexError)",
               16: "split numerics: an IST activation left the range of the f16 planes (|x| >= 8190) or is not finite -- "
                   "use numerics 'chain' or ResNet.conv_kernel = '128' for this checkpoint (GigaPose falls back by itself)"}
SPLIT_RANGE_BITS = 4 | 16
_status = {}          # device index -> the int32 word on that device
_registered = set()   # device indices whose word the ACTIVE library has in its per-device table


def _reregister():
    """Each library keeps its own per-device table of status words: hand the words this process already owns to the active one."""
    _registered.clear()
    for idx in list(_status):
        status_word(torch.device("cuda", idx))


def status_word(device=None):
    """The device int32 the kernels OR their guard-rail bits into (include/gigapose_hip.h: gp_set_status_buffer), one per GPU.
    The library keeps a per-device table (round 6; it held ONE process-wide pointer before): the word of a device is registered once,
    with that device current, and every launch takes the word of the device it is issued on."""
    dev = torch.device(device if device is not None else "cuda")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _status:
        _status[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    if idx not in _registered:
        with torch.cuda.device(idx):
            call("gp_set_status_buffer", ctypes.c_void_p(_status[idx].data_ptr()))
        _registered.add(idx)
    return _status[idx]


def take_status():
    """Read + clear the status words (a host synchronisation: call where one happens anyway); returns the OR of their bits."""
    bits = 0
    for w in _status.values():
        b = int(w.item())
        if b:
            w.zero_()
        bits |= b
    return bits


def raise_status(bits):
    if bits:
        msgs = [m for b, m in STATUS_BITS.items() if bits & b]
        raise GigaPoseHipError("device status 0x%x: %s" % (bits, "; ".join(msgs)))


def check_status():
    """Read + clear the status word and raise if a bit is set."""
    raise_status(take_status())


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
