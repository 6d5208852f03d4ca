"""GPU probe: where conv_planes_kernel's time goes on the IST layer shapes at B = 64 -- per slot: tile prologue (gather
state + first slab), k loop, tail (epilogue / hand-over), from the kernel's 100 MHz stamps (gp_conv2d_planes_set_trace);
with and without a residual."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)

dev = "cuda"
if os.environ.get("GP_LIB"):   # a library built with other macros (tools/patches/conv_epi_probe.diff, -DGP_CONV_PROBE=..)
    _lib.LIB_PATH = os.path.abspath(os.environ["GP_LIB"])
lib = _lib.lib()
lib.gp_conv2d_planes_workspace_bytes.restype = ctypes.c_size_t
nb = lib.gp_conv2d_planes_workspace_bytes()
ws = torch.zeros(nb // 4, device=dev)
trace = torch.zeros(256, 8, dtype=torch.int64, device=dev)


def planes(t, scale):
    v = t * scale
    hi = v.half()
    return hi, (v - hi.float()).half()


def run(cin, cout, k, stride, pad, hw, B, res, tall):
    g = torch.Generator(device=dev).manual_seed(1)
    xh, xl = planes(torch.randn(B, hw, hw, cin, device=dev, generator=g), 8.0)
    wh, wl = planes(torch.randn(cout, k * k * cin, device=dev, generator=g) / (cin * k * k) ** 0.5, 64.0)
    oh = (hw + 2 * pad - k) // stride + 1
    rh, rl = planes(torch.randn(B, oh, oh, cout, device=dev, generator=g), 8.0) if res else (None, None)
    o1, o2 = torch.empty(B, oh, oh, cout, dtype=torch.float16, device=dev), torch.empty(B, oh, oh, cout, dtype=torch.float16, device=dev)
    al, be = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)

    def call():
        _lib.call("gp_conv2d_planes", _lib.ptr(xh), _lib.ptr(xl), _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(al), _lib.ptr(be), _lib.ptr(rh), _lib.ptr(rl),
                  _lib.i(B), _lib.i(hw), _lib.i(hw), _lib.i(cin), _lib.i(cout), _lib.i(k), _lib.i(k), _lib.i(stride), _lib.i(pad), _lib.i(1),
                  _lib.ptr(o1), _lib.ptr(o2), None, _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    lib.gp_conv2d_planes_set_trace(ctypes.c_void_p(trace.data_ptr()))
    trace.zero_()
    call()
    torch.cuda.synchronize()
    lib.gp_conv2d_planes_set_trace(None)
    t = trace.cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    seg, steps, pro, loop, tail, life = t[:, 0], t[:, 1], t[:, 2] / 100, t[:, 3] / 100, t[:, 4] / 100, (t[:, 6] - t[:, 5]) / 100
    fl = 2.0 * cout * B * oh * oh * k * k * cin
    mf = (2 if tall or cout < 192 else (3 if cout < 256 else 4))
    nstep_mfma_us = 12 * mf * 32 * 2 / 1.62e3   # matrix cycles per SIMD and k-step (12 NI instructions per wave, 2 waves) at 1.62 GHz, in us
    print(f"{cin:4d}->{cout:4d} k{k} s{stride} {hw:3d}x{hw:<3d} {'tall' if tall else '    '} {us:8.1f} us = {fl / us / 1e6:6.1f} TF-eq | slots {len(t):3d}: lifetime mean {life.mean():7.1f} max {life.max():7.1f} | "
          f"per slot: {seg.mean():5.2f} segments, {steps.mean():6.1f} k-steps | prologue {pro.mean():6.1f} us ({pro.mean() / seg.mean():5.2f} each) | "
          f"k loop {loop.mean():7.1f} us = {loop.mean() / steps.mean():5.3f} us/step (matrix-only {nstep_mfma_us:5.3f}) | tail {tail.mean():6.1f} us ({tail.mean() / seg.mean():5.2f} each)")


# IST ResNet layer shapes at B = 64 (resize to 256 x 256, stem stride 2 -> 128 x 128; reference resnet.py:364-381).  3 x 3 / stride 1 layers
# run conv_halo_kernel unless HALO=0 (then, like the stride-2 layers, conv_planes_kernel: one gather per tap).
lib.gp_conv2d_planes_set_halo(int(os.environ.get("HALO", "1")))
print("# halo kernel for 3 x 3 / stride 1:", os.environ.get("HALO", "1"))
run(128, 128, 3, 1, 1, 128, 64, True, False)
run(128, 128, 3, 1, 1, 128, 64, False, False)
run(128, 192, 3, 2, 1, 128, 64, False, False)
run(192, 192, 3, 1, 1, 64, 64, True, False)
run(192, 256, 3, 2, 1, 64, 64, False, False)
run(256, 256, 3, 1, 1, 32, 64, True, False)
run(256, 512, 3, 2, 1, 32, 64, False, False)
run(512, 512, 3, 1, 1, 16, 64, True, False)
