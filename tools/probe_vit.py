"""GPU probe: time the HIP ViT forward (ViT-L/14, B=64) and a few GEMM shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd.vit import Dinov2ViT
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)

dev = "cuda"
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

import ctypes
lib = _lib.lib()
lib.gp_gemm_streamk_workspace_bytes.restype = ctypes.c_size_t
nbytes = lib.gp_gemm_streamk_workspace_bytes()
ws = torch.zeros(nbytes // 4, device=dev)
for mode in []:
    tot = 0.0
    for (I, J, K, epi) in [(1024, 16512, 1024, 3), (4096, 16512, 1024, 2), (1024, 16512, 4096, 3), (2048, 16512, 1024, 1), (16512, 1024, 1024, 4)]:
        A = torch.randn(K, I, device=dev); Bm = torch.randn(K, J, device=dev); D = torch.randn(I, J, device=dev)
        bias = torch.randn(max(I, J), device=dev); sc = torch.randn(I, device=dev)
        if mode == "plain":
            fn = lambda: _lib.call("gp_gemm_kmajor", _lib.ptr(A), _lib.i(I), _lib.ptr(Bm), _lib.i(J), _lib.ptr(D), _lib.i(J),
                                   _lib.i(I), _lib.i(J), _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(D), _lib.i(J), _lib.stream_ptr())
        else:
            fn = lambda: _lib.call("gp_gemm_kmajor_sk", _lib.ptr(A), _lib.i(I), _lib.ptr(Bm), _lib.i(J), _lib.ptr(D), _lib.i(J),
                                   _lib.i(I), _lib.i(J), _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(D), _lib.i(J),
                                   _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.stream_ptr())
        ms = timeit(fn)
        tot += ms
        print(f"{mode} gemm I={I} J={J} K={K} epi={epi}: {ms:.3f} ms {2.0*I*J*K/ms/1e9:.1f} TF")
    print(f"{mode} layer total {tot:.3f} ms -> {415.5/tot:.1f} TF  err={lib.gp_gemm_streamk_error(_lib.ptr(ws), _lib.stream_ptr())}")

name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitl14"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
vit = Dinov2ViT.from_name(name)
for p in vit.parameters():
    torch.nn.init.normal_(p, std=0.02)
vit = vit.to(dev)
x = torch.randn(B, 3, 224, 224, device=dev)
fl = {"dinov2_vitl14": 162.0e9, "dinov2_vits14": 12.25e9, "dinov2_vitb14": 0}[name] * B
outs = {}
for mode in ["chain", "split"]:
    vit.set_numerics(mode)
    ms = timeit(lambda: vit.patch_features(x), iters=3, warm=1)
    outs[mode] = vit.patch_features(x)
    print(f"{name} B={B} numerics={mode}: {ms:.2f} ms/forward -> {B/ms*1e3:.1f} crops/s, {fl/ms/1e9:.1f} TFLOP/s-equivalent")
d = (outs["chain"] - outs["split"]).abs()
print(f"unit-norm patch features chain vs split: max |diff| {d.max().item():.3e}, rms {d.pow(2).mean().sqrt().item():.3e}")
