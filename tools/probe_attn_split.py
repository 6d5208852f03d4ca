"""GPU probe: split-numerics attention (attention_split_kernel) isolated time at ViT-L B=64; f16-subnormal behaviour of the
plane GEMM's MFMA inputs."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
dev = "cuda"
lib = _lib.lib()
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H = 64, 16
C = 64 * H
M = B * 257
Mpad = (M + 255) // 256 * 256
qkv = torch.randn(Mpad, 3 * C, device=dev)
hi = torch.empty(Mpad, 3 * C, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
_lib.call("gp_split_planes", _lib.ptr(qkv), ctypes.c_size_t(qkv.numel()), _lib.f(8.0), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
ohi = torch.zeros(Mpad, C, dtype=torch.float16, device=dev); olo = torch.zeros_like(ohi)
run = lambda: _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad), _lib.f(8.0), _lib.stream_ptr())
ms = timeit(run)
fl = 4.0 * B * H * 257 * 257 * 64
print(f"attention_split_kernel B={B} H={H}: {ms*1e3:.1f} us isolated = {fl/ms/1e9:.1f} TF-equivalent (f32 kernel: 286 us in the bench)")
if "--plain" in sys.argv:   # PMC passes: only the product kernel
    sys.exit(0)
from test_gpu_split import planes256_gemm
I, J, K = 4096, 4096, 64
for mag in (1e-3, 3e-6, 3e-7):
    A = torch.full((I, K), mag, device=dev); Bm = torch.ones(J, K, device=dev)
    D = planes256_gemm(A, Bm, 0)
    print(f"plane GEMM, A = {mag:g} (x64 = {mag*64:g}; f16 min normal 6.1e-5): D[0,0] = {D[0,0].item():.6e}, exact {mag*K:.6e}")
