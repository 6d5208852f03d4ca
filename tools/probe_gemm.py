"""GPU probe: f32 MFMA GEMM tile-shape variants (gp_gemm_probe) on the ViT-L shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)

dev = "cuda"
lib = _lib.lib()
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
names = {0: "128x128 KS16 4w", 1: "128x128 KS32 4w", 2: "128x256 KS16 8w", 3: "256x128 KS16 8w", 4: "256x256 KS16 8w",
         5: "256x128 KS16 4w", 6: "128x256 KS16 4w", 7: "128x128 KS8 4w", 8: "128x128 KS16 4w prefetch", 9: "128x128 KS16 8w", 10: "128x128 KS8 4w prefetch", 11: "256x256 8w prefetch"}
J = 16640 - 256  # 128 * 128: no ragged tail, isolates the main loop
for (I, K) in [(1024, 1024), (2048, 1024), (4096, 1024), (1024, 4096)]:
    A = torch.randn(K, I, device=dev); Bm = torch.randn(K, J, device=dev); D = torch.empty(I, J, device=dev)
    ref = None
    for v in [0, 7, 8, 9, 10, 4, 11]:
        rc = lib.gp_gemm_probe(v, _lib.ptr(A), I, _lib.ptr(Bm), J, _lib.ptr(D), J, I, J, K, _lib.stream_ptr())
        if rc != 0:
            print(f"I={I} K={K} variant {v} ({names[v]}): n/a"); continue
        ms = timeit(lambda: lib.gp_gemm_probe(v, _lib.ptr(A), I, _lib.ptr(Bm), J, _lib.ptr(D), J, I, J, K, _lib.stream_ptr()))
        if ref is None: ref = D.clone()
        ok = torch.equal(ref, D)
        print(f"I={I} J={J} K={K} variant {v} ({names[v]}): {ms:.3f} ms {2.0*I*J*K/ms/1e9:.1f} TF bit-identical={ok}")
