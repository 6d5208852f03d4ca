"""layernorm_planes_reg_kernel: launch time against the number of 32-token blocks (two blocks are resident per CU = 512 slots; the ViT-L
step at 64 crops has 520).  Shows whether the 8 blocks past one round cost a second round."""
import torch

from gigapose_amd import _lib

dev = torch.device("cuda", 0)
C = 1024
g = torch.randn(C, device=dev)
b = torch.randn(C, device=dev)
print("# blocks  Mpad   us/launch   GB/s (8 C Mpad bytes)")
for blocks in (256, 384, 448, 504, 512, 520, 528, 544, 576, 640, 768, 1024, 1040):
    Mpad = 32 * blocks
    X = torch.randn(C, Mpad, device=dev)
    hi = torch.empty(Mpad, C, dtype=torch.float16, device=dev)
    lo = torch.empty_like(hi)

    def run():
        _lib.call("gp_layernorm_planes", _lib.ptr(X), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(g), _lib.ptr(b), _lib.i(C), _lib.i(Mpad),
                  _lib.f(1e-6), _lib.stream_ptr())

    for _ in range(5):
        run()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    print(f"{blocks:7d} {Mpad:6d} {best:9.2f} {8.0 * C * Mpad / best / 1e3:9.0f}")
_lib.check_status()
