"""GPU probe: ViT-L forward with 1 vs 2 query tiles per attention wave (bit-identical results expected)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd.vit import Dinov2ViT
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
dev = "cuda"
lib = _lib.lib()
def timeit(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
torch.manual_seed(0)
vit = Dinov2ViT.from_name("dinov2_vitl14")
for p in vit.parameters(): torch.nn.init.normal_(p, std=0.02)
vit = vit.to(dev).set_numerics("split")
x = torch.randn(64, 3, 224, 224, device=dev)
outs = {}
for nq in [1, 0, 1, 0]:
    lib.gp_attention_set_nq(nq)
    ms = timeit(lambda: vit.patch_features(x))
    outs[nq] = vit.patch_features(x)
    print(f"nq={nq}: ViT-L B=64 split forward {ms:.2f} ms")
print("bit-identical:", torch.equal(outs[1], outs[0]))
