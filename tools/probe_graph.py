"""GPU probe: does replaying the ViT forward (one C call = ~175 kernel launches) from a HIP graph shrink the inter-kernel gaps?
Times vit.patch_features eagerly and as a captured graph (torch.cuda.CUDAGraph = hipGraph on ROCm)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gigapose_amd import _lib

from gigapose_testing import factory

dev = "cuda"
model = factory.build_model("dinov2_vitl14", k=5, device=dev, seed=2)
vit = model.ae_net.dinov2_model
vit.set_numerics("split")
x = torch.randn(64, 3, 224, 224, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_eager = timed(lambda: vit.patch_features(x))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        vit.patch_features(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = vit.patch_features(x)
t_graph = timed(g.replay)
ref = vit.patch_features(x)
g.replay()
torch.cuda.synchronize()
print(f"ViT-L forward, B = 64: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms; outputs equal {torch.equal(ref, out)}")
_lib.check_status()
