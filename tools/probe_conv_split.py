"""GPU probe: IST backbone chain vs split numerics (time), layer-1-shaped split conv alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_testing import factory
dev = "cuda"
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
model = factory.build_model("dinov2_vits14", k=4, device=dev, seed=3)
x = torch.randn(64, 3, 224, 224, device=dev)
net = model.ist_net.backbone
for mode in ["chain", "split", "chain", "split"]:
    net.set_numerics(mode)
    ms = timeit(lambda: net(x))
    print(f"IST backbone B=64 numerics={mode}: {ms:.2f} ms -> {39.1*64/ms:.1f} TF-equivalent")
