"""ViT-L forward in a loop (for rocprofv3 --kernel-trace): python tools/probe_vit_loop.py [B] [iters].  Env: GIGAPOSE_LN_FOLD, GIGAPOSE_PLANES_DP
(bits 2..5 = probe mask of the folded producer epilogue: results are then garbage, only the timing is of interest)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapose_amd import _lib, factory  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
model = factory.build_model("dinov2_vitl14", k=5, device="cuda", seed=0, numerics="split")
vit = model.ae_net.dinov2_model
q = factory.TemplateSet(1, 8, seed=100).crops(9, B, "cuda")
for _ in range(2):
    vit.patch_features(q["tar_img"])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    vit.patch_features(q["tar_img"])
e1.record()
torch.cuda.synchronize()
print(f"ViT-L forward, {B} crops: {e0.elapsed_time(e1) / iters:.3f} ms  (status bits {_lib.take_status()})")
