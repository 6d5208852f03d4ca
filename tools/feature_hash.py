"""sha256 of everything GigaPose.predict returns for a seeded batch (ViT-L, 162 templates, split numerics) -- run it with two libraries
(GIGAPOSE_LIB=...) to check that a kernel change which must not move a bit did not:   python tools/feature_hash.py [n_crops ...]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gigapose_amd import _lib
from gigapose_testing import factory

dev = torch.device("cuda", 0)
tset = factory.TemplateSet(1, 162, seed=70)
model = factory.build_model("dinov2_vitl14", k=5, device=dev, seed=5)
model.set_numerics("split")
model.template_datasets = {"syn": tset}
model.set_template_data("syn")
for n in [int(v) for v in sys.argv[1:]] or [64, 8, 24]:
    q = tset.crops(71, n, dev)
    p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    torch.cuda.synchronize()
    _lib.check_status()
    h = hashlib.sha256()
    for name in sorted(p.tensors):
        h.update(p.tensors[name].cpu().contiguous().numpy().tobytes())
    feat = model.ae_net(q["tar_img"])
    h2 = hashlib.sha256(feat.cpu().contiguous().numpy().tobytes())
    print(f"{n:3d} crops: predict {h.hexdigest()[:16]}  ViT features {h2.hexdigest()[:16]}")
