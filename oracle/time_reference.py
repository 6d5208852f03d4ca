"""TEST / BENCH INFRASTRUCTURE ONLY (build container; imports the unmodified reference through oracle/ref_shim.py).

Anchors bench.py's `cpu_baseline` (kind "port": oracle/torch_port.py, the reference's operators restated in torch -- the Python
reference cannot travel to the GPU box) to the reference's own wall clock: the UNMODIFIED `GigaPose.eval_retrieval`
(reference src/models/gigaPose.py:481-633, onboarding excluded as the reference excludes it, gigaPose.py:396-398) and the port, on the
same 32 crops x 162 templates, ViT-L/14 stand-in, the same thread count, in the same process.

    python oracle/time_reference.py [threads] [crops]      ->  profiles/r03_cpu_reference_vs_port.txt (append by hand)
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_goldens as mg, ref_shim, torch_port  # noqa: E402
from gigapose_testing import synthetic as syn  # noqa: E402


def main(threads=8, n_crops=32, n_templates=162):
    ref_shim.install()
    torch.set_num_threads(threads)
    import pandas as pd
    from src.megapose.utils.tensor_collection import PandasTensorCollection
    from src.models.gigaPose import GigaPose
    from src.models.matching import LocalSimilarity
    from src.models.network.ae_net import AENet

    backbone = ref_shim.HFDinov2Backbone.build(1024, 24, 16, seed=0)
    syn.fill_state_dict(backbone.m, 302)
    ist = mg.build_ref_ist(seed=303, conditioned=True)
    model = GigaPose("large", AENet("dinov2_vitl14", backbone, 1024, 64), ist, None, LocalSimilarity(k=5, sim_threshold=0.5, patch_threshold=3),
                     None, 1000, tempfile.mkdtemp(), max_num_dets_per_forward=4).eval()
    items, q = mg.e2e_inputs(311, 1, n_templates, n_crops)
    model.template_datasets = {"syn": mg._FakeTemplates(items)}
    t0 = time.time()
    with torch.no_grad():
        model.set_template_data("syn")
    t_onboard = time.time() - t0
    infos = pd.DataFrame(dict(label=[str(l) for l in q["labels"]], scene_id=[1] * n_crops, view_id=[7] * n_crops))
    batch = PandasTensorCollection(infos=infos, **{n: torch.from_numpy(q[n]) for n in ["tar_img", "tar_mask", "tar_K", "tar_M"]})
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(im_id=[7], scene_id=[1], obj_id=[1], inst_count=[n_crops], detection_time=[0.1])))
    t0 = time.time()
    with torch.no_grad():
        model.eval_retrieval(batch, 0, "syn")
    t_ref = time.time() - t0
    # the port on the same crops, fed the reference's own onboarded banks
    td = model.template_datas["syn"]
    crops = {n: torch.from_numpy(q[n]) for n in ["tar_img", "tar_mask", "tar_K", "tar_M"]}
    crops["labels"] = torch.from_numpy(q["labels"])
    geom = (td.K.numpy(), td.M.numpy(), td.poses.numpy())
    from gigapose_testing import factory
    ist_port = factory.build_model("dinov2_vits14", k=5, device="cpu", seed=0).ist_net      # reference-shaped ISTNet mirror (same operators)
    t0 = time.time()
    torch_port.eval_retrieval(backbone.m, ist_port, td.ae_features, td.ist_features, td.mask, geom, crops, 5, dets_per_forward=4)
    t_port = time.time() - t0
    print(f"threads {threads}, {n_crops} crops x {n_templates} templates, ViT-L/14 stand-in, f32, onboarding excluded ({t_onboard:.0f} s):")
    print(f"  unmodified reference GigaPose.eval_retrieval  {t_ref:7.1f} s = {n_crops / t_ref:.3f} crops/s")
    print(f"  oracle/torch_port.eval_retrieval (cpu_baseline) {t_port:7.1f} s = {n_crops / t_port:.3f} crops/s   (port / reference = {t_ref / t_port:.2f} x)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 32)
