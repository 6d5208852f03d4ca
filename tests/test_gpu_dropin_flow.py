"""GPU: the flow of the reference's test.py (test.py:43-79) without Hydra / Lightning (absent from the image), driven by the
reference's OWN model config: tests/golden/model_cfg_resolved.json is configs/model/large.yaml + its defaults with the `_target_`s
swapped as INTEGRATION.md section 1 says (written and kept current by tests/test_dropin_hydra_flow.py in the build container).
instantiate(cfg.model) -> attributes test.py assigns -> the loop Trainer.test runs (test_step per image, on_test_epoch_end) ->
the BOP csv files; the poses in the csv are the ones a direct predict() returns."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

import dropin_flow as df
from gigapose_amd import factory
from gigapose_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def image_batch(tset, seed, n, view_id):
    """One image's detections, as GigaPoseTestSet.collate_fn hands them over (dataloader/test.py:308-315)."""
    from gigapose_amd.tensor_collection import PandasTensorCollection

    q = tset.crops(seed, n, DEV)
    labels = q["labels"].numpy()
    infos = pd.DataFrame(dict(label=[str(l) for l in labels], scene_id=[2] * n, view_id=[view_id] * n))
    batch = PandasTensorCollection(infos=infos, **{k: q[k] for k in ["tar_img", "tar_mask", "tar_K", "tar_M"]})
    objs = sorted(set(int(l) for l in labels))
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(
        im_id=[view_id] * len(objs), scene_id=[2] * len(objs), obj_id=objs,
        inst_count=[int((labels == o).sum()) for o in objs], detection_time=[0.05] * len(objs))))
    return batch, q


def test_reference_test_py_flow_from_the_reference_model_config(tmp_path):
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_cfg_resolved.json")))
    cfg["log_dir"] = str(tmp_path)
    cfg["test_setting"] = "localization"                                  # test.py:46
    model = df.instantiate(cfg).to(DEV)                                   # test.py:47  instantiate(cfg.model)
    syn.fill_state_dict(model.ae_net.dinov2_model, 11)                    # no network for gigaPose_v1.ckpt: deterministic random weights
    syn.fill_state_dict(model.ist_net, 12)
    model.set_numerics("split")
    tset = factory.TemplateSet(2, 12, seed=90)
    model.template_datasets = {"syn": tset}                               # test.py:67-74
    model.test_dataset_name = "syn"
    model.max_num_dets_per_forward = 4
    model.run_id = "r0"
    model.log_interval = 1
    batches = [image_batch(tset, 91, 5, view_id=3), image_batch(tset, 92, 9, view_id=4)]
    df.trainer_test(model, [b for b, _ in batches])                       # test.py:77-79  trainer.test(model, dataloaders=...)
    pred_dir = os.path.join(str(tmp_path), "predictions")
    csvs = sorted(f for f in os.listdir(pred_dir) if f.endswith(".csv"))
    assert csvs == ["large-pbrreal-rgb-mmodel_syn-test_r0.csv", "large-pbrreal-rgb-mmodel_syn-test_r0MultiHypothesis.csv"]
    top1 = pd.read_csv(os.path.join(pred_dir, csvs[0]))
    multi = pd.read_csv(os.path.join(pred_dir, csvs[1]))
    assert len(top1) == 5 + 9 and len(multi) == (5 + 9) * 5 and set(top1.im_id) == {3, 4}
    # the csv carries the poses of a direct predict() on the same detections (localization mode keeps all of them here)
    row = 0
    for (batch, q), n in zip(batches, (5, 9)):
        p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
        poses, labels = p.pred_poses[:, 0].cpu().numpy(), q["labels"].numpy()
        sel = []
        for o in sorted(set(int(l) for l in labels)):                     # filter_and_save groups by object id (gigaPose.py:408-425)
            cand = np.flatnonzero(labels == o)
            sel += cand[np.argsort(-p.scores[cand, 0].cpu().numpy(), kind="stable")].tolist()
        for d in sel:
            t = np.array(top1.t[row].split(), dtype=np.float64)
            R = np.array(top1.R[row].split(), dtype=np.float64).reshape(3, 3)
            np.testing.assert_allclose(t, poses[d][:3, 3], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(R, poses[d][:3, :3], rtol=1e-6, atol=1e-6)
            assert int(top1.obj_id[row]) == int(labels[d])
            row += 1
    assert row == len(top1)
