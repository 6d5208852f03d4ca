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
from gigapose_testing import factory
from gigapose_testing import synthetic as syn

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


_MODELS = {}


def build_from_reference_cfg(log_dir, numerics, accumulate):
    """instantiate(cfg.model) from the reference's own (resolved, target-swapped) config.  One ViT-L instance per numerics for the
    whole module (303 M random parameters each); a later call only re-points log_dir and the accumulation key."""
    if numerics not in _MODELS:
        cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_cfg_resolved.json")))
        cfg["log_dir"] = str(log_dir)
        cfg["test_setting"] = "localization"                              # test.py:46
        cfg["accumulate_crops"] = accumulate                              # optional key next to `numerics`
        model = df.instantiate(cfg).to(DEV)                               # test.py:47  instantiate(cfg.model)
        assert model.accumulate_crops == accumulate
        syn.fill_state_dict(model.ae_net.dinov2_model, 11)                # no network for gigaPose_v1.ckpt: deterministic random weights
        syn.condition_ist(syn.fill_state_dict(model.ist_net, 12))         # the regime a trained ISTNet works in (synthetic.condition_ist)
        model.set_numerics(numerics)
        _MODELS[numerics] = model
    model = _MODELS[numerics]
    model.log_dir, model.accumulate_crops = str(log_dir), accumulate
    os.makedirs(os.path.join(model.log_dir, "predictions"), exist_ok=True)
    return model


@pytest.mark.parametrize("numerics,accumulate", [("split", 64), ("split", 0), ("chain", 64), ("chain", 0)])
def test_reference_test_py_flow_from_the_reference_model_config(tmp_path, numerics, accumulate):
    """Both numerics (the product default `split` AND the verification mode), with the cross-image accumulation of test_step on (the
    default: 64) and off (the reference's one-predict-per-image flow)."""
    model = build_from_reference_cfg(tmp_path, numerics, accumulate)
    assert model.accumulate_crops == accumulate
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
    if accumulate:   # one predict over both images (5 + 9 < 64 crops: flushed by on_test_epoch_end) -- the same batch composition
        cat = {k: torch.cat([q[k] for _, q in batches]) for k in ["tar_img", "tar_mask", "tar_K", "tar_M", "labels"]}
        p_all = model.predict(cat["tar_img"], cat["tar_mask"], cat["tar_K"], cat["tar_M"], cat["labels"], "syn")
    first = 0
    for (batch, q), n in zip(batches, (5, 9)):
        if accumulate:
            import types

            p = types.SimpleNamespace(pred_poses=p_all.pred_poses[first:first + n], scores=p_all.scores[first:first + n])
        else:
            p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
        first += n
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


def _run_flow(log_dir, numerics, accumulate, sizes, n_obj=2, n_tmpl=12):
    model = build_from_reference_cfg(log_dir, numerics, accumulate)
    tset = factory.TemplateSet(n_obj, n_tmpl, seed=90)
    model.template_datasets = {"syn": tset}
    model.test_dataset_name = "syn"
    model.max_num_dets_per_forward = 4
    model.run_id = "r0"
    batches = [image_batch(tset, 300 + i, n, view_id=20 + i)[0] for i, n in enumerate(sizes)]
    df.trainer_test(model, batches)
    pred_dir = os.path.join(str(log_dir), "predictions")
    files = {}
    for i in range(len(sizes)):
        with np.load(os.path.join(pred_dir, f"{i}.npz")) as z:
            files[i] = {k: z[k] for k in z.files}
    csvs = {f: pd.read_csv(os.path.join(pred_dir, f)) for f in sorted(os.listdir(pred_dir)) if f.endswith(".csv")}
    return files, csvs


SIZES = [5, 9, 3, 7, 12, 4, 8]


def test_accumulated_test_steps_write_the_files_of_the_per_image_flow_chain(tmp_path):
    """Crops are independent until filter_and_save (reference gigaPose.py:408-425), and in `chain` numerics every dot product is
    a fixed fmaf chain whatever the batch: the npz files and the BOP csv written by the accumulated flow (whole images, one predict
    per >= 16 pending crops here) equal the per-image flow's BYTE FOR BYTE except the `time` field / column (the flush's device time
    apportioned by crop count instead of the image's own wall time)."""
    a_files, a_csv = _run_flow(tmp_path / "per_image", "chain", 0, SIZES)
    b_files, b_csv = _run_flow(tmp_path / "accumulated", "chain", 16, SIZES)
    for i, n in enumerate(SIZES):
        assert sorted(a_files[i]) == sorted(b_files[i])
        for key in a_files[i]:
            assert a_files[i][key].dtype == b_files[i][key].dtype and a_files[i][key].shape == b_files[i][key].shape
            if key != "time":
                assert a_files[i][key].tobytes() == b_files[i][key].tobytes(), f"image {i}: {key}"
        assert (b_files[i]["time"] > 0).all() and len(set(b_files[i]["time"].tolist())) == 1
    assert list(a_csv) == list(b_csv) and len(a_csv) == 2
    for name in a_csv:
        ca, cb = a_csv[name], b_csv[name]
        assert list(ca.columns) == list(cb.columns) and len(ca) == len(cb)
        for col in ca.columns:
            if col != "time":
                assert ca[col].tolist() == cb[col].tolist(), f"{name}: column {col}"


def test_accumulated_test_steps_in_split_numerics(tmp_path):
    """`split` (the product default): a crop's ViT round-off depends on the GEMM partition its batch selects (tile / ragged strip /
    parallel split-K), as any GPU BLAS does -- the accumulated flow's files agree with the per-image flow's within f32 round-off:
    same detections kept, inlier scores equal for nearly all hypotheses, poses within 1e-4 where the score is."""
    a_files, _ = _run_flow(tmp_path / "per_image", "split", 0, SIZES)
    b_files, _ = _run_flow(tmp_path / "accumulated", "split", 64, SIZES)
    same = total = 0
    for i in range(len(SIZES)):
        for key in ("scene_id", "im_id", "object_id", "detection_time"):
            np.testing.assert_array_equal(a_files[i][key], b_files[i][key])
        eq = a_files[i]["scores"] == b_files[i]["scores"]
        same += int(eq.sum())
        total += eq.size
        pa, pb = a_files[i]["poses"][eq], b_files[i]["poses"][eq]
        close = np.abs(pa - pb).reshape(len(pa), -1).max(1) <= 1e-4 * (1 + np.abs(pa).reshape(len(pa), -1).max(1))
        assert close.mean() >= 0.9, f"image {i}: {int((~close).sum())} of {len(close)} equal-score hypotheses moved (a RANSAC tie at 14.000 px moves a few)"
    assert same >= 0.9 * total, f"{same} / {total} hypothesis scores equal"


def test_accumulated_flow_edge_cases_empty_image_and_oversized_image(tmp_path):
    """An image without detections (its npz is written, empty), an image with more crops than the threshold (one flush of its own, chunked
    by AENet's max_batch_size inside predict), a flush that ends exactly on the threshold, and flush_pending() called twice."""
    model = build_from_reference_cfg(tmp_path, "chain", 16)
    tset = factory.TemplateSet(2, 12, seed=90)
    model.template_datasets = {"syn": tset}
    model.test_dataset_name = "syn"
    model.run_id = "r0"
    sizes = [5, 0, 11, 70, 16, 3]
    batches = []
    for i, n in enumerate(sizes):
        b, _ = image_batch(tset, 400 + i, max(n, 1), view_id=30 + i)
        if n == 0:   # no detection in this image: empty tensors, empty infos, empty test list
            b = b[[]]
            b.test_list = b.test_list[[]] if hasattr(b, "test_list") else None
            from gigapose_amd.tensor_collection import PandasTensorCollection

            b.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(im_id=[], scene_id=[], obj_id=[], inst_count=[], detection_time=[])))
        batches.append(b)
    for i, b in enumerate(batches):
        assert model.test_step(b, i) == 0
    model.flush_pending()
    model.flush_pending()
    assert model._pending == [] and model._in_flight is None and model._pending_crops == 0
    pred_dir = os.path.join(str(tmp_path), "predictions")
    for i, n in enumerate(sizes):
        with np.load(os.path.join(pred_dir, f"{i}.npz")) as z:
            assert z["poses"].shape == (n, 5, 4, 4) and z["scores"].shape == (n, 5) and len(z["object_id"]) == n
            assert np.isfinite(z["poses"]).all()
