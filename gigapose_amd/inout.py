"""BOP result writer behind the reference interface: the step right after the hot path.

`save_predictions_from_batched_predictions` mirrors src/utils/inout.py:278-367 (called from
GigaPose.on_test_epoch_end, gigaPose.py:644-653): it merges the per-batch `<idx>.npz` files GigaPose.filter_and_save
writes (gigaPose.py:436-447) into the two BOP csv files (`...csv` top-1, `...MultiHypothesis.csv` all k
hypotheses + instance_id) with the BOP run-time accounting of calculate_runtime_per_image (inout.py:217-275:
time of an image = its detection time + the sum of the coarse times of the distinct batches it appears in).
Host-side file formatting only (no GPU work); the csv text is byte-identical to the reference's
(tests/test_inout.py against tests/golden/bop_csv.npz, written by the unmodified reference).
"""
import os
import os.path as osp

import numpy as np

LMO_INDEX_TO_ID = ["1", "5", "6", "8", "9", "10", "11", "12"]   # src/utils/dataset.py:18 (strings, as there)


def _floats(a):
    """' '.join(map(str, x.flatten().tolist())): float32 widened to Python floats, shortest repr."""
    return " ".join(map(str, np.asarray(a).flatten().tolist()))


def _rows_from_batches(prediction_dir, dataset_name, is_refined):
    """One record per (detection, hypothesis), in file order; hypothesis 0 first."""
    extra = "refinement_time" if is_refined else "detection_time"
    files = sorted(f for f in os.listdir(prediction_dir) if f.endswith(".npz"))
    rows, instance_id, multi = [], 0, False
    for batch_id, name in enumerate(files):
        d = np.load(osp.join(prediction_dir, name))
        poses = d["poses"]
        if poses.ndim not in (3, 4):
            raise AssertionError(f"{name}: poses must be (n,4,4) or (n,k,4,4)")
        top1_only = poses.ndim == 3
        multi = multi or not top1_only
        if top1_only:
            poses, scores = poses[:, None], d["scores"][:, None]
        else:
            scores = d["scores"]
        for i in range(len(d["im_id"])):
            obj_id = int(d["object_id"][i])
            if not is_refined and "lmo" in dataset_name:
                obj_id = LMO_INDEX_TO_ID[obj_id - 1]
            for h in range(poses.shape[1]):
                rows.append(dict(scene_id=int(d["scene_id"][i]), im_id=int(d["im_id"][i]), obj_id=obj_id,
                                 score=scores[i][h], R=poses[i][h][:3, :3].reshape(-1), t=poses[i][h][:3, 3].reshape(-1),
                                 time=d["time"][i], extra=d[extra][i], batch_id=batch_id, instance_id=instance_id,
                                 hyp=h))
            instance_id += 1
    return rows, multi


def _image_times(rows, is_refined):
    """BOP run time per image (inout.py:217-275).  Coarse: detection_time (of the last new batch seen for that
    image) + sum of `time` over its distinct batches; refined: sum(refinement_time) + sum(time) over them."""
    acc = {}
    for r in rows:
        key = (r["scene_id"], r["im_id"])
        a = acc.setdefault(key, dict(batches=[], time=[], extra=[]))
        if r["batch_id"] not in a["batches"]:
            a["batches"].append(r["batch_id"])
            a["time"].append(r["time"])
            a["extra"].append(r["extra"])
    total = {}
    for key, a in acc.items():
        if is_refined:
            total[key] = np.sum(a["extra"]) + np.sum(a["time"])
        else:
            total[key] = a["extra"][-1] + np.sum(a["time"])
    return total


def _write_csv(path, rows, times, with_instance):
    head = "scene_id,im_id,obj_id,score,R,t,time" + (",instance_id" if with_instance else "")
    lines = [head]
    for r in rows:
        line = "{},{},{},{},{},{},{}".format(r["scene_id"], r["im_id"], r["obj_id"], r["score"], _floats(r["R"]),
                                             _floats(r["t"]), times[(r["scene_id"], r["im_id"])])
        if with_instance:
            line += ",{}".format(r["instance_id"])
        lines.append(line)
    with open(path, "w") as f:
        f.write("\n".join(lines))


def save_predictions_from_batched_predictions(prediction_dir, dataset_name, model_name, run_id, is_refined):
    rows, multi = _rows_from_batches(prediction_dir, dataset_name, is_refined)
    stem = osp.join(prediction_dir, f"{model_name}-pbrreal-rgb-mmodel_{dataset_name}-test_{run_id}")
    top1 = [r for r in rows if r["hyp"] == 0]
    _write_csv(stem + ".csv", top1, _image_times(top1, is_refined), with_instance=False)
    paths = [stem + ".csv"]
    if multi:
        _write_csv(stem + "MultiHypothesis.csv", rows, _image_times(rows, is_refined), with_instance=True)
        paths.append(stem + "MultiHypothesis.csv")
    return paths
