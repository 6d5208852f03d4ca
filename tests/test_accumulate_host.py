"""CPU: the cross-image accumulation behind `GigaPose.test_step` (gigapose_amd/gigaPose.py, round 5) -- host logic only.

The reference's test.py feeds one image per test_step (reference test.py:55-60); the product queues whole images and runs one
predict() per >= accumulate_crops pending crops.  Here the device part (`_run_flush`: predict + stream-ordered downloads) is
replaced by a CPU double that computes every crop's result from the crop alone, so what runs as shipped is: the queue, the whole-image
cut, the one-flush-in-flight pipeline, the order files are written in, on_test_epoch_end's drain, the range-fallback redo of BOTH
queued flushes, and `_save_image` -- whose npz must equal filter_and_save's (reference gigaPose.py:400-449) field for field."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from gigapose_amd import _lib
from gigapose_amd.gigaPose import GigaPose
from gigapose_amd.tensor_collection import PandasTensorCollection

K_HYP = 5


class _Ev:
    def __init__(self, ms):
        self.ms = ms

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return other.ms - self.ms


def crop_result(img):
    """Deterministic per-crop 'prediction' from the crop's own pixels only (any batch composition gives the same rows)."""
    seed = int(abs(float(img.sum())) * 1e3) % (2 ** 31)
    rs = np.random.RandomState(seed)
    scores = np.sort(rs.randint(0, 200, K_HYP).astype(np.float32) / 256.0)[::-1].copy()
    poses = rs.standard_normal((K_HYP, 4, 4)).astype(np.float32)
    return scores, poses


class _Model(GigaPose):
    """GigaPose with the device half of a flush replaced (see module docstring)."""

    def __init__(self, log_dir, accumulate):
        torch.nn.Module.__init__(self)
        self.log_dir, self.test_setting, self.test_dataset_name = log_dir, "localization", "syn"
        os.makedirs(os.path.join(log_dir, "predictions"), exist_ok=True)
        self.template_datas, self.template_shard = {"syn": object()}, None
        self.accumulate_crops, self._pending, self._pending_crops, self._in_flight = accumulate, [], 0, None
        self.image_ownership, self._sharded_flow = None, None
        self.flushes, self.trip_on_flush, self.widened, self.clock = [], None, 0, 0.0
        self.model_name, self.run_id = "large", "r0"

    def _accumulating(self, batch):
        return self.accumulate_crops > 0 and getattr(batch, "test_list", None) is not None

    def _drain_device(self):
        pass

    def _recover_range(self, bits, images):
        assert images is not None and len(images) > 0     # the offending crops are handed over for re-calibration
        self.widened += 1
        return self.widened == 1

    def _run_flush(self, images, dataset_name):
        imgs = torch.cat([im[0].tar_img for im in images])
        res = [crop_result(i) for i in imgs]
        status = 0
        if self.trip_on_flush is not None and len(self.flushes) == self.trip_on_flush and not self.widened:
            status = 4                                           # the range bit of the split planes
        self.flushes.append([im[1] for im in images])
        t0 = self.clock
        self.clock += 10.0
        labels = np.concatenate([np.asarray(im[0].infos.label).astype(np.int32) for im in images])
        pred = PandasTensorCollection(infos=pd.DataFrame(), scores=torch.from_numpy(np.stack([r[0] for r in res])),
                                      pred_poses=torch.from_numpy(np.stack([r[1] for r in res])))
        host = dict(scores=pred.scores, pred_poses=pred.pred_poses, status=torch.tensor([status], dtype=torch.int32),
                    bad_crop_M=torch.zeros(1, dtype=torch.int32))
        return dict(images=images, labels=labels, pred=pred, host=host, ev=(_Ev(t0), _Ev(self.clock)), dataset_name=dataset_name, device="cpu")

    def eval_retrieval(self, batch, idx_batch, dataset_name, sort_pred_by_inliers=True):   # the per-image flow (accumulate_crops = 0)
        res = [crop_result(i) for i in batch.tar_img]
        pred = PandasTensorCollection(infos=batch.infos, scores=torch.from_numpy(np.stack([r[0] for r in res])),
                                      pred_poses=torch.from_numpy(np.stack([r[1] for r in res])))
        self.flushes.append([idx_batch])
        return self.filter_and_save(pred, batch.test_list, time=0.01, save_path=os.path.join(self.log_dir, "predictions", f"{idx_batch}.npz"),
                                    keep_only_testing_instances=True)


def image(seed, n, view_id, n_obj=3, cap=None):
    rs = np.random.RandomState(seed)
    labels = rs.randint(1, n_obj + 1, n)
    infos = pd.DataFrame(dict(label=[str(l) for l in labels], scene_id=[2] * n, view_id=[view_id] * n))
    batch = PandasTensorCollection(infos=infos, tar_img=torch.from_numpy(rs.standard_normal((n, 3, 4, 4)).astype(np.float32)))
    objs = sorted(set(int(l) for l in labels))
    counts = [int((labels == o).sum()) for o in objs]
    if cap:   # localisation keeps the best `inst_count` detections per object (fewer than detected)
        counts = [min(c, cap) for c in counts]
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(im_id=[view_id] * len(objs), scene_id=[2] * len(objs), obj_id=objs,
                                                                     inst_count=counts, detection_time=[0.05 + 0.01 * o for o in objs])))
    return batch


def run(tmp, accumulate, images, trip=None):
    m = _Model(str(tmp), accumulate)
    m.trip_on_flush = trip
    for idx, b in enumerate(images):
        assert m.test_step(b, idx) == 0
    m.flush_pending()
    return m


def load(tmp, idx):
    with np.load(os.path.join(str(tmp), "predictions", f"{idx}.npz")) as z:
        return {k: z[k] for k in z.files}


SIZES = [5, 9, 3, 7, 16, 2, 11, 4, 6]


def test_files_equal_the_per_image_flow_field_for_field(tmp_path):
    imgs = [image(100 + i, n, view_id=10 + i, cap=2 if i % 2 else None) for i, n in enumerate(SIZES)]
    a = run(tmp_path / "per_image", 0, imgs)
    b = run(tmp_path / "accumulated", 16, imgs)
    assert a.flushes == [[i] for i in range(len(SIZES))]
    # whole images, at least one, never more than 16 crops unless a single image is larger; order kept
    assert b.flushes == [[0, 1], [2, 3], [4], [5, 6], [7, 8]]
    for idx, n in enumerate(SIZES):
        fa, fb = load(tmp_path / "per_image", idx), load(tmp_path / "accumulated", idx)
        assert sorted(fa) == sorted(fb) == ["detection_time", "im_id", "object_id", "poses", "scene_id", "scores", "time"]
        for key in fa:
            assert fa[key].dtype == fb[key].dtype and fa[key].shape == fb[key].shape, key
            if key != "time":
                np.testing.assert_array_equal(fa[key], fb[key], err_msg=f"image {idx}: {key}")
    # `time`: the flush's device time (10 ms per flush in the double) apportioned by crop count -- sums back to it
    for flush in b.flushes:
        crops = sum(SIZES[i] for i in flush)
        for i in flush:
            t = load(tmp_path / "accumulated", i)["time"]
            assert np.allclose(t, 0.010 * SIZES[i] / crops) and t.dtype == np.float64


def test_one_flush_stays_in_flight_and_files_appear_in_order(tmp_path):
    imgs = [image(200 + i, 8, view_id=i) for i in range(6)]
    m = _Model(str(tmp_path), 16)
    seen = []
    for idx, b in enumerate(imgs):
        m.test_step(b, idx)
        seen.append(sorted(int(f[:-4]) for f in os.listdir(os.path.join(str(tmp_path), "predictions"))))
    # flush j's files are written while flush j + 1 runs: after the launch of flush 1 (step 3) the files of flush 0 exist, ...
    assert seen == [[], [], [], [0, 1], [0, 1], [0, 1, 2, 3]]
    assert m._in_flight is not None and m._pending == []
    m.flush_pending()
    assert m._in_flight is None and sorted(os.listdir(os.path.join(str(tmp_path), "predictions"))) == [f"{i}.npz" for i in range(6)]


def test_range_fallback_redoes_the_tripping_flush_and_the_one_queued_behind_it(tmp_path):
    imgs = [image(300 + i, 8, view_id=i) for i in range(6)]
    clean = run(tmp_path / "clean", 16, imgs)
    tripped = run(tmp_path / "tripped", 16, imgs, trip=0)
    assert clean.flushes == [[0, 1], [2, 3], [4, 5]]
    # flush 0 trips; flush 1 was already queued with the narrow planes: both run again, in order, then the rest
    assert tripped.flushes == [[0, 1], [2, 3], [0, 1], [2, 3], [4, 5]] and tripped.widened == 1
    for idx in range(6):
        fa, fb = load(tmp_path / "clean", idx), load(tmp_path / "tripped", idx)
        for key in fa:
            if key != "time":
                np.testing.assert_array_equal(fa[key], fb[key])


def test_a_second_trip_raises(tmp_path):
    m = _Model(str(tmp_path), 8)
    m._recover_range = lambda bits, images: False    # scales cover the inputs and the kernels are already wide: nothing left
    m.trip_on_flush = 0
    m.test_step(image(1, 8, 0), 0)
    with pytest.raises(_lib.GigaPoseHipError):
        m.flush_pending()


def test_an_image_larger_than_the_threshold_is_one_flush_and_zero_means_the_reference_flow(tmp_path):
    imgs = [image(400, 40, 0), image(401, 3, 1)]
    m = run(tmp_path / "a", 16, imgs)
    assert m.flushes == [[0], [1]]
    m0 = run(tmp_path / "b", 0, imgs)
    assert m0.flushes == [[0], [1]] and m0._in_flight is None
