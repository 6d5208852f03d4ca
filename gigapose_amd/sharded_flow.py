"""`GigaPose.test_step` for a template-SHARDED model: every flush is exactly F crop rows on every rank.

No reference counterpart (the reference's inference is single-GPU); what it has to serve is the reference's test
loop: ONE image per `test_step` with however many detections that image has (reference test.py:55-60,
src/dataloader/test.py:104-107), different images -- hence different counts -- on every rank.  The two exchanges of a sharded
step (gigapose_amd/sharding.py) are fixed-size collectives, so:

  * the queue is CROP-granular: a flush takes exactly F = `accumulate_crops` crop rows off the front, cutting through images
    where it has to (detections are independent until filter_and_save, reference gigaPose.py:408-425); an image's <idx>.npz is
    written once its last crop has come back.  Only a draining rank pads: dummy crops (zero image, zero mask => no live patch,
    label of the first onboarded object, identity K / M) whose rows are dropped when files are written;
  * every rank runs the SAME sequence of collectives: L(0), L(1), F(0), L(2), F(1), ... -- flush j + 1 is launched, then flush
    j is finished (waited for, checked, files written while j + 1 runs).  Ranks reach their flushes at different `test_step`s;
    the collectives pair by their position in that sequence;
  * the end is agreed inside the flushes: each crop row of exchange #1 carries a word whose bit 0 says "this rank is draining
    and its queue is empty after this flush".  `drain()` (on_test_epoch_end / flush_pending) keeps flushing -- all-dummy rows
    once nothing is left -- until a flush in which EVERY rank said so; every rank sees that in the same F(j), one flush L(j+1)
    (all-dummy everywhere) is already queued, is finished, and all ranks stop after the same number of flushes;
  * the guard-rail word of flush j is all-gathered at the end of flush j (one 4-byte-per-rank collective) so that every rank
    takes the range-recovery decision (re-calibration = an all-reduce; re-onboarding) in the same F(j): then all ranks drop
    the results of j and j + 1, recover together, put both flushes' crops back at the front of their queues and carry on.
"""
import collections
import os.path as osp

import numpy as np
import torch

from . import _lib

DONE = 1   # bit 0 of the exchange word: this rank is draining and has nothing queued behind this flush


class _Image:
    """One queued image: its batch, and the host rows that have come back so far."""

    def __init__(self, batch, idx_batch, log_dir, test_setting):
        self.batch, self.idx_batch, self.n = batch, idx_batch, len(batch.infos)
        self.log_dir, self.test_setting = log_dir, test_setting      # bound when queued (a driver may re-point the model later)
        self.got, self.scores, self.poses, self.seconds, self.attempts = 0, None, None, 0.0, 0

    def reset(self):
        self.got, self.seconds = 0, 0.0


class ShardedFlow:
    def __init__(self, model):
        self.model = model
        self.queue = collections.deque()      # segments (image, a, b): crops [a, b) of that image, in arrival order
        self.queued = 0
        self.in_flight = None
        self.dummies = None
        self.last_all_done = False

    # -------------------------------------------------------------------------------------------------------------- host side
    @property
    def rows(self):
        return self.model.accumulate_crops if self.model.accumulate_crops > 0 else 64

    def push(self, batch, idx_batch):
        m = self.model
        img = _Image(batch, idx_batch, m.log_dir, m.test_setting)
        if img.n == 0:                        # an image without detections: its (empty) file, now
            k = m.testing_metric.k
            self._save(img, np.zeros((0, k), np.float32), np.zeros((0, k, 4, 4), np.float32))
            return
        self.queue.append((img, 0, img.n))
        self.queued += img.n
        while self.queued >= self.rows:
            self._tick(final=False)

    def drain(self):
        """Collective: every rank of the shard group must call it (on_test_epoch_end does).  Returns once every rank's queue is
        empty and every file is written."""
        while True:
            self._tick(final=True)
            if self.last_all_done:
                break
        job, self.in_flight = self.in_flight, None
        self._finish(job)                     # the all-dummy flush every rank queued before it saw the end
        self.last_all_done = False

    def _tick(self, final):
        job = self._launch(final)
        prev, self.in_flight = self.in_flight, job
        if prev is not None:
            self._finish(prev)

    def _launch(self, final):
        m = self.model
        dataset_name = m.test_dataset_name
        if dataset_name not in m.template_datas:
            m.set_template_data(dataset_name)
        F, segs, n = self.rows, [], 0
        while self.queue and n < F:
            img, a, b = self.queue.popleft()
            if n + (b - a) > F:               # cut through the image: the rest stays at the front
                cut = a + (F - n)
                self.queue.appendleft((img, cut, b))
                b = cut
            segs.append((img, a, b))
            n += b - a
        self.queued -= n
        flag = DONE if (final and not self.queue) else 0
        inputs, labels = self._inputs(segs, n, F)
        job = m._run_rows(inputs, labels, dataset_name, aux=flag, live_rows=n)
        job.update(segs=segs, n_live=n, rows=F)
        return job

    def _inputs(self, segs, n, F):
        """The F input rows of a flush: the queued crops, then dummy crops."""
        names = ("tar_img", "tar_mask", "tar_K", "tar_M")
        parts = {k: [img.batch.tensors[k][a:b] for img, a, b in segs] for k in names}
        labels = [np.asarray(img.batch.infos.label).astype(np.int32)[a:b] for img, a, b in segs]
        if n < F:
            like = segs[0][0].batch if segs else None
            d = self._dummies(like, F)
            for k in names:
                parts[k].append(d[k][: F - n])
            labels.append(np.ones(F - n, np.int32))   # any onboarded object: a dummy's zero mask leaves it no live patch
        inputs = {k: (v[0] if len(v) == 1 else torch.cat(v, dim=0)) for k, v in parts.items()}
        return inputs, np.concatenate(labels)

    def _dummies(self, like, F):
        if self.dummies is None or self.dummies["tar_img"].shape[0] < F:
            m = self.model
            if like is not None:
                t = like.tensors
                dev, dt = t["tar_img"].device, {k: t[k].dtype for k in ("tar_img", "tar_mask", "tar_K", "tar_M")}
                shp = {k: tuple(t[k].shape[1:]) for k in ("tar_img", "tar_mask", "tar_K", "tar_M")}
            else:                             # a rank that never saw an image
                dev = m.device
                dt = dict(tar_img=torch.float32, tar_mask=torch.float32, tar_K=torch.float32, tar_M=torch.float32)
                shp = dict(tar_img=(3, 224, 224), tar_mask=(224, 224), tar_K=(3, 3), tar_M=(3, 3))
            d = {k: torch.zeros((F,) + shp[k], dtype=dt[k], device=dev) for k in ("tar_img", "tar_mask")}
            for k in ("tar_K", "tar_M"):
                d[k] = torch.eye(shp[k][-1], dtype=dt[k], device=dev).expand(F, *shp[k]).contiguous()
            self.dummies = d
        return self.dummies

    def _finish(self, job):
        m = self.model
        job["ev"][1].synchronize()
        words = job["host"]["status"].numpy().reshape(-1)
        bits = 0
        for w in words:
            bits |= int(w)                    # every rank's word of THIS flush: all ranks hold the same `bits`
        flags = job["host"]["aux_all"].numpy().reshape(-1, job["rows"])[:, 0]
        self.last_all_done = bool((flags & DONE).all())
        if bits & _lib.SPLIT_RANGE_BITS:
            self._recover(job, bits)
            return
        _lib.raise_status(bits)
        if int(job["host"]["bad_crop_M"][0]) != 0:   # reference lib3d/torch.py:54-55
            m._clear_crop_flag(job["dataset_name"])
            raise AssertionError("tar_M must be an isotropic scale + translation")
        total_s = 1e-3 * job["ev"][0].elapsed_time(job["ev"][1])
        scores, poses = job["host"]["scores"].numpy(), job["host"]["pred_poses"].numpy()
        row = 0
        for img, a, b in job["segs"]:
            if img.scores is None:
                img.scores = np.empty((img.n,) + scores.shape[1:], scores.dtype)
                img.poses = np.empty((img.n,) + poses.shape[1:], poses.dtype)
            img.scores[a:b], img.poses[a:b] = scores[row:row + b - a], poses[row:row + b - a]
            img.seconds += total_s * (b - a) / max(job["n_live"], 1)
            img.got += b - a
            row += b - a
            if img.got == img.n:
                self._save(img, img.scores, img.poses)

    def _save(self, img, scores, poses):
        save_path = osp.join(img.log_dir, "predictions", f"{img.idx_batch}.npz")
        self.model._save_image(img.batch.infos, img.batch.test_list, scores, poses, img.seconds, save_path, img.test_setting == "localization")

    def _recover(self, job, bits):
        """A plane value left f16's range in flush `job` on SOME rank (`bits` is the OR over the ranks, so every rank is here, in
        the same F(j), with flush j + 1 queued behind).  All ranks: wait for j + 1, drop both results, recover together
        (GigaPose._recover_range: plane scales re-calibrated on both flushes' own rows -- two all-reduces on every rank -- or
        the wide kernels + re-onboarding), put the crops of both flushes back at the front of the queue.  Images whose crops
        have tripped three times raise (on every rank: the bits are shared)."""
        m = self.model
        nxt, self.in_flight = self.in_flight, None
        m._drain_device()
        jobs = [job] + ([nxt] if nxt is not None else [])
        images = [j["inputs"]["tar_img"] for j in jobs]
        while len(images) < 2:                # keep the number of calibration collectives equal on all ranks (the last F has no successor)
            images.append(images[0][:0])
        worst = max([img.attempts for j in jobs for img, _, _ in j["segs"]] + [job.get("attempts", 0)])
        if worst >= 2 or not m._recover_range(bits, images):
            _lib.raise_status(bits)
        back = []
        for j in jobs:
            for img, a, b in j["segs"]:
                back.append((img, a, b))
        touched = {id(img): img for img, _, _ in back}
        for img in touched.values():
            img.attempts += 1
        # an image cut by an EARLIER flush keeps the rows it already has; only these segments run again
        for seg in reversed(back):
            self.queue.appendleft(seg)
            self.queued += seg[2] - seg[1]
        self.last_all_done = False
