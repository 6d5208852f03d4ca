"""`GigaPose` inference orchestration behind the reference interface
(src/models/gigaPose.py:34-77 ctor, :357-398 set_template_data, :481-633 eval_retrieval,
:400-449 filter_and_save, :635-653 test_step / on_test_epoch_end; Hydra target
configs/model/large.yaml:1).  Training/validation steps are out of scope.

What changes relative to the reference control flow (results are identical):
  * the template bank stays resident, matcher-normalised once (MatchBank); kernels index it by
    label -- no `ae_features[label-1]` gather (gigaPose.py:520), no per-batch re-normalisation;
  * the IST backbone runs once per crop, not k times (gigaPose.py:553);
  * IST regression, RANSAC and pose recovery run for all k hypotheses in one launch each;
  * the only host syncs per image are the label upload and the final result download.
"""
import os
import ctypes
import os.path as osp
import time

import numpy as np
import pandas as pd
import torch
from torch import nn

from . import _lib
from .matching import MatchBank
from .poses import ObjectPoseRecovery
from .tensor_collection import PandasTensorCollection

try:  # drop into pytorch_lightning.Trainer.test() when Lightning is installed
    import pytorch_lightning as pl

    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning is absent in this image
    class _Base(nn.Module):
        global_rank = 0
        logger = None

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")


def stable_argsort_desc(score):
    """argsort(score, dim=1, descending=True) with ties -> lower hypothesis index first
    (the reference's torch.argsort leaves tie order unspecified, gigaPose.py:591)."""
    return torch.sort(score, dim=1, descending=True, stable=True).indices


def rank_hypotheses(pred, sort=True):
    """gigaPose.py:588-594 in one launch (gp_rank_hypotheses): scores = sum(ransac_scores, dim=2) / num_patches
    (an int64 sum: exact), and -- when `sort` -- every tensor of the collection reordered along k by descending score,
    ties keeping the lower hypothesis index (stable_argsort_desc).  Registers "scores"; returns the order (B,k)."""
    isc = pred.ransac_scores.contiguous()
    B, k, num_patches = isc.shape
    dev = isc.device
    if not isc.is_cuda or isc.dtype != torch.int64:
        # the launch reinterprets the buffer: a float score tensor would rank garbage.  No torch fallback (the hot path is HIP only)
        raise _lib.GigaPoseHipError(f"rank_hypotheses needs the int64 device tensor gp_ransac writes, got {isc.dtype} on {isc.device}")
    score = torch.empty(B, k, dtype=torch.float32, device=dev)
    order = torch.empty(B, k, dtype=torch.int64, device=dev)
    names = list(pred.tensors) if sort else []
    srcs = [pred.tensors[n].contiguous() for n in names]
    for v in srcs:
        if v.shape[:2] != (B, k):
            raise ValueError(f"per-hypothesis tensors must be (B, k, ...), got {tuple(v.shape)}")
    dsts = [torch.empty_like(v) for v in srcs]
    for s0 in range(0, max(len(srcs), 1), 16):                                   # the launch takes <= 16 tensors
        a, b = srcs[s0:s0 + 16], dsts[s0:s0 + 16]
        n = len(a)
        src_t = (ctypes.c_void_p * max(n, 1))(*[v.data_ptr() for v in a])
        dst_t = (ctypes.c_void_p * max(n, 1))(*[v.data_ptr() for v in b])
        rb_t = (ctypes.c_int * max(n, 1))(*[v[0, 0].numel() * v.element_size() if B else 1 for v in a])
        _lib.call("gp_rank_hypotheses", _lib.ptr(isc), _lib.i(B), _lib.i(k), _lib.i(num_patches), _lib.i(1 if sort else 0),
                  _lib.ptr(score), _lib.ptr(order), _lib.i(n), src_t, dst_t, rb_t, _lib.stream_ptr())
    for name, v in zip(names, dsts):
        pred.register_tensor(name, v)
    pred.register_tensor("scores", score)
    return order


class GigaPose(_Base):
    def __init__(self, model_name, ae_net, ist_net, training_loss, testing_metric, optim_config, log_interval,
                 log_dir, max_num_dets_per_forward=None, test_setting="localization", **kwargs):
        """Reference signature (gigaPose.py:35-48).  One optional key is read from **kwargs (where the reference's own YAML already
        passes `refiner` / `checkpoint_path`): `numerics` ("split" | "chain"; a `numerics:` line in the model YAML next to the
        `_target_` edits of INTEGRATION.md section 1).  Absent = _lib.default_numerics(): "split" unless GIGAPOSE_NUMERICS=chain
        -- the drop-in runs the numerics bench.py measures; "chain" is the verification mode."""
        super().__init__()
        numerics = kwargs.pop("numerics", None)
        self.model_name = model_name
        self.ae_net = ae_net
        self.ist_net = ist_net
        self.training_loss = training_loss
        self.testing_metric = testing_metric
        self.max_num_dets_per_forward = max_num_dets_per_forward
        self.test_setting = test_setting
        self.log_interval = log_interval
        self.log_dir = log_dir
        os.makedirs(osp.join(self.log_dir, "predictions"), exist_ok=True)
        self.optim_config = optim_config
        self.template_datas = {}
        self.match_banks = {}
        self.pose_recovery = {}
        self.run_id = None
        self.template_datasets = None
        self.test_dataset_name = None
        self.last_predictions = None  # full (unfiltered) predictions of the last eval_retrieval call
        self.template_shard = None    # (rank, world, group) when the template bank is sharded
        # IST backbone on a side stream, concurrently with ViT + matching?  At 64 crops (BENCH_r01, profiles/r02_*) both chains are
        # matrix-core / power bound, so the overlap only stretches every kernel (attention 3.4 -> 7.7 ms, GEMMs 30.6 -> 36.4 ms per
        # step inside the two-stream region) for a 0-1 % gain in step time.  Below that neither chain fills the chip and the two
        # streams do: 825 -> 884 crops/s at 8 crops, 1136 -> 1168 at 16, 1349 -> 1370 at 32 (round 4, tools/gpu_r04_overlap.sh).  Same
        # kernels, same results.  False (default) / True / "auto" (two streams up to 32 crops); GIGAPOSE_OVERLAP_IST=0 / 1 / auto.
        env = os.environ.get("GIGAPOSE_OVERLAP_IST", "0").strip().lower()
        self.overlap_ist = "auto" if env == "auto" else env in ("1", "true", "on")
        self._side_stream = None
        if numerics is not None:
            self.set_numerics(numerics)

    def set_numerics(self, mode):
        """"split" (default): the ViT linear layers + attention, the template matcher and the IST convolutions / MLP run as
        3 x f16 MFMA on split f32 operands (f32-equivalent accuracy).  "chain": f32 fmaf-chain kernels,
        bit-exact vs the CPU oracle (the verification mode).  Call before set_template_data (the bank is stored in the
        matcher's format; banks already onboarded are dropped)."""
        if mode not in _lib.NUMERICS:
            raise ValueError(f"numerics must be one of {_lib.NUMERICS}")
        self.ae_net.dinov2_model.set_numerics(mode)
        self.testing_metric.numerics = mode
        self.ist_net.backbone.set_numerics(mode)
        self.template_datas, self.match_banks, self.pose_recovery = {}, {}, {}
        return self

    def _widen_split_range(self, bits):
        """Automatic fallback of the split numerics' range guard.  The single-accumulator plane kernels hold activations x 8 in
        f16 (|x| < 8190); a checkpoint with larger activations (DINOv2's massive channels) trips the guard -- instead of failing,
        move the network that tripped it to its two-accumulator 128 x 128 kernels (range 65504) for the rest of this model's life,
        drop the onboarded banks (they are rebuilt with the same kernels) and let the caller run again.  Returns True if something
        was widened (False: already wide, or not a range bit -- the caller raises)."""
        import warnings

        changed = []
        vit = getattr(self.ae_net, "dinov2_model", None)
        if bits & 4 and vit is not None and getattr(vit, "numerics", None) == "split" and getattr(vit, "split_gemm", "128") != "128":
            vit.set_split_gemm("128")
            changed.append("ViT linear layers -> 128 x 128 two-accumulator kernels (GIGAPOSE_SPLIT_GEMM=128)")
        ist = getattr(self.ist_net, "backbone", None)
        if bits & 16 and ist is not None and getattr(ist, "numerics", None) == "split" and getattr(ist, "conv_kernel", "128") != "128":
            ist.conv_kernel = "128"
            ist.invalidate()
            changed.append("IST convolutions -> 128 x 128 two-accumulator kernel (GIGAPOSE_SPLIT_CONV=128)")
        if changed:
            warnings.warn("split numerics: an activation left the range of the f16 planes (|x| >= 8190); falling back for this model: "
                          + "; ".join(changed) + ".  Template banks are rebuilt.", RuntimeWarning)
            # every bank was built by the kernels that just left: rebuild ALL of them now (a later predict() on another dataset must
            # not hit a KeyError, and one model must not hold banks made by different kernels); outside any step timing
            names = list(self.template_datas)
            self.template_datas, self.match_banks, self.pose_recovery = {}, {}, {}
            for name in names:
                self.set_template_data(name)
        return bool(changed)

    def _collect_status(self):
        """Read + clear the guard-rail bits at a point where the host synchronises anyway.  With a sharded template bank the
        decision they drive (range fallback = re-onboarding + a second pass through the collectives) must be taken by ALL ranks
        together: each rank runs the ViT on its own crops, so one rank may see the range bit while its peers see none -- it would
        re-enter predict() and pair its all-gather with the peers' NEXT step (a hang, or rows of different steps exchanged
        silently).  The bits are therefore OR-ed over the group first (one tiny all-reduce per call, MAX over per-bit flags:
        RCCL has no bitwise OR), so every rank widens and retries, or none does."""
        bits = _lib.take_status()
        if self.template_shard is not None:
            import torch.distributed as dist

            _, world, group = self.template_shard
            if world > 1:
                dev = self.device if dist.get_backend(group) != "gloo" else torch.device("cpu")
                flags = torch.tensor([(bits >> b) & 1 for b in range(16)], dtype=torch.int32, device=dev)
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
                bits = sum(int(f) << b for b, f in enumerate(flags.tolist()))
        return bits

    def enable_template_sharding(self, group=None):
        """Shard the template bank over the ranks of `group` (gigapose_amd/sharding.py).  Call before set_template_data; every rank
        must then call predict() with the SAME batch size (the exchanges are fixed-size collectives: eval_retrieval verifies it
        with one tiny all-reduce and raises on all ranks together; a raw predict() loop must guarantee it, as bench.py does)."""
        import torch.distributed as dist

        self.template_shard = (dist.get_rank(group), dist.get_world_size(group), group)

    # ------------------------------------------------------------------ onboarding
    @torch.no_grad()
    def set_template_data(self, dataset_name):
        """Build the resident template bank (reference gigaPose.py:357-398)."""
        t0 = time.time()
        template_dataset = self.template_datasets[dataset_name]
        dev = self.device
        if torch.device(dev).type == "cuda":
            _lib.status_word(dev)
        cols = {n: [] for n in ["mask", "K", "M", "poses", "ae_features", "ist_features"]}
        lo, hi = 0, None
        if self.template_shard is not None:
            from .sharding import ShardedMatcher, shard_bounds

            rank, world, group = self.template_shard
        for idx in range(len(template_dataset)):
            item = template_dataset[idx]
            templates = item.rgb.to(dev)
            if self.template_shard is not None:
                # This rank KEEPS templates [lo, hi) but computes the features of all of them, in the chunks an unsharded model
                # uses: in split numerics a feature's round-off depends on the GEMM shapes its chunk selects (tile vs ragged strip,
                # serial vs parallel split-K), and the sharded path promises results equal to the unsharded one bit for bit
                # (tests/test_gpu_world2.py).  Onboarding is one-off and outside every timed region (0.13 s per object).
                lo, hi = shard_bounds(templates.shape[0], world, rank)
            cols["ae_features"].append(self.ae_net(templates)[lo:hi].contiguous())
            cols["ist_features"].append(self.ist_net.forward_by_chunk(templates))
            for n in ["mask", "K", "M", "poses"]:
                cols[n].append(getattr(item, n).to(dev))
        data = {n: torch.stack(v, dim=0) for n, v in cols.items()}
        self.template_datas[dataset_name] = PandasTensorCollection(infos=pd.DataFrame(), **data)
        if self.template_shard is None:
            self.match_banks[dataset_name] = MatchBank(data["ae_features"], data["mask"], self.testing_metric.numerics,
                                                       self.testing_metric.bank_dtype)
        else:  # the matcher bank holds only this rank's slice of every object's templates
            shard = MatchBank(data["ae_features"], data["mask"][:, lo:hi].contiguous(), self.testing_metric.numerics,
                              self.testing_metric.bank_dtype)
            self.match_banks[dataset_name] = ShardedMatcher(self.testing_metric, shard, lo, group)
        self.pose_recovery[dataset_name] = ObjectPoseRecovery(template_K=data["K"], template_Ms=data["M"],
                                                              template_poses=data["poses"])
        torch.cuda.synchronize()
        bits = self._collect_status()
        if bits & _lib.SPLIT_RANGE_BITS and self._widen_split_range(bits):
            if dataset_name in self.template_datas:              # _widen_split_range rebuilt every bank, this one included
                return
            return self.set_template_data(dataset_name)          # once more with the wide-range kernels (then any bit raises)
        _lib.raise_status(bits)
        self.onboarding_time = (time.time() - t0) / max(1, len(template_dataset))

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def predict(self, tar_img, tar_mask, tar_K, tar_M, labels, dataset_name, sort_pred_by_inliers=True):
        """Device-only part of eval_retrieval (gigaPose.py:511-604): crops -> sorted pose hypotheses.
        labels: (B) int tensor of 1-based object labels.  Returns a PandasTensorCollection with
        id_src, score_src, score_pts, tar_pts, src_pts, relScale, relInplane, idx_failed, M,
        ransac_*, scores (B,k), pred_poses (B,k,4,4)."""
        bank = self.match_banks[dataset_name]
        template_data = self.template_datas[dataset_name]
        n_obj = template_data.ist_features.shape[0]
        if not labels.is_cuda and labels.numel() and (int(labels.min()) < 1 or int(labels.max()) > n_obj):
            raise IndexError(f"detection label outside 1..{n_obj} (the reference indexes ae_features[label - 1], gigaPose.py:520)")
        if tar_img.is_cuda:
            _lib.status_word(tar_img.device)  # device labels / hand-offs / split range are checked by the kernels (check_status)
        if not labels.is_cuda and tar_img.is_cuda:
            # stream-ordered upload from pinned memory: a pageable-memory copy would block the host until the stream drains,
            # i.e. a host sync per call that exposes the launch latency of everything after it
            labels = labels.pin_memory().to(tar_img.device, non_blocking=True)
        labels0 = (labels.to(tar_img.device) - 1).to(torch.int32).contiguous()
        side = None
        if tar_img.is_cuda and (self.overlap_ist is True or (self.overlap_ist == "auto" and tar_img.shape[0] <= 32)):
            # IST backbone on a second HIP stream: both chains are matrix-core bound, the overlap fills
            # the partial last wave of workgroups ("tail") of each other's launches
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=tar_img.device)
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                tar_ist = self.ist_net.forward_by_chunk(tar_img)                 # stage 4a: IST backbone (once)
        tar_ae = self.ae_net(tar_img)                                            # stage 1: ViT features
        if self.template_shard is None:
            pred = self.testing_metric.test_bank(bank, tar_ae, tar_mask, labels0)  # stage 3: matching
            if side is None:
                tar_ist = self.ist_net.forward_by_chunk(tar_img)                 # stage 4a: IST backbone (once)
        else:
            # sharded bank: exchange #1 (query features to every rank) travels while the IST backbone runs
            pending = bank.start_exchange(tar_ae, tar_mask, labels0)
            if side is None:
                tar_ist = self.ist_net.forward_by_chunk(tar_img)
            pred = bank.finish(pending)                                          # match the shard + exchange #2 + merge
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            tar_ist.record_stream(torch.cuda.current_stream())
        rel_scale, rel_inplane = self.ist_net.regress_bank(template_data.ist_features, labels0, pred.id_src,
                                                           tar_ist, pred.src_pts, pred.tar_pts)
        pred.register_tensor("relScale", rel_scale)
        pred.register_tensor("relInplane", rel_inplane)
        pred = self.pose_recovery[dataset_name].forward_ransac(predictions=pred)  # stage 5
        rank_hypotheses(pred, sort_pred_by_inliers)                              # gigaPose.py:588-594, one launch
        poses = self.pose_recovery[dataset_name].forward_recovery(                # stage 6
            tar_label=labels, tar_K=tar_K, tar_M=tar_M, pred_src_views=pred.id_src, pred_M=pred.M)
        pred.register_tensor("pred_poses", poses)
        return pred

    def eval_retrieval(self, batch, idx_batch, dataset_name, sort_pred_by_inliers=True):
        if dataset_name not in self.template_datas:
            self.set_template_data(dataset_name)
        labels_np = np.asarray(batch.infos.label).astype(np.int32)
        if self.template_shard is not None:   # fixed-size collectives ahead: all ranks agree on the batch size, or all raise (sharding.py)
            from .sharding import require_same_batch

            require_same_batch(len(labels_np), batch.tar_img.device, self.template_shard[2])
        t0 = time.time()
        labels = torch.from_numpy(labels_np)
        predictions = self.predict(batch.tar_img, batch.tar_mask, batch.tar_K, batch.tar_M, labels, dataset_name,
                                   sort_pred_by_inliers)
        torch.cuda.synchronize()
        bits = self._collect_status()  # guard rails: lost hand-off / split range / label range -> GigaPoseHipError, never silent garbage
        if bits & _lib.SPLIT_RANGE_BITS and self._widen_split_range(bits):
            return self.eval_retrieval(batch, idx_batch, dataset_name, sort_pred_by_inliers)   # re-onboards, runs again; a second trip raises
        _lib.raise_status(bits)
        predictions.infos = batch.infos
        total_time = time.time() - t0
        self.last_predictions = predictions
        save_path = osp.join(self.log_dir, "predictions", f"{idx_batch}.npz")
        if getattr(batch, "test_list", None) is not None:
            return self.filter_and_save(predictions, test_list=batch.test_list, time=total_time, save_path=save_path,
                                        keep_only_testing_instances=(self.test_setting == "localization"))
        return None, predictions

    def filter_and_save(self, predictions, test_list, time, save_path, keep_only_testing_instances=True):
        """Keep the top-`inst_count` detections per target object and write the per-image npz the
        BOP writer consumes (reference gigaPose.py:400-449)."""
        labels = np.asarray(predictions.infos.label).astype(np.int32)
        assert len(np.unique(labels)) == len(np.unique(test_list.infos.obj_id))
        top_scores = predictions.scores[:, 0].cpu().numpy()
        selected, detection_times = list(range(len(labels))), [0.0] * len(labels)
        if keep_only_testing_instances:
            selected, detection_times = [], []
            for row, obj_id in enumerate(test_list.infos.obj_id):
                num_inst = int(test_list.infos.inst_count[row])
                cand = np.flatnonzero(labels == obj_id)
                best = cand[np.argsort(-top_scores[cand], kind="stable")[:num_inst]]
                selected.extend(best.tolist())
                detection_times.extend([test_list.infos.detection_time[row]] * num_inst)
        predictions = predictions[selected]
        det_t = torch.from_numpy(np.asarray(detection_times, dtype=np.float64)).to(predictions.scores.device)
        predictions.register_tensor("detection_time", det_t)
        predictions.register_tensor("time", torch.ones_like(det_t) * time)
        np.savez(save_path,
                 scene_id=np.asarray(predictions.infos.scene_id).astype(np.int32),
                 im_id=np.asarray(predictions.infos.view_id).astype(np.int32),
                 object_id=np.asarray(predictions.infos.label).astype(np.int32),
                 time=predictions.time.cpu().numpy(), detection_time=predictions.detection_time.cpu().numpy(),
                 poses=predictions.pred_poses.cpu().numpy(), scores=predictions.scores.cpu().numpy())
        return selected, predictions

    @torch.no_grad()
    def test_step(self, batch, idx_batch):
        self.eval_retrieval(batch, idx_batch=idx_batch, dataset_name=self.test_dataset_name)
        return 0

    def on_test_epoch_end(self):
        """Merge the per-batch npz files into the BOP csv files (reference gigaPose.py:644-653 ->
        src/utils/inout.py:278-367; here gigapose_amd/inout.py, byte-identical output)."""
        if self.global_rank != 0:
            return
        from .inout import save_predictions_from_batched_predictions

        save_predictions_from_batched_predictions(osp.join(self.log_dir, "predictions"),
                                                  dataset_name=self.test_dataset_name, model_name=self.model_name,
                                                  run_id=self.run_id, is_refined=False)
