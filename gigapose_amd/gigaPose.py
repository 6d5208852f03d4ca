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
        accumulate = kwargs.pop("accumulate_crops", None)
        ownership = kwargs.pop("image_ownership", None)
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
        # Default "auto" since round 5 (the full GPU suite runs with it: tests/conftest.py leaves it at the product default).
        env = os.environ.get("GIGAPOSE_OVERLAP_IST", "auto").strip().lower()
        self.overlap_ist = "auto" if env == "auto" else env in ("1", "true", "on")
        self._side_stream = None
        # Cross-image accumulation behind the unchanged test_step API (round 5).  test.py feeds ONE image per test_step (batch_size = 1,
        # reference test.py:55-60; at most 16 detections per object id, dataloader/test.py:104-107): 4-30 crops per call, where the
        # chip runs at 0.55-0.9 of its 64-crop rate.  Detections are independent until filter_and_save (gigaPose.py:408-425 of the
        # reference is the first cross-detection step, and it is per image), so test_step queues each image's crops and runs ONE
        # predict() over whole images once >= accumulate_crops are pending (and in on_test_epoch_end); the per-image <idx>.npz files are
        # written then.  0 = the reference's flow (one predict per image, file written before test_step returns).  `accumulate_crops:`
        # in the model YAML (read from **kwargs like `numerics`) or GIGAPOSE_ACCUMULATE_CROPS.
        self.accumulate_crops = int(os.environ.get("GIGAPOSE_ACCUMULATE_CROPS", "64")) if accumulate is None else int(accumulate)
        self._pending = []        # queued images: (batch, idx_batch, (dataset_name, log_dir, test_setting) when queued)
        self._pending_crops = 0
        self._in_flight = None    # the flush whose kernels are queued on the GPU while the host writes the previous flush's files
        self._feat_shapes = {}     # output shape of the two backbones (predict: all-padding batches skip them)
        self._sharded_flow = None  # sharded_flow.ShardedFlow: test_step's fixed-size flushes when the template bank is sharded
        # Who computes which image when several ranks run the reference's test loop?  The reference's webdataset pipeline has no
        # split_by_node (reference src/custom_megapose/web_scene_dataset.py:207-215): under a multi-process launch EVERY rank
        # replays EVERY image and overwrites predictions/<idx>.npz (reference gigaPose.py:611), i.e. N GPUs give 1 x throughput.
        # `image_ownership: round_robin` (model YAML, next to `numerics`; env GIGAPOSE_IMAGE_OWNERSHIP=round_robin): test_step
        # skips images with idx_batch % world != rank (ranks of the default process group, or of the shard group when the bank is
        # sharded); rank 0 merges all ranks' files in on_test_epoch_end after a barrier.  Off (None) by default: a single process,
        # or a loader that already splits the images, needs nothing.
        self.image_ownership = (os.environ.get("GIGAPOSE_IMAGE_OWNERSHIP") or None) if ownership is None else (ownership or None)
        if self.image_ownership not in (None, "round_robin"):
            raise ValueError("image_ownership must be None or 'round_robin'")
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
            changed.append("ViT linear layers -> 128 x 128 two-accumulator kernels (Dinov2ViT.set_split_gemm('128'))")
        ist = getattr(self.ist_net, "backbone", None)
        if bits & 16 and ist is not None and getattr(ist, "numerics", None) == "split" and getattr(ist, "conv_kernel", "128") != "128":
            ist.conv_kernel = "128"
            ist.invalidate()
            changed.append("IST convolutions -> 128 x 128 two-accumulator kernel (ResNet.conv_kernel = '128')")
        if changed:
            warnings.warn("split numerics: an activation left the range of the f16 planes (|x| >= 8190); falling back for this model: "
                          + "; ".join(changed) + ".  Template banks are rebuilt.", RuntimeWarning)
            # every bank was built by the kernels that just left: rebuild ALL of them now (a later predict() on another dataset must
            # not hit a KeyError, and one model must not hold banks made by different kernels); outside any step timing
            # (only names whose dataset is still attached: a bank onboarded from a dataset the caller has since replaced cannot be
            # rebuilt -- it is dropped, and a later predict() on it onboards or raises KeyError as for any unknown name)
            names = [n for n in self.template_datas if n in (self.template_datasets or {})]
            self.template_datas, self.match_banks, self.pose_recovery = {}, {}, {}
            for name in names:
                self.set_template_data(name)
        return bool(changed)

    def _calibrate_planes(self, images):
        """Run the ViT's plane-scale calibration (vit.py: calibrate_plane_scales) over `images` (a tensor, or an iterable of tensors);
        with a sharded bank the ranks calibrate together (maxima all-reduced: every rank must hold the same scales).  Returns True if a
        scale changed.  A calibration pass that itself trips a guard rail (non-finite activations) raises."""
        vit = getattr(self.ae_net, "dinov2_model", None)
        if vit is None or not hasattr(vit, "calibrate_plane_scales"):
            return False
        sharded = self.template_shard is not None
        group = self.template_shard[2] if sharded else None            # None under sharding = the default process group
        changed = False
        for x in ([images] if torch.is_tensor(images) else images):
            changed = vit.calibrate_plane_scales(x.to(self.device), group=group, sync_ranks=sharded) or changed
        _lib.raise_status(_lib.take_status() & ~_lib.SPLIT_RANGE_BITS)   # hand-over bits of the pass raise; its range bits are what it measures
        return changed

    def _needs_calibration(self):
        vit = getattr(self.ae_net, "dinov2_model", None)
        return (vit is not None and getattr(vit, "numerics", None) == "split" and getattr(vit, "split_gemm", "128") != "128"
                and hasattr(vit, "calibrate_plane_scales") and vit.plane_amax is None)

    def _recover_range(self, bits, images):
        """A plane value left f16's range (GP_STATUS_SPLIT_RANGE).  First remedy (round 5): re-calibrate the per-tensor plane scales on
        the inputs that tripped the guard -- the scales of the offending tensors drop by powers of two, every GEMM stays on the fast
        kernels, banks are kept (a feature computed under another scale differs by f32 round-off, as it does between two batch
        shapes).  If no scale changed (already covered: not a range problem of the ViT planes) fall back to the wide kernels
        (_widen_split_range).  Returns "recalibrated" / "widened" (the caller runs again; after "widened" every bank has been rebuilt) or False."""
        import warnings

        if bits & 4 and images is not None:
            vit = self.ae_net.dinov2_model
            if getattr(vit, "numerics", None) == "split" and getattr(vit, "split_gemm", "128") != "128" and self._calibrate_planes(images):
                warnings.warn("split numerics: a ViT activation left the range of its f16 planes; plane scales re-calibrated on the offending "
                              f"inputs: {vit.plane_scale_report()} (tensor: (max |x|, scale)); all GEMMs stay on the 256 x 256 kernels.", RuntimeWarning)
                return "recalibrated"
        return "widened" if self._widen_split_range(bits) else False

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
        if self._needs_calibration() and len(template_dataset):
            # first bank of this model: the templates are the calibration set of the ViT's per-tensor plane scales (vit.py)
            self._calibrate_planes(template_dataset[idx].rgb for idx in range(len(template_dataset)))
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
        if bits & _lib.SPLIT_RANGE_BITS:
            how = self._recover_range(bits, (template_dataset[idx].rgb for idx in range(len(template_dataset))))
            if how == "widened" and dataset_name in self.template_datas:   # _widen_split_range rebuilt every bank, this one included
                return
            if how:
                self.template_datas.pop(dataset_name, None)
                return self.set_template_data(dataset_name)      # once more (re-calibrated planes, or the wide-range kernels); a further trip of the same kind raises
        _lib.raise_status(bits)
        self.onboarding_time = (time.time() - t0) / max(1, len(template_dataset))

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def predict(self, tar_img, tar_mask, tar_K, tar_M, labels, dataset_name, sort_pred_by_inliers=True, exchange_aux=None, live_rows=None):
        """Device-only part of eval_retrieval (gigaPose.py:511-604): crops -> sorted pose hypotheses.
        labels: (B) int tensor of 1-based object labels.  Returns a PandasTensorCollection with
        id_src, score_src, score_pts, tar_pts, src_pts, relScale, relInplane, idx_failed, M,
        ransac_*, scores (B,k), pred_poses (B,k,4,4).  `exchange_aux` (sharded bank only): a (B,) int32 word per crop that
        travels with exchange #1; all ranks' words are in `self.match_banks[dataset_name].last_aux` afterwards.  `live_rows`
        (fixed-size sharded flushes): only the first `live_rows` crops are real, the rest are padding (zero image, zero mask) -- the
        two backbones run on the real rows only and the padding rows get zero features (their zero masks leave them no live
        patch, so nothing downstream reads those features)."""
        bank = self.match_banks[dataset_name]
        template_data = self.template_datas[dataset_name]
        n_obj = template_data.ist_features.shape[0]
        if not labels.is_cuda and labels.numel() and (int(labels.min()) < 1 or int(labels.max()) > n_obj):
            raise IndexError(f"detection label outside 1..{n_obj} (the reference indexes ae_features[label - 1], gigaPose.py:520)")
        if tar_img.is_cuda:
            _lib.status_word(tar_img.device)  # device labels / hand-offs / split range are checked by the kernels (check_status)
        if not labels.is_cuda and tar_img.is_cuda:
            # host labels: the 0-based int32 form every kernel takes is made on the HOST and uploaded once, stream-ordered from pinned
            # memory (a pageable-memory copy would block the host until the stream drains, i.e. a host sync per call that exposes the
            # launch latency of everything after it; converting on the device cost three ATen launches per step)
            labels0 = (labels.to(torch.int64) - 1).to(torch.int32).contiguous().pin_memory().to(tar_img.device, non_blocking=True)
        else:
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
                tar_ist = self._backbone_rows("ist", self.ist_net.forward_by_chunk, tar_img, live_rows)   # stage 4a: IST backbone (once)
        tar_ae = self._backbone_rows("ae", self.ae_net, tar_img, live_rows)            # stage 1: ViT features
        if self.template_shard is None:
            pred = self.testing_metric.test_bank(bank, tar_ae, tar_mask, labels0)  # stage 3: matching
            if side is None:
                tar_ist = self._backbone_rows("ist", self.ist_net.forward_by_chunk, tar_img, live_rows)   # stage 4a: IST backbone (once)
        else:
            # sharded bank: exchange #1 (query features to every rank) travels while the IST backbone runs
            pending = bank.start_exchange(tar_ae, tar_mask, labels0, exchange_aux)
            if side is None:
                tar_ist = self._backbone_rows("ist", self.ist_net.forward_by_chunk, tar_img, live_rows)
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
            tar_label=labels, tar_K=tar_K, tar_M=tar_M, pred_src_views=pred.id_src, pred_M=pred.M, labels0=labels0)
        pred.register_tensor("pred_poses", poses)
        return pred

    def _backbone_rows(self, key, net, tar_img, live_rows):
        """net(tar_img) computed on the first `live_rows` rows only, zero features for the padding rows behind them (an all-padding
        batch does not run the network at all once its output shape is known)."""
        B = tar_img.shape[0]
        if live_rows is None or live_rows >= B:
            out = net(tar_img)
            self._feat_shapes[key] = (tuple(out.shape[1:]), out.dtype)
            return out
        if live_rows == 0 and key in self._feat_shapes:
            shape, dtype = self._feat_shapes[key]
            return torch.zeros((B,) + shape, dtype=dtype, device=tar_img.device)
        feat = net(tar_img[:max(live_rows, 1)])
        self._feat_shapes[key] = (tuple(feat.shape[1:]), feat.dtype)
        out = feat.new_zeros((B,) + tuple(feat.shape[1:]))
        if live_rows > 0:
            out[:live_rows] = feat[:live_rows]
        return out

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
        if bits & _lib.SPLIT_RANGE_BITS and self._recover_range(bits, batch.tar_img):
            return self.eval_retrieval(batch, idx_batch, dataset_name, sort_pred_by_inliers)   # re-calibrated (or re-onboarded wide): run again; a further trip raises
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
        """Reference signature and return value (gigaPose.py:635-642).  With accumulate_crops > 0 the image is queued and the work
        happens at the next flush (see __init__); eval_retrieval itself is unchanged and immediate."""
        if self.image_ownership == "round_robin":
            rank, world = self._image_ranks()
            if idx_batch % world != rank:
                return 0                       # another rank's image (see __init__)
        if self.template_shard is not None and self._flushable(batch):
            # sharded bank: ranks see different images with different detection counts, the exchanges are fixed-size collectives --
            # crop-granular queue, flushes of exactly accumulate_crops rows, collectively agreed end (sharded_flow.py)
            self._flow().push(batch, idx_batch)
            return 0
        if not self._accumulating(batch):
            if self.template_shard is None:
                self.flush_pending()   # keep file order if the mode is switched mid-run
            if self.accumulate_crops == 0 and self.template_shard is None and self._flushable(batch):
                # the reference's flow -- one predict() per image, its file on disk when test_step returns -- through the lean host
                # writer of the flushes (scores / poses downloaded once, selection in numpy) instead of filter_and_save's ~20 device
                # gathers for a `predictions[selected]` nobody reads here: 1.4 ms per image (bench.py: dropin_flow.per_image)
                if self.test_dataset_name not in self.template_datas:
                    self.set_template_data(self.test_dataset_name)
                t0 = time.time()
                job = self._run_flush([(batch, idx_batch, self._bind())], self.test_dataset_name)
                job["ev"][1].synchronize()
                job["wall_s"] = time.time() - t0       # `time` of the npz = this call's wall clock, as the reference measures it
                self._finish_flush(job)
                return 0
            self.eval_retrieval(batch, idx_batch=idx_batch, dataset_name=self.test_dataset_name)
            return 0
        bound = self._bind()
        if self._pending and self._pending[-1][2] != bound:
            self.flush_pending()               # the driver re-pointed the model (dataset / log_dir / setting): what is queued belongs to the old one
        self._pending.append((batch, idx_batch, bound))
        self._pending_crops += len(batch.infos)
        if self._pending_crops >= self.accumulate_crops:
            self._launch_flush()
        return 0

    def _flushable(self, batch):
        return getattr(batch, "test_list", None) is not None and batch.tar_img.is_cuda

    def _accumulating(self, batch):
        return self.accumulate_crops > 0 and self.template_shard is None and self._flushable(batch)

    def _bind(self):
        """What a queued image belongs to, taken when it is queued (a driver may re-point the model between test_steps)."""
        return (self.test_dataset_name, self.log_dir, self.test_setting)

    def _flow(self):
        if self._sharded_flow is None:
            from .sharded_flow import ShardedFlow

            self._sharded_flow = ShardedFlow(self)
        return self._sharded_flow

    def _image_ranks(self):
        """(rank, world) for image ownership and for the end-of-epoch barrier: the shard group if the bank is sharded, else the
        default process group, else a single process."""
        import torch.distributed as dist

        if self.template_shard is not None:
            return self.template_shard[0], self.template_shard[1]
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    @torch.no_grad()
    def _launch_flush(self):
        """Queue ONE predict() over the pending images (whole images, at least one, up to accumulate_crops crops) on the GPU without
        waiting for it, then write the files of the PREVIOUS flush while this one runs."""
        dataset_name = self._pending[0][2][0] if self._pending else self.test_dataset_name
        if dataset_name not in self.template_datas:
            self.set_template_data(dataset_name)
        take, n = [], 0
        while self._pending and (not take or n + len(self._pending[0][0].infos) <= max(self.accumulate_crops, 1)):
            take.append(self._pending.pop(0))
            n += len(take[-1][0].infos)
        self._pending_crops -= n
        job = self._run_flush(take, dataset_name)
        prev, self._in_flight = self._in_flight, job
        if prev is not None:
            self._finish_flush(prev)

    def _run_flush(self, images, dataset_name):
        batches = [im[0] for im in images]
        cat = (lambda name: batches[0].tensors[name]) if len(batches) == 1 else (lambda name: torch.cat([b.tensors[name] for b in batches], dim=0))
        labels_np = np.concatenate([np.asarray(b.infos.label).astype(np.int32) for b in batches])
        job = self._run_rows({n: cat(n) for n in ("tar_img", "tar_mask", "tar_K", "tar_M")}, labels_np, dataset_name)
        job["images"] = images
        return job

    def _run_rows(self, inputs, labels_np, dataset_name, aux=None, live_rows=None):
        """The device half of a flush: ONE predict() over the rows of `inputs`, queued without a host wait, and what the host needs
        afterwards copied to pinned memory BEHIND the kernels.  `aux` (sharded flow only): an int the ranks exchange with this
        flush (sharded_flow.py)."""
        dev = inputs["tar_img"].device
        n = len(labels_np)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        recovery = self.pose_recovery[dataset_name]
        keep_asserts, recovery.check_asserts = recovery.check_asserts, ("deferred" if recovery.check_asserts else False)
        sharded = self.template_shard is not None
        aux_t = None
        if sharded:
            aux_t = torch.full((n,), int(aux or 0), dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
        try:
            pred = self.predict(inputs["tar_img"], inputs["tar_mask"], inputs["tar_K"], inputs["tar_M"], torch.from_numpy(labels_np), dataset_name,
                                exchange_aux=aux_t, live_rows=live_rows)
        finally:
            recovery.check_asserts = keep_asserts
        # what the files and the guard rails need, to pinned host memory BEHIND the kernels (stream order; no host wait here): scores,
        # poses, the crop-transform assert flag (reference lib3d/torch.py:54-55) and this flush's snapshot of the status word, which is
        # cleared in stream order so that the next flush starts from zero
        word = _lib.status_word(dev)
        flag = recovery.deferred_flag if keep_asserts else None
        host = {"scores": torch.empty(pred.scores.shape, dtype=pred.scores.dtype, pin_memory=True),
                "pred_poses": torch.empty(pred.pred_poses.shape, dtype=pred.pred_poses.dtype, pin_memory=True),
                "bad_crop_M": torch.zeros(1, dtype=torch.int32, pin_memory=True)}
        host["scores"].copy_(pred.scores, non_blocking=True)
        host["pred_poses"].copy_(pred.pred_poses, non_blocking=True)
        if sharded:
            # every rank's status word of THIS flush (one 4-byte-per-rank all-gather), so that all ranks take the same recovery
            # decision in the same place; and all ranks' exchange words (they arrived with exchange #1)
            from .sharding import all_gather_rows

            words, _ = all_gather_rows(word.view(1, 1), self.template_shard[2])
            host["status"] = torch.zeros(words.shape[0], dtype=torch.int32, pin_memory=True)
            host["status"].copy_(words.reshape(-1), non_blocking=True)
            aux_all = self.match_banks[dataset_name].last_aux
            host["aux_all"] = torch.zeros(aux_all.shape[0], dtype=torch.int32, pin_memory=True)
            host["aux_all"].copy_(aux_all, non_blocking=True)
        else:
            host["status"] = torch.zeros(1, dtype=torch.int32, pin_memory=True)
            host["status"].copy_(word, non_blocking=True)
        word.zero_()
        if flag is not None:
            host["bad_crop_M"].copy_(flag, non_blocking=True)
        ev1.record()
        return dict(inputs=inputs, labels=labels_np, pred=pred, host=host, ev=(ev0, ev1), dataset_name=dataset_name, device=dev)

    def _clear_crop_flag(self, dataset_name):
        flag = self.pose_recovery[dataset_name]._flag   # the device flag is persistent: clear it before raising
        if flag is not None:
            flag.zero_()

    def _finish_flush(self, job):
        """Wait for a flush, check the guard rails, write one <idx>.npz per image (contents = filter_and_save's, reference
        gigaPose.py:400-449) with `time` = the flush's device time apportioned by crop count."""
        job["ev"][1].synchronize()
        bits = int(job["host"]["status"][0])           # this flush's own bits (snapshot + clear in stream order, _run_rows)
        redone = False
        while bits & _lib.SPLIT_RANGE_BITS and self._recover_range(bits, [im[0].tar_img for im in job["images"]]):
            # a remedy applied (plane scales re-calibrated, or a network moved to its wide kernels): run the flush again.  The loop ends
            # when the flush is clean or no remedy is left (a trip with both the ViT and the IST bit may need both, one per pass)
            if not redone:   # the kernels of the NEXT flush (already queued) ran with the old planes too: redo both, in order
                nxt, self._in_flight = self._in_flight, None
                self._drain_device()
                redone = True
            wall = "wall_s" in job
            t0 = time.time()
            job = self._run_flush(job["images"], job["dataset_name"])
            job["ev"][1].synchronize()
            if wall:
                job["wall_s"] = time.time() - t0       # the per-image flow reports wall clock: the redo's own
            bits = int(job["host"]["status"][0])
        _lib.raise_status(bits)
        if redone and nxt is not None:
            self._in_flight = self._run_flush(nxt["images"], nxt["dataset_name"])
        if int(job["host"]["bad_crop_M"][0]) != 0:   # reference lib3d/torch.py:54-55
            self._clear_crop_flag(job["dataset_name"])
            raise AssertionError("tar_M must be an isotropic scale + translation")
        total_ms = 1e3 * job["wall_s"] if "wall_s" in job else job["ev"][0].elapsed_time(job["ev"][1])
        scores, poses = job["host"]["scores"].numpy(), job["host"]["pred_poses"].numpy()
        n_all, a = len(job["labels"]), 0
        for batch, idx_batch, (_, log_dir, test_setting) in job["images"]:
            n = len(batch.infos)
            save_path = osp.join(log_dir, "predictions", f"{idx_batch}.npz")
            self._save_image(batch.infos, batch.test_list, scores[a:a + n], poses[a:a + n], 1e-3 * total_ms * n / max(n_all, 1), save_path,
                             test_setting == "localization")
            a += n
        pred = job["pred"]
        pred.infos = pd.concat([im[0].infos for im in job["images"]], axis=0, sort=False).reset_index(drop=True)
        self.last_predictions = pred

    @staticmethod
    def _drain_device():
        """Before a redo: let the flush queued behind the tripping one finish and drop whatever bits it raised (it runs again)."""
        torch.cuda.synchronize()
        _lib.take_status()

    @staticmethod
    def _save_image(infos, test_list, scores, poses, time, save_path, keep_only_testing_instances):
        """filter_and_save on host arrays (same selection, same npz fields and dtypes)."""
        labels = np.asarray(infos.label).astype(np.int32)
        assert len(np.unique(labels)) == len(np.unique(test_list.infos.obj_id))
        selected, detection_times = list(range(len(labels))), [0.0] * len(labels)
        if keep_only_testing_instances:
            selected, detection_times = [], []
            for row, obj_id in enumerate(test_list.infos.obj_id):
                num_inst = int(test_list.infos.inst_count[row])
                cand = np.flatnonzero(labels == obj_id)
                best = cand[np.argsort(-scores[cand, 0], kind="stable")[:num_inst]]
                selected.extend(best.tolist())
                detection_times.extend([test_list.infos.detection_time[row]] * num_inst)
        det_t = np.asarray(detection_times, dtype=np.float64)
        sel = np.asarray(selected, dtype=np.int64)
        np.savez(save_path,
                 scene_id=np.asarray(infos.scene_id).astype(np.int32)[sel], im_id=np.asarray(infos.view_id).astype(np.int32)[sel],
                 object_id=labels[sel], time=np.ones_like(det_t) * time, detection_time=det_t,
                 poses=poses[sel], scores=scores[sel])
        return selected

    def flush_pending(self):
        """Run what is queued and write every outstanding file (on_test_epoch_end calls it; a caller that reads the npz files between
        test_steps calls it too).  With a sharded bank this is a COLLECTIVE: every rank of the shard group must call it (the ranks
        agree inside the flushes on when all queues are empty, sharded_flow.py)."""
        if self.template_shard is not None:
            if self._sharded_flow is not None or self.test_dataset_name is not None:
                self._flow().drain()
            return
        while self._pending:
            self._launch_flush()
        if self._in_flight is not None:
            job, self._in_flight = self._in_flight, None
            self._finish_flush(job)
            if self._in_flight is not None:       # a range fallback re-queued the following flush
                job, self._in_flight = self._in_flight, None
                self._finish_flush(job)

    def on_test_epoch_end(self):
        """Merge the per-batch npz files into the BOP csv files (reference gigaPose.py:644-653 ->
        src/utils/inout.py:278-367; here gigapose_amd/inout.py, byte-identical output)."""
        self.flush_pending()
        # The reference wrote every file inside test_step; here each rank's last flushes are written by flush_pending() just above:
        # rank 0 must not start merging before every rank's files are on disk -- one barrier when several ranks run.
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            group = self.template_shard[2] if self.template_shard is not None else None
            if dist.get_world_size(group) > 1:
                dist.barrier(group=group)
        if self.global_rank != 0:
            return
        from .inout import save_predictions_from_batched_predictions

        save_predictions_from_batched_predictions(osp.join(self.log_dir, "predictions"),
                                                  dataset_name=self.test_dataset_name, model_name=self.model_name,
                                                  run_id=self.run_id, is_refined=False)
