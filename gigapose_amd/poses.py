"""2-D similarity voting + 6-D pose recovery behind the reference interfaces
`RANSAC` (src/models/ransac.py:9-172) and `ObjectPoseRecovery` (src/models/poses.py:12-163).
One workgroup per (detection, hypothesis) in libgigapose_hip.so (gp_ransac, gp_recover_poses)
replaces the reference's Python loops over detections x hypotheses."""
import pandas as pd
import torch

from . import _lib
from .tensor_collection import PandasTensorCollection

P = 256


class RANSAC(torch.nn.Module):
    def __init__(self, pixel_threshold, patch_size=14):
        super().__init__()
        self.patch_size = patch_size
        self.pixel_threshold = pixel_threshold

    @torch.no_grad()
    def run(self, src_pts, tar_pts, rel_scale, rel_inplane, scores=None):
        """src_pts/tar_pts (...,256,2) int64, rel_scale (...,256), rel_inplane (...,256,2) [, scores (...,256): the weights of
        RANSAC.forward's `scores` argument, None = ones] ->
        M (...,3,3), failed (...) bool, inlier src/tar pts (...,256,2) int64, inlier scores (...,256) int64."""
        lead = tuple(src_pts.shape[:-2])
        R = 1
        for d in lead:
            R *= d
        dev = src_pts.device
        M = torch.empty(*lead, 3, 3, dtype=torch.float32, device=dev)
        failed = torch.empty(*lead, dtype=torch.uint8, device=dev)
        isrc = torch.empty(*lead, P, 2, dtype=torch.int64, device=dev)
        itar = torch.empty(*lead, P, 2, dtype=torch.int64, device=dev)
        isc = torch.empty(*lead, P, dtype=torch.int64, device=dev)
        weights = None if scores is None else scores.to(dev).float().contiguous()
        if weights is not None and tuple(weights.shape) != tuple(src_pts.shape[:-1]):
            raise ValueError(f"scores must have shape {tuple(src_pts.shape[:-1])}, got {tuple(weights.shape)}")
        _lib.call("gp_ransac_scored", _lib.ptr(src_pts.contiguous()), _lib.ptr(tar_pts.contiguous()),
                  _lib.ptr(rel_scale.contiguous().float()), _lib.ptr(rel_inplane.contiguous().float()), _lib.ptr(weights), _lib.i(R),
                  _lib.f(self.patch_size), _lib.f(self.pixel_threshold), _lib.ptr(M), _lib.ptr(failed),
                  _lib.ptr(isrc), _lib.ptr(itar), _lib.ptr(isc), _lib.stream_ptr())
        return M, failed.view(torch.bool), isrc, itar, isc   # the kernel writes 0 / 1 bytes: a reinterpretation, not a conversion launch

    def forward(self, batch, scores=None, direction="src2tar"):
        """Reference signature (ransac.py:108-121): batch has src_pts / tar_pts (B,P,2), relScale, relInplane -- or, for
        direction="tar2src", the same four with the suffix `_inv`; `scores` (B,P) weights the correspondences (default: ones).
        Inference (poses.py:143) uses neither; both are built so that the class is a full stand-in (round 5)."""
        if direction == "src2tar":
            fields = (batch.src_pts, batch.tar_pts, batch.relScale, batch.relInplane)
        elif direction == "tar2src":
            fields = (batch.src_pts_inv, batch.tar_pts_inv, batch.relScale_inv, batch.relInplane_inv)
        else:
            raise ValueError(f"direction must be 'src2tar' or 'tar2src', got {direction!r}")   # the reference fails later with an UnboundLocalError
        M, failed, isrc, itar, isc = self.run(*fields, scores=scores)
        out = PandasTensorCollection(src_pts=isrc, tar_pts=itar, scores=isc, infos=batch.infos)
        return M, failed, out


class ObjectPoseRecovery(torch.nn.Module):
    def __init__(self, template_K, template_Ms, template_poses, pixel_threshold=14):
        super().__init__()
        self.template_K = template_K.contiguous().float()          # (O,3,3)
        self.template_Ms = template_Ms.contiguous().float()        # (O,N,3,3)
        self.template_poses = template_poses.contiguous().float()  # (O,N,4,4)
        self.ransac = RANSAC(pixel_threshold=pixel_threshold)
        # True: read the crop-transform flag back after every call (a host sync, as the reference's assert is); False: never;
        # "deferred": leave the device flag in self.deferred_flag for a caller that synchronises later (GigaPose's flush pipeline)
        self.check_asserts = True
        self.deferred_flag = None
        self._flag = None   # the kernel only ORs into it; it stays zero unless an assert is about to fire: no per-call fill launch

    @torch.no_grad()
    def forward_recovery(self, tar_label, tar_K, tar_M, pred_src_views, pred_M, labels0=None):
        """tar_label (B) 1-based object labels (reference indexes template tensors with label-1,
        poses.py:111-113); returns pred_poses (B,k,4,4).  labels0 (optional, not in the reference signature): the same labels already
        as a 0-based int32 device tensor (GigaPose.predict has it: saves two conversion launches per step)."""
        B, k = pred_src_views.shape
        O, N = self.template_Ms.shape[:2]
        dev = pred_M.device
        if labels0 is None:
            labels0 = (tar_label.to(dev) - 1).to(torch.int32).contiguous()
        poses = torch.empty(B, k, 4, 4, dtype=torch.float32, device=dev)
        if self._flag is None or self._flag.device != dev:
            self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        flag = self._flag
        _lib.call("gp_recover_poses", _lib.ptr(labels0), _lib.ptr(tar_K.contiguous().float()),
                  _lib.ptr(tar_M.contiguous().float()), _lib.ptr(pred_src_views.contiguous().long()),
                  _lib.ptr(pred_M.contiguous().float()), _lib.ptr(self.template_K), _lib.ptr(self.template_Ms),
                  _lib.ptr(self.template_poses), _lib.i(B), _lib.i(O), _lib.i(N), _lib.i(k), _lib.ptr(poses),
                  _lib.ptr(flag), _lib.stream_ptr())
        if self.check_asserts == "deferred":
            self.deferred_flag = flag
        elif self.check_asserts and B > 0:
            # reference lib3d/torch.py:54-55 asserts the crop transform is isotropic scale + translation
            bad = int(flag.item())
            if bad:
                flag.zero_()
            assert bad == 0, "tar_M must be an isotropic scale + translation"
        return poses

    @torch.no_grad()
    def forward_ransac(self, predictions):
        """predictions: src_pts/tar_pts (B,k,P,2), relScale (B,k,P), relInplane (B,k,P,2); registers
        idx_failed, M, ransac_scores, ransac_src_pts, ransac_tar_pts (poses.py:124-163)."""
        M, failed, isrc, itar, isc = self.ransac.run(predictions.src_pts, predictions.tar_pts,
                                                     predictions.relScale, predictions.relInplane)
        predictions.register_tensor("idx_failed", failed)
        predictions.register_tensor("M", M)
        predictions.register_tensor("ransac_scores", isc)
        predictions.register_tensor("ransac_src_pts", isrc)
        predictions.register_tensor("ransac_tar_pts", itar)
        return predictions
