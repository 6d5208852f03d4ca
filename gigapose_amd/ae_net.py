"""`AENet` behind the reference interface (src/models/network/ae_net.py:18-73; Hydra target
configs/model/ae_net/dinov2_l.yaml): chunk to <= max_batch_size crops, run the DINOv2 backbone,
drop CLS, (b,256,C) -> (b,C,16,16), L2-normalise over C.  On MI355X all of that is one C-ABI
call per chunk (gp_vit_forward; the rearrange + F.normalize are its epilogue)."""
import torch
from torch import nn

from .vit import Dinov2ViT, VARIANTS

descriptor_sizes = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024}


def as_hip_backbone(model, model_name=None):
    """Accept what the reference passes as `dinov2_model`: our Dinov2ViT, a hub-style DINOv2
    module (state_dict with blocks.{i}.attn.qkv...), or a transformers.Dinov2Model."""
    if isinstance(model, Dinov2ViT):
        return model
    if model is None:
        return Dinov2ViT.from_name(model_name)
    if hasattr(model, "config") and hasattr(model, "embeddings"):
        return Dinov2ViT.from_hf(model)
    sd = model.state_dict()
    if "blocks.0.attn.qkv.weight" in sd:
        dim = sd["cls_token"].shape[-1]
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        vit = Dinov2ViT(dim, depth, dim // 64)
        vit.load_state_dict(sd, strict=False)
        return vit.eval()
    raise TypeError(f"cannot convert {type(model).__name__} into the HIP DINOv2 backbone")


class AENet(nn.Module):
    def __init__(self, model_name, dinov2_model, descriptor_size, max_batch_size, patch_size=14, **kwargs):
        super().__init__()
        if patch_size != 14:
            raise NotImplementedError("kernels are specialised for patch_size 14 @ 224x224")
        self.model_name = model_name
        self.dinov2_model = as_hip_backbone(dinov2_model, model_name)
        self.descriptor_size = descriptor_size
        self.max_batch_size = max_batch_size
        self.patch_size = patch_size
        assert self.dinov2_model.dim == descriptor_size

    def compute_features(self, images):
        return self.dinov2_model.forward_features(images)

    @torch.no_grad()
    def forward_by_chunk(self, processed_rgbs, patch_dim=(2, 3)):
        outs = [self.dinov2_model.patch_features(processed_rgbs[s:s + self.max_batch_size], normalize=True)
                for s in range(0, processed_rgbs.shape[0], self.max_batch_size)]
        if not outs:
            return torch.empty(0, self.descriptor_size, 16, 16, device=processed_rgbs.device)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, images):
        return self.forward_by_chunk(images)
