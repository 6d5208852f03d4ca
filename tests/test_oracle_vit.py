"""CPU: numpy ViT restatement (oracle/vit_numpy.py) vs the HF Dinov2Model stand-in, and the
hub<->HF weight conversion of gigapose_amd.vit.Dinov2ViT."""
import numpy as np
import pytest
import torch

from gigapose_amd.vit import Dinov2ViT
from oracle import vit_numpy


def hf_model(dim, depth, heads, seed=0):
    from transformers import Dinov2Config, Dinov2Model

    cfg = Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                       image_size=224, patch_size=14)
    torch.manual_seed(seed)
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():  # make LayerScale / LN affine / biases non-trivial so layout bugs show
        for n, p in m.named_parameters():
            if "lambda1" in n:
                p.copy_(torch.rand_like(p) * 0.5 + 0.75)
            elif n.endswith("bias") or "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    return m


def sd_numpy(vit):
    return {k: v.detach().float().numpy() for k, v in vit.state_dict().items()}


def test_numpy_vit_matches_hf():
    torch.set_num_threads(8)
    hf = hf_model(128, 3, 2, seed=1)
    vit = Dinov2ViT.from_hf(hf)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = hf(pixel_values=x, output_hidden_states=True).hidden_states[-1].numpy()
    got = vit_numpy.forward_x_prenorm(sd_numpy(vit), x.numpy(), 3, 2)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)
    feats = vit_numpy.patch_features(sd_numpy(vit), x.numpy(), 3, 2)
    ref_f = torch.nn.functional.normalize(torch.from_numpy(ref[:, 1:]).permute(0, 2, 1), dim=1).reshape(2, 128, 16, 16)
    np.testing.assert_allclose(feats, ref_f.numpy(), rtol=0, atol=2e-6)


def test_pos_embed_resampling_on_load():
    vit = Dinov2ViT(128, 1, 2)
    sd = vit.state_dict()
    sd["pos_embed"] = torch.randn(1, 1 + 37 * 37, 128)
    vit.load_state_dict(sd)
    assert tuple(vit.pos_embed.shape) == (1, 257, 128)
