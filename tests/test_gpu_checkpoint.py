"""GPU: a HUB-FORMAT DINOv2 state dict through the path a Lightning checkpoint takes (VERDICT r5 missing 4 / 5).

The reference builds its backbone with `torch.hub.load("facebookresearch/dinov2", "dinov2_vitl14")` (reference
configs/model/ae_net/dinov2_l.yaml:1-10) and restores `gigaPose_v1.ckpt` through the PARENT module's load_state_dict (keys
`ae_net.dinov2_model.*`, reference src/models/network/ae_net.py:44-47).  The hub model is un-vendored and the image is offline, so the
checkpoint FORMAT is what can be tested: `_HubViT` below is an independent module tree with the hub model's parameter names -- fused
`blocks.N.attn.qkv`, `ls1.gamma` / `ls2.gamma`, `mask_token`, and the released checkpoints' 1 + 37 x 37 `pos_embed` (trained at 518 px)
-- filled from the weights of the HF `Dinov2Model` stand-in.  Its state dict, prefixed as a GigaPose checkpoint prefixes it, goes through
`GigaPose.load_state_dict`; the features must equal, BIT FOR BIT in chain numerics, those of `Dinov2ViT.from_hf` of the same HF weights
whose position table was resampled by the ORACLE's restatement of the hub arithmetic (oracle/vit_numpy.interpolate_pos_encoding).
ViT-S, ViT-B (C = 768: three whole 256-tiles per row panel, 12 heads -- a shape no other test covers) and ViT-L."""
import numpy as np
import pytest
import torch
from torch import nn

from oracle import vit_numpy
from test_oracle_vit import hf_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _HubBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim)      # fused, as facebookresearch/dinov2 layers/attention.py has it
        self.attn.proj = nn.Linear(dim, dim)
        self.ls1 = nn.Module()
        self.ls1.gamma = nn.Parameter(torch.ones(dim))
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)
        self.ls2 = nn.Module()
        self.ls2.gamma = nn.Parameter(torch.ones(dim))


class _HubViT(nn.Module):
    """Parameter names of the hub's DinoVisionTransformer (patch 14, no register tokens), position table for 518 x 518."""

    def __init__(self, dim, depth):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + 37 * 37, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, dim, kernel_size=14, stride=14)
        self.blocks = nn.ModuleList([_HubBlock(dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)


def hub_from_hf(hf, table37):
    """The published correspondence between the two public formats (HF convert_dinov2_to_hf.py, read backwards)."""
    cfg = hf.config
    hub = _HubViT(cfg.hidden_size, cfg.num_hidden_layers)
    sd = hf.state_dict()
    with torch.no_grad():
        hub.cls_token.copy_(sd["embeddings.cls_token"])
        hub.pos_embed.copy_(table37)
        hub.mask_token.copy_(sd["embeddings.mask_token"])
        hub.patch_embed.proj.weight.copy_(sd["embeddings.patch_embeddings.projection.weight"])
        hub.patch_embed.proj.bias.copy_(sd["embeddings.patch_embeddings.projection.bias"])
        hub.norm.weight.copy_(sd["layernorm.weight"])
        hub.norm.bias.copy_(sd["layernorm.bias"])
        for i, blk in enumerate(hub.blocks):
            p = f"encoder.layer.{i}."
            a = p + "attention.attention."
            blk.norm1.weight.copy_(sd[p + "norm1.weight"]); blk.norm1.bias.copy_(sd[p + "norm1.bias"])
            blk.attn.qkv.weight.copy_(torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]]))
            blk.attn.qkv.bias.copy_(torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]]))
            blk.attn.proj.weight.copy_(sd[p + "attention.output.dense.weight"]); blk.attn.proj.bias.copy_(sd[p + "attention.output.dense.bias"])
            blk.ls1.gamma.copy_(sd[p + "layer_scale1.lambda1"])
            blk.norm2.weight.copy_(sd[p + "norm2.weight"]); blk.norm2.bias.copy_(sd[p + "norm2.bias"])
            blk.mlp.fc1.weight.copy_(sd[p + "mlp.fc1.weight"]); blk.mlp.fc1.bias.copy_(sd[p + "mlp.fc1.bias"])
            blk.mlp.fc2.weight.copy_(sd[p + "mlp.fc2.weight"]); blk.mlp.fc2.bias.copy_(sd[p + "mlp.fc2.bias"])
            blk.ls2.gamma.copy_(sd[p + "layer_scale2.lambda1"])
    return hub


@pytest.mark.parametrize("variant", ["dinov2_vits14", "dinov2_vitb14", "dinov2_vitl14"])
@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_hub_format_checkpoint_through_the_parent_load_state_dict(variant, numerics):
    from gigapose_testing import factory
    from gigapose_amd.ae_net import AENet
    from gigapose_amd.vit import VARIANTS, Dinov2ViT

    dim, depth, heads = VARIANTS[variant]
    hf = hf_model(dim, depth, heads, seed=31)
    table37 = 0.02 * torch.randn(1, 1 + 37 * 37, dim, generator=torch.Generator().manual_seed(32))
    hub = hub_from_hf(hf, table37)

    # the product: a GigaPose assembled as the reference's config assembles it, restored from a checkpoint-shaped state dict
    model = factory.build_model(variant, k=5, device=DEV, seed=7, numerics=numerics)
    ckpt = {k: v for k, v in model.state_dict().items() if not k.startswith("ae_net.dinov2_model.")}
    ckpt.update({"ae_net.dinov2_model." + k: v for k, v in hub.state_dict().items()})
    missing, unexpected = model.load_state_dict(ckpt, strict=True)            # nn.Module.load_state_dict of the PARENT, as Lightning calls it
    assert not missing and not unexpected
    vit = model.ae_net.dinov2_model
    assert tuple(vit.pos_embed.shape) == (1, 257, dim) and vit._packed is None and vit.plane_amax is None

    # the yardstick: from_hf of the same weights, position table resampled by the oracle's restatement of the hub arithmetic
    with torch.no_grad():
        hf.embeddings.position_embeddings.copy_(torch.from_numpy(vit_numpy.interpolate_pos_encoding(table37.numpy()).astype(np.float32)))
    ref = AENet(variant, Dinov2ViT.from_hf(hf).set_numerics(numerics).to(DEV), descriptor_size=dim, max_batch_size=64)
    # (the product resamples with torch's bicubic kernel, the oracle in float64 numpy: equal to ~1e-6, tests/test_oracle_vit.py; for the
    # bit-for-bit comparison both models get the SAME table -- the one the load hook produced -- and the oracle's is checked beside it)
    np.testing.assert_allclose(vit.pos_embed.detach().cpu().numpy(), ref.dinov2_model.pos_embed.detach().cpu().numpy(), rtol=0, atol=2e-6)
    with torch.no_grad():
        ref.dinov2_model.pos_embed.copy_(vit.pos_embed)
    ref.dinov2_model.invalidate()

    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(33)).to(DEV)
    got, want = model.ae_net(x), ref(x)
    torch.cuda.synchronize()
    if numerics == "chain":
        assert torch.equal(got, want), "hub-format weights through the parent load differ from from_hf of the same weights"
    else:   # split: the same kernels on the same weights and the same (uncalibrated) plane scales -- equal too
        assert torch.equal(got, want)
    # and against the HF model itself in f32 on the CPU (tolerance: f32 round-off of two different f32 implementations)
    with torch.no_grad():
        hf.embeddings.position_embeddings.copy_(vit.pos_embed.detach().cpu())
        h = hf(pixel_values=x[:2].cpu(), output_hidden_states=True).hidden_states[-1]
    ref_f = torch.nn.functional.normalize(h[:, 1:].permute(0, 2, 1), dim=1).reshape(2, dim, 16, 16)
    np.testing.assert_allclose(got[:2].cpu().numpy(), ref_f.numpy(), rtol=0, atol=3e-5)


def test_vit_base_end_to_end_predict_both_numerics():
    """dinov2_vitb14 through the whole path (onboarding, calibration, predict): chain == split in every discrete output on an easy bank."""
    from gigapose_amd import _lib
    from gigapose_testing import factory

    outs = {}
    for numerics in ("chain", "split"):
        model = factory.build_model("dinov2_vitb14", k=4, device=DEV, seed=3, numerics=numerics)
        tset = factory.TemplateSet(2, 9, seed=60)
        model.template_datasets = {"syn": tset}
        model.set_template_data("syn")
        q = tset.crops(61, 6, DEV)
        p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
        torch.cuda.synchronize()
        _lib.check_status()
        outs[numerics] = {n: v.cpu() for n, v in p.tensors.items()}
        assert tuple(p.pred_poses.shape) == (6, 4, 4, 4) and torch.isfinite(p.pred_poses).all()
    assert torch.equal(outs["chain"]["id_src"][:, 0], outs["split"]["id_src"][:, 0]), "best template differs between the numerics"
