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


@pytest.mark.parametrize("offset,antialias", [(0.1, False), (0.0, False), (0.1, True), (0.0, True)])
def test_pos_embed_resampling_follows_the_hub_model(offset, antialias):
    """A 518-px checkpoint's 1 + 37*37 table -> 1 + 16*16 as facebookresearch/dinov2's interpolate_pos_encoding does it
    (released models: interpolate_offset = 0.1 -> scale_factor 16.1/37, NOT size=(16,16); antialias off): the load hook
    of gigapose_amd.vit against the numpy restatement of the published arithmetic (oracle/vit_numpy.py), both branches
    of the hub code and both antialias settings."""
    rs = np.random.RandomState(3)
    table = rs.standard_normal((1, 1 + 37 * 37, 24)).astype(np.float32)
    vit = Dinov2ViT(128, 1, 2)
    vit.interpolate_offset, vit.interpolate_antialias = offset, antialias
    got = vit.resample_pos_embed(torch.from_numpy(table)).numpy()
    ref = vit_numpy.interpolate_pos_encoding(table, offset, antialias)
    assert got.shape == (1, 257, 24)
    np.testing.assert_array_equal(got[:, 0], table[:, 0])            # class token row untouched
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)   # torch sums and evaluates the Keys coefficients in f32


def test_hub_offset_and_plain_size_sample_different_positions():
    """The two branches differ by far more than round-off (why VERDICT r1 / ADVICE flagged `size=(16,16)`)."""
    rs = np.random.RandomState(4)
    table = rs.standard_normal((1, 1 + 37 * 37, 8)).astype(np.float32)
    a = vit_numpy.interpolate_pos_encoding(table, 0.1, False)
    b = vit_numpy.interpolate_pos_encoding(table, 0.0, False)
    assert np.abs(a - b).max() > 1e-2


def test_pos_embed_resampling_on_load_and_cache_invalidation():
    vit = Dinov2ViT(128, 1, 2)
    assert vit.interpolate_offset == 0.1 and vit.interpolate_antialias is False   # the released hub models' settings
    sd = vit.state_dict()
    sd["pos_embed"] = torch.randn(1, 1 + 37 * 37, 128)
    vit._packed = "stale"
    parent = torch.nn.Module()
    parent.backbone = vit
    parent.load_state_dict({"backbone." + k: v for k, v in sd.items()})   # through a PARENT, as a Lightning checkpoint load does
    assert tuple(vit.pos_embed.shape) == (1, 257, 128) and vit._packed is None
    np.testing.assert_allclose(vit.pos_embed.detach().numpy(), vit_numpy.interpolate_pos_encoding(sd["pos_embed"].numpy()), atol=1e-5)
    vit._packed = "stale"
    vit.float()
    assert vit._packed is None


def test_ist_packed_weights_invalidate_through_a_parent_load():
    """nn.Module.load_state_dict on a parent never calls a child's load_state_dict override; the folded-BN /
    split-plane copies of the IST net must drop anyway (via _load_from_state_dict / _apply)."""
    from gigapose_amd.ist_net import ISTNet, Regressor, ResNet

    cfg = dict(n_heads=0, input_dim=3, input_size=256, initial_dim=128, block_dims=[128, 192, 256, 512], descriptor_size=256)
    net = ISTNet("resnet", ResNet(cfg), Regressor(256, 256, True, True), 64)
    parent = torch.nn.Module()
    parent.ist_net = net
    net._packed, net.backbone._packed, net.backbone._split = "stale", "stale", "stale"
    parent.load_state_dict(parent.state_dict())
    assert net._packed is None and net.backbone._packed is None and net.backbone._split is None
    net._packed, net.backbone._packed, net.backbone._split = "stale", "stale", "stale"
    parent.float()
    assert net._packed is None and net.backbone._packed is None and net.backbone._split is None
