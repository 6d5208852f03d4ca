"""GPU: the split-f16 numerics mode (gp_split.hip, gp_match_tiles_split) -- opt-in, NOT bit-identical to the fmaf
chain.  What is asserted: (1) its error against an f64 reference is not above the chain kernel's own error (i.e.
the mode is f32-equivalent, not reduced precision); (2) ViT features agree with the chain mode to f32 round-off
and with the HF stand-in to the same tolerance as the chain mode; (3) the matcher's index outputs agree with the
chain mode except for a handful of threshold/argmax ties at config-2 size (the agreement rate is printed).
Parity of the split mode with the REFERENCE goldens: tests/test_gpu_matcher.py, tests/test_gpu_e2e.py (both modes)."""
import ctypes

import numpy as np
import pytest
import torch

from gigapose_amd import _lib
from gigapose_testing import synthetic as syn
from oracle import ist_torch
from test_gpu_vit import hip_gemm, run_vit

# stage-level tests of the GEMM machinery: plain-f32 epilogues, A/B switches and the scratch error word live in the probe build
pytestmark = [pytest.mark.gpu, pytest.mark.probes]
DEV = "cuda"


def split_gemm(act, W, act_is_b, epi=0, bias=None, scale=None, res=None):
    """act (K, n_act) f32 k-major, W (n_w, K) f32 [out][in]."""
    from gigapose_amd import _lib
    from gigapose_amd.vit import split_planes

    K = act.shape[0]
    I, J = (W.shape[0], act.shape[1]) if act_is_b else (act.shape[1], W.shape[0])
    hi, lo = split_planes(torch.from_numpy(W).to(DEV))
    ta = torch.from_numpy(act).to(DEV)
    D = torch.empty(I, J, device=DEV)
    tb = None if bias is None else torch.from_numpy(bias).to(DEV)
    ts = None if scale is None else torch.from_numpy(scale).to(DEV)
    tr = None if res is None else torch.from_numpy(res).to(DEV)
    _lib.call("gp_gemm_split", _lib.ptr(ta), _lib.i(act.shape[1]), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I),
              _lib.i(J), _lib.i(K), _lib.i(1 if act_is_b else 0), _lib.i(epi), _lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tr), _lib.i(J),
              _lib.stream_ptr())
    torch.cuda.synchronize()
    return D.cpu().numpy()


@pytest.mark.parametrize("act_is_b", [True, False])
@pytest.mark.parametrize("K", [96, 1024])
def test_split_gemm_error_vs_f64_not_above_the_chain(act_is_b, K):
    rs = np.random.RandomState(80 + K)
    I, J = 256, 384
    n_act, n_w = (J, I) if act_is_b else (I, J)
    act = (rs.standard_normal((K, n_act)) * rs.uniform(0.01, 30, (K, 1))).astype(np.float32)   # wide dynamic range
    W = (rs.standard_normal((n_w, K)) * 0.05).astype(np.float32)
    got = split_gemm(act, W, act_is_b)
    A, B = (np.ascontiguousarray(W.T), act) if act_is_b else (act, np.ascontiguousarray(W.T))
    chain = hip_gemm(A, B)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    e_split, e_chain = np.abs(got - ref) / mag, np.abs(chain - ref) / mag
    print(f"K={K} act_is_b={act_is_b}: max err/sum|ab| split {e_split.max():.2e} chain {e_chain.max():.2e}; "
          f"rms split {np.sqrt((e_split**2).mean()):.2e} chain {np.sqrt((e_chain**2).mean()):.2e}")
    assert e_split.max() < 2.5e-7                                   # a few ulp of f32 relative to sum |a||b|
    assert np.sqrt((e_split ** 2).mean()) <= 1.25 * np.sqrt((e_chain ** 2).mean())
    assert e_split.max() <= 1.5 * e_chain.max()


@pytest.mark.parametrize("epi", [1, 2, 3, 4, 5])
def test_split_gemm_epilogues_match_chain_kernel(epi):
    rs = np.random.RandomState(90 + epi)
    I, J, K = 256, 256, 64
    act = rs.standard_normal((K, J)).astype(np.float32)
    W = rs.standard_normal((I, K)).astype(np.float32)
    bias = rs.standard_normal(J if epi == 4 else I).astype(np.float32)
    scale = rs.standard_normal(I).astype(np.float32)
    res = rs.standard_normal((I, J)).astype(np.float32)
    got = split_gemm(act, W, True, epi, bias, scale, res)
    ref = hip_gemm(np.ascontiguousarray(W.T), act, epi, bias, scale, res)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


def test_split_weights_kernel_equals_host_split():
    from gigapose_amd import _lib
    from gigapose_amd.vit import split_planes

    rs = np.random.RandomState(3)
    K, n = 100, 77                                                   # ragged: exercises the tile guards
    Wt = (rs.standard_normal((K, n)) * rs.uniform(1e-6, 100, (K, 1))).astype(np.float32)
    t = torch.from_numpy(Wt).to(DEV)
    hi = torch.empty(n, K, dtype=torch.float16, device=DEV)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_weights", _lib.ptr(t), _lib.i(K), _lib.i(n), _lib.i(n), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    h2, l2 = split_planes(t.t())
    assert torch.equal(hi, h2) and torch.equal(lo, l2)
    back = hi.float() + lo.float() / 2048.0
    assert ((back - t.t()).abs() <= 2.0 ** -21 * t.t().abs() + 1e-12).all()   # 22 significant bits survive


def test_vit_split_features_vs_chain_and_hf():
    """ViT-S/14 stand-in: unit-norm patch features of the two numerics modes agree to f32 round-off, and the
    split mode is as close to the HF Dinov2Model as the chain mode."""
    hf, vit, x = run_vit(384, 12, 6, 3, seed=70)
    chain = vit.set_numerics("chain").patch_features(x.to(DEV)).cpu()
    split = vit.set_numerics("split").patch_features(x.to(DEV)).cpu()
    with torch.no_grad():
        hs = hf(pixel_values=x, output_hidden_states=True).hidden_states[-1][:, 1:]
    ref = torch.nn.functional.normalize(hs.transpose(1, 2).reshape(3, 384, 16, 16), dim=1)
    d_modes = (chain - split).abs().max().item()
    e_chain, e_split = (chain - ref).abs().max().item(), (split - ref).abs().max().item()
    print(f"ViT-S features: chain vs split {d_modes:.2e}; vs HF: chain {e_chain:.2e}, split {e_split:.2e}")
    assert d_modes < 2e-6 and e_split < 5e-5 and e_split <= 1.5 * e_chain + 1e-6


def test_matcher_split_agreement_at_config2_size():
    """B=64, N=162, C=1024 with planted, graded matches: template ranking identical, patch indices identical up to a
    handful of exact ties at the 0.5 threshold / argmax (f32 round-off class: the reference's own GEMM would differ
    from the fmaf chain in the same way)."""
    from gigapose_amd.matching import LocalSimilarity, MatchBank, patch_grid_mask

    B, N, C = 64, 162, 1024
    case = syn.matcher_case(seed=5, B=B, O=1, N=N, C=C)
    feats, masks = torch.from_numpy(case["src_feats"]).to(DEV), torch.from_numpy(case["src_masks"]).to(DEV)
    qf = torch.from_numpy(case["tar_feat"]).to(DEV)
    qmask = patch_grid_mask(torch.from_numpy(case["tar_mask"]).to(DEV))
    labels = torch.from_numpy(case["labels"]).to(DEV)
    out = {}
    for mode in ["chain", "split"]:
        m = LocalSimilarity(5, 0.5, 3)
        m.numerics = mode
        out[mode] = m.match_tiles(m.normalize(qf), qmask, MatchBank(feats, masks, mode), labels)
    (i0, s0, m0, a0), (i1, s1, m1, a1) = out["chain"], out["split"]
    n_idx, n_mask = (i0 != i1).sum().item(), (m0 != m1).sum().item()
    print(f"matcher chain vs split at B={B} N={N} C={C}: {n_idx} of {i0.numel()} patch ids differ, {n_mask} mask bits differ, "
          f"sim_avg max |diff| {(a0 - a1).abs().max().item():.2e}")
    assert (s0 != 0).sum().item() > 100000                              # the case is not degenerate
    assert n_idx <= 40 and n_mask <= 40                                 # < 2e-5 of the entries
    both = (s0 != 0) & (s1 != 0)
    assert (s0 - s1)[both].abs().max().item() < 6e-6                    # K=1024 f32 chain round-off is ~1e-6 by itself
    # which mode is closer to the truth?  f64 similarities of a few tiles from the SAME normalised f32 features
    qn = LocalSimilarity(5, 0.5, 3).normalize(qf).double()                                   # (B, C, 256)
    bn = MatchBank(feats, masks, "chain").features.double()                                  # (1, N, C, 256)
    e_chain, e_split = [], []
    for b, n in [(0, 0), (5, 17), (33, 100), (63, 161)]:
        sim = qn[b].t() @ bn[0, n]                                                            # (t, s)
        sim = sim * qmask[b].double()[:, None] * patch_grid_mask(masks)[0, n].double()[None, :]
        row = sim.gather(1, i0[b, n].long()[:, None])[:, 0]                                  # f64 value at the chosen s
        ok = both[b, n] & (i0[b, n] == i1[b, n])
        e_chain.append((s0[b, n].double() - row)[ok].abs())
        e_split.append((s1[b, n].double() - row)[ok].abs())
    e_chain, e_split = torch.cat(e_chain), torch.cat(e_split)
    print(f"score error vs f64 over {e_chain.numel()} matched patches: chain rms {e_chain.pow(2).mean().sqrt().item():.2e} "
          f"max {e_chain.max().item():.2e}; split rms {e_split.pow(2).mean().sqrt().item():.2e} max {e_split.max().item():.2e}")
    assert e_split.pow(2).mean().sqrt().item() <= 1.25 * e_chain.pow(2).mean().sqrt().item()
    assert (a0 - a1).abs().max().item() < 2e-3                           # one flipped patch moves sim_avg by <= 1/256
    assert torch.equal(torch.topk(a0, 5, dim=1).indices, torch.topk(a1, 5, dim=1).indices)


def test_split_normalize_planes_reconstruct_unit_vectors():
    from gigapose_amd.matching import LocalSimilarity, normalize_split

    rs = np.random.RandomState(8)
    x = (rs.standard_normal((5, 48, 256)) * rs.uniform(0.1, 30, (5, 1, 256))).astype(np.float32)   # C=48 -> Cp=64
    hi, lo = normalize_split(torch.from_numpy(x).to(DEV))
    assert hi.shape == (5, 256, 64) and (hi[..., 48:] == 0).all() and (lo[..., 48:] == 0).all()
    chain = LocalSimilarity(5, 0.5, 3).normalize(torch.from_numpy(x).to(DEV))            # (5, 48, 256) f32
    back = (hi.float() + lo.float())[..., :48].transpose(1, 2) / 32.0
    assert (back - chain).abs().max().item() < 2e-7


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw,B,res", [(32, 64, 3, 1, 1, 16, 2, True), (64, 192, 3, 2, 1, 16, 2, False),
                                                           (128, 128, 1, 2, 0, 16, 2, False), (32, 256, 1, 1, 0, 8, 2, True)])
def test_split_conv_vs_f64(cin, cout, k, stride, pad, hw, B, res):
    """gp_conv2d_nhwc_split (channel-last f16 planes) against an f64 convolution + BN + residual + ReLU."""
    from gigapose_amd import _lib
    from gigapose_amd.vit import split_planes

    rs = np.random.RandomState(cin + cout + k)
    X = rs.standard_normal((B, hw, hw, cin)).astype(np.float32)                       # NHWC
    Wt = (rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    alpha = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    beta = rs.standard_normal(cout).astype(np.float32)
    oh = (hw + 2 * pad - k) // stride + 1
    R = rs.standard_normal((B, oh, oh, cout)).astype(np.float32) if res else None
    wp = np.zeros(((cout + 127) // 128 * 128, k * k * cin), np.float32)
    wp[:cout] = Wt.transpose(0, 2, 3, 1).reshape(cout, -1)
    xh, xl = split_planes(torch.from_numpy(X).to(DEV))
    wh, wl = split_planes(torch.from_numpy(wp).to(DEV))
    rh, rl = split_planes(torch.from_numpy(R).to(DEV)) if res else (None, None)
    oh_, ol_ = torch.empty(B, oh, oh, cout, dtype=torch.float16, device=DEV), torch.empty(B, oh, oh, cout, dtype=torch.float16, device=DEV)
    of32 = torch.empty(B, cout, oh, oh, device=DEV)
    ta, tb = torch.from_numpy(alpha).to(DEV), torch.from_numpy(beta).to(DEV)
    for out_f32 in (None, of32):
        _lib.call("gp_conv2d_nhwc_split", _lib.ptr(xh), _lib.ptr(xl), _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(ta), _lib.ptr(tb),
                  _lib.ptr(rh), _lib.ptr(rl), _lib.i(B), _lib.i(hw), _lib.i(hw), _lib.i(cin), _lib.i(cout), _lib.i(k), _lib.i(k),
                  _lib.i(stride), _lib.i(pad), _lib.i(1), _lib.ptr(oh_), _lib.ptr(ol_), _lib.ptr(out_f32), _lib.stream_ptr())
    torch.cuda.synchronize()
    t = torch.nn.functional.conv2d(torch.from_numpy(X).double().permute(0, 3, 1, 2), torch.from_numpy(Wt).double(),
                                   stride=stride, padding=pad)
    t = t * torch.from_numpy(alpha).double()[None, :, None, None] + torch.from_numpy(beta).double()[None, :, None, None]
    if res:
        t = t + torch.from_numpy(R).double().permute(0, 3, 1, 2)
    t = torch.relu(t).numpy()
    got_planes = (oh_.float() + ol_.float() / 2048.0).cpu().numpy().transpose(0, 3, 1, 2)
    np.testing.assert_allclose(of32.cpu().numpy(), t, rtol=0, atol=2e-6 * max(1.0, np.abs(t).max()))
    np.testing.assert_allclose(got_planes, t, rtol=0, atol=3e-6 * max(1.0, np.abs(t).max()))


def planes8(t, scale=8.0):
    """f32 tensor -> (hi, lo) planes of scale * t in the single-accumulator convention (gp_split_planes)."""
    t = t.contiguous()
    hi = torch.empty(t.shape, dtype=torch.float16, device=t.device)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(t), ctypes.c_size_t(t.numel()), _lib.f(scale), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw,B,res", [
    (128, 128, 3, 1, 1, 32, 2, True),     # NI = 2 (layer1 shape), 8 whole tiles on few slots
    (128, 128, 3, 1, 1, 56, 8, True),     # NI = 2, 98 tiles on 64 slots: stream-K hand-overs, residual
    (64, 128, 3, 2, 1, 64, 2, False),     # NI = 2, stride 2, Cin = 64
    (128, 128, 3, 1, 1, 16, 3, False),    # 768 pixels = 3 tiles
    (128, 192, 3, 2, 1, 32, 4, False),    # NI = 3, stride 2 (layer2 entry); 4 tiles
    (128, 192, 1, 2, 0, 32, 4, False),    # 1 x 1 stride-2 shortcut, K = 128 (4 k-steps)
    (256, 512, 3, 2, 1, 32, 4, False),    # two co tiles (NI = 4), K = 2304; 8 tiles x 72 steps: few tiles -> cut into k ranges with hand-overs
    (512, 256, 1, 1, 0, 16, 4, False),    # the head: 1 x 1, f32 NCHW output too
    (192, 192, 3, 1, 1, 64, 16, True),    # 256 tiles = one whole tile per slot, residual (halo kernel: 3 x 3 / stride 1 / multiples of 16)
    (256, 256, 3, 1, 1, 32, 4, True),     # halo kernel, NI = 4, 8 channel blocks, residual
    (512, 512, 3, 1, 1, 16, 8, False),    # halo kernel, two co tiles: 16 tiles x 16 channel blocks cut into ranges with hand-overs (layer4)
    (128, 128, 3, 1, 1, 128, 1, True),    # halo kernel, one 128 x 128 image = 64 blocks: halo rows cross block borders, zeros at the image border
    (32, 64, 3, 1, 1, 16, 2, False),      # halo kernel, a single channel block (no halo prefetch), Cout = 64 (one matrix column block)
])
def test_conv_planes_vs_f64(cin, cout, k, stride, pad, hw, B, res):
    """gp_conv2d_planes (gp_conv256.hip) against an f64 convolution + BN + residual + ReLU: both outputs (planes, f32 NCHW)."""
    rs = np.random.RandomState(cin + cout + k + hw)
    X = (rs.standard_normal((B, hw, hw, cin)) * rs.uniform(0.2, 3.0)).astype(np.float32)        # NHWC
    Wt = (rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    alpha = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    beta = rs.standard_normal(cout).astype(np.float32)
    oh = (hw + 2 * pad - k) // stride + 1
    R = rs.standard_normal((B, oh, oh, cout)).astype(np.float32) if res else None
    xh, xl = planes8(torch.from_numpy(X).to(DEV))
    wh, wl = planes8(torch.from_numpy(np.ascontiguousarray(Wt.transpose(0, 2, 3, 1).reshape(cout, -1))).to(DEV), 64.0)
    rh, rl = planes8(torch.from_numpy(R).to(DEV)) if res else (None, None)
    oh_, ol_ = torch.zeros(B, oh, oh, cout, dtype=torch.float16, device=DEV), torch.zeros(B, oh, oh, cout, dtype=torch.float16, device=DEV)
    of32 = torch.zeros(B, cout, oh, oh, device=DEV)
    ta, tb = torch.from_numpy(alpha).to(DEV), torch.from_numpy(beta).to(DEV)
    lib = _lib.lib()
    lib.gp_conv2d_planes_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_conv2d_planes_workspace_bytes()
    ws = torch.zeros(nb // 4, device=DEV)
    _lib.status_word(DEV).zero_()
    for out_f32 in (None, of32):
        _lib.call("gp_conv2d_planes", _lib.ptr(xh), _lib.ptr(xl), _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(rh), _lib.ptr(rl),
                  _lib.i(B), _lib.i(hw), _lib.i(hw), _lib.i(cin), _lib.i(cout), _lib.i(k), _lib.i(k), _lib.i(stride), _lib.i(pad), _lib.i(1),
                  _lib.ptr(oh_), _lib.ptr(ol_), _lib.ptr(out_f32), _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())
    torch.cuda.synchronize()
    _lib.check_status()
    t = torch.nn.functional.conv2d(torch.from_numpy(X).double().permute(0, 3, 1, 2), torch.from_numpy(Wt).double(), stride=stride, padding=pad)
    t = t * torch.from_numpy(alpha).double()[None, :, None, None] + torch.from_numpy(beta).double()[None, :, None, None]
    if res:
        t = t + torch.from_numpy(R).double().permute(0, 3, 1, 2)
    t = torch.relu(t).numpy()
    got_planes = ((oh_.double() + ol_.double()) / 8.0).cpu().numpy().transpose(0, 3, 1, 2)
    tol = max(1.0, np.abs(t).max())
    print(f"conv planes {cin}->{cout} k{k} s{stride} {hw}x{hw} B={B}: max err / max|y| f32 out {np.abs(of32.cpu().numpy() - t).max() / tol:.2e}, planes {np.abs(got_planes - t).max() / tol:.2e}")
    np.testing.assert_allclose(of32.cpu().numpy(), t, rtol=0, atol=2e-6 * tol)
    np.testing.assert_allclose(got_planes, t, rtol=0, atol=3e-6 * tol)


def test_conv_halo_kernel_matches_the_gather_kernel():
    """3 x 3 / stride 1 convolutions take conv_halo_kernel (16 x 16 pixel blocks, halo in LDS, k order channel-block major); the
    same launch through conv_planes_kernel (gp_conv2d_planes_set_halo(0): one gather per tap) must agree to f32 round-off."""
    rs = np.random.RandomState(77)
    lib = _lib.lib()
    lib.gp_conv2d_planes_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_conv2d_planes_workspace_bytes()
    ws = torch.zeros(nb // 4, device=DEV)
    for (cin, cout, hw, B) in [(128, 128, 64, 4), (192, 192, 32, 8), (512, 512, 16, 16)]:
        X = torch.from_numpy(rs.standard_normal((B, hw, hw, cin)).astype(np.float32)).to(DEV)
        Wt = torch.from_numpy((rs.standard_normal((cout, 9 * cin)) / np.sqrt(9 * cin)).astype(np.float32)).to(DEV)
        al, be = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV)
        R = torch.randn(B, hw, hw, cout, device=DEV)
        (xh, xl), (wh, wl), (rh, rl) = planes8(X), planes8(Wt, 64.0), planes8(R)
        outs = []
        for halo in (0, 1):
            lib.gp_conv2d_planes_set_halo(halo)
            of32 = torch.zeros(B, cout, hw, hw, device=DEV)
            oh_, ol_ = torch.zeros(B, hw, hw, cout, dtype=torch.float16, device=DEV), torch.zeros(B, hw, hw, cout, dtype=torch.float16, device=DEV)
            for o32 in (None, of32):
                _lib.call("gp_conv2d_planes", _lib.ptr(xh), _lib.ptr(xl), _lib.ptr(wh), _lib.ptr(wl), _lib.ptr(al), _lib.ptr(be), _lib.ptr(rh), _lib.ptr(rl),
                          _lib.i(B), _lib.i(hw), _lib.i(hw), _lib.i(cin), _lib.i(cout), _lib.i(3), _lib.i(3), _lib.i(1), _lib.i(1), _lib.i(1),
                          _lib.ptr(oh_), _lib.ptr(ol_), _lib.ptr(o32), _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())
            outs.append((of32, oh_.float() + ol_.float()))
        lib.gp_conv2d_planes_set_halo(1)
        torch.cuda.synchronize()
        _lib.check_status()
        scale = outs[0][0].abs().max().item()
        d32, dpl = (outs[0][0] - outs[1][0]).abs().max().item() / scale, (outs[0][1] - outs[1][1]).abs().max().item() / (8 * scale)
        print(f"halo vs gather kernel {cin}->{cout} {hw}x{hw} B={B}: max |diff| / max|y| f32 {d32:.2e}, planes {dpl:.2e}")
        assert d32 < 4e-6 and dpl < 4e-6   # each kernel is within 2e-6 / 3e-6 of float64 (test above)


def test_ist_backbone_split_vs_chain_and_torch():
    """The whole ResNet in split numerics: as close to the torch f32 reference as the chain kernels."""
    from test_oracle_pose_ist import build_ist

    net = build_ist(101)
    tmpl, _ = syn.template_images(102, 2)
    x = torch.from_numpy(np.concatenate([tmpl, tmpl[:1] * 0.5]))      # B=3
    with torch.no_grad():
        ref = ist_torch.resnet_forward(net.backbone.double(), x.double()).numpy()
    net = net.float().to(DEV)
    chain = net.backbone.set_numerics("chain")(x.to(DEV)).cpu().numpy()
    split = net.backbone.set_numerics("split")(x.to(DEV)).cpu().numpy()
    net.backbone.conv_kernel = "128"                                   # first-generation 128 x 128 two-accumulator kernel
    split128 = net.backbone(x.to(DEV)).cpu().numpy()
    net.backbone.conv_kernel = "256"
    scale = np.abs(ref).max()
    e_chain, e_split, e_128 = (np.abs(v - ref).max() / scale for v in (chain, split, split128))
    print(f"IST backbone vs f64 torch: chain {e_chain:.2e}, split {e_split:.2e} (conv_planes_kernel), split-128 {e_128:.2e} (relative to max |feature|)")
    assert e_split < 2e-5 and e_split <= 1.5 * e_chain + 1e-7 and e_128 <= 1.5 * e_chain + 1e-7


def split256_gemm(act, W, act_is_b, epi=0, bias=None, scale=None, res=None):
    """gp_gemm_split256 through the C-ABI.  act (K, n_act) f32 k-major, W (n_w, K) f32 [out][in]."""
    from gigapose_amd import _lib
    from gigapose_amd.vit import split_planes_x64

    lib = _lib.lib()
    lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_gemm_split256_workspace_bytes()
    K = act.shape[0]
    I, J = (W.shape[0], act.shape[1]) if act_is_b else (act.shape[1], W.shape[0])
    hi, lo = split_planes_x64(torch.from_numpy(W).to(DEV))
    ta = torch.from_numpy(act).to(DEV)
    D = torch.from_numpy(res).to(DEV).clone() if epi == 3 else torch.empty(I, J, device=DEV)
    tb = None if bias is None else torch.from_numpy(bias).to(DEV)
    ts = None if scale is None else torch.from_numpy(scale).to(DEV)
    ws = torch.full((nb // 4,), float("nan"), device=DEV)
    for _ in range(2):  # second launch: stale flags of the first must not satisfy it
        if epi == 3:
            D.copy_(torch.from_numpy(res))
        _lib.call("gp_gemm_split256", _lib.ptr(ta), _lib.i(act.shape[1]), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I),
                  _lib.i(J), _lib.i(K), _lib.i(1 if act_is_b else 0), _lib.i(epi), _lib.ptr(tb), _lib.ptr(ts), _lib.ptr(D if epi == 3 else None),
                  _lib.i(J), _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())
    assert lib.gp_gemm_split256_error(_lib.ptr(ws), _lib.stream_ptr()) == 0, "a stream-K hand-off timed out"
    return D.cpu().numpy()


@pytest.mark.parametrize("shape", ["proj", "v", "qk"])
def test_split256_gemm_vs_f64_and_vs_the_128_tile_kernel(shape):
    """ViT-L shapes at B=64 (4 x 65, 65 x 4, 8 x 65 tiles of 256 x 256 on 256 slots: every slot hands a partial tile
    over): error vs f64 at the level of the two-accumulator kernel, epilogues included."""
    rs = np.random.RandomState(17)
    M = 65 * 256
    I, J, K, epi, act_is_b = {"proj": (1024, M, 96, 3, True), "v": (M, 1024, 64, 4, False), "qk": (2048, M, 160, 1, True)}[shape]
    n_act, n_w = (J, I) if act_is_b else (I, J)
    act = (rs.standard_normal((K, n_act)) * rs.uniform(0.05, 20, (K, 1))).astype(np.float32)
    W = (rs.standard_normal((n_w, K)) * 0.03).astype(np.float32)
    bias = rs.standard_normal(J if epi == 4 else I).astype(np.float32)
    scale = rs.standard_normal(I).astype(np.float32)
    res = rs.standard_normal((I, J)).astype(np.float32)
    # (1) the contraction itself against f64, next to the two-accumulator 128-tile kernel
    got = split256_gemm(act, W, act_is_b)
    ref128 = split_gemm(act, W, act_is_b)
    rows = np.r_[0:48, I // 2:I // 2 + 16, I - 48:I]
    A = (W.T if act_is_b else act).astype(np.float64)[:, rows]
    B = (act if act_is_b else W.T).astype(np.float64)
    ref, mag = A.T @ B, np.abs(A).T @ np.abs(B)
    e256, e128 = np.abs(got[rows] - ref) / mag, np.abs(ref128[rows] - ref) / mag
    print(f"{shape}: err vs f64 / sum|ab|: 256-tile rms {np.sqrt((e256**2).mean()):.2e} max {e256.max():.2e}; "
          f"128-tile rms {np.sqrt((e128**2).mean()):.2e} max {e128.max():.2e}")
    assert e256.max() < 2.5e-7 and np.sqrt((e256 ** 2).mean()) <= 2.0 * np.sqrt((e128 ** 2).mean()) + 1e-9
    # (2) the fused epilogue (in-place residual for epi 3) against the 128-tile kernel's, whole output
    got = split256_gemm(act, W, act_is_b, epi, bias, scale, res)
    ref128 = split_gemm(act, W, act_is_b, epi, bias, scale, res)
    np.testing.assert_allclose(got, ref128, rtol=3e-5, atol=3e-5)


def test_vit_large_split_uses_256_tiles_and_matches_chain():
    """ViT-L/14, B=64 (BASELINE config 2): the split forward (256-tile stream-K GEMMs) vs the chain forward."""
    from gigapose_amd.vit import Dinov2ViT

    torch.manual_seed(0)
    vit = Dinov2ViT.from_name("dinov2_vitl14")
    for p in vit.parameters():
        torch.nn.init.normal_(p, std=0.02)
    vit = vit.to(DEV)
    x = torch.randn(64, 3, 224, 224, device=DEV)
    chain = vit.set_numerics("chain").patch_features(x)
    split = vit.set_numerics("split").patch_features(x)
    d = (chain - split).abs().max().item()
    print(f"ViT-L B=64 unit-norm features chain vs split (256-tile GEMMs): max |diff| {d:.2e}")
    assert d < 2e-6
    # the default split forward at this size = activation planes + ping-pong plane x plane GEMMs + attention in split
    # numerics (mode 2).  Mode 1 keeps the f32 attention: the same values in the same order as the f32-activation
    # lock-step 256-tile kernels (mode 0) -- bit for bit on the tiled rows; the 64 ragged-edge tokens (strip_phase: K
    # summed in eight slices) differ by f32 round-off, which attention spreads to every token at the 1e-8 level.
    lib = _lib.lib()
    try:
        lib.gp_vit_set_planes(0)
        lockstep = vit.patch_features(x)
        lib.gp_vit_set_planes(1)
        planes_f32_attention = vit.patch_features(x)
    finally:
        lib.gp_vit_set_planes(2)
    d1 = (planes_f32_attention - lockstep).abs().max().item()
    print(f"ViT-L B=64 planes (ragged strip) vs f32-activation lock-step split forward: max |diff| {d1:.2e}")
    assert d1 < 2e-7
    d2 = (split - lockstep).abs().max().item()
    print(f"ViT-L B=64 split attention vs f32 attention (both split GEMMs): max |diff| {d2:.2e}")
    assert d2 < 1e-6


def test_attention_split_matches_f64():
    """attention_split_kernel (Q | K | V planes -> output planes) vs float64 softmax attention on the same plane values."""
    torch.manual_seed(11)
    B, H = 3, 6
    C = 64 * H
    M = B * 257
    Mpad = (M + 255) // 256 * 256
    qkv = torch.zeros(Mpad, 3 * C, device=DEV)
    qkv[:M] = torch.randn(M, 3 * C, device=DEV) * torch.tensor([1.5] * (2 * C) + [1.0] * C, device=DEV)
    hi = torch.empty(Mpad, 3 * C, dtype=torch.float16, device=DEV)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(qkv), ctypes.c_size_t(qkv.numel()), _lib.f(8.0), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    ohi = torch.zeros(Mpad, C, dtype=torch.float16, device=DEV)
    olo = torch.zeros_like(ohi)
    _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad),
              _lib.f(8.0), _lib.stream_ptr())
    torch.cuda.synchronize()
    got = ((ohi.double() + olo.double()) / 8.0)[:M].view(B, 257, H, 64)
    x = ((hi.double() + lo.double()) / 8.0)[:M].view(B, 257, 3, H, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    err256 = (got[:, 256] - ref[:, 256]).abs().max().item() / ref.abs().max().item()   # query 256: the vector-ALU path
    print(f"split attention vs f64: max |err| / max |ref| = {err:.2e} (query 256 alone {err256:.2e})")
    assert err < 2e-6, err
    assert torch.count_nonzero(ohi[M:]) == 0  # pad rows untouched


def planes256_gemm(A, Bm, epi, bias=None, scale=None, res=None, a_scale=64.0, b_scale=8.0, j_valid=None):
    """A [I][K], Bm [J][K] f32 -> D[i][j] through gp_split_planes + gp_gemm_planes256 (epi 6: returns the (hi, lo) planes O[j][i]).
    j_valid: rows of Bm that carry data (gp_gemm_planes256_ragged: tiles below floor(j_valid / 256) * 256, strip above)."""
    lib = _lib.lib()
    lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_gemm_split256_workspace_bytes()
    ws = torch.zeros(nb // 4, device=DEV)

    def planes(W, sc):
        hi = torch.empty(W.shape, dtype=torch.float16, device=DEV)
        lo = torch.empty_like(hi)
        _lib.call("gp_split_planes", _lib.ptr(W), ctypes.c_size_t(W.numel()), _lib.f(sc), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
        return hi, lo

    (ahi, alo), (bhi, blo) = planes(A, a_scale), planes(Bm, b_scale)
    I, J, K = A.shape[0], Bm.shape[0], A.shape[1]
    D = res.clone() if res is not None else torch.zeros(I, J, device=DEV)
    ohi = torch.zeros(J, I, dtype=torch.float16, device=DEV)
    olo = torch.zeros_like(ohi)
    _lib.call("gp_gemm_planes256_scaled", _lib.ptr(ahi), _lib.ptr(alo), _lib.ptr(bhi), _lib.ptr(blo), _lib.ptr(D), _lib.i(J), _lib.ptr(ohi),
              _lib.ptr(olo), _lib.i(I), _lib.i(I), _lib.i(J), _lib.i(J if j_valid is None else j_valid), _lib.i(K), _lib.i(epi), _lib.ptr(bias),
              _lib.ptr(scale), _lib.ptr(D), _lib.i(J), _lib.f(1.0 / (a_scale * b_scale)), _lib.f(8.0), _lib.ptr(None), _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())
    torch.cuda.synchronize()
    if hasattr(lib, "gp_gemm_split256_error"):   # the scratch's error word: probe library only (the product raises through the status word)
        assert lib.gp_gemm_split256_error(_lib.ptr(ws), _lib.stream_ptr()) == 0
    return (ohi, olo) if epi in (6, 7) else D


@pytest.mark.parametrize("I,J,K", [(4096, 4096, 64), (2048, 8192, 96), (4352, 4096, 32), (8192, 8448, 64)])
def test_planes256_gemm_matches_f64(I, J, K):
    """Ping-pong plane x plane GEMM (short k loops: 1-3 steps per segment, ragged stream-K ranges; the last shape has 132 tiles
    per XCD = 3 data-parallel rounds + a stream-K remainder) vs f64."""
    torch.manual_seed(I + K)
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV)
    D = planes256_gemm(A, Bm, 0)
    ref = A.double() @ Bm.double().t()
    mag = A.double().abs() @ Bm.double().abs().t()
    e = ((D.double() - ref).abs() / mag).max().item()
    assert e < 2e-6, e


def test_planes256_gemm_epilogues():
    torch.manual_seed(5)
    I, J, K = 4096, 4096, 128
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV)
    bias = torch.randn(max(I, J), device=DEV)
    scale = torch.randn(I, device=DEV)
    res = torch.randn(I, J, device=DEV)
    base = (A.double() @ Bm.double().t())
    tol = dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(planes256_gemm(A, Bm, 1, bias).cpu().double(), (base + bias[:I, None].double()).cpu(), **tol)
    np.testing.assert_allclose(planes256_gemm(A, Bm, 4, bias).cpu().double(), (base + bias[None, :J].double()).cpu(), **tol)
    np.testing.assert_allclose(planes256_gemm(A, Bm, 5, bias).cpu().double(), torch.relu(base + bias[:I, None].double()).cpu(), **tol)
    gelu = torch.nn.functional.gelu(base + bias[:I, None].double())
    np.testing.assert_allclose(planes256_gemm(A, Bm, 2, bias).cpu().double(), gelu.cpu(), **tol)
    np.testing.assert_allclose(planes256_gemm(A, Bm, 3, bias, scale, res).cpu().double(),
                               (res.double() + scale[:, None].double() * (base + bias[:I, None].double())).cpu(), **tol)
    ohi, olo = planes256_gemm(A, Bm, 6, bias)   # GELU output as activation planes O[j][i] = 8 x, hi + lo
    back = (ohi.double() + olo.double()) / 8.0
    np.testing.assert_allclose(back.t().cpu(), gelu.cpu(), **tol)
    ohi, olo = planes256_gemm(A, Bm, 7, bias)   # bias only, as planes (the Q | K | V producer)
    back = (ohi.double() + olo.double()) / 8.0
    np.testing.assert_allclose(back.t().cpu(), (base + bias[:I, None].double()).cpu(), **tol)


def test_gelu_epilogue_on_a_grid_vs_float64():
    """The GELU of the plane kernels' epilogues (gp_common.h: gp_gelu_scaled, erfc = 2^-Q(|x|), one fma for both signs) evaluated on a grid:
    acc = 1 * x exactly (A = e_0, B = the grid in column 0), so D[i][j] = GELU(x~_j) with x~ the 22-bit plane value of x.  Both the scalar form
    (f32 epilogue 2) and the packed form inside the plane epilogue 6; the bound is |error| <= 7e-8 (|x| + 1) -- the f32 rounding of the result
    alone is 4.8e-8 there -- plus, for the planes, their own 2^-21 relative representation."""
    torch.manual_seed(11)
    I, J, K = 4096, 4096, 32          # 256 whole tiles: the data-parallel launch of the ViT
    xs = torch.cat([torch.linspace(-12.0, 12.0, J - 16, device=DEV, dtype=torch.float64).float(),
                    torch.tensor([0.0, -0.0, 1e-30, -1e-30, 9.0, -9.0, 9.5, -9.5, 50.0, -50.0, 1000.0, -1000.0, 1e-4, -1e-4, 0.5, -0.5], device=DEV)])
    A = torch.zeros(I, K, device=DEV)
    A[:, 0] = 1.0
    Bm = torch.zeros(J, K, device=DEV)
    Bm[:, 0] = xs
    hi = (xs * 8.0).half()
    x22 = (hi.double() + (xs * 8.0 - hi.float()).half().double()) / 8.0          # what the planes carry
    want = x22 * 0.5 * (1.0 + torch.erf(x22 / 2.0 ** 0.5))
    far = x22 < -5.0                                                               # 1 + erf cancels in float64 out there: erfc
    want[far] = x22[far] * 0.5 * torch.special.erfc(-x22[far] / 2.0 ** 0.5)
    bias = torch.zeros(I, device=DEV)
    D = planes256_gemm(A, Bm, 2, bias).double()                                    # f32 out, scalar form
    e2 = ((D - want[None, :]).abs() / (x22.abs()[None, :] + 1.0)).max().item()
    ohi, olo = planes256_gemm(A, Bm, 6, bias)                                      # planes out (x 8), packed form
    back = (ohi.double() + olo.double()) / 8.0                                     # [J][I]
    e6 = (((back - want[:, None]).abs() - 2.0 ** -21 * want.abs()[:, None]).clamp_min(0.0) / (x22.abs()[:, None] + 1.0)).max().item()
    print(f"GELU epilogues vs float64 on [-12, 12] + extremes: max |err| / (|x| + 1): f32 out {e2:.2e}, planes out (beyond their 2^-21) {e6:.2e}")
    assert e2 < 7e-8 and e6 < 7e-8
    assert torch.equal(D[0], D[I - 1]) and torch.isfinite(D).all()
    assert (D[0][x22 >= 9.0] == x22[x22 >= 9.0]).all() and (D[0][x22 <= -9.0].abs() < 1e-15).all()   # clamped tails: x and -|x| 2^-63


@pytest.mark.parametrize("I,J,jv,K", [(1024, 16640, 16448, 64), (2048, 8448, 8224, 96), (1024, 16640, 16385, 32), (4096, 4352, 4350, 64),
                                      (1024, 16640, 16448, 1024)])
def test_planes256_ragged_rows(I, J, jv, K):
    """257 tokens per crop: the rows above floor(J_valid / 256) * 256 are computed as 32 x 32 fragments (strip_phase: K split
    over the eight waves, partial sums added in wave order).  Tiled rows must be BIT-identical to the fully tiled launch;
    strip rows agree with it to f32 round-off (different summation order over K) and with f64; rows >= round_up(J_valid, 32)
    stay untouched.  (1024, 16640, 16448, .) is ViT-L at B = 64: 256 whole tiles, no hand-over."""
    torch.manual_seed(I + jv)
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV) * 1.3
    bias, scale = torch.randn(I, device=DEV), torch.randn(I, device=DEV)
    res = torch.randn(I, J, device=DEV)
    jm, top = jv // 256 * 256, (jv + 31) // 32 * 32
    mag = (A.double().abs() @ Bm[jm:jv].double().abs().t())              # sum |a||b| per strip output
    for epi in (0, 3):
        full = planes256_gemm(A, Bm, epi, bias, scale, res)
        rag = planes256_gemm(A, Bm, epi, bias, scale, res, j_valid=jv)
        assert torch.equal(rag[:, :jm], full[:, :jm]), f"epilogue {epi}: tiled columns differ"
        sc = scale[:, None].abs().double() if epi == 3 else 1.0
        diff = (rag[:, jm:jv].double() - full[:, jm:jv].double()).abs()
        bound = 4e-7 * sc * mag + 2.4e-7 * full[:, jm:jv].double().abs()    # summation order + 2 ulp of the stored value
        assert (diff <= bound).all(), f"epilogue {epi}: strip columns differ from the tiled result by up to {(diff / bound).max().item():.1f} x the bound"
        assert torch.equal(rag[:, top:], res[:, top:]), f"epilogue {epi}: columns beyond the strip were written"
    ref = A.double() @ Bm[jm:jv].double().t()
    rag0 = planes256_gemm(A, Bm, 0, bias, scale, torch.zeros_like(res), j_valid=jv)
    e64 = ((rag0[:, jm:jv].double() - ref).abs() / mag).max().item()
    print(f"ragged I={I} J_valid={jv} K={K}: strip vs f64 max err / sum|a||b| = {e64:.2e}")
    assert e64 < 4e-7
    for epi in (6, 7):
        fh, fl = planes256_gemm(A, Bm, epi, bias)
        rh, rl = planes256_gemm(A, Bm, epi, bias, j_valid=jv)
        assert torch.equal(rh[:jm], fh[:jm]) and torch.equal(rl[:jm], fl[:jm]), f"epilogue {epi}: tiled rows differ"
        v_f = fh[jm:jv].double() + fl[jm:jv].double()
        v_r = rh[jm:jv].double() + rl[jm:jv].double()
        assert ((v_f - v_r).abs() <= 8.0 * 4e-7 * mag.t() + 1e-5 * v_f.abs()).all(), f"epilogue {epi}: strip rows differ"
        assert not rh[top:].any() and not rl[top:].any()


# ViT-L shapes below 64 crops: fewer 256 x 256 tiles than slots -> the slots of a tile split its K in PARALLEL and the slot with
# the last range adds the published partial accumulators (gp_split256.hip).  (I, J, J_valid, K): proj / fc2 / q|k|v / fc1 at
# B = 16 (J_valid = 4112), proj at B = 8 and B = 33, a 9-tile problem (one XCD holds two tiles, seven hold one)
PAR_SHAPES = [(1024, 4352, 4112, 1024), (1024, 4352, 4112, 4096), (3072, 4352, 4112, 1024), (4096, 4352, 4112, 1024),
              (1024, 2304, 2056, 1024), (1024, 8704, 8481, 1024), (768, 768, 768, 1024),
              (1024, 8448, 8224, 1024), (1024, 8448, 8224, 4096),   # proj / fc2 at B = 32: two slots per tile
              (3072, 2304, 2056, 1024), (4096, 2304, 2056, 1024)]   # q|k|v / fc1 at B = 8: 256 x 128 tiles, unsplit (below)


@pytest.mark.parametrize("I,J,jv,K", [(1024, 2304, 2056, 1024), (1024, 2304, 2056, 4096), (3072, 2304, 2056, 1024), (4096, 2304, 2056, 1024),
                                      (1024, 4352, 4112, 4096)])
def test_planes256_half_width_tiles_vs_full_width(I, J, jv, K):
    """Launches whose 256 x 256 tiles fill at most half the slots take 256 x 128 tiles (epilogues 3 / 6 / 7; ViT-L at 8 crops: all four
    GEMMs of a layer, at 16: proj / fc2).  Same products, a different partition of K over the slots of a tile: the two forms agree to the
    summation-order bound of the strip test, each is deterministic, and the A/B hook selects between them."""
    torch.manual_seed(I + jv + K + 1)
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV) * 1.3
    bias, scale = torch.randn(I, device=DEV), torch.randn(I, device=DEV)
    res = torch.randn(I, J, device=DEV)
    mag = A.double().abs() @ Bm[:jv].double().abs().t()
    lib = _lib.lib()
    out = {}
    try:
        for half in (1, 0):
            lib.gp_gemm_planes256_set_half_tiles(half)
            d3 = planes256_gemm(A, Bm, 3, bias, scale, res, j_valid=jv)
            assert torch.equal(d3, planes256_gemm(A, Bm, 3, bias, scale, res, j_valid=jv)), "two launches differ"
            out[half] = (d3, planes256_gemm(A, Bm, 6, bias, j_valid=jv), planes256_gemm(A, Bm, 7, bias, j_valid=jv))
    finally:
        lib.gp_gemm_planes256_set_half_tiles(1)
    d_h, d_f = out[1][0][:, :jv].double(), out[0][0][:, :jv].double()
    bound = 4e-7 * scale[:, None].abs().double() * mag + 2.4e-7 * d_f.abs()
    assert ((d_h - d_f).abs() <= bound).all(), f"epilogue 3: {((d_h - d_f).abs() / bound).max().item():.2f} x the bound"
    n_diff = int((d_h != d_f).sum())
    print(f"half-width tiles I={I} J_valid={jv} K={K}: epilogue 3 differs from the 256-wide form in {n_diff} of {d_f.numel()} values (round-off)")
    for k, epi in ((1, 6), (2, 7)):
        v_h = (out[1][k][0][:jv].double() + out[1][k][1][:jv].double()).t()
        v_f = (out[0][k][0][:jv].double() + out[0][k][1][:jv].double()).t()
        assert ((v_h - v_f).abs() <= 8.0 * 4e-7 * mag + 1e-5 * v_f.abs()).all(), f"epilogue {epi}"


@pytest.mark.parametrize("I,J,jv,K", PAR_SHAPES)
def test_planes256_parallel_split_k(I, J, jv, K):
    """Fewer tiles than slots: every epilogue against float64 on the tiled columns AND the strip, run twice -- the partial
    accumulators are added in a fixed order, so two launches must agree bit for bit."""
    torch.manual_seed(I + jv + K)
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV) * 1.3
    bias, scale = torch.randn(I, device=DEV), torch.randn(I, device=DEV)
    res = torch.randn(I, J, device=DEV)
    top = (jv + 31) // 32 * 32
    ref = A.double() @ Bm[:jv].double().t()
    mag = A.double().abs() @ Bm[:jv].double().abs().t()
    d0 = planes256_gemm(A, Bm, 0, bias, scale, torch.zeros_like(res), j_valid=jv)
    e64 = ((d0[:, :jv].double() - ref).abs() / mag).max().item()
    print(f"parallel split-K I={I} J_valid={jv} K={K}: max err / sum|a||b| vs f64 = {e64:.2e}")
    assert e64 < 4e-7
    assert torch.equal(d0, planes256_gemm(A, Bm, 0, bias, scale, torch.zeros_like(res), j_valid=jv)), "two launches differ"
    d3 = planes256_gemm(A, Bm, 3, bias, scale, res, j_valid=jv)
    want = res[:, :jv].double() + scale[:, None].double() * (ref + bias[:, None].double())
    assert ((d3[:, :jv].double() - want).abs() <= 4e-7 * scale[:, None].abs().double() * (mag + bias[:, None].abs().double()) + 2.4e-7 * want.abs()).all()
    assert torch.equal(d3[:, top:], res[:, top:]), "columns beyond the strip were written"
    for epi in (6, 7):
        oh, ol = planes256_gemm(A, Bm, epi, bias, j_valid=jv)
        x = ref + bias[:, None].double()
        want = torch.nn.functional.gelu(x) if epi == 6 else x
        got = (oh[:jv].double() + ol[:jv].double()).t() / 8.0
        assert ((got - want).abs() <= 1.5e-6 * (mag + bias[:, None].abs().double()) + 1e-6 * want.abs()).all(), f"epilogue {epi}"
        assert not oh[top:].any() and not ol[top:].any()
