"""GPU parity of the implicit-GEMM convolution (gp_conv2d_cm) and the bilinear resize: bit-exact vs
the oracle's fmaf chain on small shapes; full IST ResNet vs the plain PyTorch fp32 statement."""
import numpy as np
import pytest
import torch

from gigapose_testing import synthetic as syn
from oracle import cpu as oracle
from oracle import ist_torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hip_conv(X, W, alpha, beta, res, stride, pad, relu, nchw=False):
    from gigapose_amd import _lib

    Cin, B, H, Wd = X.shape
    Cout, _, KH, KW = W.shape
    K = Cin * KH * KW
    wt = torch.zeros((K + 15) // 16 * 16, Cout)
    wt[:K] = torch.from_numpy(W).reshape(Cout, K).t()
    OH, OW = (H + 2 * pad - KH) // stride + 1, (Wd + 2 * pad - KW) // stride + 1
    Y = torch.empty((B, Cout, OH, OW) if nchw else (Cout, B, OH, OW), device=DEV)
    d = lambda a: None if a is None else torch.from_numpy(a).to(DEV)
    tx, twt, ta, tb, tr = d(X), wt.to(DEV), d(alpha), d(beta), d(res)
    _lib.call("gp_conv2d_cm", _lib.ptr(tx), _lib.ptr(twt), _lib.ptr(Y), _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(tr),
              _lib.i(Cin), _lib.i(B), _lib.i(H), _lib.i(Wd), _lib.i(Cout), _lib.i(KH), _lib.i(KW), _lib.i(stride),
              _lib.i(pad), _lib.i(1 if relu else 0), _lib.i(1 if nchw else 0), _lib.stream_ptr())
    torch.cuda.synchronize()
    return Y.cpu().numpy()


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw,B,bn,res,relu", [
    (3, 128, 7, 2, 3, 64, 1, True, False, True),     # stem shape (K=147 -> padded 160)
    (8, 64, 3, 1, 1, 16, 2, True, True, True),       # BasicBlock conv2 + residual (direct kernel, 16x16 window)
    (24, 128, 3, 1, 1, 32, 2, True, True, True),     # direct kernel, 32x8 window, 3 channel chunks, 2 co tiles
    (16, 64, 3, 1, 1, 64, 1, True, False, True),     # direct kernel, several windows per image (halo borders)
    (16, 192, 3, 2, 1, 32, 1, True, False, True),    # strided block entry, Cout = 3*64
    (24, 64, 1, 2, 0, 32, 1, True, False, False),    # downsample 1x1 stride 2
    (32, 64, 1, 1, 0, 16, 1, False, False, False),   # layer4_outconv (no BN)
])
@pytest.mark.probes
def test_conv_bit_exact_vs_oracle(cin, cout, k, stride, pad, hw, B, bn, res, relu):
    rs = np.random.RandomState(cin * 7 + cout)
    X = rs.standard_normal((cin, B, hw, hw)).astype(np.float32)
    W = (rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    alpha = rs.uniform(0.5, 1.5, cout).astype(np.float32) if bn else None
    beta = rs.standard_normal(cout).astype(np.float32) if bn else None
    oh = (hw + 2 * pad - k) // stride + 1
    R = rs.standard_normal((cout, B, oh, oh)).astype(np.float32) if res else None
    got = hip_conv(X, W, alpha, beta, R, stride, pad, relu)
    ref = oracle.conv2d_cm(X, W, alpha, beta, R, stride, pad, relu)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    # and against torch's convolution (different summation order): tolerance
    t = torch.nn.functional.conv2d(torch.from_numpy(X).permute(1, 0, 2, 3), torch.from_numpy(W), stride=stride, padding=pad)
    t = t.permute(1, 0, 2, 3).numpy()
    if bn:
        t = t * alpha[:, None, None, None] + beta[:, None, None, None]
    if res:
        t = t + R
    if relu:
        t = np.maximum(t, 0)
    np.testing.assert_allclose(got, t, rtol=1e-4, atol=1e-5)
    if not res:
        nchw = hip_conv(X, W, alpha, beta, None, stride, pad, relu, nchw=True)
        np.testing.assert_array_equal(nchw, got.transpose(1, 0, 2, 3))
    if k == 3 and stride == 1:  # the generic im2col-gather kernel must agree with the direct one bit-for-bit
        from gigapose_amd import _lib

        _lib.lib().gp_conv_set_direct(0)
        try:
            generic = hip_conv(X, W, alpha, beta, R, stride, pad, relu)
        finally:
            _lib.lib().gp_conv_set_direct(1)
        np.testing.assert_array_equal(generic.view(np.uint32), got.view(np.uint32))


def test_resize_matches_torch():
    from gigapose_amd import _lib

    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    out = torch.empty(3, 3, 256, 256, device=DEV)
    xd = x.to(DEV)
    _lib.call("gp_resize_bilinear_cm", _lib.ptr(xd), _lib.ptr(out), _lib.i(3), _lib.i(3), _lib.i(224), _lib.i(224),
              _lib.i(256), _lib.stream_ptr())
    ref = torch.nn.functional.interpolate(x, (256, 256), mode="bilinear", align_corners=True).permute(1, 0, 2, 3)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


def test_resnet_hip_vs_torch_reference_and_golden(golden_dir):
    import os
    from test_oracle_pose_ist import build_ist

    g = np.load(os.path.join(golden_dir, "ist.npz"))
    net = build_ist(101)
    tmpl, _ = syn.template_images(102, 2)
    x = torch.from_numpy(np.concatenate([tmpl, tmpl[:1] * 0.5]))  # B=3 (odd batch)
    with torch.no_grad():
        ref = ist_torch.resnet_forward(net.backbone, x).numpy()
    got = net.to(DEV).forward_by_chunk(x.to(DEV)).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-5, np.abs(got - ref).max() / scale
    np.testing.assert_allclose(got[:2], g["resnet_feat"], rtol=1e-4, atol=2e-5 * scale)
