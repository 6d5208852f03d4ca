"""GPU probe: 256x256 single-accumulator stream-K split GEMM (gp_split256.hip) vs f64 and vs the 128-tile kernels."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
dev = "cuda"
lib = _lib.lib()
lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
NB = lib.gp_gemm_split256_workspace_bytes()
ws = torch.zeros(NB // 4, device=dev)
def timeit(fn, iters=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def planes256(W):  # W [n][K] f32
    hi = torch.empty(W.shape, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    _lib.call("gp_split256_weights", _lib.ptr(W), ctypes.c_size_t(W.numel()), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo
torch.manual_seed(0)
M = 16640
for (I, J, K, act_is_b, name, epi) in [(1024, M, 1024, 1, "proj", 3), (4096, M, 1024, 1, "fc1", 2), (4096, M, 1024, 1, "fc1-bias-only", 1), (1024, M, 4096, 1, "fc2", 3), (2048, M, 1024, 1, "qk", 1), (M, 1024, 1024, 0, "v", 4)]:
    nw, na = (I, J) if act_is_b else (J, I)
    W = torch.randn(nw, K, device=dev) * 0.03
    X = torch.randn(K, na, device=dev) * 1.5
    hi, lo = planes256(W)
    bias = torch.randn(max(I, J), device=dev); sc = torch.randn(I, device=dev)
    D = torch.randn(I, J, device=dev); D0 = D.clone()
    def run():
        _lib.call("gp_gemm_split256", _lib.ptr(X), _lib.i(na), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I), _lib.i(J),
                  _lib.i(K), _lib.i(act_is_b), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(D), _lib.i(J), _lib.ptr(ws),
                  ctypes.c_size_t(NB), _lib.stream_ptr())
    D.copy_(D0); run(); torch.cuda.synchronize()
    err = lib.gp_gemm_split256_error(_lib.ptr(ws), _lib.stream_ptr())
    rows = slice(0, 128) if act_is_b else slice(0, 128)
    if act_is_b:
        ref = W[rows].double() @ X.double(); mag = W[rows].double().abs() @ X.double().abs()
        if epi == 3: ref = D0[rows].double() + sc[rows, None].double() * (ref + bias[rows, None].double())
        elif epi == 2: ref = torch.nn.functional.gelu(ref + bias[rows, None].double())
        elif epi == 1: ref = ref + bias[rows, None].double()
    else:
        ref = X[:, rows].double().T @ W.double().T + bias[None, :J].double(); mag = X[:, rows].double().abs().T @ W.double().abs().T
    e = ((D[rows].double() - ref).abs() / mag)
    ms = timeit(run)
    print(f"{name:5s} I={I} J={J} K={K}: {ms:.3f} ms = {2.0*I*J*K/ms/1e9:.0f} TF-equivalent | err/sum|ab| rms {e.pow(2).mean().sqrt().item():.2e} max {e.max().item():.2e} | hand-off error word {err}")

I, J, K = 1024, M, 4096
W = torch.randn(I, K, device=dev) * 0.03; X = torch.randn(K, J, device=dev); hi, lo = planes256(W); D = torch.empty(I, J, device=dev)
out = (ctypes.c_ulonglong * 6)()
for _ in range(2):
    lib.gp_gemm_split256_timing(_lib.ptr(X), J, _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), J, I, J, K, _lib.ptr(ws), out, _lib.stream_ptr())
n = max(1, out[5])
print("fc2 shape, wave 0 of block 100, cycles per k-step: stage %d, k16-0 (reads+24 MFMA+12 loads) %d, k16-1 (reads+24 MFMA) %d, drain %d, barrier %d  (steps %d; one wave's MFMAs occupy the pipe 1536)"
      % (out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[4] / n, n))
