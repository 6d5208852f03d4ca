"""GPU probe: ping-pong plane x plane split GEMM (gemm_planes256_kernel) vs the lock-step 256 kernel (bitwise) and f64."""
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
def planes(W, scale):
    hi = torch.empty(W.shape, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(W), ctypes.c_size_t(W.numel()), _lib.f(scale), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo
torch.manual_seed(0)
M = 16640
only = sys.argv[1:] or None
for (nw, K, w_is_a, name, epi) in [(1024, 1024, 1, "proj", 3), (4096, 1024, 1, "fc1", 2), (4096, 1024, 1, "fc1-planes", 6),
                                   (1024, 4096, 1, "fc2", 3), (2048, 1024, 1, "qk", 1), (1024, 1024, 0, "v", 4)]:
    if only and name not in only: continue
    W = torch.randn(nw, K, device=dev) * 0.03          # [out][in]
    Xt = torch.randn(M, K, device=dev) * 1.5           # token-major activations
    Xk = Xt.t().contiguous()                           # k-major copy for the lock-step kernel
    whi, wlo = planes(W, 64.0)
    xhi, xlo = planes(Xt, 8.0)
    I, J = (nw, M) if w_is_a else (M, nw)
    bias = torch.randn(max(I, J), device=dev); sc = torch.randn(I, device=dev)
    D0 = torch.randn(I, J, device=dev)
    Dn, Do = D0.clone(), D0.clone()
    ohi = torch.zeros(M, nw, dtype=torch.float16, device=dev); olo = torch.zeros_like(ohi)
    a_hi, a_lo, b_hi, b_lo = (whi, wlo, xhi, xlo) if w_is_a else (xhi, xlo, whi, wlo)
    def run_new():
        _lib.call("gp_gemm_planes256_scaled", _lib.ptr(a_hi), _lib.ptr(a_lo), _lib.ptr(b_hi), _lib.ptr(b_lo), _lib.ptr(Dn), _lib.i(J),
                  _lib.ptr(ohi), _lib.ptr(olo), _lib.i(nw), _lib.i(I), _lib.i(J), _lib.i(J), _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc),
                  _lib.ptr(Dn), _lib.i(J), _lib.f(1.0 / 512.0), _lib.f(8.0), _lib.ptr(None), _lib.ptr(ws), ctypes.c_size_t(NB), _lib.stream_ptr())
    def run_old():
        _lib.call("gp_gemm_split256", _lib.ptr(Xk), _lib.i(M), _lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(Do), _lib.i(J), _lib.i(I), _lib.i(J),
                  _lib.i(K), _lib.i(w_is_a), _lib.i(2 if epi == 6 else epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(Do), _lib.i(J), _lib.ptr(ws),
                  ctypes.c_size_t(NB), _lib.stream_ptr())
    run_new(); torch.cuda.synchronize()
    err_n = lib.gp_gemm_split256_error(_lib.ptr(ws), _lib.stream_ptr())
    run_old(); torch.cuda.synchronize()
    err_o = lib.gp_gemm_split256_error(_lib.ptr(ws), _lib.stream_ptr())
    if epi == 6:
        v = Do * 8.0
        h = v.half(); l = (v - h.float()).half()
        same = bool((h.t() == ohi).all().item() and (l.t() == olo).all().item())
        nbad = int((h.t() != ohi).sum().item() + (l.t() != olo).sum().item())
    else:
        same = bool(torch.equal(Dn, Do)); nbad = int((Dn != Do).sum().item())
    rows = slice(0, 128)
    if epi != 6:
        if w_is_a:
            ref = W[rows].double() @ Xt.double().t(); mag = W[rows].double().abs() @ Xt.double().abs().t()
            if epi == 3: ref = D0[rows].double() + sc[rows, None].double() * (ref + bias[rows, None].double())
            elif epi == 2: ref = torch.nn.functional.gelu(ref + bias[rows, None].double())
            elif epi == 1: ref = ref + bias[rows, None].double()
        else:
            ref = Xt[rows].double() @ W.double().t() + bias[None, :J].double(); mag = Xt[rows].double().abs() @ W.double().abs().t()
        e = ((Dn[rows].double() - ref).abs() / mag)
        es = f"err/sum|ab| rms {e.pow(2).mean().sqrt().item():.2e} max {e.max().item():.2e}"
    else:
        es = "planes output"
    fl = 2.0 * I * J * K / 1e9
    t_new = timeit(run_new); t_old = timeit(run_old)
    print(f"{name:10s} I={I} J={J} K={K}: ping-pong {t_new:.3f} ms = {fl/t_new:.0f} TF-eq | lock-step {t_old:.3f} ms = {fl/t_old:.0f} TF-eq | "
          f"bitwise equal {same} ({nbad} differ) | {es} | hand-off errors {err_n} {err_o}", flush=True)

I, J, K = 1024, M, 4096
W = torch.randn(I, K, device=dev) * 0.03; Xt = torch.randn(J, K, device=dev)
whi, wlo = planes(W, 64.0); xhi, xlo = planes(Xt, 8.0); D = torch.empty(I, J, device=dev)
out = (ctypes.c_ulonglong * 8)()
def timed(): lib.gp_gemm_planes256_timing(_lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(xhi), _lib.ptr(xlo), _lib.ptr(D), J, I, J, K, _lib.ptr(ws), out, _lib.stream_ptr())
def plain():
    _lib.call("gp_gemm_planes256_scaled", _lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(xhi), _lib.ptr(xlo), _lib.ptr(D), _lib.i(J), _lib.ptr(None), _lib.ptr(None),
              _lib.i(0), _lib.i(I), _lib.i(J), _lib.i(J), _lib.i(K), _lib.i(0), _lib.ptr(None), _lib.ptr(None), _lib.ptr(None), _lib.i(0), _lib.f(1.0 / 512.0),
              _lib.f(8.0), _lib.ptr(None), _lib.ptr(ws), ctypes.c_size_t(NB), _lib.stream_ptr())
for label, heat in (("cold (single launch)", 0), ("sustained (after 30 back-to-back launches)", 30)):
    torch.cuda.synchronize()
    for _ in range(heat): plain()
    timed()
    n = max(1, out[5]) / 2
    print("fc2 shape, %s: wave 0 of block 100, cycles per k-step: matrix phase %d, barrier after it %d, memory phase %d, barrier after it %d (steps %d; one wave's MFMAs "
          "occupy the pipe 1536) | whole kernel %d cycles in %.1f us -> %.0f MHz; in-loop share %.0f %%"
          % (label, out[0] / n, out[1] / n, out[2] / n, out[3] / n, n, out[6], out[7] / 100.0, out[6] / (out[7] / 100.0),
             100.0 * (out[0] + out[1] + out[2] + out[3]) / out[6]))
