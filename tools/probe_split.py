"""GPU probe: split-f16 (3 x f16 MFMA) GEMM vs the exact f32 fmaf-chain GEMM: error against an f64 reference and time."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
dev = "cuda"
lib = _lib.lib()
def timeit(fn, iters=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def split_w(Wt):  # Wt [K][n] f32 -> hi, lo [n][K] f16
    K, n = Wt.shape
    hi = torch.empty(n, K, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    _lib.call("gp_split_weights", _lib.ptr(Wt), _lib.i(K), _lib.i(n), _lib.i(n), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo
torch.manual_seed(0)
# correctness on a small asymmetric case, both operand roles
for act_is_b in [1, 0]:
    I, J, K = 256, 384, 96
    if act_is_b:
        Wt = torch.randn(K, I, device=dev); X = torch.randn(K, J, device=dev) * 3
        ref = (Wt.double().T @ X.double())
    else:
        X = torch.randn(K, I, device=dev) * 3; Wt = torch.randn(K, J, device=dev)
        ref = (X.double().T @ Wt.double())
    hi, lo = split_w(Wt)
    D = torch.empty(I, J, device=dev)
    _lib.call("gp_gemm_split", _lib.ptr(X), _lib.i(X.shape[1]), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I), _lib.i(J),
              _lib.i(K), _lib.i(act_is_b), _lib.i(0), None, None, None, _lib.i(J), _lib.stream_ptr())
    A_, B_ = (Wt, X) if act_is_b else (X, Wt)
    De = torch.empty(I, J, device=dev)
    _lib.call("gp_gemm_kmajor", _lib.ptr(A_), _lib.i(I), _lib.ptr(B_), _lib.i(J), _lib.ptr(De), _lib.i(J), _lib.i(I), _lib.i(J),
              _lib.i(K), _lib.i(0), None, None, None, _lib.i(J), _lib.stream_ptr())
    torch.cuda.synchronize()
    scale = (A_.double().abs().T @ B_.double().abs())
    es = ((D.double() - ref).abs() / scale).max().item(); ee = ((De.double() - ref).abs() / scale).max().item()
    print(f"act_is_b={act_is_b} small: max |err| / sum|a||b|: split {es:.3e}  exact-f32-chain {ee:.3e}  max abs split {(D.double()-ref).abs().max().item():.3e}")
# ViT-L shapes: error + time
for (I, J, K, act_is_b, name) in [(1024, 16512, 1024, 1, "proj"), (4096, 16512, 1024, 1, "fc1"), (1024, 16512, 4096, 1, "fc2"), (2048, 16512, 1024, 1, "qk"), (16512, 1024, 1024, 0, "v")]:
    if act_is_b:
        Wt = torch.randn(K, I, device=dev) * 0.05; X = torch.randn(K, J, device=dev)
    else:
        X = torch.randn(K, I, device=dev); Wt = torch.randn(K, J, device=dev) * 0.05
    hi, lo = split_w(Wt)
    A_, B_ = (Wt, X) if act_is_b else (X, Wt)
    D = torch.empty(I, J, device=dev); De = torch.empty(I, J, device=dev)
    fs = lambda: _lib.call("gp_gemm_split", _lib.ptr(X), _lib.i(X.shape[1]), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I), _lib.i(J),
                           _lib.i(K), _lib.i(act_is_b), _lib.i(0), None, None, None, _lib.i(J), _lib.stream_ptr())
    fe = lambda: _lib.call("gp_gemm_kmajor", _lib.ptr(A_), _lib.i(I), _lib.ptr(B_), _lib.i(J), _lib.ptr(De), _lib.i(J), _lib.i(I), _lib.i(J),
                           _lib.i(K), _lib.i(0), None, None, None, _lib.i(J), _lib.stream_ptr())
    ms_s, ms_e = timeit(fs), timeit(fe)
    rows = slice(0, 256)
    ref = (A_[:, rows].double().T @ B_.double())
    scale = (A_[:, rows].double().abs().T @ B_.double().abs())
    es = ((D[rows].double() - ref).abs() / scale); ee = ((De[rows].double() - ref).abs() / scale)
    print(f"{name:5s} I={I} J={J} K={K}: split {ms_s:.3f} ms = {2.0*I*J*K/ms_s/1e9:.0f} TF-equivalent | exact {ms_e:.3f} ms = {2.0*I*J*K/ms_e/1e9:.0f} TF | "
          f"err/sum|ab| max: split {es.max().item():.2e} exact {ee.max().item():.2e}; rms: split {es.pow(2).mean().sqrt().item():.2e} exact {ee.pow(2).mean().sqrt().item():.2e}")

# per-phase cycle breakdown of one mid-grid block (fc1 shape)
I, J, K = 4096, 16512, 1024
Wt = torch.randn(K, I, device=dev) * 0.05; X = torch.randn(K, J, device=dev)
hi, lo = split_w(Wt); D = torch.empty(I, J, device=dev)
out = (ctypes.c_ulonglong * 23)()
for rep in range(2):
    lib.gp_gemm_split_timing(_lib.ptr(X), J, _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), J, I, J, K, out, _lib.stream_ptr())
names = ["gload issue", "LDS read + MFMA issue", "MFMA drain", "convert + LDS write", "barrier"]
for w in range(4):
    v = [out[w * 5 + p] for p in range(5)]
    print(f"wave {w}: " + ", ".join(f"{n} {x / 32:.0f}" for n, x in zip(names, v)) + f"  (cycles per k-step; total {sum(v) / 32:.0f}; MFMA pipe time 768)")
print(f"timed block: {out[20]} shader cycles in {out[21] * 10} ns -> {out[20] / (out[21] * 10.0):.3f} GHz effective clock; occupancy API: {out[22]} workgroups/CU")

# per-block trace: how many workgroups are really resident, per CU?
import numpy as np
grid = 8 * ((32 * 129 + 7) // 8)
trace = torch.zeros(grid * 3, dtype=torch.int64, device=dev)
lib.gp_gemm_split_set_trace(ctypes.c_void_p(trace.data_ptr()))
lib.gp_gemm_split_timing(_lib.ptr(X), J, _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), J, I, J, K, out, _lib.stream_ptr())
lib.gp_gemm_split_set_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy().reshape(grid, 3)
t = t[t[:, 1] > 0]
start, end, hw = t[:, 0], t[:, 1], t[:, 2]
t0 = start.min(); dur = (end - start) * 10e-3  # us (100 MHz ticks)
print(f"{len(t)} blocks; kernel span {(end.max() - t0) * 10e-3:.1f} us; block duration us: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f}")
# concurrency over time
ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0], append=ev[-1, 0])
print(f"time-averaged resident workgroups: {(conc * dt).sum() / dt.sum():.1f} (of 512 slots); max {conc.max()}")
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
print("distinct (xcc,se,sh,cu):", len(np.unique(key)), " blocks per key min/max:", np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max())
