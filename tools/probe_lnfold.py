"""Launch times of the folded-LayerNorm plane GEMMs (epilogues 8 / 9 / 10) next to the epilogues they replace (7 / 6 / 3), ViT-L
shapes at B crops, each alone in a loop (torch events); epilogue 10 also with parts of its epilogue switched off at run time
(gp_gemm_planes256_set_dp bits 2..5: statistics / plane stores / D stores / residual loads).  Usage: python tools/probe_lnfold.py [B]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapose_amd import _lib  # noqa: E402
from gigapose_amd.vit import split_planes_x64  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
C, MLP = 1024, 4096
Mtok = B * 257
Mpad = (Mtok + 255) // 256 * 256
lib = _lib.lib()
lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
nb = lib.gp_gemm_split256_workspace_bytes()
ws = torch.zeros(nb // 4, device=DEV)
_lib.status_word(DEV)


def planes(rows, cols, scale=1.0):
    x = torch.randn(rows, cols, device=DEV) * scale
    hi = torch.empty(rows, cols, dtype=torch.float16, device=DEV)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(x), ctypes.c_size_t(x.numel()), _lib.f(8.0), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st_main = torch.zeros(C // 256 + 1, Mpad, 2, device=DEV)
st_main[0, :, 1] = 1024.0
st_strip = torch.zeros(C // 32 + 1, 256, 2, device=DEV)
st_strip[0, :, 1] = 1024.0
xa, xb = torch.randn(Mpad, C, device=DEV), torch.zeros(Mpad, C, device=DEV)
xcm = torch.randn(C, Mpad, device=DEV)


def gemm(I, K, epi, whi, wlo, bhi, blo, bias, scale, out, D=None, ldd=0, res=None, ldr=0):
    ohi, olo = out if out is not None else (None, None)

    def f():
        _lib.call("gp_gemm_planes256_ln", _lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(bhi), _lib.ptr(blo), _lib.ptr(D), _lib.i(ldd), _lib.ptr(ohi), _lib.ptr(olo),
                  _lib.i(I), _lib.i(I), _lib.i(Mpad), _lib.i(Mtok), _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(res), _lib.i(ldr),
                  _lib.f(1.0 / 512.0), _lib.ptr(st_main), _lib.ptr(st_strip), _lib.ptr(st_main), _lib.ptr(st_strip), _lib.i(Mpad), _lib.f(1e-6),
                  _lib.ptr(ws), ctypes.c_size_t(nb), _lib.stream_ptr())
    return f


rows = []
h = planes(Mpad, C)
f4 = planes(Mpad, MLP, 0.5)
for name, I, K, act, old, new in (("q|k|v", 3 * C, C, h, 7, 8), ("fc1 (GELU)", MLP, C, h, 6, 9)):
    W = torch.randn(I, K, device=DEV) / K ** 0.5
    whi, wlo = split_planes_x64(W)
    bias, s = torch.zeros(I, device=DEV), torch.randn(I, device=DEV)
    out = (torch.zeros(Mpad, I, dtype=torch.float16, device=DEV), torch.zeros(Mpad, I, dtype=torch.float16, device=DEV))
    t_old = timeit(gemm(I, K, old, whi, wlo, act[0], act[1], bias, None, out))
    t_new = timeit(gemm(I, K, new, whi, wlo, act[0], act[1], bias, s, out))
    rows.append((name, f"epilogue {old}", t_old, f"epilogue {new}", t_new))
for name, K, act in (("proj", C, h), ("fc2", MLP, f4)):
    W = torch.randn(C, K, device=DEV) / K ** 0.5
    whi, wlo = split_planes_x64(W)
    bias, ls = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    out = (torch.zeros(Mpad, C, dtype=torch.float16, device=DEV), torch.zeros(Mpad, C, dtype=torch.float16, device=DEV))
    t_old = timeit(gemm(C, K, 3, whi, wlo, act[0], act[1], bias, ls, None, D=xcm, ldd=Mpad, res=xcm, ldr=Mpad))
    lib.gp_gemm_planes256_set_dp(1)
    t_new = timeit(gemm(C, K, 10, whi, wlo, act[0], act[1], bias, ls, out, D=xb, ldd=C, res=xa, ldr=C))
    rows.append((name, "epilogue 3", t_old, "epilogue 10", t_new))
    for mask, what in ((1, "no statistics"), (2, "no plane stores"), (4, "no D stores"), (8, "no residual loads"), (3, "no statistics, no plane stores"),
                       (15, "arithmetic + LDS turn only")):
        lib.gp_gemm_planes256_set_dp(1 | (mask << 2))
        rows.append((name, "", 0.0, f"  epilogue 10, {what}", timeit(gemm(C, K, 10, whi, wlo, act[0], act[1], bias, ls, out, D=xb, ldd=C, res=xa, ldr=C))))
    lib.gp_gemm_planes256_set_dp(1)
_lib.take_status()
print(f"ViT-L plane GEMMs at {B} crops (Mtok {Mtok}), microseconds per launch")
for name, a, ta, b, tb in rows:
    print(f"{name:12s} {a:12s} {ta:8.1f}   {b:44s} {tb:8.1f}")
