"""GPU probe: per-slot timeline of the plane GEMM (gp_gemm_planes256_trace) for the five ViT-L GEMM shapes at B = 64:
where a slot's time goes -- accumulator hand-over waits, k loop, epilogue / publish -- next to the event-timed launch."""
import os, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
dev = "cuda"
if os.environ.get("GP_LIB"):   # a library built with other macros (e.g. -DGP_EPI_PROBE=.., tools/patches/epi_probe.diff)
    _lib.LIB_PATH = os.path.abspath(os.environ["GP_LIB"])
lib = _lib.lib()
lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
NB = lib.gp_gemm_split256_workspace_bytes()
ws = torch.zeros(NB // 4, device=dev)
def planes(W, scale):
    hi = torch.empty(W.shape, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(W), ctypes.c_size_t(W.numel()), _lib.f(scale), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def timeit_cold(fn, iters=10):
    """Each launch after 2 x 400 MB of unrelated traffic (the 256 MB Infinity Cache and the L2s hold nothing of the GEMM's operands or
    of the residual its epilogue reads: the state a launch finds inside the ViT, where ~650 MB pass between two uses of X)."""
    a_, b_ = torch.empty(100 << 20, device=dev), torch.zeros(100 << 20, device=dev)
    tot = 0.0
    for _ in range(iters):
        a_.copy_(b_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3
torch.manual_seed(0)
if os.environ.get("PAR_MIN_STEPS"):   # parallel split-K (fewer tiles than slots): at least this many k-steps per slot of a split tile
    lib.gp_gemm_planes256_set_par(int(os.environ["PAR_MIN_STEPS"]))
M = int(os.environ.get("MPAD", 16640))           # padded rows of the token dimension (ViT-L at B = 64)
MV = int(os.environ.get("MTOK", 16448))          # rows that carry tokens (64 x 257); MTOK=16640 probes the fully tiled launch
for (nw, K, name, epi) in [(3072, 1024, "qkv", 7), (1024, 1024, "proj", 3), (4096, 1024, "fc1", 6), (1024, 4096, "fc2", 3)]:
    W = torch.randn(nw, K, device=dev) * 0.03
    Xt = torch.randn(M, K, device=dev) * 1.5
    whi, wlo = planes(W, 64.0); xhi, xlo = planes(Xt, 8.0)
    I, J = nw, M
    bias = torch.randn(I, device=dev); sc = torch.randn(I, device=dev)
    D = torch.randn(I, J, device=dev)
    ohi = torch.zeros(M, nw, dtype=torch.float16, device=dev); olo = torch.zeros_like(ohi)
    trace = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
    args = (_lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(xhi), _lib.ptr(xlo), _lib.ptr(D), _lib.i(J), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(nw),
            _lib.i(I), _lib.i(J), _lib.i(MV), _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(D), _lib.i(J), _lib.f(1.0 / 512.0), _lib.ptr(ws))
    def plain(): _lib.call("gp_gemm_planes256_scaled", *args[:-1], _lib.f(8.0), _lib.ptr(None), args[-1], ctypes.c_size_t(NB), _lib.stream_ptr())
    def traced(): _lib.call("gp_gemm_planes256_trace", *args, _lib.ptr(trace), _lib.stream_ptr())
    t_plain = timeit(plain)
    if os.environ.get("COLD"):
        t_one = timeit_cold(plain, 1) * 0 + sum(timeit(plain, 1, 0) for _ in range(10)) / 10   # one launch per event pair, caches warm
        print(f"{name:5s} event-timed one launch at a time: caches warm {t_one:6.1f} us, after 800 MB of other traffic {timeit_cold(plain):6.1f} us", flush=True)
    for _ in range(20): plain()          # sustained clocks
    traced(); torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(256, 32).astype(np.int64)
    start, nseg = t[:, 0], t[:, 1]
    t0 = start.min()
    ends, kl, ep, wait, steps, pub, strip, ep_own = [], [], [], [], [], [], [], []
    by_seg = {}                                   # SEGSTAT=1: epilogue durations by segment index (aligned whole-tile rounds vs staggered stream-K ones)
    for p in range(256):
        prev = start[p]
        for s in range(min(int(nseg[p]), 7)):
            kind, ns = int(t[p, 2 + 4 * s] >> 32), int(t[p, 2 + 4 * s] & 0xffffffff)
            a, b, c = t[p, 3 + 4 * s], t[p, 4 + 4 * s], t[p, 5 + 4 * s]
            wait.append((a - prev) / 100.0); kl.append((b - a) / 100.0); steps.append(ns)
            (pub if kind in (1, 3) else (ep_own if kind == 2 else ep)).append((c - b) / 100.0)
            if kind in (0, 2): by_seg.setdefault((s, kind), []).append(((c - b) / 100.0, (b - t0) / 100.0))
            prev = c
        strip.append((t[p, 31] - prev) / 100.0)
        ends.append((t[p, 31] - t0) / 100.0)
    kl, steps = np.array(kl), np.array(steps)
    fl = 2.0 * I * MV * K
    print(f"{name:5s} I={I} J={J} K={K} epi {epi}: event-timed {t_plain:6.1f} us = {fl / t_plain / 1e6:5.0f} TF-eq | traced slots: start spread "
          f"{(start.max() - t0) / 100.0:5.1f} us, lifetime min/mean/max {min(ends):6.1f}/{np.mean(ends):6.1f}/{max(ends):6.1f} us | per slot: segments "
          f"{nseg.mean():.2f}, k loop {kl.sum() / 256:6.1f} us ({kl.sum() / steps.sum():.3f} us/step, {steps.sum() / 256:.1f} steps), "
          f"epilogues {np.sum(ep) / 256:5.1f} us ({np.mean(ep) if ep else 0:5.1f} each x {len(ep) / 256:.2f}), owner reduce + epilogue {np.mean(ep_own) if ep_own else 0:5.1f} each x {len(ep_own) / 256:.2f}, publishes {np.sum(pub) / 256:5.1f} us "
          f"({np.mean(pub) if pub else 0:5.1f} each), waits + prologue {np.sum(wait) / 256:5.1f} us (max single {np.max(wait):5.1f}), "
          f"strip mean {np.mean(strip):4.1f} max {np.max(strip):4.1f} us", flush=True)
    if os.environ.get("PER_XCD"):   # slot p runs on XCD p & 7 (dispatch order): is the spread of lifetimes systematic?
        e = np.array(ends)
        print("      lifetime by XCD (us):", " ".join(f"{e[x::8].mean():6.1f}" for x in range(8)), "| by slot-in-XCD quartile:",
              " ".join(f"{e.reshape(32, 8)[q * 8:(q + 1) * 8].mean():6.1f}" for q in range(4)),
              "| k-loop us/step by XCD:", " ".join(f"{(kl.reshape(256, -1).sum(1)[x::8].mean() / (steps.sum() / 256)):5.3f}" for x in range(8)) if len(kl) % 256 == 0 else "", flush=True)
    if os.environ.get("SEGSTAT"):
        for (sidx, kind), v in sorted(by_seg.items()):
            d, at = np.array([x[0] for x in v]), np.array([x[1] for x in v])
            print(f"      segment {sidx} ({'whole tile' if kind == 0 else 'continued tile'}): {len(v):3d} epilogues, duration min/mean/max {d.min():5.1f}/{d.mean():5.1f}/{d.max():5.1f} us, "
                  f"starting {at.min():6.1f} .. {at.max():6.1f} us after the launch (spread {at.std():5.1f})", flush=True)
