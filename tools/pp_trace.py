"""Per-workgroup timeline of gemm_pp_kernel (debug build into ab/: tools/build_variant.sh trace conv_halo.hip -DDD_TRACE, run with DD_LIB):
per tile: wait for its first K-step -> K loop -> epilogue of the persistent kernel."""
import collections, ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
from distdiff_amd._lib import lib as load_library

L = load_library()


def run(name, B, H, Cin, Cout, res=False, geglu=False):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    pk = ops.PackedConv(w, 0, geglu=geglu, bias=torch.randn(Cout, generator=g))
    M = B * H * H
    ncol = Cout // 2 if geglu else Cout
    x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, ncol, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, ncol, device="cuda").to(torch.bfloat16) if res else None
    for _ in range(3):
        ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r)
    e1.record()
    torch.cuda.synchronize()
    tiles = min((M // 256) * (Cout // (256 if geglu else 320)), 8192)
    buf = (ctypes.c_ulonglong * (tiles * 6))()
    L.dd_debug_read_pp_trace(buf, tiles * 6)
    rec = [[buf[i * 6 + j] for j in range(4)] for i in range(tiles)]
    t0 = min(r_[0] for r_ in rec)
    last = max(r_[3] for r_ in rec)
    top = sum(r_[1] - r_[0] for r_ in rec) / tiles / 100.0
    kl = sum(r_[2] - r_[1] for r_ in rec) / tiles / 100.0
    ep = sum(r_[3] - r_[2] for r_ in rec) / tiles / 100.0
    print("%-26s kernel %.1f us (events), %d tiles: wait for the first K-step %.2f, K loop %.2f, epilogue %.2f us per tile; first entry -> last exit %.1f us = %.2f us x %.1f rounds"
          % (name, e0.elapsed_time(e1) * 1000, tiles, top, kl, ep, (last - t0) / 100.0, (last - t0) / 100.0 / max(tiles / 256.0, 1), tiles / 256.0), flush=True)


run("320->960 @64", 64, 64, 320, 960)
run("320->320 +res @64", 64, 64, 320, 320, res=True)
run("640->640 +res @32", 64, 32, 640, 640, res=True)
run("1280->1280 +res @16", 64, 16, 1280, 1280, res=True)
run("geglu 320->2560 @64", 64, 64, 320, 2560, geglu=True)
run("geglu 640->5120 @32", 64, 32, 640, 5120, geglu=True)
run("geglu 1280->10240 @16", 64, 16, 1280, 10240, geglu=True)
