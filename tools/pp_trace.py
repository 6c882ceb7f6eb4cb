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
    all_tiles = (M // 256) * (Cout // (256 if geglu else 320))
    tiles = min(all_tiles, 8192)                      # the trace buffer holds 8192 tiles
    buf = (ctypes.c_ulonglong * (tiles * 6))()
    L.dd_debug_read_pp_trace(buf, tiles * 6)
    rec = [[buf[i * 6 + j] for j in range(4)] for i in range(tiles)]
    t0 = min(r_[0] for r_ in rec)
    last = max(r_[3] for r_ in rec)
    top = sum(r_[1] - r_[0] for r_ in rec) / tiles / 100.0
    kl = sum(r_[2] - r_[1] for r_ in rec) / tiles / 100.0
    ep = sum(r_[3] - r_[2] for r_ in rec) / tiles / 100.0
    print("%-26s kernel %.1f us (events), %d tiles: wait for the first K-step %.2f, K loop %.2f, epilogue %.2f us per tile; first entry -> last exit %.1f us = %.2f us x %.1f rounds"
          % (name, e0.elapsed_time(e1) * 1000, tiles, top, kl, ep, (last - t0) / 100.0, (last - t0) / 100.0 / max(all_tiles / 256.0, 1), all_tiles / 256.0), flush=True)


run("320->960 @64", 64, 64, 320, 960)
run("320->320 +res @64", 64, 64, 320, 320, res=True)
run("640->640 +res @32", 64, 32, 640, 640, res=True)
run("1280->1280 +res @16", 64, 16, 1280, 1280, res=True)
run("geglu 320->2560 @64", 64, 64, 320, 2560, geglu=True)
run("geglu 640->5120 @32", 64, 32, 640, 5120, geglu=True)
run("geglu 1280->10240 @16", 64, 16, 1280, 10240, geglu=True)


def spread(name, B, H, Cin, Cout, res=False, geglu=False):
    """per-workgroup view of the same trace: when each workgroup of the persistent grid finished, by XCD"""
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
    tiles = (M // 256) * (Cout // (256 if geglu else 320))
    assert tiles <= 8192
    buf = (ctypes.c_ulonglong * (tiles * 6))()
    L.dd_debug_read_pp_trace(buf, tiles * 6)
    rec = [[buf[i * 6 + j] for j in range(4)] for i in range(tiles)]
    t0 = min(r_[0] for r_ in rec)
    grid, per = 256, 32
    q, rr = tiles >> 3, tiles & 7
    ends, starts, durs = {}, {}, collections.defaultdict(list)
    for xcd in range(8):
        xbase = xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q
        xcount = q + (1 if xcd < rr else 0)
        for slot in range(per):
            ts = list(range(xbase + slot, xbase + xcount, per))
            if not ts:
                continue
            starts[(xcd, slot)] = (rec[ts[0]][0] - t0) / 100.0
            ends[(xcd, slot)] = (rec[ts[-1]][3] - t0) / 100.0
            for t in ts:
                durs[xcd].append((rec[t][3] - rec[t][0]) / 100.0)
    print(name)
    for xcd in range(8):
        e = sorted(v for (x, s), v in ends.items() if x == xcd)
        st = sorted(v for (x, s), v in starts.items() if x == xcd)
        d = sorted(durs[xcd])
        print("  xcd %d: workgroups start %.1f .. %.1f us, finish min %.1f  median %.1f  max %.1f us; tile time min %.2f median %.2f p90 %.2f max %.2f us"
              % (xcd, st[0], st[-1], e[0], e[len(e) // 2], e[-1], d[0], d[len(d) // 2], d[int(len(d) * 0.9)], d[-1]), flush=True)


if os.environ.get("PP_SPREAD"):
    spread("geglu 320->2560 @64 (32 images)", 32, 64, 320, 2560, geglu=True)
    spread("geglu 1280->10240 @16", 64, 16, 1280, 10240, geglu=True)
    spread("320->960 @64", 64, 64, 320, 960)
    spread("640->640 +res @32", 64, 32, 640, 640, res=True)
