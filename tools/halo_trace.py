"""Per-tile timeline of conv_halo_persist_kernel (debug build: tools/build_variant.sh trace conv_halo.hip -DDD_TRACE, run with DD_LIB=ab/trace.so):
wave 0 of every workgroup stamps: loop top -> first stage landed -> K loop done -> next tile's first stage issued -> epilogue done."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
from distdiff_amd._lib import lib as load_library

L = load_library()


def run(name, B, H, Cin, Cout, res=False):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    pk = ops.PackedConv(w, 1, bias=torch.randn(Cout, generator=g))
    M = B * H * H
    x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, Cout, device="cuda").to(torch.bfloat16) if res else None
    part = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device="cuda")
    f = lambda: ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r, partial=part)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    tiles = min((M // 512) * (Cout // 128), 8192)
    buf = (ctypes.c_ulonglong * (tiles * 6))()
    L.dd_debug_read_pp_trace(buf, tiles * 6)
    rec = [[buf[i * 6 + j] for j in range(5)] for i in range(tiles)]
    rec = [r_ for r_ in rec if r_[4] > r_[0] > 0]
    n = len(rec)
    seg = [sum(r_[j + 1] - r_[j] for r_ in rec) / n / 100.0 for j in range(4)]
    chunks = Cin // 64
    print("%-16s kernel %.0f us; %d tiles traced: wait for the first stage %.2f | K loop %.2f (%.2f per chunk) | issue of the next tile's first stage %.2f | "
          "epilogue %.2f | tile %.2f us" % (name, e0.elapsed_time(e1) * 1000, n, seg[0], seg[1], seg[1] / chunks, seg[2], seg[3], sum(seg)), flush=True)


run("128>128@512", 32, 512, 128, 128, True)
run("256>128@512", 32, 512, 256, 128)
run("256>256@256", 32, 256, 256, 256, True)
run("512>512@128", 32, 128, 512, 512, True)
