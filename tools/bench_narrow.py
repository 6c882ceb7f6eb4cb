"""conv_out shapes (3x3, N = 3 / 4) through the op-level ABI: the 512 x 32 form of the halo kernel against the tiled kernels (DD_HALO_NARROW=0)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(B, Cin, Cout, H, W, iters=10):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    pk = ops.PackedConv(w, 1, bias=torch.randn(Cout, generator=g))
    M = B * H * W
    x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, 8, dtype=torch.float32, device="cuda")
    f = lambda: ops.conv_gemm(x, pk, B, H, W, H, W, y=y[:, :Cout], out_f32=True, ksplit=1)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    print("B %3d Cin %4d N %d %4dx%-4d: %8.1f us  %5.2f TB/s of input" % (B, Cin, Cout, H, W, us, M * Cin * 2 / us / 1e6), flush=True)


print("DD_HALO_NARROW=" + os.environ.get("DD_HALO_NARROW", "1"))
run(32, 128, 3, 512, 512)
run(64, 320, 4, 64, 64)
