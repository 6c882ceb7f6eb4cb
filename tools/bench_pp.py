"""Fixed cost per tile of the pointwise ping-pong GEMM: time against K at fixed M, N (one round of 256 tiles), with / without residual."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(B, H, Cin, Cout, res=False, bias=True, iters=20):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    pk = ops.PackedConv(w, 0, bias=torch.randn(Cout, generator=g) if bias else None)
    M = B * H * H
    x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, Cout, device="cuda").to(torch.bfloat16) if res else None
    part = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device="cuda")
    f = lambda: ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r, partial=part)
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
    tiles = (M // 256) * (Cout // 320) if Cout % 320 == 0 else 0
    print("M %6d N %5d K %5d res %d bias %d: %7.1f us %6.0f TF/s  tiles %d rounds %.2f us/round %.1f" % (M, Cout, Cin, res, bias, us, 2.0 * M * Cout * Cin / us / 1e6, tiles, tiles / 256, us / max(tiles / 256, 1)), flush=True)


print("pp=" + os.environ.get("DD_GEMM_PP", "1"))
for K in (320, 640, 1280, 2560, 5120):
    run(64, 16, K, 1280)
run(64, 16, 1280, 1280, res=True)
run(64, 16, 1280, 1280, bias=False)
for K in (320, 640, 1280):
    run(64, 32, K, 1920)
run(64, 32, 640, 640, res=True)
run(64, 64, 320, 960)
run(64, 64, 320, 320, res=True)
