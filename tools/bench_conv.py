"""Micro-benchmark of single conv/linear shapes through the op-level ABI (TFLOP/s, GB/s)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops

def run(name, B, H, Cin, Cout, k, geglu=False, bias=True, res=False, raw=False, force_small=False, iters=20, ksplit=0):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    pk = ops.PackedConv(w, k // 2, geglu=geglu, bias=torch.randn(Cout, generator=g) if bias else None)
    M = B * H * H
    x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).cuda()
    ncol = Cout // 2 if geglu else Cout
    y = torch.empty(M, ncol, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, ncol, generator=g).to(torch.bfloat16).cuda() if res else None
    rw = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda") if raw else None
    part = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    f = lambda: ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r, raw=rw, force_small=force_small, partial=part, ksplit=ksplit)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    flops = 2.0 * M * Cout * Cin * k * k
    byts = (M * Cin + M * ncol * (2 if res else 1) + (M * Cout if raw else 0)) * 2 + Cout * Cin * k * k * 2
    print("%-44s %8.1f us  %7.1f TF/s  %7.1f GB/s" % (name, us, flops / us / 1e6, byts / us / 1e3))

B = 16
run("1x1 320->320 plain", B, 64, 320, 320, 1, bias=False)
run("1x1 320->320 +bias", B, 64, 320, 320, 1)
run("1x1 320->320 +bias+res", B, 64, 320, 320, 1, res=True)
run("1x1 320->320 +bias+res small-kernel", B, 64, 320, 320, 1, res=True, force_small=True)
run("1x1 320->2560 plain", B, 64, 320, 2560, 1)
run("1x1 320->2560 geglu", B, 64, 320, 2560, 1, geglu=True)
run("1x1 320->2560 geglu+raw", B, 64, 320, 2560, 1, geglu=True, raw=True)
run("1x1 320->960 (qkv)", B, 64, 320, 960, 1, bias=False)
run("1x1 1280->320 (ff2) +res", B, 64, 1280, 320, 1, res=True)
run("3x3 320->320", B, 64, 320, 320, 3)
run("3x3 320->320 ABLATE no-DMA", B, 64, 320, 320, 3, force_small=2)
run("3x3 320->320 ABLATE no-compute", B, 64, 320, 320, 3, force_small=4)
run("3x3 320->320 ABLATE neither", B, 64, 320, 320, 3, force_small=6)
run("3x3 320->320 small-kernel", B, 64, 320, 320, 3, force_small=True)
run("3x3 640->640 @32", B, 32, 640, 640, 3)
run("3x3 1280->1280 @16", B, 16, 1280, 1280, 3)
run("3x3 1280->1280 @8", B, 8, 1280, 1280, 3)
run("1x1 1280->1280 @16", B, 16, 1280, 1280, 1)
run("1x1 640->640 @32", B, 32, 640, 640, 1)
run("3x3 128->128 @512 (vae)", 8, 512, 128, 128, 3, iters=5)
run("3x3 128->128 @512 small-kernel", 8, 512, 128, 128, 3, iters=5, force_small=True)
run("3x3 256->256 @256 (vae)", 8, 256, 256, 256, 3, iters=5)
run("3x3 512->512 @128 (vae)", 8, 128, 512, 512, 3, iters=5)
