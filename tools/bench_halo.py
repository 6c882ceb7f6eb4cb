"""The bench step's deep 3x3 convolutions (engine batch 32: CFG batch 64 for the UNet, 32 images for the VAE) through the op-level ABI,
for A/B of the halo-resident kernel: DD_CONV_HALO=0 python tools/bench_halo.py vs DD_CONV_HALO=1."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(name, B, H, Cin, Cout, res=False, iters=6):
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
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    return "%s %.0f us %.0f TF/s" % (name, us, 2.0 * M * Cout * Cin * 9 / us / 1e6)


tag = "halo=" + os.environ.get("DD_CONV_HALO", "1")
out = [run("960>320@64", 64, 64, 960, 320), run("320>320@64", 64, 64, 320, 320, True), run("640>320@64", 64, 64, 640, 320), run("640>640@32", 64, 32, 640, 640, True),
       run("1280>640@32", 64, 32, 1280, 640), run("1280>1280@16", 64, 16, 1280, 1280, True), run("2560>1280@16", 64, 16, 2560, 1280),
       run("512>512@64", 32, 64, 512, 512, True, 3), run("512>512@128", 32, 128, 512, 512, True, 2), run("256>256@256", 8, 256, 256, 256, True, 2), run("128>128@512", 8, 512, 128, 128, True, 2), run("256>128@512", 8, 512, 256, 128, False, 2)]
print(tag, " | ".join(out))
