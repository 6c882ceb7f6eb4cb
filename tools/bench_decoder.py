"""The decoder's 3x3 convolutions of the bench step (32 images) through the op-level ABI: the persistent 512 x 128 halo form
(conv_halo_persist_kernel) against the one-tile-per-workgroup form:  DD_HALO_PERSIST=0 python tools/bench_decoder.py  vs  =1 (default)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(name, B, H, Cin, Cout, res=False, iters=3):
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
    print("%-14s %8.0f us %6.0f TF/s" % (name, us, 2.0 * M * Cout * Cin * 9 / us / 1e6), flush=True)


print("persist=" + os.environ.get("DD_HALO_PERSIST", "1"))
run("128>128@512", 32, 512, 128, 128, True)
run("256>128@512", 32, 512, 256, 128)
run("256>256@256", 32, 256, 256, 256, True)
run("512>256@256", 32, 256, 512, 256)
run("512>512@128", 32, 128, 512, 512, True)
run("512>512@64", 32, 64, 512, 512, True)
