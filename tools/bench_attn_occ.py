"""Attention kernels at the bench shapes (SD-1.5: d = 40 / 80 / 160 self + 77-key cross; SDXL: d = 64), forward and forward + backward:
for same-device A/B of occupancy targets (DD_AW_FWD / DD_AW_DQ / DD_AW_DKV builds via tools/build_variant.sh, DD_LIB)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(B, H, Nq, Nk, D, bwd=False, iters=5):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B * Nq, H * D, generator=g).to(torch.bfloat16).cuda()
    k = torch.randn(B * Nk, H * D, generator=g).to(torch.bfloat16).cuda()
    v = torch.randn(B * Nk, H * D, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * Nq, H * D, generator=g).to(torch.bfloat16).cuda() if bwd else None
    f = lambda: ops.attention(q, k, v, B, H, Nq, Nk, D, 1 / math.sqrt(D), d_o=do, need_dkv=(Nk != 77))
    for _ in range(2):
        o = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        o = f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    chk = sum(float(t.float().abs().mean()) for t in o if t is not None and t.dtype == torch.bfloat16)
    return "%s%dx%d %8.1f us (%.5f)" % ("bwd " if bwd else "", D, Nk, us, chk)


name = os.path.basename(os.environ.get("DD_LIB", "default"))
res = [run(32, 8, 4096, 4096, 40), run(64, 8, 1024, 1024, 80), run(64, 8, 256, 256, 160), run(64, 8, 4096, 77, 40), run(64, 8, 1024, 77, 80),
       run(16, 10, 4096, 4096, 64), run(8, 8, 4096, 4096, 40, bwd=True, iters=3), run(16, 8, 1024, 1024, 80, bwd=True, iters=3),
       run(8, 8, 4096, 77, 40, bwd=True, iters=3)]
print("%-14s" % name, " | ".join(res))
