"""The cross-attention launches of the bench step (77 keys) through the op-level ABI: attention_shortk.hip against the streaming forward
(DD_ATTN_SHORTK=0).  64 = 32 images x 2 CFG halves."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(B, H, Nq, Nk, D, iters=20):
    q = torch.randn(B * Nq, H * D, device="cuda").to(torch.bfloat16)
    k = torch.randn(B * Nk, H * D, device="cuda").to(torch.bfloat16)
    v = torch.randn(B * Nk, H * D, device="cuda").to(torch.bfloat16)
    f = lambda: ops.attention(q, k, v, B, H, Nq, Nk, D, 0.6931471805599453, q_prescaled=True)
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
    print("B %3d H %2d Nq %5d Nk %3d d %3d: %8.1f us   Q + O at %.2f TB/s" % (B, H, Nq, Nk, D, us, 2 * B * Nq * H * D * 2 / us / 1e6), flush=True)


print("DD_ATTN_SHORTK=" + os.environ.get("DD_ATTN_SHORTK", "1"))
run(64, 8, 4096, 77, 40)
run(64, 8, 1024, 77, 80)
run(8, 10, 4096, 77, 64)
run(8, 20, 1024, 77, 64)
