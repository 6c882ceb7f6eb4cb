"""Diagnostic: prescaled attention forward on a few shapes against fp32 softmax; prints the error and where non-finite outputs sit.
DD_LIB selects the library build.  python tools/attn_dbg.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from distdiff_amd import ops

bf = lambda t: t.to(torch.bfloat16).float()
CASES = [(2, 2, 256, 256, 32), (1, 2, 256, 256, 32), (2, 2, 128, 256, 32), (2, 8, 256, 256, 40), (2, 2, 512, 512, 32), (2, 2, 256, 64, 32),
         (2, 2, 256, 128, 32), (2, 1, 256, 256, 64)]
for (B, H, Nq, Nk, D) in CASES:
    g = torch.Generator().manual_seed(3)
    c = math.log2(math.e) / math.sqrt(D)
    qp = bf(torch.randn(B, Nq, H, D, generator=g) * c)
    k = bf(torch.randn(B, Nk, H, D, generator=g))
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    s = torch.einsum("bqhd,bkhd->bhqk", qp, k) * math.log(2.0)
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    # fused-QKV-like strides: embed q / k / v as column views of wider buffers
    o, lse = ops.attention(dev(qp, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, math.log(2.0), q_prescaled=True)
    torch.cuda.synchronize()
    o = o.float().cpu().reshape(B, Nq, H, D)
    bad = ~torch.isfinite(o)
    msg = ""
    if bad.any():
        idx = bad.any(-1).nonzero()
        msg = " NON-FINITE at (b, q, h) %s ... %s  count %d" % (idx[0].tolist(), idx[-1].tolist(), idx.shape[0])
        o = torch.where(bad, torch.zeros_like(o), o)
    print("B%d H%d Nq%d Nk%d D%d: max err %.4f%s" % (B, H, Nq, Nk, D, float((o - ref).abs().max()), msg), flush=True)
