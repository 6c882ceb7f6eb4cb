"""The d = 40 / 4096-key self-attention forward of the bench through the op ABI, a few launches, for rocprofv3 --pmc passes (tools/attn_pmc.sh)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops

B, H, N, D = 32, 8, 4096, 40
g = torch.Generator().manual_seed(0)
q = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
k = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
v = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
for _ in range(4):
    ops.attention(q, k, v, B, H, N, N, D, 1 / math.sqrt(D))
torch.cuda.synchronize()
