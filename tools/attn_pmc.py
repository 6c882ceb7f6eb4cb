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
# the engine's form: the query carries 1/sqrt(d) * log2(e) (folded into the to_q weights), scale = ln 2 -> lazy-reference forward
qp = (q.float() * (math.log2(math.e) / math.sqrt(D))).to(torch.bfloat16)
for _ in range(4):
    ops.attention(qp, k, v, B, H, N, N, D, math.log(2.0), q_prescaled=True)
torch.cuda.synchronize()
