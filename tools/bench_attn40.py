"""d = 40 self-attention forward at the bench shape (CFG batch 32, 8 heads, 4096 tokens): time + checksum, for A/B runs with DD_LIB."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
B, H, N, D = 32, 8, 4096, 40
g = torch.Generator().manual_seed(0)
q = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
k = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
v = torch.randn(B * N, H * D, generator=g).to(torch.bfloat16).cuda()
f = lambda: ops.attention(q, k, v, B, H, N, N, D, 1 / math.sqrt(D))
for _ in range(3): o = f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o = f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
o0 = o[0] if isinstance(o, (tuple, list)) else o
print("%-28s d40 fwd %8.1f us  %6.1f TF/s  checksum %.6f" % (os.path.basename(os.environ.get("DD_LIB", "default")), us, 4.0 * B * H * N * N * D / us / 1e6,
      float(o0.float().abs().mean())))
