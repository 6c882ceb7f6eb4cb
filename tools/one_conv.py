"""Runs one conv shape repeatedly (for rocprofv3 --pmc).  usage: one_conv.py B H Cin Cout k [iters]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
B, H, Cin, Cout, k = map(int, sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = torch.Generator().manual_seed(0)
w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
pk = ops.PackedConv(w, k // 2, bias=torch.randn(Cout, generator=g))
M = B * H * H
x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).cuda()
y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
part = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device="cuda")
for _ in range(iters):
    ops.conv_gemm(x, pk, B, H, H, H, H, y=y, partial=part)
torch.cuda.synchronize()
print("done", M, Cout, Cin * k * k)
