import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
def run(B, H, Nq, Nk, D, bwd=False, iters=10, gemm=False):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B * Nq, H * D, generator=g).to(torch.bfloat16).cuda()
    k = torch.randn(B * Nk, H * D, generator=g).to(torch.bfloat16).cuda()
    v = torch.randn(B * Nk, H * D, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * Nq, H * D, generator=g).to(torch.bfloat16).cuda() if bwd else None
    f = (lambda: ops.attention_gemm(q, k, v, B, H, Nq, Nk, D, 1 / math.sqrt(D), d_o=do)) if gemm else \
        (lambda: ops.attention(q, k, v, B, H, Nq, Nk, D, 1 / math.sqrt(D), d_o=do, need_dkv=(Nk != 77)))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    fl = 4.0 * B * H * Nq * Nk * D * (3.5 if bwd and Nk != 77 else 2.5 if bwd else 1)
    print("attn B%d H%d Nq%d Nk%d D%d %s%s  %9.1f us %7.1f TF/s" % (B, H, Nq, Nk, D, "fwd+bwd" if bwd else "fwd", " (GEMM path)" if gemm else "", us, fl / us / 1e6))
run(16, 8, 4096, 4096, 40); run(16, 8, 1024, 1024, 80); run(16, 8, 256, 256, 160); run(16, 8, 4096, 77, 40)
run(8, 1, 4096, 4096, 512, iters=3); run(8, 1, 4096, 4096, 512, bwd=True, iters=3); run(8, 8, 4096, 4096, 40, bwd=True, iters=3)
run(8, 1, 4096, 4096, 512, iters=3, gemm=True); run(8, 1, 4096, 4096, 512, bwd=True, iters=3, gemm=True)
