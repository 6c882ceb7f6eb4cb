"""K = 320 pointwise shapes of the 64x64 level (bench batch: M = 262144 / 131072) through the op-level ABI: weight-stationary GEMM
(gemm_ws.hip) against the streaming ping-pong GEMM (DD_GEMM_WS=0).  python tools/bench_ws.py [K]"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(M, K, N, res=False, lnfold=False, geglu=False, rowstats=False, raw=False, iters=20):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    pk = ops.PackedConv(w, 0, geglu=geglu, bias=torch.randn(N, generator=g))
    pc = ops.PackedConv(w, 0, geglu=geglu, bias=w.sum(dim=1))
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ncols = N // 2 if geglu else N
    y = torch.empty(M, ncols, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    st = torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], 1).contiguous() if lnfold else None
    part = torch.zeros((M, N // 40, 2), device="cuda") if rowstats else None
    rw = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if raw else None
    f = lambda: ops.conv_gemm(x, pk, 1, M, 1, M, 1, y=y, res=r, ksplit=1, ln_stats=st, ln_c1=pc.bias if lnfold else None, rowpart=part, raw=rw)
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
    byt = (M * K + M * ncols + (M * N if res else 0) + (M * N if raw else 0)) * 2
    print("M %6d N %5d K %4d res %d ln %d geglu %d rs %d raw %d: %7.1f us %6.0f TF/s %5.2f TB/s" % (M, N, K, res, lnfold, geglu, rowstats, raw, us, 2.0 * M * N * K / us / 1e6, byt / us / 1e6), flush=True)


print("ws=" + os.environ.get("DD_GEMM_WS", "1"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 320
if os.environ.get("WS_SHORT"):
    run(262144, K, 320)
    run(262144, K, 320, res=True, rowstats=True)
    run(262144, K, 960, lnfold=True)
    run(262144, K, 2560, lnfold=True, geglu=True)
    sys.exit(0)
for M in (262144, 131072):
    run(M, K, 320)
    run(M, K, 320, res=True, rowstats=True)
    run(M, K, 320, lnfold=True)
    run(M, K, 960, lnfold=True)
    run(M, K, 2560, lnfold=True, geglu=True)
run(262144, K, 2560, lnfold=True, geglu=True, raw=True)
