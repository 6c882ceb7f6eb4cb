"""Two deep 3x3 convolutions through the op ABI, a few launches each, for rocprofv3 --pmc passes (tools/conv_pmc.sh)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops


def run(B, H, Cin, Cout, k, iters=6):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    pk = ops.PackedConv(w, k // 2, bias=torch.randn(Cout, generator=g))
    M = B * H * H
    x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).cuda()
    y = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    for _ in range(iters):
        ops.conv_gemm(x, pk, B, H, H, H, H, y=y, partial=part)
    torch.cuda.synchronize()


B = int(os.environ.get("PMC_B", "64"))   # the bench's CFG batch
run(B, 64, 320, 320, 3)      # round 2: 128x160 tiles, two 4-wave workgroups per CU, 45 K-steps per item; round 3: halo kernel, 512x160 tiles
run(B, 16, 1280, 1280, 3)    # round 2: 256x160 tiles, one 8-wave workgroup per CU, 180 K-steps per item; round 3: halo kernel, 256x320 tiles
