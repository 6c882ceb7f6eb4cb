import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from distdiff_amd import ops
def bf(x): return x.to(torch.bfloat16).float()
B, Cin, Cout, H, W, k = 2, 1280, 320, 24, 24, 3
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
x = bf(torch.randn(B, Cin, H, W, generator=g)); w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * 9)); bias = torch.randn(Cout, generator=g)
ref = F.conv2d(x, w, bias, padding=1)
xd = ops.to_nhwc_bf16(x).cuda(); pk = ops.PackedConv(w, 1, bias=bias)
for ks in (1, 2, 4, 5, 10, 20, 0):
    for rep in range(2):
        y = ops.conv_gemm(xd, pk, B, H, W, H, W, ksplit=ks)
        torch.cuda.synchronize()
        got = ops.from_nhwc(y, B, H, W).cpu()
        err = (got - ref).abs()
        bad = (err > 0.15).nonzero()
        msg = ""
        if len(bad):
            e2 = (y.float().cpu() - ref.permute(0, 2, 3, 1).reshape(-1, Cout)).abs() > 0.15
            rows = e2.any(1).nonzero().flatten(); cols = e2.any(0).nonzero().flatten()
            msg = "rows %d..%d (%d) cols %d..%d (%d)" % (rows.min(), rows.max(), len(rows), cols.min(), cols.max(), len(cols))
        print("ksplit", ks, "rep", rep, "max err %.3f" % err.max().item(), msg)
