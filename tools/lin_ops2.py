import sys, os, math, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops, _lib
def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()
g = torch.Generator().manual_seed(0)
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
for (B, Cin, Cout, H, k, stride, pad, up) in [(2, 64, 128, 16, 3, 1, 1, 0), (2, 320, 320, 16, 3, 1, 1, 0), (2, 128, 64, 16, 3, 1, 1, 1), (2, 64, 64, 16, 3, 2, 1, 0),
                                               (2, 4, 64, 16, 3, 1, 1, 0), (2, 64, 4, 16, 3, 1, 1, 0), (2, 1280, 128, 8, 3, 1, 1, 0), (8, 320, 320, 64, 3, 1, 1, 0)]:
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    Hl = H << up
    Ho = (Hl + 2 * pad - k) // stride + 1
    dy = torch.randn(B * Ho * Ho, (Cout + 7) // 8 * 8, generator=g).to(torch.bfloat16).cuda()
    if Cout % 8: dy[:, Cout:] = 0
    pkd = ops.PackedConv(w, pad, mode=1)
    kw = dict(stride=1, shift=1, parity=1) if stride == 2 else dict(stride=1)
    d1 = ops.conv_gemm(dy, pkd, B, Ho, Ho, Hl, Hl, **kw)
    d2 = ops.conv_gemm((dy.float() * -4).to(torch.bfloat16), pkd, B, Ho, Ho, Hl, Hl, **kw)
    # accumulate form
    acc = torch.randn_like(d1)
    a1 = ops.conv_gemm(dy, pkd, B, Ho, Ho, Hl, Hl, y=acc.clone(), res=None, **kw)
    print("dgrad", (B, Cin, Cout, H, k, stride, up), rel(d2, -4 * d1.float()))
    x = torch.randn(B * H * H, (Cin + 7) // 8 * 8, generator=g).to(torch.bfloat16).cuda()
    pk = ops.PackedConv(w, pad, mode=0)
    f1 = ops.conv_gemm(x, pk, B, H, H, Ho, Ho, stride=stride, shift=up)
    f2 = ops.conv_gemm((x.float() * -4).to(torch.bfloat16), pk, B, H, H, Ho, Ho, stride=stride, shift=up)
    print("  fwd", rel(f2, -4 * f1.float()))
M, F = 300, 256
raw = torch.randn(M, 2 * F, generator=g).to(torch.bfloat16).cuda(); d = torch.randn(M, F, generator=g).to(torch.bfloat16).cuda()
o1 = torch.zeros(M, 2 * F, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros_like(o1)
d4 = (d.float() * -4).to(torch.bfloat16)
L.dd_op_geglu_bwd(P(raw), 2 * F, P(d), F, P(o1), 2 * F, M, F, None); L.dd_op_geglu_bwd(P(raw), 2 * F, P(d4), F, P(o2), 2 * F, M, F, None)
print("geglu", rel(o2, -4 * o1.float()))
