import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops

def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()
g = torch.Generator().manual_seed(0)
M, K, N = 512, 4608, 256
x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = torch.randn(N, K, generator=g) / math.sqrt(K)
pk = ops.PackedConv(w, 0)
for ks in (1, 4):
    for sc in (-4.0, 2.0, -1.0):
        y1 = ops.conv_gemm(x, pk, 1, M, 1, M, 1, ksplit=ks)
        y2 = ops.conv_gemm((x.float() * sc).to(torch.bfloat16), pk, 1, M, 1, M, 1, ksplit=ks)
        print("gemm ksplit", ks, "scale", sc, rel(y2, sc * y1.float()))
B, HW, C, G = 2, 1024, 320, 32
xg = (torch.randn(B * HW, C, generator=g) * 2).to(torch.bfloat16).cuda()
ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
dy = torch.randn(B * HW, C, generator=g).to(torch.bfloat16).cuda()
y, st = ops.groupnorm(xg, ga, be, B, HW, G, 1e-5, True)
d1 = ops.groupnorm(xg, ga, be, B, HW, G, 1e-5, True, dy=dy, stats=st)
d2 = ops.groupnorm(xg, ga, be, B, HW, G, 1e-5, True, dy=(dy.float() * -4).to(torch.bfloat16), stats=st)
print("gn bwd", rel(d2, -4 * d1.float()))
xl = torch.randn(300, 320, generator=g).to(torch.bfloat16).cuda()
gl, bl = torch.randn(320, generator=g).cuda(), torch.randn(320, generator=g).cuda()
yl, sl = ops.layernorm(xl, gl, bl, 1e-5)
dl = torch.randn(300, 320, generator=g).to(torch.bfloat16).cuda()
e1 = ops.layernorm(xl, gl, bl, 1e-5, dy=dl, stats=sl); e2 = ops.layernorm(xl, gl, bl, 1e-5, dy=(dl.float() * -4).to(torch.bfloat16), stats=sl)
print("ln bwd", rel(e2, -4 * e1.float()))
Bq, H, Nq, D = 1, 8, 256, 40
q, k, v, do = (torch.randn(Bq * Nq, H * D, generator=g).to(torch.bfloat16).cuda() for _ in range(4))
o1 = ops.attention(q, k, v, Bq, H, Nq, Nq, D, 1 / math.sqrt(D), d_o=do)
o2 = ops.attention(q, k, v, Bq, H, Nq, Nq, D, 1 / math.sqrt(D), d_o=(do.float() * -4).to(torch.bfloat16))
for i, n in zip((2, 3, 4), ("dq", "dk", "dv")):
    print("attn", n, rel(o2[i], -4 * o1[i].float()))
for (Bq, H, Nq, Nk, D) in [(2, 2, 256, 256, 32), (2, 2, 64, 64, 64), (2, 2, 256, 13, 32), (1, 8, 256, 77, 40), (1, 8, 64, 64, 160), (1, 8, 64, 77, 160), (1, 8, 256, 256, 80), (1, 1, 256, 256, 512), (2, 1, 256, 256, 64)]:
    q, do = (torch.randn(Bq * Nq, H * D, generator=g).to(torch.bfloat16).cuda() for _ in range(2))
    k, v = (torch.randn(Bq * Nk, H * D, generator=g).to(torch.bfloat16).cuda() for _ in range(2))
    cross = Nk in (13, 77)
    o1 = ops.attention(q, k, v, Bq, H, Nq, Nk, D, 1 / math.sqrt(D), d_o=do, need_dkv=not cross)
    o2 = ops.attention(q, k, v, Bq, H, Nq, Nk, D, 1 / math.sqrt(D), d_o=(do.float() * -4).to(torch.bfloat16), need_dkv=not cross)
    print("attn", (Bq, H, Nq, Nk, D), [rel(o2[i], -4 * o1[i].float()) for i in ((2,) if cross else (2, 3, 4))])
for (B, HW, C, G, silu, eps) in [(2, 256, 64, 8, 1, 1e-5), (2, 256, 128, 8, 0, 1e-6), (1, 4096, 320, 32, 1, 1e-5), (2, 16, 128, 8, 1, 1e-5)]:
    xg = (torch.randn(B * HW, C, generator=g) * 2).to(torch.bfloat16).cuda()
    ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    dy = torch.randn(B * HW, C, generator=g).to(torch.bfloat16).cuda()
    y, st = ops.groupnorm(xg, ga, be, B, HW, G, eps, silu)
    d1 = ops.groupnorm(xg, ga, be, B, HW, G, eps, silu, dy=dy, stats=st)
    d2 = ops.groupnorm(xg, ga, be, B, HW, G, eps, silu, dy=(dy.float() * -4).to(torch.bfloat16), stats=st)
    print("gn bwd", (B, HW, C, G, silu), rel(d2, -4 * d1.float()))
