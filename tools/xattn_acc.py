import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from distdiff_amd import ops
torch.manual_seed(0)
for (B,H,Nq,Nk,D,gain) in [(2,8,4096,77,40,1.0),(2,8,4096,77,40,2.5),(2,8,1024,77,80,1.0),(2,8,1024,77,80,2.5)]:
    q=(torch.randn(B,Nq,H,D)*gain).to(torch.bfloat16); k=(torch.randn(B,Nk,H,D)*gain).to(torch.bfloat16); v=torch.randn(B,Nk,H,D).to(torch.bfloat16)
    c=1/math.sqrt(D)*1.4426950408889634
    qs=(q.float()*c).to(torch.bfloat16)
    s=torch.einsum("bqhd,bkhd->bhqk", qs.float(), k.float())*0.6931471805599453
    ref=torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.float())
    dev=lambda t,n: t.reshape(B*n,H*D).cuda()
    o,lse=ops.attention(dev(qs,Nq),dev(k,Nk),dev(v,Nk),B,H,Nq,Nk,D,0.6931471805599453,q_prescaled=True)
    torch.cuda.synchronize()
    e=(o.float().cpu().reshape(B,Nq,H,D)-ref)
    print("SHORTK=%s d=%d gain %.1f: O rel-L2 %.5f  max abs %.5f  lse max err %.5f" % (os.environ.get("DD_ATTN_SHORTK","1"),D,gain,float(e.norm()/ref.norm()),float(e.abs().max()),float((lse.cpu()-torch.logsumexp(s,-1)).abs().max())))
