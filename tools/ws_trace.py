"""Per-wave timeline of gemm_ws_kernel, workgroup 0 (debug build: tools/build_variant.sh ws_trace gemm_ws.hip -DWS_TRACE, run with DD_LIB=ab/ws_trace.so):
per tile and wave: DMA issue, then MFMA (+ next fragment reads / ring hand-over) and epilogue of the four 16-row steps, in shader cycles."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
from distdiff_amd._lib import lib as load_library

L = load_library()


def run(name, M, N, res=False, geglu=False, lnfold=False):
    K = 320
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    pk = ops.PackedConv(w, 0, geglu=geglu, bias=torch.randn(N, generator=g))
    pc = ops.PackedConv(w, 0, geglu=geglu, bias=w.sum(dim=1))
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    st = torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], 1).contiguous() if lnfold else None
    f = lambda: ops.conv_gemm(x, pk, 1, M, 1, M, 1, y=y, res=r, ksplit=1, ln_stats=st, ln_c1=pc.bias if lnfold else None)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 16 * 10))()
    L.dd_debug_read_ws_trace(buf, 8 * 16 * 10)
    print("== %s" % name)
    t00 = min(buf[(w_ * 16) * 10] for w_ in range(8))
    for w_ in range(8):
        rows = []
        for it in range(2, 8):
            b = [buf[(w_ * 16 + it) * 10 + i] for i in range(10)]
            nxt = buf[(w_ * 16 + it + 1) * 10]
            rows.append([b[1] - b[0]] + [b[i + 1] - b[i] for i in range(1, 9)] + [nxt - b[0]])
        avg = [sum(r_[i] for r_ in rows) / len(rows) for i in range(10)]
        print("wave %d: start %6d | dma %5.0f | mfma/epi %s | tile %6.0f cycles" % (w_, buf[(w_ * 16 + 2) * 10] - t00, avg[0], " ".join("%5.0f/%5.0f" % (avg[1 + 2 * a], avg[2 + 2 * a]) for a in range(4)), avg[9]))


run("N 320", 262144, 320)
run("N 320 + res", 262144, 320, res=True)
run("N 960 lnfold", 262144, 960, lnfold=True)
run("GEGLU 2560 lnfold", 262144, 2560, geglu=True, lnfold=True)
