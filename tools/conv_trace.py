"""Timeline of workgroup 0 of the persistent conv kernel (debug build: DD_EXTRA_CFLAGS=-DDD_TRACE python -m distdiff_amd.build --force):
per work item the time of its K loop and of its epilogue, in microseconds."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from distdiff_amd import ops
from distdiff_amd._lib import lib as load_library

L = load_library()


def run(name, B, H, Cin, Cout, k, res=False, geglu=False):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    pk = ops.PackedConv(w, k // 2, geglu=geglu, bias=torch.randn(Cout, generator=g))
    M = B * H * H
    x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).cuda()
    ncol = Cout // 2 if geglu else Cout
    y = torch.empty(M, ncol, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(M, ncol, generator=g).to(torch.bfloat16).cuda() if res else None
    for _ in range(3):
        ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r)
    torch.cuda.synchronize()
    L.dd_debug_clear_trace()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_gemm(x, pk, B, H, H, H, H, y=y, res=r)
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    L.dd_debug_read_trace(buf, 512)
    t0 = buf[0]
    out = []
    for o, who in ((0, "first wg"), (256, "last wg")):
        ev = [((buf[o + 2 * i] - t0) / 100.0, buf[o + 2 * i + 1]) for i in range(127) if buf[o + 2 * i]]
        entry, loop0, exit_ = ev[0][0], ev[1][0], ev[-1][0]
        items, last, i = [], loop0, 2
        while i + 1 < len(ev) and ev[i][1] == 1 and ev[i + 1][1] == 2:
            items.append((ev[i][0] - last, ev[i + 1][0] - ev[i][0]))
            last = ev[i + 1][0]
            i += 2
        out.append("%s: entry %+.1f, prologue %.1f, %d items (K loop, epilogue) %s, exit %+.1f" %
                   (who, entry, loop0 - entry, len(items), " ".join("(%.1f %.1f)" % it for it in items[:6]), exit_))
    print("%-28s kernel %.1f us (events)\n    %s\n    %s" % (name, e0.elapsed_time(e1) * 1000, out[0], out[1]))


B = 32
run("1x1 320->320 +res @64", B, 64, 320, 320, 1, res=True)
run("1x1 320->320 @64", B, 64, 320, 320, 1)
run("1x1 320->2560 geglu @64", B, 64, 320, 2560, 1, geglu=True)
run("1x1 640->640 +res @32", B, 32, 640, 640, 1, res=True)
run("1x1 1280->320 +res @64", B, 64, 1280, 320, 1, res=True)
run("3x3 320->320 @64", B, 64, 320, 320, 3)
run("3x3 640->640 @32", B, 32, 640, 640, 3)
run("3x3 1280->1280 @16", B, 16, 1280, 1280, 3)
