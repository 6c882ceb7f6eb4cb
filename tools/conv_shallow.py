"""shallow-K shapes of the UNet at CFG batch 32: big persistent kernel vs the 4-wave 128x128 kernel (force_small)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
src = open(os.path.join(os.path.dirname(__file__), "bench_conv.py")).read().split("\nB = 16\n")[0]
exec(src)
for small in (False, True):
    tag = " small" if small else ""
    run("1x1 320->320 +res M=131072" + tag, 32, 64, 320, 320, 1, res=True, force_small=small)
    run("1x1 320->960 qkv M=131072" + tag, 32, 64, 320, 960, 1, bias=False, force_small=small)
    run("1x1 640->640 +res M=32768" + tag, 32, 32, 640, 640, 1, res=True, force_small=small)
    run("1x1 1280->1280 +res M=8192" + tag, 32, 16, 1280, 1280, 1, res=True, force_small=small)
    run("1x1 1280->320 +res M=131072" + tag, 32, 64, 1280, 320, 1, res=True, force_small=small)
    run("1x1 320->2560 geglu M=131072" + tag, 32, 64, 320, 2560, 1, geglu=True, force_small=small)
