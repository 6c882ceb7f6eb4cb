"""deep-K / small-M and GEGLU shapes of the UNet at CFG batch 32 (B=16 images)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
src = open(os.path.join(os.path.dirname(__file__), "bench_conv.py")).read().split("\nB = 16\n")[0]
exec(src)
run("3x3 1280->1280 @16 M=8192", 32, 16, 1280, 1280, 3)
run("3x3 2560->1280 @16 M=8192", 32, 16, 2560, 1280, 3)
run("3x3 1280->1280 @8 M=2048", 32, 8, 1280, 1280, 3)
run("3x3 1920->640 @32 M=32768", 32, 32, 1920, 640, 3)
run("3x3 640->640 @32 M=32768", 32, 32, 640, 640, 3)
run("1x1 320->2560 geglu M=131072", 32, 64, 320, 2560, 1, geglu=True, raw=True)
run("1x1 640->5120 geglu M=32768", 32, 32, 640, 5120, 1, geglu=True, raw=True)
run("1x1 1280->10240 geglu M=8192", 32, 16, 1280, 10240, 1, geglu=True, raw=True)
