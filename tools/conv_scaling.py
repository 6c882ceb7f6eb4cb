"""time = a + b * K-steps ? 3x3 conv at fixed M/N, growing Cin (K-steps per tile) and batch (tiles per CU)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(__file__), "bench_conv.py"))
src = open(spec.origin).read().split("\nB = 16\n")[0]
exec(src)
for B in (8, 16, 32):
    for cin in (320, 640, 1280):
        run("B=%d 3x3 %d->320 @64 (%d K-steps)" % (B, cin, cin * 9 // 64), B, 64, cin, 320, 3)
for cin in (320, 640, 1280, 2560):
    run("B=16 1x1 %d->320 @64 (%d K-steps)" % (cin, cin // 64), 16, 64, cin, 320, 1)
