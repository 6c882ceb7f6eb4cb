"""Output stage throughput: 512x512 RGB PNG files per second of `save_png` on a host thread pool.
    python tools/png_rate.py [threads ...]
The expansion needs ~11 PNG/s per GPU (88 per 8-GPU node); `AsyncPNGWriter` runs 4 encoder threads per rank."""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from distdiff_amd.generate_data import save_png  # noqa: E402

rng = np.random.RandomState(0)
x = np.linspace(0, 1, 512)
base = (np.sin(8 * x)[:, None] * np.cos(5 * x)[None, :] * 0.4 + 0.5)[..., None] * np.ones(3)
images = {"smooth + 5 % noise": (np.clip(base + rng.randn(512, 512, 3) * 0.05, 0, 1) * 255).astype(np.uint8),
          "pure noise (worst case)": rng.randint(0, 256, (512, 512, 3), dtype=np.uint8)}
d = tempfile.mkdtemp()
for name, im in images.items():
    for th in [int(v) for v in sys.argv[1:]] or [1, 4, 8]:
        n = 16 * th
        t0 = time.time()
        with ThreadPoolExecutor(th) as ex:
            list(ex.map(lambda i: save_png(im, os.path.join(d, "a%d.png" % i)), range(n)))
        print("%-24s %2d threads: %6.1f PNG/s  (%d KB per file)" % (name, th, n / (time.time() - t0), os.path.getsize(os.path.join(d, "a0.png")) // 1024))
