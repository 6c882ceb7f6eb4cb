"""Per-IMAGE time of every op shape at several engine batches, from DD_PROFILE_DUMP csvs of the bench step (is a
producer -> consumer pair faster when the tensor between them fits the 256 MB Infinity Cache, i.e. at 4 / 8 images instead of 32?).

    python tools/per_image_compare.py 32:ops_b32.csv 8:ops_b8.csv 4:ops_b4.csv > profiles/r06_depth_first_per_image.txt

Rows are keyed by (family, bwd, M per image, N, K); columns: ms per image at each batch, and the ratio to the first batch."""
import collections
import csv
import sys

FAM = {"0": "conv/linear", "1": "attention", "2": "norm", "3": "other"}
runs = []
for arg in sys.argv[1:]:
    b, path = arg.split(":", 1)
    b = int(b)
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        M = int(r["M"])
        # conv / norm rows: M = rows of the whole (CFG-doubled or not) batch -> per image; attention rows carry Nq (already per image)
        per = M // b if r["fam"] != "1" and M % b == 0 else M
        k = (r["fam"], r["bwd"], per, int(r["N"]), int(r["K"]))
        acc[k][0] += 1
        acc[k][1] += float(r["ms"])
        acc[k][2] += float(r["flops"])
    runs.append((b, acc))
b0, a0 = runs[0]
keys = sorted(a0.keys(), key=lambda k: -a0[k][1])
print("# family bwd M/image N K launches | ms per image at B = %s | ratio to B = %d" % (", ".join(str(b) for b, _ in runs), b0))
tot = collections.defaultdict(lambda: [0.0] * len(runs))
for k in keys:
    ms = [a[k][1] / b if k in a else float("nan") for b, a in runs]
    for i, v in enumerate(ms):
        if v == v:
            tot[k[0]][i] += v
            if k[0] == "0" and k[4] <= 320:
                tot["0:K<=320"][i] += v
    if a0[k][1] / b0 < 0.02:
        continue
    print("%-12s %s %8d %6d %6d %5d | %s | %s" % (FAM.get(k[0], k[0]), k[1], k[2], k[3], k[4], a0[k][0], " ".join("%8.3f" % v for v in ms),
                                                 " ".join("%5.2f" % (v / ms[0]) for v in ms[1:])))
print("# totals, ms per image")
for f, v in sorted(tot.items()):
    print("%-12s | %s | %s" % (FAM.get(f, f), " ".join("%8.2f" % x for x in v), " ".join("%5.2f" % (x / v[0]) for x in v[1:])))
