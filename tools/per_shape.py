"""Aggregates a DD_PROFILE_DUMP csv (one HIP-event-timed row per op of one bench step) per shape:  python tools/per_shape.py ops.csv > per_shape.txt"""
import collections
import csv
import sys

FAM = {"0": "conv/linear", "1": "attention", "2": "norm", "3": "other"}
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["fam"], r["bwd"], int(r["M"]), int(r["N"]), int(r["K"]))
    acc[k][0] += 1
    acc[k][1] += float(r["ms"])
    acc[k][2] += float(r["flops"])
print("# family bwd M N K launches total_ms TFLOP/s   (conv/linear: M = output pixels x batch, N = Cout, K = taps x Cin; attention: M = Nq, N = Nk, K = d)")
for k, (n, ms, fl) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-12s %s %8d %6d %6d %5d %9.2f %7.0f" % (FAM.get(k[0], k[0]), k[1], k[2], k[3], k[4], n, ms, fl / ms / 1e9 if ms > 0 else 0))
