"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into a per-kernel-family HBM traffic summary.

  python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, so wide streaming reads are doubled; WRITE_SIZE is exact.  Each pass is its own run.
"""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = (("conv_gemm_big_kernel", "conv_gemm_big_kernel"), ("conv_gemm_kernel", "conv_gemm_kernel"),
            ("attn_fwd_kernel", "attn_fwd"), ("attn_bwd", "attn_bwd"), ("gn_", "gn_"),
            ("ln_", "ln_"), ("conv_halo_kernel", "conv_halo_"), ("gemm_pp_kernel", "gemm_pp"), ("gemm_ws_kernel", "gemm_ws"), ("splitk", "splitk"))


def fold(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            fam = next((k for k, pat in FAMILIES if pat in name), "other")
            tot[fam] += float(r["Counter_Value"]) * 1024.0
            cnt[fam] += 1
    return tot, cnt


def main():
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, _ = fold(sys.argv[2], "WRITE_SIZE")
    out = {}
    for fam in nf:
        out[fam] = {"launches": nf[fam], "hbm_fetch_bytes_corrected": 2.0 * fetch[fam],
                    "hbm_write_bytes": write.get(fam, 0.0)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
