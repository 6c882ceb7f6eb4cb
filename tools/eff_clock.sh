#!/bin/bash
# effective shader clock per kernel family during the bench: GRBM_GUI_ACTIVE cycles / kernel duration (rocprofv3 --pmc, its own run)
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/effclk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/run -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_profile $EFFCLK_ARGS > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
ls $OUT/run/* | head
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/effclk/run"
cc = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cc)))
print(rows[0].keys())
fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    n = r["Kernel_Name"]
    f = "conv_halo" if "conv_halo" in n else "gemm_pps" if "gemm_pps" in n else "conv_old" if "conv_gemm" in n else "attn" if "attn" in n else "norm" if ("gn_" in n or "ln_" in n) else "other"
    fam[f][0] += float(r["Counter_Value"]); fam[f][1] += dur; fam[f][2] += 1
for f, (cyc, ns, k) in fam.items():
    print("%-6s %6d launches  %.1f ms  GRBM_GUI_ACTIVE/ns = %.3f (x8 XCDs if summed: %.3f GHz per XCD)" % (f, k, ns / 1e6, cyc / ns, cyc / ns / 8))
PY
rm -rf $OUT/run
