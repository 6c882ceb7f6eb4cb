#!/bin/bash
# SQ counters of the attention forward (d = 40, 4096 keys): bash tools/attn_pmc.sh -> gpurun_out/attn_pmc.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/attn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/tools/attn_pmc.py > /dev/null 2> $OUT/p$i.err || true
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/attn_pmc"
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open(root + "/../attn_pmc.txt", "w") as out:
    out.write("== attn_fwd_dma_kernel<40, 2, 64, 8, lazy>, B 32 x 8 heads x 4096 x 4096, prescaled query (per launch)\n")
    for c in sorted(acc):
        out.write("  %-32s %16.0f\n" % (c, acc[c] / cnt[c]))
print(open(root + "/../attn_pmc.txt").read())
PY
rm -rf $OUT
