#!/bin/bash
# SQ / TCP counters of the conv kernels (round 3: the halo-resident ping-pong kernel) on two deep 3x3 shapes: bash tools/conv_pmc.sh -> gpurun_out/conv_pmc.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/conv_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/tools/conv_pmc.py > /dev/null 2> $OUT/p$i.err || true
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/conv_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_gemm_big_kernel" not in n and "conv_halo_kernel" not in n: continue
        key = "2wg 128x160" if "<2, 2, 4, 5, 2" in n else "8-wave 256x160" if "<4, 2, 4, 5, 3" in n else "halo 512x160" if "conv_halo_kernel<5, 2>" in n else "halo 256x320" if "conv_halo_kernel<5, 4>" in n else n[:60]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
with open(root + "/../conv_pmc.txt", "w") as out:
    for k in acc:
        out.write("== %s (per launch)\n" % k)
        for c in sorted(acc[k]):
            out.write("  %-32s %16.0f\n" % (c, acc[k][c] / cnt[k][c]))
print(open(root + "/../conv_pmc.txt").read())
PY
rm -rf $OUT
