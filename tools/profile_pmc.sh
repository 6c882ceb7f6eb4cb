#!/bin/bash
# the two PMC passes of tools/profile_bench.sh on their own (each its own rocprofv3 run): bash tools/profile_pmc.sh <tag>
TAG=${1:-b32}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 --no_profile $PMC_BENCH_ARGS > $OUT/fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?" >> $OUT/progress
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 --no_profile $PMC_BENCH_ARGS > $OUT/write.json 2> $OUT/write.err
echo "write rc=$?" >> $OUT/progress
cd $ROOT
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python3 tools/pmc_summary.py $F $W > $OUT/pmc_traffic.json; fi
rm -rf $OUT/fetch $OUT/write
cat $OUT/progress; ls -la $OUT
