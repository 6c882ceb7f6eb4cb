#!/bin/bash
# rocprofv3 passes of the default bench command (one timed step): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs.
#   bash tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -e
TAG=${1:-b32}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 > $OUT/stats.json 2> $OUT/stats.err
echo "stats done" > $OUT/progress
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 --no_profile > $OUT/fetch.json 2> $OUT/fetch.err
echo "fetch done" >> $OUT/progress
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 --no_profile > $OUT/write.json 2> $OUT/write.err
echo "write done" >> $OUT/progress
cd $ROOT
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W > $OUT/pmc_traffic.json
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $S $OUT/kernel_stats.csv
# the raw traces are large: keep only the summaries
rm -rf $OUT/fetch $OUT/write $OUT/stats
ls -la $OUT
