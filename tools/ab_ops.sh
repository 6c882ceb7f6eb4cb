#!/bin/bash
# same-device per-op comparison of two library builds on the bench workload: tools/ab_ops.sh ab/a.so ab/b.so  -> gpurun_out/ops_<name>.csv
for L in "$@"; do
  n=$(basename $L .so)
  DD_LIB=$PWD/$L DD_PROFILE_DUMP=$PWD/gpurun_out/ops_$n.csv timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', d['value'], d['roofline']['family_ms'])"
done
