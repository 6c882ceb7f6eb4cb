#!/bin/bash
# same-device A/B of library builds on the bench workload (interleaved, two rounds): tools/ab_lib.sh ab/a.so ab/b.so
for round in 1 2; do
  for L in "$@"; do
    n=$(basename $L .so)
    DD_LIB=$PWD/$L DD_PROFILE_DUMP=$PWD/gpurun_out/ops_$n.csv timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'img/s %.3f' % d['value'], 'ms %.1f' % d['ms_per_step'], d['roofline']['family_ms'])"
  done
done
