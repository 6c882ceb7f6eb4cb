#!/bin/bash
# same-device A/B of engine-level switches on the bench workload: tools/ab_env.sh "NAME=ENV=VAL ..." ...   (interleaved, two rounds)
# e.g. tools/ab_env.sh "fold=" "nofold=DD_NO_LN_FOLD=1" "norows=DD_NO_LN_ROWSTATS=1"   -> gpurun_out/ab_<name>.json, ops_<name>.csv
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs DD_PROFILE_DUMP=$PWD/gpurun_out/ops_$name.csv timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 2>gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_$name.json || exit 1
    python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read())
print(sys.argv[1], "img/s %.3f" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["roofline"].get("family_ms"))
PY
  done
done
