for round in 1 2; do
  for spec in "pregn:ab/pregn.so:" "gncode_off:ab/gncode.so:DD_NO_GN_APPLY_FUSION=1" "gncode_on:ab/gncode.so:"; do
    n=${spec%%:*}; rest=${spec#*:}; L=${rest%%:*}; E=${rest#*:}
    env $E DD_LIB=$PWD/$L timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'img/s %.3f' % d['value'], 'ms %.1f' % d['ms_per_step'], d['roofline']['family_ms'])"
  done
done
