for v in 1 0 1 0; do
  DD_LN_ROWS=$v timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ln_rows', $v, round(d['value'],3), round(d['roofline']['family_ms']['norm'],1))"
done
