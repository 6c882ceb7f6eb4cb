for v in 2048 1024 4096 8192 2048; do
  DD_GN_BLOCKS=$v timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_blocks', $v, round(d['value'],3), round(d['roofline']['family_ms']['norm'],1))"
done
