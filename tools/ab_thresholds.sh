for v in 100 60 160 240; do
  DD_CONV_2WG=$v DD_PROFILE_DUMP=$PWD/gpurun_out/ops_2wg$v.csv timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2wg', $v, d['value'], d['roofline']['family_ms']['conv_gemm'])"
done
for v in 12 48; do
  DD_CONV_2WG128=$v timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2wg128', $v, d['value'], d['roofline']['family_ms']['conv_gemm'])"
done
