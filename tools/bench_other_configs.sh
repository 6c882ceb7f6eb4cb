#!/bin/bash
# the other configurations of SURVEY.md 8(d) with the default batch: one JSON line each under gpurun_out/other_*.json
python bench.py --strength 1.0 --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 > gpurun_out/other_strength1_transform.json 2>/dev/null
python bench.py --classes 196 --guidance direct_guidance --guidance_step 10 --guidance_period 10 --strength 1.0 --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 > gpurun_out/other_config4_direct_last10.json 2>/dev/null
python bench.py --guidance none --strength 1.0 --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 > gpurun_out/other_guidance_off.json 2>/dev/null
for f in gpurun_out/other_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],3), d['config']['images_per_step_per_gpu'], round(d['config']['algorithmic_tflop_per_image'],1), round(d['e2e_tflops_per_gpu'],1), round(d['config']['workspace_gb'],1))"; done
python bench.py --config sdxl --steps 2 --warmup 1 --no_cpu_baseline --no_cli --no_strength1 > gpurun_out/other_sdxl_1024.json 2>/dev/null
python -c "
import json,glob; print(json.dumps({f.split('other_')[1][:-5]: json.load(open(f)) for f in sorted(glob.glob('gpurun_out/other_*.json'))}, indent=1))" > gpurun_out/bench_other_configs.json
