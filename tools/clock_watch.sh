#!/bin/bash
# samples the shader clock and the socket power while the bench workload runs: tools/clock_watch.sh -> gpurun_out/clock_watch.log
python bench.py --steps 6 --warmup 1 --no_cpu_baseline --no_profile > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
rocm-smi --showpower --showclocks > gpurun_out/clock_raw.txt 2>&1
for i in $(seq 1 80); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -iE 'sclk|power' | sed 's/GPU\[0\]//; s/\s\+/ /g' | tr '\n' ' ')" >> gpurun_out/clock_watch.log
  sleep 0.5
done
wait $BP
