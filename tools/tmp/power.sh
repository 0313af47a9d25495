#!/bin/bash
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/pw_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 16); do
  sleep 2
  echo "t=$((i*2))s $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i -E 'Package Power|sclk' | sed 's/.*: //' | tr '\n' ' ')"
done
wait $BP
tail -1 gpurun_out/pw_bench.json | cut -c1-160
