#!/bin/bash
# Host launch path: eager vs hipGraph replay at several batch sizes (configs[1] workload), one line each.
for B in 32 256 1024; do
  for mode in "" "--graph" "--serial-streams" "--graph --serial-streams"; do
    timeout -k 10 200 python bench.py --workload config1 --batch $B $mode --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=$B mode=[$mode]', 'windows/s', round(d['value']), 'ms/step', d['ms_per_step'], 'host_enqueue_ms', d['config'].get('host_enqueue_ms_per_step'))" || exit 1
  done
done
