#!/bin/bash
# same-box A/B of one environment switch: tools/ab_env.sh VAR [bench.py args]  -> alternating runs with VAR=0 / VAR unset
VAR=$1; shift
for i in 1 2; do
  for v in 0 1; do
    if [ $v = 0 ]; then export $VAR=0; else unset $VAR; fi
    python bench.py "$@" --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null > gpurun_out/ab_env.json
    python -c "import json;d=json.load(open('gpurun_out/ab_env.json'));print('$VAR=$v run $i:',d['value'],d['ms_per_step'])"
  done
done
