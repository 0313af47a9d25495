#!/bin/bash
# same-box A/B of the fused upsample: tools/ab_fuse.sh [bench.py args]  -> alternating runs with SVAE_FUSE_UPSAMPLE=0 / force
for i in 1 2; do
  for v in 0 force; do
    SVAE_FUSE_UPSAMPLE=$v python bench.py "$@" --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null > gpurun_out/ab_env.json
    python -c "import json;d=json.load(open('gpurun_out/ab_env.json'));print('FUSE=$v run $i:',d['value'],d['ms_per_step'])"
  done
done
