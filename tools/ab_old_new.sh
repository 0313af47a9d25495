#!/bin/bash
# Same-box A/B of the working tree against an older revision at three batch sizes (configs[1] at B = 32 / 1024, configs[2] at 4096).
# Prepare the old tree next to the new one before sending both to the GPU box (.ab_old/ is git-ignored, its .so travels with gpurun):
#   rm -rf .ab_old && mkdir .ab_old && git archive <rev> | tar -x -C .ab_old && (cd .ab_old && python -c "from scrubvae_amd import build; build.build()")
#   gpurun -- 'bash tools/ab_old_new.sh'
set -o pipefail
run() { # dir tag args...
  d=$1; tag=$2; shift 2
  (cd $d && python bench.py "$@" --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null) > gpurun_out/ab_$tag.json
  python -c "import json;d=json.load(open('gpurun_out/ab_$tag.json'));print('$tag',d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'])"
}
for i in 1 2; do
  run .ab_old old_b32_$i --workload config1 --batch 32 --steps 100
  run . new_b32_$i --workload config1 --batch 32 --steps 100
  run .ab_old old_b1024_$i --workload config1 --steps 40
  run . new_b1024_$i --workload config1 --steps 40
  run .ab_old old_b4096_$i
  run . new_b4096_$i
done
