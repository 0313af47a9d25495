#!/bin/bash
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for t in old new; do
  if [ $t = old ]; then export SVAE_TILE_TABLE=$ROOT/tools/tmp/tuned_old.json; else unset SVAE_TILE_TABLE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/p_$t -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --no-roofline --serial-streams --steps 20 --warmup 5 > $ROOT/gpurun_out/p_$t.json 2> $ROOT/gpurun_out/p_$t.err || exit 1
  cp $(find $ROOT/gpurun_out/p_$t -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/kstats_$t.csv
  rm -rf $ROOT/gpurun_out/p_$t
done
