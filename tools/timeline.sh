#!/bin/bash
# Per-dispatch timeline of a few default-schedule steps: rocprofv3 --kernel-trace, the raw trace CSV reduced by tools/timeline.py
# (GPU-busy union, idle gaps, overlap) -> gpurun_out/timeline_<tag>.txt.   tools/timeline.sh <tag> [bench.py args]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary --no-roofline --steps 12 --warmup 5 > "$OUT/bench.json" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
cd "$ROOT"
python3 tools/timeline.py "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > gpurun_out/timeline_$TAG.txt
rm -rf "$OUT/trace"
tail -40 gpurun_out/timeline_$TAG.txt
