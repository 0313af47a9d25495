#!/bin/bash
# Profiles of one bench.py workload on the GPU box (run from the repo root):
#   tools/profile_bench.sh <tag> [bench.py args]      e.g.  tools/profile_bench.sh r02_config2_b4096
# 1. rocprofv3 --kernel-trace --stats of the default (3-stream) schedule and of --serial-streams (a launch's duration is the
#    kernel's own time only when the side streams are serialised);
# 2. separate --pmc passes (never combined with trace domains): matrix-core / issue counters, FETCH_SIZE, WRITE_SIZE;
# 3. summaries into profiles/<tag>_*.{csv,json} (tools/summarize_pmc.py, tools/summarize_pmc_util.py).
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" "$@" $COMMON --steps 100 --warmup 5 > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -o t -- python3 "$ROOT/bench.py" "$@" $COMMON --serial-streams --steps 100 --warmup 5 > "$OUT/trace_serial.json" 2> "$OUT/trace_serial.err" || { tail -5 "$OUT/trace_serial.err"; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_util" -o t -- python3 "$ROOT/bench.py" "$@" $COMMON --serial-streams --steps 3 --warmup 2 > "$OUT/pmc_util.json" 2> "$OUT/pmc_util.err" || { tail -5 "$OUT/pmc_util.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- python3 "$ROOT/bench.py" "$@" $COMMON --serial-streams --steps 3 --warmup 2 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- python3 "$ROOT/bench.py" "$@" $COMMON --serial-streams --steps 3 --warmup 2 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
cd "$ROOT"
find "$OUT" -name '*.csv' | head -20
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "profiles/${TAG}_kernel_stats.csv"
cp "$(find "$OUT/trace_serial" -name '*kernel_stats.csv' | head -1)" "profiles/${TAG}_serial_streams_kernel_stats.csv"
cp "$OUT/trace.json" "profiles/${TAG}_bench_under_rocprof.json"
python3 tools/summarize_pmc_util.py "$OUT/pmc_util" "$TAG" > "$OUT/pmc_util_summary.txt"
python3 tools/summarize_pmc.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$TAG" > "$OUT/pmc_traffic_summary.txt"
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/
head -12 "$OUT/pmc_util_summary.txt"
# the raw traces are large (gpurun copies back at most 64 MiB): keep the summaries and the small logs only
rm -rf "$OUT/trace" "$OUT/trace_serial" "$OUT/pmc_util" "$OUT/pmc_fetch" "$OUT/pmc_write"
