#!/bin/bash
# rocprofv3 kernel trace of one bench.py workload, cut to the TIMED (graph-replayed) steps by marker kernels (bench.py --profile-markers
# + tools/prof_window.py): calls per step are integers and the per-kernel "us inside the step" can be recomputed from the CSV.
# -> gpurun_out/<tag>_window_kernels.csv, <tag>_window_summary.txt, <tag>_bench.json (run on the GPU box from the repo root)
# usage: tools/prof_bench.sh <tag> [bench.py args...]       (steps: the --steps given, default 20)
TAG=$1; shift
OUT=$PWD/gpurun_out
mkdir -p "$OUT/$TAG"
export TMPDIR=/tmp
STEPS=20
for ((i = 1; i <= $#; i++)); do [ "${!i}" = "--steps" ] && j=$((i + 1)) && STEPS=${!j}; done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/$TAG" -o prof --output-format csv -- python "$OLDPWD/bench.py" --no-cpu-baseline --profile-markers "$@" > "$OUT/$TAG/bench.log" 2>&1)
tail -1 "$OUT/$TAG/bench.log" > "$OUT/${TAG}_bench.json"
python tools/prof_window.py "$OUT/$TAG" "$STEPS" "$OUT/${TAG}_window"
F=$(find "$OUT/$TAG" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/${TAG}_kernel_stats.csv"  # (whole process: build, warm-up, roofline replays included)
find "$OUT/$TAG" -name "*.csv" -size +500k -delete
