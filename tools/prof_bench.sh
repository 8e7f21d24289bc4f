#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py workload -> gpurun_out/<tag>_kernel_stats.csv (run on the GPU box from the repo root)
# usage: tools/prof_bench.sh <tag> [bench.py args...]
TAG=$1; shift
OUT=$PWD/gpurun_out
mkdir -p "$OUT/$TAG"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$TAG" -o prof --output-format csv -- python "$OLDPWD/bench.py" --no-cpu-baseline "$@" > "$OUT/$TAG/bench.log" 2>&1)
F=$(find "$OUT/$TAG" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_kernel_stats.csv"
tail -1 "$OUT/$TAG/bench.log" > "$OUT/${TAG}_bench.json"
find "$OUT/$TAG" -name "*.csv" -size +500k -delete
python - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    nm = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:95]
    print(f"{nm:95s} calls {int(r['Calls']):6d} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}%")
PY
