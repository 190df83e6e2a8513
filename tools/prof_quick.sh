#!/bin/bash
# per-kernel averages of one bench command: tools/prof_quick.sh <tag> <bench args...>
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
TAG=$1; shift
D=gpurun_out/pq_$TAG; rm -rf $D
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$D -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > /dev/null 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
find $D -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
echo "== $TAG: $*"
python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("bv_","void bv_")) and "synth" not in r["Name"]:
        print("  %-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
