#!/bin/bash
# kernel-trace stats of pop-group runs at several group counts: tools/experiments/r4_groups_trace.sh <out-dir>
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/$1; mkdir -p $O
for G in ${GROUPS_LIST:-8 16 32}; do
  rm -rf /tmp/gt_$G
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gt_$G -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --samples ${SAMPLES:-10000} --batch-sites ${SITES:-100000} --groups $G > /dev/null 2>> $O/err.log )
  echo "== groups $G" >> $O/trace.txt
  python3 - $(find /tmp/gt_$G -name '*kernel_stats.csv' | head -1) >> $O/trace.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].replace("void ", "").startswith("bv_") and "synth" not in r["Name"]:
        print("%-60s calls %3s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cat $O/trace.txt
