#!/bin/bash
# SQ counters of the pop-group kernels: tools/experiments/r4_groups_sq.sh <out-dir> <groups>
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/$1; G=$2; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  rm -rf /tmp/gsq_$i
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/gsq_$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --samples ${SAMPLES:-10000} --batch-sites ${SITES:-100000} --groups $G > /dev/null 2>> $O/err.log )
  python3 - $(find /tmp/gsq_$i -name '*counter_collection.csv' | head -1) >> $O/sq_g$G.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "")
    if k.startswith("bv_p2g") or k.startswith("bv_pass2_kernel"):
        a = acc[(k[:40], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-42s %-22s %14.4g per launch (%d dispatch rows)" % (k, c, v / max(1, n) , n))
PY
  i=$((i+1))
done
cat $O/sq_g$G.txt
