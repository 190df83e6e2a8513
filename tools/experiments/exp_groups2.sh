#!/bin/bash
# pop-group pass 2 on short rows: the streaming group tally (default) against the workgroup-per-row kernels (flag 0x20)
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_groups2_$N.txt; : > $OUT
for g in ${GROUPS_LIST:-1 2 3 4 5 6 7}; do
 for fl in 0 32; do
   timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --groups $g --flags $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('groups %-2s flags %-3s sites/s %.4g pass1 %.4f ms | pass2 %.4f ms' % ('$g', '$fl', d['value'], r['pass1_avg_ms'], r['pass2_avg_launch_ms']))" >> $OUT
 done
done
cat $OUT
