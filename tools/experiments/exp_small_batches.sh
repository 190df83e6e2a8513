#!/bin/bash
# where does the fixed per-launch cost of the long-row pass 1 come from?  tools/exp_small_batches.sh
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_small_batches.txt; : > $OUT
run() {
  timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 4 --batch-sites $2 --flags $3 $4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s B=%-7d p1 %.4f ms frac %.3f | pass2 %.4f ms | sites/s %.4g' % ('$1', $2, r['avg_launch_ms'], r['frac'], r['pass2_avg_launch_ms'], d['value']))" >> $OUT
}
for B in 8192 32768; do
  run "full" $B 0
  run "tally only" $B 1
  run "skip fisher" $B 2
  run "skip lrt" $B 4
  run "full, 2 streams" $B 0 "--streams 2"
done
cat $OUT
