#!/bin/bash
# ablation of the short-row solve kernel: tools/exp_short_abl.sh [samples] [batch]
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_short_abl_$N.txt; : > $OUT
run() {
  timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --flags $2 $3 2>gpurun_out/exp_err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-34s stream %.4f ms frac %.3f | solve %.4f ms | pass2 %.4f ms  nvar %d' % ('$1', r['avg_launch_ms'], r['frac'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms'], d['config']['variant_sites_last_batch']))
except Exception as e:
    print('$1 FAILED', e)
" >> $OUT
  grep -v amdgpu.ids gpurun_out/exp_err.txt | tail -3 >> $OUT
}
run "full" $((0x2000))
run "tally only (no candidates)" $((0x2001))
run "skip fisher" $((0x2002))
run "skip lrt" $((0x2004))
run "skip lrt+fisher" $((0x2006))
run "full, coverage 0.02" $((0x2000)) "--coverage 0.02"
run "full, coverage 0.3" $((0x2000)) "--coverage 0.3"
cat $OUT
