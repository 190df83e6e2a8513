#!/bin/bash
# A/B of the short-row streaming kernel shapes on the GPU box: tools/exp_short.sh [samples] -> gpurun_out/exp_short_<N>.txt
N=${1:-10000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_short_$N.txt; : > $OUT
run() {  # label, batch, flags
  timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $2 --flags $3 2>gpurun_out/exp_err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-34s B=%-7d sites/s %.4g  stream %.4f ms frac %.3f | pass1 %.4f ms | pass2 %.4f ms' % ('$1', $2, d['value'], r['avg_launch_ms'], r['frac'], r['pass1_avg_ms'], r['pass2_avg_launch_ms']))
except Exception as e:
    print('$1 B=$2 FAILED', e)
" >> $OUT
  grep -v amdgpu.ids gpurun_out/exp_err.txt | tail -3 >> $OUT
}
for B in 100000 524288; do
  run "U1 K4 8w (default)" $B 0
  run "U1 K6 8w" $B $((0x2000))
  run "U2 K3 8w" $B $((0x6000))
  run "U2 K4 8w" $B $((0x7000))
  run "U2 K2 12w" $B $((0x8000))
  run "U2 K4 8w 2-wave WGs" $B $((0x9000))
done
cat $OUT
