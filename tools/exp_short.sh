#!/bin/bash
# A/B of the short-row pass-1 kernels on the GPU box: tools/exp_short.sh [samples] -> gpurun_out/exp_short.txt
N=${1:-10000}
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_short_$N.txt; : > $OUT
run() {  # label, batch, flags
  timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $2 --flags $3 2>gpurun_out/exp_err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-28s B=%-7d sites/s %.4g  stream %.4f ms frac %.3f | pass1 %.4f ms frac %.3f | pass2 %.4f ms  nvar %d' % ('$1', $2, d['value'], r['avg_launch_ms'], r['frac'], r['pass1_avg_ms'], r['pass1_frac'], r['pass2_avg_launch_ms'], d['config']['variant_sites_last_batch']))
except Exception as e:
    print('$1 B=$2 FAILED', e)
" >> $OUT
  [ -s gpurun_out/exp_err.txt ] && grep -v amdgpu.ids gpurun_out/exp_err.txt | tail -3 >> $OUT
}
for B in 100000 524288; do
  run "old fused (1 kernel)" $B $((0x900))
  run "2-kernel K4 12w (default)" $B 0
  run "2-kernel K3 16w" $B $((0x1000))
  run "2-kernel K6 8w" $B $((0x2000))
  run "2-kernel K8 8w" $B $((0x3000))
  run "2-kernel K4 12w (2-wave WG)" $B $((0x4000))
  run "2-kernel K4 8w" $B $((0x5000))
done
cat $OUT
