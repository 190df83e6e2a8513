#!/bin/bash
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_solve16_$N.txt; : > $OUT
for v in "" _OCC2; do
 for fl in 0 16; do
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --flags $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s flags %-3s sites/s %.4g stream %.4f ms | solve %.4f ms | pass2 %.4f' % ('lib$v', '$fl', d['value'], r['avg_launch_ms'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms']))" >> $OUT
 done
done
cat $OUT
