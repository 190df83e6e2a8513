#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_exp4.txt; : > $OUT
run() { # lib args...
  lib=$1; shift
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>>gpurun_out/r3_exp4.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-50s sites/s %.4g ms/step %.4f stream %.4f | p1 %.4f | pass2 %.4f | p1frac %.3f whole %.3f' % ('$lib', '$*', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['pass1_avg_ms'], r['pass2_avg_launch_ms'], r['pass1_frac'], r['whole_path_frac']))" >> $OUT
}
for lib in libbasevar_amd.so; do
for sp in 1 2 3 4; do
  run $lib --samples 10000 --batch-sites 100000 --flags $((sp << 24))
done
  run $lib --samples 10000 --batch-sites 524288 --flags $((2 << 24))
  run $lib --samples 10000 --batch-sites 524288 --flags $((4 << 24))
done
cat $OUT
