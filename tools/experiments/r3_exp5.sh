#!/bin/bash
# A/B of library builds on configs[1] (unsplit launch): tools/r3_exp5.sh lib1 lib2 ...
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_exp5.txt; : > $OUT
run() { # lib args...
  lib=$1; shift
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>>gpurun_out/r3_exp5.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-44s sites/s %.4g ms/step %.4f stream %.4f | solve %.4f | pass2 %.4f | p1frac %.3f whole %.3f' % ('$lib', '$*', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms'], r['pass1_frac'], r['whole_path_frac']))" >> $OUT
}
for round in 1 2; do
for lib in "$@"; do
  run $lib --samples 10000 --batch-sites 100000 --flags $((1 << 24))
done
done
for lib in "$@"; do
  run $lib --samples 10000 --batch-sites 524288 --flags $((1 << 24))
  run $lib --samples 40000 --batch-sites 65536 --flags $((1 << 24))
  run $lib --samples 3000 --batch-sites 262144 --flags $((1 << 24))
done
cat $OUT
