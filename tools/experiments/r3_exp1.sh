#!/bin/bash
# round 3, first look: configs[1] baseline and the 16-lane solver at 2 waves per SIMD
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_exp1.txt; : > $OUT
run() { # lib args...
  lib=$1; shift
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-44s sites/s %.4g ms/step %.4f stream %.4f | solve %.4f | pass2 %.4f | p1frac %.3f whole %.3f' % ('$lib', '$*', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms'], r['pass1_frac'], r['whole_path_frac']))" >> $OUT
}
for round in 1 2; do
for lib in libbasevar_amd.so libbasevar_amd_occ2.so; do
  run $lib --samples 10000 --batch-sites 100000
  run $lib --samples 10000 --batch-sites 100000 --streams 2
done
done
run libbasevar_amd.so --samples 10000 --batch-sites 524288
run libbasevar_amd.so --samples 100000 --batch-sites 131072
cat $OUT
