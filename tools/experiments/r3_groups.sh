#!/bin/bash
# pop-group configurations, A/B over library builds: tools/r3_groups.sh lib1 lib2 ...
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_groups.txt; : > $OUT
run() { # lib args...
  lib=$1; shift
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>>gpurun_out/r3_groups.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-52s sites/s %.4g ms/step %.4f p1 %.4f | pass2 %.4f' % ('$lib', '$*', d['value'], d['ms_per_step'], r['pass1_avg_ms'], r['pass2_avg_launch_ms']))" >> $OUT
}
for lib in "$@"; do
  run $lib --samples 10000 --batch-sites 100000 --groups 1
  run $lib --samples 10000 --batch-sites 100000 --groups 2
  run $lib --samples 10000 --batch-sites 100000 --groups 8
  run $lib --samples 10000 --batch-sites 50000 --groups 32
  run $lib --samples 100000 --batch-sites 65536 --groups 2
  run $lib --samples 1000 --batch-sites 262144 --groups 5
done
cat $OUT
