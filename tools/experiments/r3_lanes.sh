#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_lanes.txt; : > $OUT
run() {
  timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>>gpurun_out/r3_lanes.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-60s sites/s %.4g ms/step %.4f stream %.4f | p1 %.4f | pass2 %.4f | p1frac %.3f whole %.3f' % ('$*', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['pass1_avg_ms'], r['pass2_avg_launch_ms'], r['pass1_frac'], r['whole_path_frac']))" >> $OUT
}
for l in 1 2; do
  run --samples 10000 --batch-sites 100000 --lanes $l
  run --samples 10000 --batch-sites 524288 --lanes $l
  run --samples 10000 --batch-sites 8192 --lanes $l --steps 60
  run --samples 40000 --batch-sites 65536 --lanes $l
  run --samples 100000 --batch-sites 131072 --lanes $l
  run --samples 100000 --batch-sites 8192 --lanes $l --steps 40
  run --samples 10000 --batch-sites 100000 --groups 2 --lanes $l
done
run --samples 10000 --batch-sites 100000 --streams 2
cat $OUT
