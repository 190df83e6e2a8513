#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_p2nt.txt; : > $OUT
for g in 6 8 12 16 24 32; do
 for v in "" _BIG6 _HUGE12; do
   BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 300 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --batch-sites 32768 --groups $g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('groups %-2s lib%-8s sites/s %.4g pass2 %.4f ms' % ('$g', '$v', d['value'], r['pass2_avg_launch_ms']))" >> $OUT
 done
done
cat $OUT
