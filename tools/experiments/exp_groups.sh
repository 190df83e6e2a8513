#!/bin/bash
# pop-group pass 2: the 16-lane group solver (default) against the one-wave-per-group solver (flag 16), two occupancies
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_groups_$N.txt; : > $OUT
for v in "" _P2GOCC3; do
 [ -f basevar_amd/lib/libbasevar_amd$v.so ] || continue
 for g in 1 2 8; do
  for fl in 0 16; do
   for extra in "" "--no-rank-planes"; do
   BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --groups $g --flags $fl $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s groups %-2s flags %-3s %-16s sites/s %.4g pass1 %.4f ms | pass2 %.4f ms' % ('lib$v', '$g', '$fl', '$extra', d['value'], r['pass1_avg_ms'], r['pass2_avg_launch_ms']))" >> $OUT
   done
  done
 done
done
cat $OUT
