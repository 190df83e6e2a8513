#!/bin/bash
# A/B of two builds on a list of bench configurations: tools/exp_ab.sh <libA suffix> <libB suffix> -- whole-path numbers
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_ab.txt; : > $OUT
while read -r name args; do
  [ -z "$name" ] && continue
  for v in "$1" "$2"; do
    BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s lib%-8s sites/s %.4g | pass1 %.4f ms | pass2 %.4f ms' % ('$name', '$v', d['value'], r['pass1_avg_ms'], r['pass2_avg_launch_ms']))" >> $OUT
  done
done <<'CFG'
n64 --samples 64 --batch-sites 300000 --coverage 0.5
n100 --samples 100 --batch-sites 200000 --coverage 0.3
n1000g5 --samples 1000 --batch-sites 100000 --groups 5
n10kg8 --samples 10000 --batch-sites 100000 --groups 8
n10kg32 --samples 10000 --batch-sites 50000 --groups 32
n10k --samples 10000 --batch-sites 100000
n100k --batch-sites 65536
CFG
cat $OUT
