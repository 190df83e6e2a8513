#!/bin/bash
# attribution of the short-row streaming kernel: tools/exp_stream_abl.sh [samples] [batch]
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_stream_abl_$N.txt; : > $OUT
for v in "" _P1S_NOADD _P1S_NOTALLY; do
 for fl in 0 $((0x2000)); do
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --flags $((fl|1)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s flags %-6s stream %.4f ms frac %.3f' % ('full$v', '$fl', r['avg_launch_ms'], r['frac']))" >> $OUT
 done
done
cat $OUT
