#!/bin/bash
# attribution builds (make VARIANT=x DEFS=-DBV_ABL_x) of the short-row solve kernel
N=${1:-10000}; B=${2:-100000}
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_short_abl2_$N.txt; : > $OUT
for v in "" _NO_EM _NO_QUAL _NO_BQ _NO_VARFS; do
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd$v.so timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --samples $N --batch-sites $B --flags $((0x2000)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s stream %.4f ms | solve %.4f ms | pass2 %.4f ms  nvar %d' % ('full$v', r['avg_launch_ms'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms'], d['config']['variant_sites_last_batch']))" >> $OUT
done
cat $OUT
