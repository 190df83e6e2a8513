#!/bin/bash
# small batches at 10 k samples: one launch set per batch against K batches chained (bv_engine_submit_many)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_chain_short.txt; : > $OUT
run() {
  timeout 300 python3 bench.py --no-cpu-baseline --samples ${4:-10000} --steps $3 --warmup 3 --batch-sites $1 --chain $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=%-6s B=%-6d chain %-2d  stream %.4f ms per launch, frac %.3f | pass 1 %.4f | pass 2 %.4f ms | sites/s %.4g' % ('${4:-10000}', $1, $2, r['avg_launch_ms'], r['frac'], r['pass1_avg_ms'], r['pass2_avg_launch_ms'], d['value']))" >> $OUT
}
run 8192 1 40; run 8192 4 10; run 8192 16 8
run 32768 1 20; run 32768 4 8
run 100000 1 10
run 4096 1 40 40000; run 4096 16 8 40000
cat $OUT
