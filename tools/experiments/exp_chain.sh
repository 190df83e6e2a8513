#!/bin/bash
# small batches at 100 k samples: one launch per batch against K batches chained into one launch (bv_engine_submit_many)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/exp_chain.txt; : > $OUT
run() {
  timeout 300 python3 bench.py --no-cpu-baseline --steps $3 --warmup 3 --batch-sites $1 --chain $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('B=%-6d chain %-2d  pass 1 %.4f ms per launch, frac %.3f | pass 2 %.4f ms | whole path frac %.3f | sites/s %.4g' % ($1, $2, r['avg_launch_ms'], r['frac'], r['pass2_avg_launch_ms'], r['whole_path_frac'], d['value']))" >> $OUT
}
run 8192 1 40; run 8192 2 20; run 8192 4 10; run 8192 8 10; run 8192 16 5
run 32768 1 20; run 32768 4 5
run 131072 1 8
cat $OUT
