#!/bin/bash
# ablation of the short-row (N = 10k) pass-1 kernel: whole path, tally only, Fisher off, LRT off, both off
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for f in "" "--tally-only" "--flags 2" "--flags 4" "--flags 6"; do
  echo "== $f" >> gpurun_out/n10k.log
  timeout 300 python bench.py --samples 10000 --batch-sites 131072 --steps 10 --warmup 2 --no-cpu-baseline --no-rank-planes $f 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('sites/s %.3e  p1 %.3f ms  frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))" >> gpurun_out/n10k.log
done
cat gpurun_out/n10k.log
