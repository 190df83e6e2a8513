#!/bin/bash
# new (fused) against the round-3 kernels (flag bits 12-15 = 9) over batch shapes: tools/experiments/r4_shapes.sh <out-dir>
O=$1; mkdir -p $O
run() {  # label, bench args
  local l=$1; shift
  for f in "" "--flags 36864"; do
    timeout 200 python bench.py --no-cpu-baseline "$@" $f 2>> $O/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-28s %-14s' % ('$l', 'round-3' if '$f' else 'fused'), round(d['value'] / 1e6, 2), 'M sites/s  step', round(d['ms_per_step'], 4), 'pass1', round(r['pass1_avg_ms'], 4), 'pass2', round(r['pass2_avg_launch_ms'], 4))" | tee -a $O/shapes.txt
  done
}
run "10k x 100k" --samples 10000 --batch-sites 100000
run "10k x 524k" --samples 10000 --batch-sites 524288
run "10k x 8192" --samples 10000 --batch-sites 8192 --steps 30
run "10k x 100k groups2" --samples 10000 --batch-sites 100000 --groups 2
run "10k x 100k noranks" --samples 10000 --batch-sites 100000 --no-rank-planes
run "5k x 200k" --samples 5000 --batch-sites 200000
run "20k x 50k" --samples 20000 --batch-sites 50000
run "40k x 25k" --samples 40000 --batch-sites 25000
run "10k x 100k cov 0.3" --samples 10000 --batch-sites 100000 --coverage 0.3
run "10k x 100k cov 0.01" --samples 10000 --batch-sites 100000 --coverage 0.01
