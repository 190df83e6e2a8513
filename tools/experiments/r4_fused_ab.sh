#!/bin/bash
# interleaved A/B of the fused short-row pass 1 (round 4): library variants selected with BASEVAR_AMD_LIB
#   tools/experiments/r4_fused_ab.sh <out-dir> <rounds> <variant> [<variant> ...]     ("base" = the default build, "old" = flag bits 12-15 = 9)
O=$1; R=$2; shift 2
mkdir -p $O
for r in $(seq 1 $R); do
  for v in "$@"; do
    unset BASEVAR_AMD_LIB; F=""
    case $v in
      base) ;;
      old) F="--flags 36864" ;;      # two-kernel pass 1, pass 2 a launch of its own (round 3)
      p2sep) F="--flags 40960" ;;    # fused pass 1, pass 2 a launch of its own
      *) export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_$v.so ;;
    esac
    timeout 120 python bench.py --no-cpu-baseline --samples 10000 --batch-sites 100000 $F $BENCH_EXTRA 2>> $O/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$v', round(d['value'] / 1e6, 1), 'M sites/s  step', round(d['ms_per_step'], 4), 'pass1', round(r['pass1_avg_ms'], 4), 'pass2', round(r['pass2_avg_launch_ms'], 4))" | tee -a $O/ab.txt
  done
done
