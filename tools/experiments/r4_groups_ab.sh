#!/bin/bash
# pop-group runs, library variants side by side (BASEVAR_AMD_LIB): tools/experiments/r4_groups_ab.sh <out-dir> <rounds> <variant> ...   ("base" = the default build)
O=$1; R=$2; shift 2
mkdir -p $O
for r in $(seq 1 $R); do
  for G in ${GROUPS_LIST:-2 8 16 32}; do
    for v in "$@"; do
      unset BASEVAR_AMD_LIB
      [ $v != base ] && export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_$v.so
      timeout 200 python bench.py --no-cpu-baseline --samples ${SAMPLES:-10000} --batch-sites ${SITES:-100000} --groups $G $BENCH_EXTRA 2>> $O/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('groups %2d %-6s' % ($G, '$v'), round(d['value'] / 1e6, 2), 'M sites/s  step', round(d['ms_per_step'], 4), 'pass1', round(r['pass1_avg_ms'], 4), 'pass2', round(r['pass2_avg_launch_ms'], 4))" | tee -a $O/groups.txt
    done
  done
done
