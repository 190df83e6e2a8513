#!/bin/bash
# attribution of the fused kernel's streaming phase (round 4): per-workgroup stamps of builds with parts removed (results wrong)
#   tools/experiments/r4_abl.sh <out-dir> <variant> ...     (libraries built with make VARIANT=x DEFS="... -DBV_TEAM_DEBUG")
O=$1; shift; mkdir -p $O
for v in "$@"; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/libbasevar_amd_$v.so
  echo "== $v" | tee -a $O/abl.txt
  timeout 120 python bench.py --steps 3 --warmup 1 --samples 10000 --batch-sites 100000 --no-cpu-baseline $BENCH_EXTRA 2>&1 | grep "fused debug" | tail -8 | cut -c1-150 | tee -a $O/abl.txt
done
