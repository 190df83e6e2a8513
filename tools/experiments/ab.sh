#!/bin/bash
# usage: tools_ab.sh "<bench args>" libA libB ...   (interleaved rounds in one gpurun call)
ARGS="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 60 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), 'p1', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'p2', round(d['roofline']['pass2_avg_launch_ms'],4))"
  done
done
