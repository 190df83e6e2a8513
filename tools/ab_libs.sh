#!/bin/bash
# A/B of library variants (make VARIANT=x DEFS=...): tools/ab_libs.sh "<bench args>" <rounds> <variant> [<variant> ...]
# ("default" = the product library); prints sites/s and the dominant kernel's avg ms per run, interleaved.
ARGS="$1"; ROUNDS="$2"; shift 2
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset BASEVAR_AMD_LIB; else export BASEVAR_AMD_LIB="$PWD/basevar_amd/lib/libbasevar_amd_$v.so"; fi
    python bench.py --no-cpu-baseline --no-configs1 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %.4g sites/s  %s %.4f ms frac %.3f  pass2 %.4f ms' % ('$v', d['value'], r['kernel'], r['avg_launch_ms'], r['frac'], r['pass2_avg_launch_ms']))"
  done
done
