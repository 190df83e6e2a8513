#!/bin/bash
# Instruction attribution of pass 1 by ablation: builds must exist (make VARIANT=x DEFS=-DBV_ABL_x).
# usage: tools/experiments/ablate.sh "<bench args>" lib[:flags] ...
ARGS="$1"; shift
cd "$(dirname "$0")/../.."; ROOT=$PWD
export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; fl=0; [[ "$spec" == *:* ]] && fl=${spec##*:}
  d=$ROOT/gpurun_out/abl_${lib}_$fl; rm -rf $d
  BASEVAR_AMD_LIB=$ROOT/basevar_amd/lib/$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --flags $fl $ARGS > /dev/null 2>&1
  t=$(BASEVAR_AMD_LIB=$ROOT/basevar_amd/lib/$lib timeout 120 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --flags $fl $ARGS 2>/dev/null | python3 -c "import json,sys; print('%.4f'%json.loads(sys.stdin.read())['roofline']['avg_launch_ms'])")
  python3 - "$d" "$lib:$fl" "$t" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "pass1" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("%-36s p1 %s ms  "%(sys.argv[2],sys.argv[3])+"  ".join("%s %.4g"%(c.replace("SQ_",""),acc[c]/cnt[c]) for c in sorted(acc)))
PY
done
