#!/bin/bash
# tools/experiments/pmc_list.sh <lib.so> "<bench args>" "<counter set 1>" "<counter set 2>" ...   (one rocprofv3 pass per set)
LIB=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/../.."; ROOT=$PWD
export TMPDIR=/tmp BASEVAR_AMD_LIB=$ROOT/basevar_amd/lib/$LIB
i=0
for set in "$@"; do
  d=$ROOT/gpurun_out/pmcl_$i; rm -rf $d; i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $ARGS > /dev/null 2>$d.err || tail -3 $d.err
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for fn in glob.glob("gpurun_out/pmcl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "pass1" not in k: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for c in sorted(acc): print("%-28s %.4g"%(c,acc[c]/cnt[c]))
PY
