#!/bin/bash
# I-cache behaviour per kernel: tools/r3_icache.sh <bench args...>
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  D=gpurun_out/r3_ic; rm -rf $D
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $PWD/$D -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if not k.startswith("bv_") or "synth" in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in acc:
    print("%-40s" % k[:40], {c: "%.4g" % (v/ n[(k,c)]) for c,v in acc[k].items()})
PY
  rm -rf $D
done
