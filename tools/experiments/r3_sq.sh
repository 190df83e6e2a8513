#!/bin/bash
# SQ instruction / wait / I-cache counters per kernel: tools/r3_sq.sh <bench args...>
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_ANY"; do
  D=gpurun_out/r3_sqd; rm -rf $D
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
    print("%-36s" % k[:36], " ".join("%s=%.4g" % (c.replace("SQ_","").replace("SQC_",""), v/ n[(k,c)]) for c,v in acc[k].items()))
PY
  rm -rf $D
done
