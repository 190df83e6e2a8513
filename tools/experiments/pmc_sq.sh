#!/bin/bash
# SQ instruction-mix counters of the pass-1 kernel for one library: tools/experiments/pmc_sq.sh <tag> <lib.so> "<bench args>"
TAG=$1; LIB=$2; ARGS=$3
cd "$(dirname "$0")/../.."; ROOT=$PWD
export TMPDIR=/tmp BASEVAR_AMD_LIB=$ROOT/basevar_amd/lib/$LIB
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
  d=$ROOT/gpurun_out/pmcsq_${TAG}_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $ARGS > /dev/null 2>&1
done
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
acc=collections.defaultdict(float); cnt=collections.Counter()
for fn in glob.glob("gpurun_out/pmcsq_%s_*/**/*counter_collection.csv"%tag, recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "pass1" not in k: continue
        acc[(k,r["Counter_Name"])]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for (k,c),v in sorted(acc.items()):
    print("%s %-28s %-24s %.4g per launch"%(tag,k[:28],c,v/cnt[(k,c)]))
PY
