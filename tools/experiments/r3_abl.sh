#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
OUT=gpurun_out/r3_abl.txt; : > $OUT
run() { # lib args...
  lib=$1; shift
  BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib timeout 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>>gpurun_out/r3_abl.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s %-44s sites/s %.4g ms/step %.4f stream %.4f | solve %.4f | pass2 %.4f | p1frac %.3f whole %.3f' % ('$lib', '$*', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['pass1_avg_ms']-r['avg_launch_ms'], r['pass2_avg_launch_ms'], r['pass1_frac'], r['whole_path_frac']))" >> $OUT
}
for fl in 0 2 4 6; do
  run libbasevar_amd.so --samples 10000 --batch-sites 100000 --flags $(( (1 << 24) | fl ))
done
cat $OUT
export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"; do
  D=gpurun_out/r3_sq_$(echo $set | cut -d' ' -f1)
  rm -rf $D
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $PWD/$D -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --samples 10000 --batch-sites 100000 --flags $((1 << 24)) > /dev/null 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0]
    if "solve16" not in k and "p1s_stream" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k[:50], {c: "%.4g" % (v/ n[(k,c)]) for c,v in acc[k].items()})
PY
  rm -rf $D
done
