#!/bin/bash
# kernel timeline (start / end of every dispatch) of one short bench command: tools/trace_timeline.sh <tag> <bench args...>
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
TAG=$1; shift
D=gpurun_out/tl_$TAG; rm -rf $D
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $PWD/$D -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > /dev/null 2>&1
F=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - "$F" > gpurun_out/tl_$TAG.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "synth" not in r["Kernel_Name"] and r["Kernel_Name"].startswith(("bv_","void bv_"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%9.1f %9.1f  %7.1f us  q%-3s %s  grid %s wg %s lds %s" % (s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"].split("(")[0][:44],r.get("Grid_Size_X", r.get("Grid_Size","?")),r.get("Workgroup_Size_X", r.get("Workgroup_Size","?")),r.get("LDS_Block_Size","?")))
PY
rm -rf $D
cat gpurun_out/tl_$TAG.txt
