#!/bin/bash
# round 5: where the fused short-row kernel's HBM writes come from.  WRITE_SIZE / FETCH_SIZE and the average duration of
# bv_p1s_fused_kernel at several launch sizes, for library variants side by side (BASEVAR_AMD_LIB).
#   tools/experiments/r5_write_sweep.sh <out-dir> <variant> [<variant> ...]     ("base" = the default build)
# Separate rocprofv3 passes per counter (no tracing beside --pmc), the program itself after `--`.
O=$1; shift
mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
for v in "$@"; do
  unset BASEVAR_AMD_LIB
  [ "$v" != base ] && export BASEVAR_AMD_LIB=$ROOT/basevar_amd/lib/libbasevar_amd_$v.so
  for S in ${SWEEP_SITES:-2048 8192 32768 100000}; do
    for ctr in WRITE_SIZE FETCH_SIZE; do
      D=$O/$v.$S.$ctr; rm -rf $D
      timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --samples 10000 --batch-sites $S --steps 6 --warmup 2 $BENCH_EXTRA > /dev/null 2>> $O/err.log
    done
    D=$O/$v.$S.stats; rm -rf $D
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --samples 10000 --batch-sites $S --steps 6 --warmup 2 $BENCH_EXTRA > /dev/null 2>> $O/err.log
    python3 - $O $v $S <<'PY' | tee -a $O/sweep.txt
import csv, glob, sys, collections
O, v, S = sys.argv[1:4]
def ctr(name):
    acc = collections.defaultdict(list)
    for fn in glob.glob("%s/%s.%s.%s/**/*counter_collection.csv" % (O, v, S, name), recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
w, f = ctr("WRITE_SIZE"), ctr("FETCH_SIZE")
dur = {}
for fn in glob.glob("%s/%s.%s.stats/**/*kernel_stats.csv" % (O, v, S), recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
for k in sorted(w):
    if not k.startswith("bv_") or "synth" in k: continue
    ws = sorted(w[k]); fs = sorted(f.get(k, [0.0]))
    med = lambda a: a[len(a) // 2]
    print("%-8s sites %7s  %-34s calls %3d avg %8.1f us  WRITE_SIZE median %9.0f KiB (min %9.0f max %9.0f)  FETCH_SIZE median %10.0f KiB" % (
        v, S, k[:34], dur.get(k, (0, 0))[0], dur.get(k, (0, 0))[1], med(ws), ws[0], ws[-1], med(fs)))
PY
    find $O/$v.$S.* -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
  done
done
