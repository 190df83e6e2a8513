#!/bin/bash
# dense-row measurements (coverage 0.3 / 1.0 beside BASELINE's 0.08), one box: tools/r6_dense.sh <tag>
TAG=${1:-a}
O=gpurun_out/r6_dense_$TAG; mkdir -p $O
for cov in 0.08 0.3 1.0; do
  python bench.py --no-cpu-baseline --no-configs1 --coverage $cov --batch-sites 65536 --steps 10 > $O/n100k_$cov.json 2>> $O/err
  python bench.py --no-cpu-baseline --no-configs1 --coverage $cov --samples 10000 --batch-sites 100000 --steps 30 > $O/n10k_$cov.json 2>> $O/err
done
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print("%-22s %8.2f M sites/s  step %.3f ms  p1 %.3f  p2 %.3f  p1frac %.3f whole %.3f  nvar %d" % (os.path.basename(f), d["value"] / 1e6, d["ms_per_step"], r["pass1_avg_ms"], r["pass2_avg_launch_ms"], r["pass1_frac"], r["whole_path_frac"], d["config"]["variant_sites_last_batch"]))
    except Exception as ex:
        print(f, "failed", ex)
PY
