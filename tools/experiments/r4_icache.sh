cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ic; ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d /tmp/ic -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --samples 10000 --batch-sites 100000 > /dev/null 2>/tmp/ic.err )
python3 - $(find /tmp/ic -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "")
    if k.startswith("bv_p1s"):
        a = acc[(k[:40], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-42s %-22s %14.4g per launch (%d rows)" % (k, c, v / max(1, n), n))
PY
tail -3 /tmp/ic.err
