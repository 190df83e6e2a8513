#!/usr/bin/env python3
"""Turns the rocprofv3 output of one round (gpurun_out/prof_<tag>_{stats,fetch,write}) into the
committed summaries under profiles/: per-kernel average durations (kernel-trace --stats) and
HBM traffic per launch from the PMC passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md
(section HBM) prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced
streams -> x2; FETCH_SIZE / WRITE_SIZE are in KiB.

    python tools/summarize_profiles.py <tag> <batch_sites> <samples>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(tag, kind, pattern):
    """newest matching file only: gpurun merges every call's output into the same local directory"""
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, kind), "**", pattern), recursive=True)
    return sorted(hits, key=os.path.getmtime)[-1:]


def main():
    tag, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = {"tag": tag, "batch_sites": B, "samples": N, "kernels": {}}
    lines = ["# rocprofv3 summary, %s (batch %d sites x %d samples)" % (tag, B, N), "",
             "Commands (on the MI355X box, from /tmp with TMPDIR=/tmp):", "",
             "    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_%s_stats -- python3 bench.py --no-cpu-baseline" % tag,
             "    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_%s_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1" % tag,
             "    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_%s_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1" % tag,
             ""]
    for fn in find(tag, "stats", "*kernel_stats.csv"):
        lines += ["## kernel-trace --stats (%s)" % os.path.relpath(fn, ROOT), "", "```"]
        rows = list(csv.DictReader(open(fn)))
        for r in rows:
            name = r["Name"].split("(")[0].replace("void ", "")
            lines.append("%-40s calls %4s  avg %12.1f ns  total %14s ns  %6s %%" % (
                name[:40], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"], r["Percentage"]))
            out["kernels"].setdefault(name, {})["avg_ns"] = float(r["AverageNs"])
            out["kernels"][name]["calls"] = int(r["Calls"])
        lines += ["```", ""]
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        acc = collections.defaultdict(float)
        cnt = collections.Counter()
        for fn in find(tag, kind, "*counter_collection.csv"):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] != counter:
                    continue
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name] += float(r["Counter_Value"])
                cnt[name] += 1
        for name in acc:
            out["kernels"].setdefault(name, {})[counter + "_KiB_per_launch"] = acc[name] / cnt[name]
    lines += ["## HBM traffic per launch (PMC, separate passes)", "",
              "| kernel | FETCH_SIZE KiB (raw) | read bytes (x2 gfx950 correction) | WRITE_SIZE KiB | write bytes | algorithmic bytes | traffic / algorithmic |",
              "|---|---|---|---|---|---|---|"]
    traffic = {}
    for name, k in out["kernels"].items():
        if "FETCH_SIZE_KiB_per_launch" not in k:
            continue
        rd = 2.0 * k["FETCH_SIZE_KiB_per_launch"] * 1024.0
        wr = k.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024.0
        algo = None
        if name.startswith("bv_pass1"):
            algo = 2.0 * B * N
            traffic["pass1_%dx%d" % (B, N)] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                                               "algorithmic_bytes": algo}
        k["hbm_read_bytes"] = rd
        k["hbm_write_bytes"] = wr
        lines.append("| %s | %.0f | %.4g | %.0f | %.4g | %s | %s |" % (
            name[:40], k["FETCH_SIZE_KiB_per_launch"], rd, k.get("WRITE_SIZE_KiB_per_launch", 0.0), wr,
            ("%.4g" % algo) if algo else "-", ("%.3f" % ((rd + wr) / algo)) if algo else "-"))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.json" % tag), "w"), indent=1)
    if traffic:
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        cur = json.load(open(tf)) if os.path.exists(tf) else {}
        cur.update(traffic)
        json.dump(cur, open(tf, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
