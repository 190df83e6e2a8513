#!/usr/bin/env python3
"""Turns the rocprofv3 output of one round (gpurun_out/prof_<tag>/, written by tools/collect_profiles.sh on the GPU
box) into the committed summaries under profiles/:

    profiles/<tag>_rocprof_summary.md / .json   per configuration and kernel: calls, average duration
                                                (--kernel-trace --stats), HBM bytes per launch (PMC passes),
                                                algorithmic bytes, the two ratios, SQ counters where collected
    profiles/<tag>_bench_<config>.json          the un-profiled bench line of the same command
    profiles/pmc_traffic.json                   kernel|batch x samples -> HBM bytes per launch (bench.py's `traffic`)

HBM traffic is corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950: FETCH_SIZE
counts 64 B per 128-B request on wide coalesced streams -> x2; FETCH_SIZE / WRITE_SIZE are in KiB.

    python tools/summarize_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0  # GB/s


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return sorted(hits, key=os.path.getmtime)[-1:] if hits else []


def grid_of(r):
    return int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)


def counters(dirpat, by_grid=False):
    """average counter value per (kernel, counter) -- or per (kernel, grid size, counter): calls of one kernel with
    different dispatch sizes are different workloads and must not be averaged together"""
    acc, cnt = collections.defaultdict(float), collections.Counter()
    # (the newest file only: gpurun MERGES a call's outputs into gpurun_out/, so a directory collected twice holds both runs --
    # files of an older build averaged into a newer one's counters is how a 40 MB WRITE_SIZE once read 88 MB)
    for fn in newest(dirpat):
        for r in csv.DictReader(open(fn)):
            k = (short(r["Kernel_Name"]), grid_of(r), r["Counter_Name"]) if by_grid else (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


def dispatch_groups(dirpat):
    """{kernel: {grid size: [durations ns]}} from the per-dispatch rows collect_profiles.sh keeps"""
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in newest(dirpat):
        for r in csv.DictReader(open(fn)):
            g[short(r["Kernel_Name"])][grid_of(r)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return g


def algorithmic_bytes(kernel, cfg, nvar):
    B, N = cfg["batch_sites"] * cfg.get("chain", 1), cfg["samples"]
    if cfg.get("tile_job"):
        return None, "-"  # the tile job's finish runs the pass kernels on other rows than the bench batch: nothing attributed
    if kernel.startswith("bv_p1s_fused_kernel<true>"):
        # pass 1 and the variant sites' pass-2 rows in one kernel: SURVEY 8d's 2 B/cell + 3 B/cell of the variant rows
        # (plain rank layout: its traffic re-reads the call byte of a variant row, 4 B/cell there; tagged: 3 B/cell)
        return (2.0 * B + 3.0 * nvar) * N, "2 B/cell x %d sites x %d samples + 3 B/cell x %d variant rows" % (B, N, nvar)
    if kernel.startswith(("bv_pass1_kernel", "bv_pass1_fused_kernel", "bv_p1s_stream_kernel", "bv_p1s_fused_kernel")):
        return 2.0 * B * N, "2 B/cell x %d sites x %d samples" % (B, N)
    per = None
    if kernel.startswith("bv_pass2_dma_kernel<true>"):
        per = 3                      # tagged rank layout: mapq + 2 B rank
    elif kernel.startswith(("bv_pass2_dma_kernel", "bv_pass2_short_kernel")):
        per = 4                      # call + mapq + 2 B rank
    elif kernel.startswith("bv_p2g_stream_kernel"):
        per = 2                      # call + phred (the group plane is the same 10-50 KB for every row: L2)
    elif kernel.startswith("bv_pass2_kernel"):
        targs = [t.strip() for t in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")] if "<" in kernel else []
        rk = len(targs) > 1 and targs[1] == "true"
        gr = len(targs) > 2 and targs[2] == "true"
        tag = len(targs) > 5 and targs[5] == "true"   # BV_SLAB_RPR_TAGGED without pop-groups: the call plane is not read
        per = (0 if tag else 1) + (3 if rk else 0) + (1 if gr else 0)
    if per is not None:
        return float(per) * N * nvar, "%d B/cell x %d variant rows x %d samples" % (per, nvar, N)
    return None, "-"


def tkey(kernel, cfg):
    """pmc_traffic.json key: kernel[<variant>] | sites per launch x samples [| rank layout] [| chainK]; bench.py forms the same
    (bench.py: traffic_key).  The variant and the rank layout are part of the key where the kernel's traffic depends on them: the
    fused short-row kernel that also streams the variant sites' pass-2 rows (<p2rows>, template <true>) against the one that does
    not (<p1only>, <false>: pop-groups >= 8, no rank planes, BV_FLAG_SHORT_ROW_FORM(10)), and for <p2rows> whether those rows
    read the call plane (plain) or not (tagged)."""
    base = kernel.split("<")[0]
    var = ""
    if base == "bv_p1s_fused_kernel":
        var = "<p2rows>" if kernel.startswith("bv_p1s_fused_kernel<true>") else "<p1only>"
    key = "%s%s|%dx%d" % (base, var, cfg["batch_sites"] * cfg.get("chain", 1), cfg["samples"])
    if var == "<p2rows>":
        key += "|" + cfg.get("rank_layout", "tagged")
    return key + ("|chain%d" % cfg["chain"] if cfg.get("chain", 1) > 1 else "")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag)
    names = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(src, "*.args")))
    out = {"tag": tag, "configs": {}}
    lines = ["# rocprofv3 summary, round tag %s" % tag, "",
             "Every configuration is ONE bench command, run four times on the MI355X box (tools/collect_profiles.sh): un-profiled",
             "(the JSON line, `profiles/%s_bench_<config>.json`), under `rocprofv3 --kernel-trace --stats`, and under" % tag,
             "`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; nothing traced beside the counters).",
             "HBM bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 counts 64 B per 128-B request on wide streams) + WRITE_SIZE KiB x 1024.",
             "`frac` = algorithmic bytes / average duration / 8 TB/s.", ""]
    traffic = {}
    for name in names:
        args = open(os.path.join(src, name + ".args")).read().strip()
        cfg = {"samples": 100000, "batch_sites": 131072, "groups": 0, "ranks": True}
        toks = args.split()
        for i, t in enumerate(toks):
            if t == "--samples": cfg["samples"] = int(toks[i + 1])
            if t == "--batch-sites": cfg["batch_sites"] = int(toks[i + 1])
            if t == "--groups": cfg["groups"] = int(toks[i + 1])
            if t == "--chain": cfg["chain"] = int(toks[i + 1])
            if t == "--no-rank-planes": cfg["ranks"] = False
            if t == "--rank-layout": cfg["rank_layout"] = toks[i + 1]
            if t == "--coverage": cfg["coverage"] = float(toks[i + 1])
            if t in ("--with-tile-mode", "--tile-job"): cfg["tile_job"] = True
        nvar = 0
        bj = os.path.join(src, name + ".bench.json")
        bench = None
        if os.path.exists(bj) and os.path.getsize(bj):
            try:
                bench = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][-1])
                nvar = int(bench["config"]["variant_sites_last_batch"])
                shutil.copy(bj, os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, name)))
            except Exception as ex:  # keep going: the profile is still worth having
                print("[warn] %s: no bench line (%s)" % (name, ex), file=sys.stderr)
        kern = collections.OrderedDict()
        for fn in newest(os.path.join(src, name + ".stats", "**", "*kernel_stats.csv")):
            for r in csv.DictReader(open(fn)):
                k = short(r["Name"])
                kern[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                           "pct": float(r["Percentage"])}
        fetch = counters(os.path.join(src, name + ".fetch", "**", "*counter_collection.csv"))
        write = counters(os.path.join(src, name + ".write", "**", "*counter_collection.csv"))
        # Calls of one kernel with different dispatch sizes (a tile job's finish beside the bench's own batches, a chunked or
        # ragged last launch) are split: the stats row averages them, which is meaningless for a bytes-per-launch ratio.
        groups = dispatch_groups(os.path.join(src, name + ".stats", "**", "*bv_dispatches.csv"))
        fetch_g = counters(os.path.join(src, name + ".fetch", "**", "*counter_collection.csv"), by_grid=True)
        write_g = counters(os.path.join(src, name + ".write", "**", "*counter_collection.csv"), by_grid=True)
        mixed = {k for k, gs in groups.items() if len(gs) > 1}
        sq = {}
        for d in glob.glob(os.path.join(src, name + ".sq*")):
            sq.update(counters(os.path.join(d, "**", "*counter_collection.csv")))
        lines += ["## %s" % name, "", "    python3 bench.py --no-cpu-baseline %s" % args, ""]
        if bench:
            r = bench["roofline"]
            lines += ["un-profiled bench line: %.4g sites/s, %s avg %.4f ms = %.3f of peak (pass 1 as a whole %.4f ms, pass 2 %.4f ms), "
                      "%d variant sites in the last batch" % (bench["value"], r["kernel"], r["avg_launch_ms"], r["frac"],
                                                             r.get("pass1_avg_ms", r["avg_launch_ms"]), r["pass2_avg_launch_ms"], nvar), ""]
        lines += ["| kernel | calls | avg duration | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch | algorithmic bytes | traffic / algorithmic | frac of 8 TB/s |",
                  "|---|---|---|---|---|---|---|---|---|"]
        for k, v in kern.items():
            if not k.startswith("bv_") or k.startswith("bv_synth"):  # (the bench's data generator is not part of the path)
                continue
            if k in mixed:
                # one row per dispatch size; the algorithmic bytes (known for the bench's own batch only) are not attributed
                for gsz in sorted(groups[k]):
                    d = groups[k][gsz]
                    fg, wg = fetch_g.get((k, gsz, "FETCH_SIZE")), write_g.get((k, gsz, "WRITE_SIZE"))
                    hb = (2.0 * fg * 1024.0 if fg is not None else 0.0) + (wg * 1024.0 if wg is not None else 0.0)
                    lines.append("| %s [grid %d] | %d | %.1f us | %s | %s | %s | (calls of several dispatch sizes: not attributed) | - | - |" % (
                        k, gsz, len(d), sum(d) / len(d) / 1e3, "%.0f" % fg if fg is not None else "-", "%.0f" % wg if wg is not None else "-",
                        "%.4g" % hb if fg is not None else "-"))
                    v.setdefault("by_grid", {})[str(gsz)] = {"calls": len(d), "avg_ns": sum(d) / len(d), "FETCH_SIZE_KiB": fg, "WRITE_SIZE_KiB": wg}
                continue
            f, w = fetch.get((k, "FETCH_SIZE")), write.get((k, "WRITE_SIZE"))
            hbm = (2.0 * f * 1024.0 if f is not None else 0.0) + (w * 1024.0 if w is not None else 0.0)
            kcfg, knvar = cfg, nvar
            if bench and bench.get("configs1") and cfg["samples"] > 49152 and k.startswith("bv_p1s_"):
                # the default bench command runs BASELINE configs[1] (100,000 sites x 10,000 samples) as a second leg of the same
                # process (bench.py: configs1_leg): the short-row kernels of this profile are that leg's
                kcfg = dict(cfg, samples=10000, batch_sites=100000)
                knvar = int(bench["configs1"]["variant_sites"])
            algo, how = algorithmic_bytes(k, kcfg, knvar)
            if kcfg is not cfg:
                how += "; the configs1 leg of the same command"
            v.update({"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes": hbm if f is not None else None, "algorithmic_bytes": algo,
                      "algorithmic_how": how})
            frac = (algo / (v["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK) if algo else None
            v["frac"] = frac
            for (kk, cn), cv in sq.items():
                if kk == k:
                    v.setdefault("sq", {})[cn] = cv
            lines.append("| %s | %d | %.1f us | %s | %s | %s | %s | %s | %s |" % (
                k, v["calls"], v["avg_ns"] / 1e3, "%.0f" % f if f is not None else "-", "%.0f" % w if w is not None else "-",
                "%.4g" % hbm if f is not None else "-", ("%.4g (%s)" % (algo, how)) if algo else "-",
                "%.3f" % (hbm / algo) if (algo and f is not None) else "-", "%.3f" % frac if frac else "-"))
            if algo and f is not None and k.startswith(("bv_pass1", "bv_p1s_stream", "bv_p1s_fused")):
                # (several configurations run the same kernel on the same shape: the first one, in name order, is quoted)
                if cfg.get("coverage", 0.08) != 0.08:
                    continue  # (pmc_traffic.json is keyed by shape: BASELINE's coverage only)
                traffic.setdefault(tkey(k, kcfg), {}).update({} if traffic[tkey(k, kcfg)] else {
                    "hbm_bytes_per_launch": hbm, "read_bytes": 2.0 * f * 1024.0, "write_bytes": (w or 0.0) * 1024.0,
                    "algorithmic_bytes": algo, "source": "profiles/%s_rocprof_summary.md#%s" % (tag, name)})
        lines.append("")
        sqk = [k for k in kern if "sq" in kern[k] and not k.startswith("bv_synth")]
        if sqk:
            lines += ["SQ counters per launch (`rocprofv3 --pmc`, three passes):", "", "```"]
            for k in sqk:
                for cn in sorted(kern[k]["sq"]):
                    lines.append("%-44s %-24s %.4g" % (k[:44], cn, kern[k]["sq"][cn]))
            lines += ["```", ""]
        out["configs"][name] = {"args": args, "config": cfg, "variant_sites": nvar, "kernels": kern}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.json" % tag), "w"), indent=1)
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    cur = json.load(open(tf)) if os.path.exists(tf) else {}
    cur.update(traffic)
    json.dump(cur, open(tf, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
