#!/usr/bin/env python3
"""tools/summarize_profile.py TAG WORKLOAD -- turn the rocprofv3 output of tools/profile.sh TAG WORKLOAD
(gpurun_out/prof_TAG/{trace,pmc1..pmc4}) into the summaries kept under profiles/:

  profiles/TAG_kernel_stats.csv   the --kernel-trace --stats table
  profiles/TAG_pmc_summary.json   per kernel, every counter averaged over its launches
  profiles/TAG_bench.json         the bench line of the traced run
  profiles/traffic.json           [WORKLOAD] HBM bytes per launch of the dominant kernel, read by bench.py

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB, and gfx950 counts
128-byte fetch requests as 64 bytes, so fetch bytes = 2 x FETCH_SIZE x 1024."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        sys.exit("no kernel_stats.csv under " + src)
    # (gpurun merges a run's files into gpurun_out/ without removing older ones: the newest file of every pass counts)
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, tag + "_kernel_stats.csv"))
    bl = os.path.join(src, "bench_line.json")
    bench = None
    if os.path.exists(bl) and os.path.getsize(bl):
        bench = json.load(open(bl))
        json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
    acc = defaultdict(lambda: defaultdict(list))
    newest = {}
    for path in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        d = os.path.relpath(path, src).split(os.sep)[0]
        if d not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d]):
            newest[d] = path
    for path in sorted(newest.values()):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    summary = {k: {c: sum(v) / len(v) for c, v in sorted(ctrs.items())} | {"launches": max(len(v) for v in ctrs.values())}
               for k, ctrs in sorted(acc.items()) if k.startswith("lime::")}
    with open(os.path.join(dst, tag + "_pmc_summary.json"), "w") as g:
        json.dump(summary, g, indent=1)
    dom = max((k for k in summary if "k_scan<" in k), key=lambda k: summary[k].get("FETCH_SIZE", 0.0), default=None)
    if dom and "FETCH_SIZE" in summary[dom] and "WRITE_SIZE" in summary[dom]:
        s = summary[dom]
        tfile = os.path.join(dst, "traffic.json")
        try:
            allt = json.load(open(tfile))
            if "kernel" in allt:            # round-1 layout: one entry
                allt = {}
        except Exception:
            allt = {}
        allt[wl] = {
            "kernel": dom, "symbols": bench["config"]["symbols_total"] if bench else None,
            "hbm_bytes_per_launch": int(2 * s["FETCH_SIZE"] * 1024 + s["WRITE_SIZE"] * 1024),
            "FETCH_SIZE_KiB": s["FETCH_SIZE"], "WRITE_SIZE_KiB": s["WRITE_SIZE"], "TCC_EA0_ATOMIC_sum": s.get("TCC_EA0_ATOMIC_sum"),
            "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None,
            # all kernels of one pass (scan, resolve, bin prefix sums, partition levels, table build, long clusters): each kernel's
            # per-launch average; the kernels of a pass are launched once per pass
            "pass_hbm_bytes": int(sum(2 * v.get("FETCH_SIZE", 0.0) * 1024 + v.get("WRITE_SIZE", 0.0) * 1024 for k, v in summary.items()
                                      if not any(x in k for x in ("k_synth", "k_fill", "k_choose", "k_gather")))),
            "pass_kernels": sorted(k for k in summary if not any(x in k for x in ("k_synth", "k_fill", "k_choose", "k_gather"))),
            "from": f"profiles/{tag}_pmc_summary.json: HBM bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE x 1024, "
                    f"separate --pmc passes of tools/profile.sh {tag} {wl}, averages over {s['launches']} launches",
        }
        json.dump(allt, open(tfile, "w"), indent=1)
        print(json.dumps(allt[wl]))
    for k, s in summary.items():
        print(k, {c: round(v) for c, v in s.items()})


if __name__ == "__main__":
    main()
