#!/usr/bin/env python3
"""Condense a rocprofv3 run (kernel-trace --stats CSVs + separate --pmc passes) into the small summary
files committed under profiles/.

usage: tools/prof_summary.py <gpurun_out/prof dir> <profiles/prefix> [kernel-substring] [workload-tag]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes (TCC slot limits); on gfx950 FETCH_SIZE counts the 128-B requests of a
wide coalesced read (16 B per lane) at 64 B, so it is doubled for kernels whose reads are of that kind
(every B-row read of the SpMM kernels is a 16-B-per-lane dwordx4); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_common as PC  # noqa: E402


def pmc(dirname, kernel_sub):
    acc = collections.defaultdict(list)
    # gpurun merges new files into gpurun_out/ without deleting old ones: keep only the newest run of every pass
    newest = {}
    for f in glob.glob(os.path.join(dirname, "pmc_*", "*", "*_counter_collection.csv")):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    ksub = sys.argv[3] if len(sys.argv) > 3 else "spmm"
    stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    with open(prefix + "_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    counters, counts = pmc(src, ksub)
    out = {"kernel_substring": ksub, "workload": sys.argv[4] if len(sys.argv) > 4 else "cfg2-default",
           "source_blobs": PC.source_blobs(),       # bench.py quotes the counters only while the kernel's files still hash to these
           "counters_avg_per_launch": counters, "launches_sampled": counts}
    k = [r for r in rows if ksub in r["Name"]]
    if k:
        out["kernel"] = k[0]["Name"][:200]
        out["avg_ns"] = float(k[0]["AverageNs"])
        out["calls"] = int(k[0]["Calls"])
    # the launches of ONE grid size (one workload) when the command ran the kernel on several: the group with the most time
    traces = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
    if traces:
        groups, _ = PC.dispatch_groups(traces[0], [ksub])
        g = groups[ksub]
        if g:
            main_g = max(g, key=lambda q: sum(g[q]))
            out["grid_size"] = main_g
            out["avg_ns"] = sum(g[main_g]) / len(g[main_g])
            out["calls"] = len(g[main_g])
            cg = PC.counter_groups(src, [ksub])[ksub].get(main_g, {})
            if cg:
                counters = {n: sum(v) / len(v) for n, v in cg.items()}
                out["counters_avg_per_launch"] = counters
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch = counters["FETCH_SIZE"] * 1024.0
        write = counters["WRITE_SIZE"] * 1024.0
        out["hbm_traffic_bytes_per_launch"] = {
            "fetch_raw": fetch, "fetch_corrected_x2": 2 * fetch, "write": write,
            "total_corrected": 2 * fetch + write,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 tallies the 128-B requests of "
                    "16-B/lane reads at 64 B); Infinity-Cache hits are included in FETCH_SIZE (fabric-side counter)"}
    if "TCC_HIT_sum" in counters:
        h, m = counters["TCC_HIT_sum"], counters["TCC_MISS_sum"]
        out["l2_hit_rate"] = h / (h + m)
    json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
