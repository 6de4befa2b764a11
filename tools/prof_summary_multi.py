#!/usr/bin/env python3
"""HBM-side traffic per launch of SEVERAL kernels of one profiled command (the extras of bench.py): reads the kernel trace
and the separate FETCH_SIZE / WRITE_SIZE --pmc passes written by tools/profile.sh and writes <prefix>_pmc.json =
{"workload": tag, "source_blobs": {...}, "kernels": {substring: {kernel, avg_ns, calls, grid_size, fetch_raw,
fetch_corrected_x2, write, total_corrected, other_workloads: [...]}}}.

Per WORKLOAD: a command may launch one kernel on several workloads (bench.py's extras run
spmv_plan_kernel on cfg3's matrix and on the vignette loop's small one); the launches are grouped by grid size, the
figures quoted are those of the group with the most total time, and the other groups are listed beside it.
FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads.
usage: tools/prof_summary_multi.py <prof dir> <profiles/prefix> <workload-tag> <kernel-substring> [...]"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_common as PC  # noqa: E402

src, prefix, tag, subs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
traces = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
durs, names = PC.dispatch_groups(traces[0], subs) if traces else ({s: {} for s in subs}, {})
ctrs = PC.counter_groups(src, subs)
out = {"workload": tag, "source_blobs": PC.source_blobs(), "kernels": {},
       "note": "FETCH_SIZE / WRITE_SIZE in KiB from separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
               "tallies the 128-B requests of 16-B/lane reads at 64 B; narrower reads are uncalibrated: an upper bound there); "
               "per kernel the launches of ONE grid size (= one workload): the group with the most total time"}
for sub in subs:
    groups = durs.get(sub, {})
    if not groups:
        out["kernels"][sub] = {"kernel": None, "avg_ns": None, "calls": 0}
        continue
    main = max(groups, key=lambda g: sum(groups[g]))
    d = groups[main]
    e = {"kernel": PC.short_kernel(names[sub]), "grid_size": main, "avg_ns": sum(d) / len(d), "min_ns": min(d), "calls": len(d)}
    c = {n: sum(v) / len(v) for n, v in ctrs[sub].get(main, {}).items()}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e.update(fetch_raw=c["FETCH_SIZE"] * 1024, fetch_corrected_x2=2 * c["FETCH_SIZE"] * 1024, write=c["WRITE_SIZE"] * 1024,
                 total_corrected=2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    for n in ("TCC_HIT_sum", "TCC_MISS_sum"):
        if n in c:
            e[n] = c[n]
    other = {n: v for n, v in c.items() if n not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")}
    if other:                                   # the SQ / TA / LDS passes of `tools/profile.sh <dir> 2` (means per launch)
        e["counters"] = other
    others = [{"grid_size": g, "calls": len(v), "avg_ns": sum(v) / len(v)} for g, v in groups.items() if g != main]
    if others:
        e["other_workloads"] = sorted(others, key=lambda o: -o["calls"])[:6]
    out["kernels"][sub] = e
json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
