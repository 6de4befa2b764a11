#!/usr/bin/env python3
"""HBM-side traffic per launch of SEVERAL kernels of one profiled command (the extras of bench.py): reads the
kernel-trace stats and the separate FETCH_SIZE / WRITE_SIZE --pmc passes written by tools/profile.sh and writes
<prefix>_pmc.json = {"workload": tag, "kernels": {substring: {kernel, avg_ns, calls, fetch_raw, fetch_corrected_x2, write,
total_corrected}}}.  FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads.
usage: tools/prof_summary_multi.py <prof dir> <profiles/prefix> <workload-tag> <kernel-substring> [...]"""
import collections, csv, glob, json, os, sys

src, prefix, tag, subs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
rows = list(csv.DictReader(open(stats[0]))) if stats else []
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for sub in subs:
            if sub in r["Kernel_Name"]:
                acc[sub][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"workload": tag, "kernels": {}, "note": "FETCH_SIZE / WRITE_SIZE in KiB from separate --pmc passes; FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of 16-B/lane reads at 64 B; narrower reads are uncalibrated: "
       "an upper bound there); averages over all launches of the kernel in the run"}
for sub in subs:
    k = [r for r in rows if sub in r["Name"]]
    c = {n: sum(v) / len(v) for n, v in acc[sub].items()}
    e = {"kernel": k[0]["Name"][:160] if k else None, "avg_ns": float(k[0]["AverageNs"]) if k else None,
         "calls": int(k[0]["Calls"]) if k else 0}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e.update(fetch_raw=c["FETCH_SIZE"] * 1024, fetch_corrected_x2=2 * c["FETCH_SIZE"] * 1024, write=c["WRITE_SIZE"] * 1024,
                 total_corrected=2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    for n in ("TCC_HIT_sum", "TCC_MISS_sum"):
        if n in c:
            e[n] = c[n]
    out["kernels"][sub] = e
json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
