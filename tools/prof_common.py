"""Shared by tools/prof_summary*.py (which write the rocprofv3 summaries committed under profiles/) and bench.py (which
quotes HBM traffic from them): which source files a kernel is built from, their git blob hashes, and the per-workload
grouping of a kernel's dispatches.

A summary records the blob hash of every source file of libmxgpu.so at the time it was made; bench.py quotes a
summary's counters only when the files the kernel comes from still hash to the same value (round 3's
line cited round 2's counters for a kernel that had changed) — otherwise `traffic` is null."""
from __future__ import annotations

import collections
import csv
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "matrixextra_amd", "csrc")

# kernel-name substring -> the files it is compiled from (beside the common headers)
KERNEL_SOURCES = [
    ("spmm_plan_kernel", ["spmm_plan.hip"]), ("plan_", ["spmm_plan.hip"]), ("repack", ["spmm_slab.hip"]),
    ("spmm_rowwave_kernel", ["spmm_rowwave.hip"]), ("spmm_rowsplit_kernel", ["spmm_rowsplit.hip"]),
    ("rowsplit_cursors_kernel", ["spmm_rowsplit.hip"]), ("spmm_rowgroup_kernel", ["spmm_rowsplit.hip"]), ("spmm_slab_kernel", ["spmm_slab.hip"]),
    ("spmm_tile_kernel", ["spmm_tile.hip"]), ("tile_unsorted_rows_kernel", ["spmm_tile.hip"]),
    ("spmv_flat_kernel", ["spmv_flat.hip", "spmv_rows.h"]), ("slice_rows_kernel", ["spmv_flat.hip"]),
    ("spmv_plan", ["spmv_plan.hip"]), ("spmv_tile", ["spmv_tile.hip", "spmv_rows.h"]), ("spmv_kernel", ["spmv.hip", "spmv_rows.h"]),
    ("gather_", ["gather.hip"]), ("rows_sorted", ["gather.hip"]), ("sort_rows", ["gather.hip"]), ("is_seq", ["gather.hip"]),
    ("merge_", ["merge.hip"]), ("values_elemwise", ["merge.hip"]), ("scan", ["scan.hip"]),
    ("stream_copy_kernel", ["stream.hip"]), ("csr_by_dvec", ["dvec.hip", "r_arith.h"]), ("dvec_na", ["dvec_na.hip", "r_arith.h"]),
    ("drop_", ["dropzeros.hip"]), ("colslice", ["colslice.hip"]), ("svec", ["svec.hip"]), ("bind", ["bind.hip"]),
    ("profile_sample_kernel", ["profile.hip"]), ("profile_bins_kernel", ["profile.hip"]),
    ("spmm_longrows", ["spmm_rowsplit.hip"]),
]
COMMON = ["mx_common.h", "spmm_common.h"]


def blob_sha(path):
    """what `git hash-object` prints for the file (works without a repository: the GPU box has no .git)"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_blobs():
    """blob hash of every source file of the library"""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    return {os.path.basename(f): blob_sha(f) for f in files}


def sources_of(kernel_name):
    """the LONGEST table entry found in the name decides (`spmv_plan_kernel` contains the generic `plan_` of the SpMM plan's
    build kernels: first-hit matching hashed spmm_plan.hip for it — round 4's advisor finding)"""
    hits = [(len(sub), files) for sub, files in KERNEL_SOURCES if sub in kernel_name]
    if not hits:
        return None        # unknown kernel: every file has to match
    files = max(hits, key=lambda h: h[0])[1]
    return files + [f for f in COMMON if not (f == "spmm_common.h" and "spmm" not in kernel_name)]


def kernel_unchanged(kernel_name, recorded):
    """(ok, why): do the files `kernel_name` is built from hash to what the summary recorded?"""
    if not recorded:
        return False, "the summary records no source hashes"
    files = sources_of(kernel_name)
    now = source_blobs()
    for f in (files if files is not None else sorted(now)):
        if recorded.get(f) != now.get(f):
            return False, f"{f} changed since the profile was taken"
    return True, ""


def newest_round(pattern):
    """files matching profiles/rNN_<pattern> of the highest round NN only (older rounds' summaries are history)"""
    files = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern))
    if not files:
        return []
    rounds = collections.defaultdict(list)
    for f in files:
        rounds[int(re.match(r"r(\d\d)_", os.path.basename(f)).group(1))].append(f)
    return sorted(rounds[max(rounds)])


def short_kernel(name):
    return name[:160]


def dispatch_groups(trace_csv, subs):
    """{substring: {grid_size: [durations ns]}} from a rocprofv3 *_kernel_trace.csv: the launches of one kernel grouped by
    grid size — i.e. by WORKLOAD when a command runs the same kernel on several (cfg3's SpMV and the vignette loop's)"""
    out = {s: collections.defaultdict(list) for s in subs}
    names = {}
    for r in csv.DictReader(open(trace_csv)):
        for s in subs:
            if s in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                out[s][g].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                names[s] = r["Kernel_Name"]
    return out, names


def counter_groups(prof_dir, subs):
    """{substring: {grid_size: {counter: [values]}}} from the separate --pmc passes (newest run of every pass)"""
    acc = {s: collections.defaultdict(lambda: collections.defaultdict(list)) for s in subs}
    newest = {}
    for f in glob.glob(os.path.join(prof_dir, "pmc_*", "*", "*_counter_collection.csv")):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            for s in subs:
                if s in r["Kernel_Name"]:
                    acc[s][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
