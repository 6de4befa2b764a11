#!/bin/bash
# Per-kernel durations of any python script on the GPU box:  bash tools/prof_kernels.sh <out-dir under gpurun_out/> <filter> <script> [args]
set -u
OUT=$1; FILT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/"$@" > $R/$OUT/run.log 2>&1
tail -5 $R/$OUT/run.log
cd $R && python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in sys.argv[2].split(",")):
        print(r["Name"][:100], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
PY
