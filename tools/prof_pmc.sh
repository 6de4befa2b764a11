#!/bin/bash
# Counter passes for the kernels of any python script, on the GPU box (never combined with a trace domain):
#   bash tools/prof_pmc.sh <out-dir under gpurun_out/> <kernel filter> "<counters of pass 1>" ["<pass 2>" ...] -- <script> [args]
set -u
OUT=$1; FILT=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
i=0
for c in "${PASSES[@]}"; do
  d=$R/$OUT/pmc_$i; i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/"$@" > $d.log 2>&1
done
cd $R && python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in sys.argv[2].split(",")):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
