#!/bin/bash
# rocprofv3 passes of one bench.py command, on the GPU box:
#   bash tools/profile.sh <out-dir under gpurun_out/> <pmc: 0|1|2> -- <bench.py args>
# pass 1: --kernel-trace --stats (per-kernel durations); pmc>=1: FETCH_SIZE, WRITE_SIZE, L2 hit/miss in their own
# passes (TCC slot limits; never combined with a trace domain); pmc>=2: the SQ / TCP passes as well.
# Summaries for profiles/ are then made with tools/prof_summary.py.
set -u
OUT=$1; PMC=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py "$@" > $R/$OUT/bench_trace.log 2>&1
if [ "$PMC" -ge 1 ]; then
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$R/$OUT/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py "$@" > $d.log 2>&1
  done
fi
if [ "$PMC" -ge 2 ]; then
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
    d=$R/$OUT/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py "$@" > $d.log 2>&1
  done
fi
ls $R/$OUT
