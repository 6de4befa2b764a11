#!/bin/bash
# calibration sweeps of the planned SpMM kernel (run on the GPU box): bash tools/calib_planned.sh
run() { timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], 'ms; step', d['ms_per_step'], 'GFLOP/s', d['value'], 'err', d['parity_max_err_over_max_abs_vs_oracle'])"; }
run --algo 3 --cols 4096 --panels 1
run --algo 3 --cols 4096 --panels 8
run --algo 1 --cols 4096
for p in 6 8 10 12; do run --algo 3 --panels $p; done
