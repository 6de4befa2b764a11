#!/bin/bash
# calibration sweeps of the planned SpMM kernel (run on the GPU box): bash tools/calib_planned.sh
run() { timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], 'ms; step', d['ms_per_step'], 'GFLOP/s', d['value'], 'frac', d['roofline']['frac'], 'err', d['parity_max_err_over_max_abs_vs_oracle'])"; }
# cfg5 per-GPU shape: 1M x 200k, 64/row, f32 dense 200k x 256
run --rows 1000000 --cols 200000 --nnz-row 64 --n 256 --dtype f32 --layout rowmajor
for p in 6 8 10 12 16; do
run --rows 1000000 --cols 200000 --nnz-row 64 --n 256 --dtype f32 --layout rowmajor --algo 3 --panels $p
done
run --rows 1000000 --cols 200000 --nnz-row 64 --n 256 --dtype f32 --layout rowmajor --algo 1
