#!/bin/bash
# same-box A/B of two builds of libmxgpu.so: bash tools/ab.sh tools/ab/lib_old.so tools/ab/lib_new.so [bench args]
A=$1; B=$2; shift 2
for r in 1 2 3; do for v in $A $B; do MXGPU_LIB=$PWD/$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d.get('steady_state_cached_plan',{}).get('ms_per_step'))"; done; done
