for P in 6 8 10 12 16 24; do python bench.py --no-extras --no-cpu-baseline --panels $P 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 P=$P', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; done
for P in 12 16 20 24 32; do python bench.py --config cfg5 --no-extras --no-cpu-baseline --panels $P 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 P=$P', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; done
