for i in 1 2 3; do python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:(round(d[k]['fp32']['ms_per_step']*1e3,1), round(d[k]['bf16']['ms_per_step']*1e3,1)) for k in ('c1','c2','c2x10','c5','b100')}, round(d['fp32']['ms_per_step']*1e3,1), d['c4']['exchange_on_one_rank_us'])"; done
