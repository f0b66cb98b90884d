#!/bin/bash
# usage: bench_env.sh "<ENV=.. ENV=..>"  -- default bench with extra environment, one summary line
env $1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['value']), round(d['ms_per_tick'],1), {k: round(v,1) for k,v in d['phase_ms_per_tick'].items()}, d['engine_stats'])"
