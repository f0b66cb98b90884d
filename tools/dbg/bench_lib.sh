#!/bin/bash
# usage: bench_lib.sh <variant.so>  -- runs the default bench with the given library swapped in
cp "$1" soilmachine_amd/libsoilmx.so
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_tick'], d['phase_ms_per_tick'], d['engine_stats'])"
