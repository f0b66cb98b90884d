set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
run() { # run "<env>" "<args>"
  env $1 timeout 300 python bench.py --engine batched --steps 2 --warmup 1 --no-cpu-baseline --no-lines $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'epochs', d['engine_stats']['epochs'], 'nested/tick', round(d['steps_per_tick']['nested_particles']))" 2>&1 | tee -a $out/sweep.txt || tail -3 $out/bench_tmp.log
}
run "A=1" ""
for sc in "1,0,2" "1,0,4" "2,0,2" "2,0,4" "2,0,8" "3,0,4" "3,0,8"; do run "SMX_BATCH_SCHED=$sc" ""; done
