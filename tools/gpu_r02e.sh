set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
run() { # run "<env>" "<args>"
  env $1 timeout 300 python bench.py --engine batched --steps 2 --warmup 1 --no-cpu-baseline $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'epochs', d['engine_stats']['epochs'], 'nested/tick', round(d['steps_per_tick']['nested_particles']))" 2>&1 | tee -a $out/sweep.txt || tail -3 $out/bench_tmp.log
}
W="SMX_BATCH_SCHED_WIND=2,2,16"
for tail in "100000:0,0,1" "4096:0,2,4;100000:0,0,1" "16384:0,2,4;100000:0,0,1" "1024:0,3,8;16384:0,2,4;100000:0,0,1" "16384:0,1,2;100000:0,0,1" "16384:1,1,4;100000:0,0,1" "4096:1,2,8;16384:1,1,4;100000:0,0,1" "100000:0,1,2" "100000:1,0,1"; do
  run "$W SMX_BATCH_SCHED=0,0,1 SMX_BATCH_SCHED_TAIL=$tail" "--nwater 64000 --nwind 16000"
done
for tail in "100000:0,0,1" "100000:0,2,4" "100000:1,1,4" "100000:1,2,8" "100000:0,3,8"; do
  run "SMX_BATCH_SCHED_WIND=3,1,16 SMX_BATCH_SCHED=0,0,1 SMX_BATCH_SCHED_TAIL=$tail" "--nwater 1000 --nwind 250"
done
for w in "3,1,16" "3,1,32" "3,2,32" "4,1,32" "3,1,64"; do
  run "SMX_BATCH_SCHED_WIND=$w SMX_BATCH_SCHED=1,1,4" "--nwater 1000 --nwind 250"
done
