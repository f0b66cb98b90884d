set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 600 python -m pytest tests/test_gpu_batched.py -x -q > $out/tests.log 2>&1; note "batched gpu tests rc=$?"; tail -4 $out/tests.log
run() { # run "<env>" "<args>"
  env $1 timeout 300 python bench.py --engine batched --steps 3 --warmup 1 --no-cpu-baseline --no-lines $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'epochs', d['engine_stats']['epochs'], 'avg_launch_us', round(1e3*d['roofline']['avg_launch_ms'],1), 'sumh', d['digest']['sumh'])" 2>&1 | tee -a $out/sweep.txt || tail -3 $out/bench_tmp.log
}
for w in 0 256 1024 2048 4096; do run "SMX_BATCH_WAVES=$w" ""; done
run "SMX_BATCH_WAVES=2048 SMX_BATCH_OCC2=1" ""
run "SMX_BATCH_WAVES=4096 SMX_BATCH_OCC2=1" ""
run "SMX_BATCH_WAVES=1024 SMX_BATCH_SCHED=2,0,8" ""
run "SMX_BATCH_WAVES=1024 SMX_BATCH_SCHED=2,0,2" ""
run "SMX_BATCH_WAVES=1024 SMX_BATCH_SCHED_WIND=3,1,32" ""
for w in 0 1024; do run "SMX_BATCH_WAVES=$w" "--nwater 1000 --nwind 250"; done
for w in 0 1024; do run "SMX_BATCH_WAVES=$w" "--nwater 256000 --nwind 64000"; done
cat $out/status.txt
