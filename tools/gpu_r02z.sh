set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_strips.py -x -q > $out/tests.log 2>&1; note "batched+strips gpu tests rc=$?"; tail -5 $out/tests.log
run() { # run "<env>" "<args>"
  env $1 timeout 300 python bench.py --engine batched --steps 3 --warmup 2 --no-cpu-baseline --no-lines $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'epochs', d['engine_stats']['epochs'], 'nested', round(d['steps_per_tick']['nested_particles']), 'floods', round(d['steps_per_tick']['floods']), 'avg_launch_us', round(1e3*d['roofline']['avg_launch_ms'],1))" 2>&1 | tee -a $out/sweep.txt || tail -3 $out/bench_tmp.log
}
run "A=1" ""
for k in 2 8 16; do run "SMX_BATCH_KFLOOD=$k" ""; done
run "A=1" "--nwater 1000 --nwind 250"
run "A=1" "--size 8192"
cat $out/status.txt
