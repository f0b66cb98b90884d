set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02ab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
run() {
  env $1 timeout 600 python bench.py --engine batched --steps 2 --warmup 1 --no-cpu-baseline --no-lines $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'sumh', d['digest']['sumh'])" 2>&1 | tee -a $out/sweep.txt || tail -5 $out/bench_tmp.log
}
run "A=1" "--size 16384"
run "SMX_BATCH_OCC2_ABOVE=65536" "--size 16384"
run "SMX_BATCH_OCC2_ABOVE=32768" "--size 16384"
run "SMX_BATCH_OCC2_ABOVE=65536" "--size 8192"
run "SMX_BATCH_OCC2_ABOVE=32768" ""
