set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/et -o t -- python tools/epoch_trace.py run > $out/epoch_trace.log 2>&1; note "epoch trace rc=$?"
f=$(ls $out/et/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/epoch_trace.py post $f $out/epoch_durations.npz >> $out/epoch_trace.log 2>&1; rm -rf $out/et; tail -4 $out/epoch_trace.log
run() { # run "<env>" "<args>"
  env $1 timeout 300 python bench.py --engine batched --steps 3 --warmup 1 --no-cpu-baseline --no-lines $2 > $out/bench_tmp.log 2>&1
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_tick']; print('$1 | $2 |', 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), 'water', round(p['water'],1), 'grid', round(p['grid'],1), 'wind', round(p['wind'],1), 'epochs', d['engine_stats']['epochs'], 'lost', d['engine_stats']['children_lost'], 'sumh', d['digest']['sumh'])" 2>&1 | tee -a $out/sweep.txt || tail -3 $out/bench_tmp.log
}
for w in 128 256 512; do run "SMX_BATCH_WAVES=$w" ""; done
for w in 128 256 512; do run "SMX_BATCH_WAVES=$w" "--nwater 1000 --nwind 250"; done
run "SMX_BATCH_WAVES=256" "--nwater 256000 --nwind 64000"
timeout 400 python tools/wet_regime.py --size 256 --warm 100 --ticks 50 --engine batched --chunk 50 > $out/wet256_batched.log 2>&1; note "wet256 rc=$?"; tail -1 $out/wet256_batched.log | cut -c1-1800
timeout 900 python tools/wet_regime.py --size 1024 --nwater 4000 --warm 100 --ticks 20 --engine batched --chunk 20 > $out/wet1024_batched.log 2>&1; note "wet1024 rc=$?"; tail -1 $out/wet1024_batched.log | cut -c1-1800
cat $out/status.txt
