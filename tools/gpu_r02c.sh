set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 300 python tools/gpu_r02b.py > $out/check.log 2>&1; note "check rc=$?"; tail -8 $out/check.log
SMX_TRACE_BATCH=1 timeout 300 python bench.py --engine batched --steps 1 --warmup 1 --no-cpu-baseline --nwater 64000 --nwind 16000 > $out/bench_trace.log 2> $out/batch_trace.txt; note "trace bench rc=$?"
tail -c 600 $out/bench_trace.log; wc -l $out/batch_trace.txt
for cfg in "--nwater 1000 --nwind 250" "--nwater 64000 --nwind 16000"; do
  timeout 300 python bench.py --engine batched --steps 3 --warmup 1 --no-cpu-baseline $cfg > $out/bench_tmp.log 2>&1; note "bench $cfg rc=$?"
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'steps/s', round(d['value']), 'ms/tick', round(d['ms_per_tick'],2), d['phase_ms_per_tick'], d['steps_per_tick'], d['engine_stats'])" 2>&1 | tee -a $out/bench_lines.txt || tail -5 $out/bench_tmp.log
done
cat $out/status.txt
