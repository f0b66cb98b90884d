set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 300 python tools/gpu_r02b.py > $out/check.log 2>&1; note "check rc=$?"; tail -12 $out/check.log
for cfg in "--nwater 1000 --nwind 250" "--nwater 64000 --nwind 16000" "--nwater 256000 --nwind 64000"; do
  timeout 300 python bench.py --engine batched --steps 3 --warmup 1 --no-cpu-baseline $cfg > $out/bench_tmp.log 2>&1; note "bench $cfg rc=$?"
  tail -1 $out/bench_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'steps/s', round(d['value']), 'ms/tick', round(d['ms_per_tick'],2), d['phase_ms_per_tick'], d['steps_per_tick'], d['engine_stats'], 'kernel', d['roofline']['avg_launch_ms'], d['roofline']['launches'])" 2>&1 | tee -a $out/bench_lines.txt || tail -5 $out/bench_tmp.log
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --engine batched --steps 2 --warmup 1 --no-cpu-baseline --nwater 64000 --nwind 16000 > $out/trace.log 2>&1; note "trace rc=$?"
head -8 $out/trace/t_kernel_stats.csv | cut -c1-160
cat $out/status.txt
