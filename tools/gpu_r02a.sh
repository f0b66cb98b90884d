set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
python -c "import torch" > /dev/null 2>&1 &
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; note "smoke rc=$?"
wait
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_new.log 2>&1; note "bench_new rc=$?"; tail -c 1500 $out/bench_new.log
cp soilmachine_amd/libsoilmx.so /tmp/new.so; cp soilmachine_amd/libsoilmx_prev.so soilmachine_amd/libsoilmx.so
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_prev.log 2>&1; note "bench_prev rc=$?"; tail -c 1500 $out/bench_prev.log
cp /tmp/new.so soilmachine_amd/libsoilmx.so
SMX_TRACE_ROUNDS=1 timeout 60 python bench.py --steps 1 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/rounds_trace.txt; note "rounds rc=$?"
timeout 900 python -m pytest tests -x -q -m gpu --durations=25 > $out/fulltests.log 2>&1; note "fulltests rc=$?"; tail -40 $out/fulltests.log
timeout 300 python tools/wet_regime.py --size 256 --warm 100 --ticks 50 --engine spec --chunk 50 > $out/wet256.log 2>&1; note "wet256 rc=$?"; tail -5 $out/wet256.log
P="python bench.py --steps 3 --warmup 2 --no-cpu-baseline"
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/icache -o i -- $P > $out/icache.log 2>&1; note "icache rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $P > $out/trace.log 2>&1; note "trace rc=$?"
cat $out/status.txt
