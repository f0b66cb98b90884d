#!/bin/bash
# One GPU call, most valuable first (every step has its own timeout and leaves a status line):
#   tools/gpu_round_end.sh <tag> [budget-seconds]
# 1 smoke  2 quick parity subset  3 default bench line  4 rocprofv3 --kernel-trace --stats of the same command
# 5 the full `pytest -m gpu` suite  6 the --pmc passes.  Outputs under gpurun_out/<tag>/ (tools/profile_summary.py
# condenses them into profiles/<tag>_*). Steps that do not fit into the budget are skipped, and say so.
set -u
tag=${1:-r01b}
budget=${2:-420}
root="${GRAFT_REPO_ROOT:-/root/repo}"
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s)
left() { echo $(( budget - ($(date +%s) - t0) )); }
note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
step() {  # step <name> <need-seconds> <timeout> <cmd...>
  local name=$1 need=$2 to=$3; shift 3
  if [ "$(left)" -lt "$need" ]; then note "$name SKIPPED (left $(left)s < $need s)"; return 1; fi
  local lim=$to; [ "$(left)" -lt "$lim" ] && lim=$(left)
  timeout $lim "$@" > $out/$name.log 2> $out/$name.err; local rc=$?
  note "$name rc=$rc"
  return $rc
}
python -c "import torch" > /dev/null 2>&1 &          # page the image in while the torch-free steps run
step smoke 20 120 python -c "import __graft_entry__ as g; g.smoke()"
tail -2 $out/smoke.log
step quick 40 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "snapshots or phase_by_phase or import_resume"
tail -3 $out/quick.log
wait
step bench 60 240 python bench.py
cp $out/bench.log $out/bench.json 2>/dev/null
tail -c 400 $out/bench.log; echo
step trace 60 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline
if [ -f soilmachine_amd/libsoilmx_prev.so ]; then      # A/B: the previous build of the library on the same box
  cp soilmachine_amd/libsoilmx.so /tmp/libsoilmx_new.so && cp soilmachine_amd/libsoilmx_prev.so soilmachine_amd/libsoilmx.so
  step bench_prev 45 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
  cp /tmp/libsoilmx_new.so soilmachine_amd/libsoilmx.so
  tail -c 300 $out/bench_prev.log; echo
fi
step fulltests 120 900 python -m pytest tests -x -q -m gpu
tail -5 $out/fulltests.log
P="python bench.py --steps 3 --warmup 2 --no-cpu-baseline"
step fetch 60 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $P
step write 60 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $P
step sq 60 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/sq -o s -- $P
# instruction-cache behaviour of the particle kernels (k_spec_exec_coop<wind> is 492 KB of code)
step icache 60 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/icache -o i -- $P
SMX_TRACE_ROUNDS=1 timeout 60 python bench.py --steps 1 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/rounds_trace.txt
find $out -name "*.csv" | head -20
cat $out/status.txt
