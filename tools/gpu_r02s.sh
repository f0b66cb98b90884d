set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 600 python -m pytest tests/test_lbm.py tests/test_host_compat.py -x -q -m gpu > $out/tests.log 2>&1; note "lbm + host compat gpu tests rc=$?"; tail -15 $out/tests.log
timeout 300 python tools/lbm_bench.py > $out/lbm_bench.log 2>&1; note "lbm bench rc=$?"; cat $out/lbm_bench.log | tail -12
