set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_host_compat.py tests/test_gpu_parity.py -x -q -k "slice or checkpoint or compat or vertex or host" --durations=6 > $out/tests.log 2>&1; note "tests rc=$?"; tail -25 $out/tests.log
cat $out/status.txt
