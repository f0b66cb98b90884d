set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02ad; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout 300 python -m pytest tests/test_gpu_strips.py -x -q > $out/tests.log 2>&1; echo "strips gpu tests rc=$?"; tail -3 $out/tests.log
