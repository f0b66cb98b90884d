set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02deep; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s)
SMX_DEEP=1 timeout 900 python -m pytest "tests/test_gpu_depth.py::test_appendix_e_default256_deep[default256_t300]" -x -q --durations=3 > $out/deep300.log 2>&1; echo "deep300 rc=$? in $(( $(date +%s) - t0 )) s" | tee -a $out/deep300.log; tail -8 $out/deep300.log
