set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02f; mkdir -p $out
cd "$root"
bash tools/gpu_r02_profiles.sh > $out/profiles.log 2>&1; tail -12 $out/profiles.log
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $out/tests.log 2>&1; echo "full gpu suite rc=$? in $(( $(date +%s) - t0 )) s"; tail -22 $out/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
