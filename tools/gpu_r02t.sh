set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
for v in "0 0" "0 1" "1 0" "1 1"; do set -- $v; echo "NT=$1 XCD=$2" | tee -a $out/ab.log; SMX_LBM_NT=$1 SMX_LBM_XCD=$2 timeout 200 python tools/lbm_bench.py 2>&1 | grep -v RESULT | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('  ', d['lattice'], round(d['us_per_step'],1), 'us', round(d['GB/s']), 'GB/s', round(d['frac_of_8TBs'],3))
    except Exception: print(l.strip()[:200])" | tee -a $out/ab.log; done
timeout 300 python -m pytest tests/test_lbm.py -x -q -m gpu 2>&1 | tail -2
