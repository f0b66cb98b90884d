set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
python -c "import torch" > /dev/null 2>&1 &
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; note "smoke rc=$?"; tail -2 $out/smoke.log
wait
timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 > $out/fulltests.log 2>&1; note "fulltests rc=$?"; tail -30 $out/fulltests.log
( time timeout 600 python bench.py ) > $out/bench.log 2> $out/bench.err; note "bench rc=$?"; tail -3 $out/bench.err
tail -1 $out/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('HEAD', round(d['value']/1e6,2),'Msteps/s', round(d['ms_per_tick'],1),'ms/tick', d['phase_ms_per_tick'], d['digest'])
print('ROOF', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k not in ('phases','note')})
for k,v in d['roofline']['phases'].items(): print('  PH', k, v)
for l in d['lines']: print('LINE', l['name'], round(l['value']/1e6,2),'Msteps/s', round(l['ms_per_tick'],2),'ms/tick', {a:round(b,2) for a,b in l['phase_ms_per_tick'].items()}, l['digest']['nsec'], l['steps_per_tick']['nested_particles'])
c=d['cpu_baseline']; print('CPU', c and {k:c[k] for k in ('value','ms_per_tick','ms_water','ms_grid','ms_wind','host_cpu','wall_s')}, c and c.get('all_cores'))
" 2>&1 | tee $out/bench_summary.txt
cat $out/status.txt
