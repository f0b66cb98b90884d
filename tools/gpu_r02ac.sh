set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02ac; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout 400 python -m pytest tests/test_gpu_strips.py -x -q > $out/tests.log 2>&1; echo "strips gpu tests rc=$?"; tail -3 $out/tests.log
export SMX_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 2 --steps 2 --warmup 1 > $out/weak2.log 2>&1; echo "weak n=2 rc=$?"
tail -1 $out/weak2.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], d['scaling'], 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), d['engine_stats']['epochs'], d['engine_stats']['strip_stats'], d['work_dropped'])" 2>&1 | cut -c1-600 || tail -5 $out/weak2.log
