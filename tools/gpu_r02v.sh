set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
export SMX_DIST_BACKEND=gloo
for n in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline --no-lines > $out/strips$n.log 2>&1; echo "strips n=$n rc=$?"
tail -1 $out/strips$n.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), d['phase_ms_per_tick'], d['engine_stats'])" 2>&1 | cut -c1-900
done
