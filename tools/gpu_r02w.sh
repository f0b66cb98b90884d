set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
export SMX_DIST_BACKEND=gloo
for n in 2; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus $n --steps 2 --warmup 1 > $out/weak$n.log 2>&1; echo "weak n=$n rc=$?"
tail -1 $out/weak$n.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], d['scaling'], 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), d['config']['workload'][:160], d['engine_stats'], d['work_dropped'])" 2>&1 | cut -c1-1200 || tail -5 $out/weak$n.log
done
