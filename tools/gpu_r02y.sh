set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s)
timeout 900 python -m pytest "tests/test_gpu_batched.py::test_256_default_1000_ticks_in_the_wet_regime_equal_restated_schedule" -x -q --durations=2 > $out/deep_batched.log 2>&1; echo "deep batched rc=$? in $(( $(date +%s) - t0 )) s" | tee -a $out/deep_batched.log; tail -6 $out/deep_batched.log
export SMX_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 4 --steps 1 --warmup 1 > $out/weak4.log 2>&1; echo "weak n=4 rc=$?"
tail -1 $out/weak4.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], d['scaling'], 'Msteps/s', round(d['value']/1e6,1), 'ms/tick', round(d['ms_per_tick'],1), d['config']['workload'][:120], d['engine_stats']['strip_stats'], d['work_dropped'])" 2>&1 | cut -c1-900 || tail -5 $out/weak4.log
