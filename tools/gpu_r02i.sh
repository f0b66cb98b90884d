set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 900 python -m pytest tests/test_gpu_strips.py -x -q --durations=6 > $out/tests.log 2>&1; note "strips gpu tests rc=$?"; tail -25 $out/tests.log
# the --gpus 2 path of bench.py end to end, two ranks sharing this one GPU over gloo (RCCL needs one device per rank)
SMX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --size 1024 --nwater 4000 --nwind 1000 > $out/bench2.log 2> $out/bench2.err; note "bench 2 ranks (gloo, one GPU) rc=$?"
tail -1 $out/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 STRIPS', round(d['value']/1e6,2),'Msteps/s', round(d['ms_per_tick'],1),'ms/tick', d['config']['parallelism'], d['scaling'], d['engine_stats'])" 2>&1 || tail -5 $out/bench2.err
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lines --size 1024 --nwater 4000 --nwind 1000 > $out/bench1.log 2>&1; note "bench 1 rank rc=$?"
tail -1 $out/bench1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 STRIP ', round(d['value']/1e6,2),'Msteps/s', round(d['ms_per_tick'],1),'ms/tick', d['engine_stats'])"
cat $out/status.txt
