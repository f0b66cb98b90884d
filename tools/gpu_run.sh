#!/bin/bash
# ONE parameterised GPU job script (replaces the per-run scripts of earlier rounds):
#   gpurun --timeout T -- 'tools/gpu_run.sh <tag> <budget-seconds> step [step ...]'
# Every step has its own timeout, writes gpurun_out/<tag>/<step>.{log,err} and leaves a line in status.txt; steps that no
# longer fit the budget are skipped and say so. Steps (most are thin wrappers around tools/*.py, tests/ and bench.py):
#   smoke quick parity gpusuite relaxtest stripstest headrelax* headbatch ktrace tracerelax bench2gloo wet wetrelax wetexact wetprof headprof bench benchdrv
#   benchfull drv drvk4 steady trace tracedrv pmc pmcwet p2ref p2multi p2var exact4096 envelope deep300 deep1000 deep10k shapes megasweep
set -u
tag=$1; budget=$2; shift 2
root="${GRAFT_REPO_ROOT:-/root/repo}"
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s)
left() { echo $(( budget - ($(date +%s) - t0) )); }
note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
run() {  # run <name> <need-seconds> <timeout> <cmd...>
  local name=$1 need=$2 to=$3; shift 3
  if [ "$(left)" -lt "$need" ]; then note "$name SKIPPED (left $(left)s < $need s)"; return 1; fi
  local lim=$to; [ "$(left)" -lt "$lim" ] && lim=$(left)
  timeout $lim "$@" > $out/$name.log 2> $out/$name.err; local rc=$?
  note "$name rc=$rc"
  tail -c 1500 $out/$name.log
  [ $rc -ne 0 ] && tail -c 600 $out/$name.err
  return $rc
}
for s in "$@"; do
  case $s in
    smoke)     run smoke 20 180 python -c "import __graft_entry__ as g; g.smoke()" ;;
    quick)     run quick 40 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ;;
    parity)    run parity 60 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_depth.py tests/test_host_compat.py -x -q -m gpu ;;
    gpusuite)  run gpusuite 120 1500 python -m pytest tests -x -q -m gpu --durations=15 ;;
    batchtest) run batchtest 60 600 python -m pytest tests/test_gpu_batched.py -x -q -m gpu --durations=5 ;;
    relaxtest) run relaxtest 60 600 python -m pytest tests/test_gpu_relaxed.py -x -q -m gpu --durations=8 ;;
    headrelax) run headrelax 30 300 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3 ;;
    headrelax8) run headrelax8 30 300 env SMX_RELAX_WIND_STEPS=8 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3 ;;
    headrelaxw0) run headrelaxw0 30 300 env SMX_RELAX_WIND_MIN=4000000000 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3 ;;
    drv)       run drv 60 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-lines --no-cpu-baseline ${BENCH_EXTRA:-}; cp $out/drv.log $out/drv.json 2>/dev/null ;;
    drvk4)     run drvk4 60 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-lines --no-cpu-baseline --water-generations ${WGEN:-1}; cp $out/drvk4.log $out/drvk4.json 2>/dev/null ;;
    gaps)      run gaps 30 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/gaps -o t -- python bench.py --engine ${ENGINE:-relaxed} --no-lines --no-cpu-baseline --steps 2 --warmup ${GWARM:-6}
               python tools/kernel_stats.py $out/gaps | tee $out/gaps.kernels
               python tools/trace_gaps.py $out/gaps --last-ticks ${LASTTICKS:-1} | tee $out/gaps.summary
               rm -rf $out/gaps ;;
    ktrace)    run ktrace 30 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -o t -- python bench.py --engine ${ENGINE:-relaxed} --no-lines --no-cpu-baseline --steps ${KSTEPS:-3} --warmup ${KWARM:-2}
               python tools/kernel_stats.py $out/ktrace | tee $out/ktrace.summary ;;
    ktracew)   run ktracew 30 300 env SMX_RELAX_WIND_MIN=${WMIN:-4096} SMX_RELAX_WIND_STEPS=${WSTEPS:-8} rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktracew -o t -- python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 3 --warmup 2
               python tools/kernel_stats.py $out/ktracew | tee $out/ktracew.summary ;;
    tracew)    run tracew 30 300 env SMX_TRACE_BATCH=1 SMX_RELAX_WIND_MIN=${WMIN:-4096} SMX_RELAX_WIND_STEPS=${WSTEPS:-8} python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 1 --warmup 3 ;;
    tracex)    run tracex 30 300 env SMX_TRACE_BATCH=1 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 1 --warmup 3 ;;
    headwaves) for w in ${WAVES_LIST:-512 1024}; do run headwaves_$w 30 300 env SMX_BATCH_WAVES=$w python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3; done ;;
    headwsteps) for w in ${WSTEPS_LIST:-1 2}; do run headwsteps_$w 30 300 env SMX_RELAX_WATER_STEPS=$w python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3; done ;;
    windtail)  i=0; IFS='|'; for w in ${WT_LIST}; do i=$((i+1)); IFS=' '; run windtail_$i 30 300 env SMX_BATCH_SCHED_TAIL_WIND="$w" python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 4 --warmup 3; IFS='|'; done; IFS=' ' ;;
    chunksweep) for kv in ${CS_LIST}; do run cs_$kv 30 300 env ${kv} python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 4 --warmup 3; done ;;
    shapes)    run shapes 30 240 python -m pytest tests/test_gpu_relaxed.py -x -q -m gpu -k "launch_shapes or (host_threads and rgps)" --durations=5 ;;
    megasweep) i=0; IFS='|'; for kv in ${MS_LIST}; do i=$((i+1)); IFS=' '; run megasweep_$i 30 200 env ${kv} python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 4 --warmup 3; echo "   [$kv]" | tee -a $out/status.txt; IFS='|'; done; IFS=' ' ;;
    headmega0) run headmega0 30 300 env SMX_RELAX_MEGA=0 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3 ;;
    headmegab) for b in ${MB_LIST:-512}; do run headmegab_$b 30 300 env SMX_RELAX_MEGA_BLOCKS=$b python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3; done ;;
    headmegac) for b in ${MC_LIST:-64}; do run headmegac_$b 30 300 env SMX_RELAX_MEGA_CHUNK=$b python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 5 --warmup 3; done ;;
    steady)    run steady 60 900 python bench.py --gpus 1 --steps ${SSTEPS:-20} --warmup ${SWARM:-100} --no-lines --no-cpu-baseline; cp $out/steady.log $out/steady.json 2>/dev/null ;;
    headbatch) run headbatch 30 300 python bench.py --engine batched --no-lines --no-cpu-baseline --steps 3 --warmup 2 ;;
    tracerelax) run tracerelax 30 300 env SMX_TRACE_BATCH=1 python bench.py --engine relaxed --no-lines --no-cpu-baseline --steps 1 --warmup 5 ;;
    stripstest) run stripstest 60 900 python -m pytest tests/test_gpu_strips.py -x -q -m gpu --durations=8 ;;
    bench2gloo) run bench2gloo 60 600 env SMX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --size ${BSIZE:-2048} ;;
    wet)       run wet 30 400 python tools/gpu_wet_exact.py --run ;;
    wetrelax)  run wetrelax 30 300 python tools/wet_regime.py --engine relaxed --warm 100 --ticks 50 --cpu 0 ;;
    wetexact)  run wetexact 60 500 python tools/wet_regime.py --engine spec --warm 100 --ticks 20 --cpu 0 ;;
    wetprof)   run wetprof 30 300 env SMX_FLOOD_PROF=1 python tools/wet_regime.py --engine relaxed --warm 100 --ticks 20 --cpu 0; grep -A 40 "soilmx" $out/wetprof.err | cut -c1-400 ;;
    headprof)  run headprof 60 400 env SMX_FLOOD_PROF=1 python bench.py --gpus 1 --steps ${HSTEPS:-5} --warmup 5 --no-lines --no-cpu-baseline ${BENCH_EXTRA:-}; grep -A 40 "soilmx" $out/headprof.err | cut -c1-400 ;;
    bench)     run bench 60 400 python bench.py; cp $out/bench.log $out/bench.json 2>/dev/null ;;
    benchdrv)  run benchdrv 120 900 python bench.py --gpus 1 --steps 20 --warmup 5; cp $out/benchdrv.log $out/benchdrv.json 2>/dev/null; cp $root/gpurun_out/bench_detail_n1.json $out/benchdrv_detail.json 2>/dev/null ;;
    benchfull) run benchfull 120 900 python bench.py --full; cp $out/benchfull.log $out/benchfull.json 2>/dev/null ;;
    trace)     run trace 60 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-lines
               python tools/kernel_stats.py $out/trace | tee $out/trace.kernels; python tools/trace_gaps.py $out/trace --last-ticks 5 | tee $out/trace.summary; rm -rf $out/trace ;;
    pmc)       PB="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-lines"
               run pmc_FETCH_SIZE 60 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_FETCH_SIZE -o p -- $PB
               run pmc_WRITE_SIZE 60 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_WRITE_SIZE -o p -- $PB
               run pmc_SQ 60 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $out/pmc_SQ -o p -- $PB
               python tools/make_pmc_json.py $out "4096/64000/16000/relaxed" "$PB" | tee $out/pmc.summary
               rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ ;;   # (raw per-dispatch CSVs: > 64 MiB, more than gpurun brings back)
    pmcwet)    i=0; for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
                                  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_BRANCH"; do
                 i=$((i+1))
                 run pmcwet$i 40 300 env SMX_WET_N=5 rocprofv3 --pmc $set --output-format csv -d $out/pmcwet$i -o p -- python tools/gpu_wet_exact.py --one ${ENGINE:-serial}
                 python tools/pmc_by_kernel.py $out/pmcwet$i | tee $out/pmcwet$i.summary
               done ;;
    exacttrace) run exacttrace 60 600 env SMX_TRACE_ROUNDS=1 python bench.py --only exact_area_scaled; grep 'round' $out/exacttrace.err | head -400 > $out/exacttrace.rounds; grep -c round $out/exacttrace.err; rm -f $out/exacttrace.err ;;
    exactk)    run exactk 60 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/exactk -o t -- python bench.py --only exact_area_scaled
               python tools/kernel_stats.py $out/exactk | tee $out/exactk.kernels; python tools/trace_gaps.py $out/exactk | tee $out/exactk.summary; rm -rf $out/exactk ;;
    exactmode2) run exactmode2 60 600 env SMX_SPEC_MODE=2 python bench.py --only exact_area_scaled ;;
    p2ref)     run p2ref 60 ${PTO:-900} python tools/p2_reference.py --size ${PSIZE:-4096} --ticks ${PTICKS:-10} --nctl ${PNCTL:-8} --engine ${PENGINE:-relaxed} --water-batches ${PWB:-0} --out $out/p2ref_${PSIZE:-4096}.json; tail -12 $out/p2ref.err ;;
    p2multi)   run p2multi 60 ${PTO:-1200} python tools/p2_reference.py --size ${PSIZE:-4096} --ticks ${PTICKS:-10} --nctl ${PNCTL:-32} --engine ${PENGINE:-relaxed} --variants ${PVARS:-k4,k8} --out $out/p2multi_${PSIZE:-4096}.json; grep -v '^ *"' $out/p2multi.err | tail -120 ;;
    tracedrv)  run tracedrv 60 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tracedrv -o t -- python bench.py --gpus 1 --steps ${TSTEPS:-20} --warmup ${TWARM:-5} --no-cpu-baseline --no-lines ${BENCH_EXTRA:-}
               python tools/kernel_stats.py $out/tracedrv | tee $out/tracedrv.kernels; python tools/trace_gaps.py $out/tracedrv --last-ticks ${TSTEPS:-20} | tee $out/tracedrv.summary; rm -rf $out/tracedrv ;;
    p2var)     i=0; for kv in ${PV_LIST}; do i=$((i+1)); run p2var_$i 60 ${PTO:-600} env ${kv} python tools/p2_reference.py --size ${PSIZE:-2048} --ticks ${PTICKS:-10} --nctl ${PNCTL:-16} --engine relaxed --out $out/p2var_$i.json; tail -9 $out/p2var_$i.err; done ;;
    exact4096) run exact4096 60 600 python bench.py --only exact_area_scaled ;;
    envelope)  run envelope 60 900 python tools/p2_envelope.py --size ${ESIZE:-1024} --ticks ${ETICKS:-20} --streams ${ESTREAMS:-4} --counts ${ECOUNTS:-area} ;;
    envelope4096) run envelope4096 60 600 python tools/p2_envelope.py --size 4096 --ticks ${ETICKS:-20} --streams ${ESTREAMS:-4} --counts file ;;
    deep300)   run deep300 60 900 env SMX_DEEP=1 python -m pytest tests/test_gpu_depth.py -x -q -m gpu -k "300" ;;
    deep1000)  run deep1000 120 2400 env SMX_DEEP=1 python -m pytest tests/test_gpu_depth.py -x -q -m gpu -k "1000" ;;
    deep10k)   run deep10k 120 ${DEEPTO:-1500} env SMX_DEEP10K=1 python -m pytest tests/test_gpu_depth.py -x -q -m gpu -k ten_thousand -s ;;
    *)         note "unknown step $s" ;;
  esac
done
note "done, $(left)s of the budget left"
