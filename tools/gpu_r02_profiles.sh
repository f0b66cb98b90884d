#!/bin/bash
# Round-2 evidence in one GPU call: default bench line, rocprofv3 --kernel-trace --stats of the headline command, the
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ), the wet-regime timing. Outputs under gpurun_out/r02p; condensed into
# profiles/r02_* by tools/profile_summary_r02.py.
set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
t0=$(date +%s); note() { echo "[$(( $(date +%s) - t0 ))s] $*" | tee -a $out/status.txt; }
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; note "bench rc=$?"
P="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lines"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $P > $out/trace.log 2>&1; note "trace rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $P > $out/fetch.log 2>&1; note "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $P > $out/write.log 2>&1; note "write rc=$?"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/sq -o s -- $P > $out/sq.log 2>&1; note "sq rc=$?"
rm -f $out/*/*_agent_info.csv
# keep the merged directory small: the per-dispatch traces are large -> condense on the box
python tools/profile_summary_r02.py $out > $out/summary_tool.log 2>&1; note "summary rc=$?"
rm -rf $out/trace/*kernel_trace.csv $out/fetch $out/write $out/sq
timeout 300 python tools/wet_regime.py --size 256 --warm 100 --ticks 50 --engine batched --chunk 50 > $out/wet256_batched.log 2>&1; note "wet256 batched rc=$?"; tail -2 $out/wet256_batched.log | cut -c1-600
cat $out/status.txt; ls -la $out
