#!/bin/bash
# Collects everything profiles/ is built from, on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      e.g.  r01
# 1. rocprofv3 --kernel-trace --stats of the default bench   2. three separate --pmc passes (HBM fetch, HBM write, SQ)
# 3. the bench line itself, with the CPU baseline.            Outputs under gpurun_out/<tag>/ ; tools/profile_summary.py
# condenses them into profiles/<tag>_*.
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
P="python bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $P > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $P > $out/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
          --kernel-trace --output-format csv -d $out/sq -o s -- $P > $out/sq.log 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
ls -R $out | head -40
tail -c 600 $out/bench.json
