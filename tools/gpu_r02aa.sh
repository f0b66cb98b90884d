set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02aa; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
P="python bench.py --size 16384 --steps 1 --warmup 1 --no-cpu-baseline --no-lines"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $P > $out/trace.log 2>&1; echo "trace rc=$?"
grep '^{' $out/trace.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'Msteps/s', round(d['ms_per_tick'],1), 'ms/tick', d['phase_ms_per_tick'], d['roofline']['phases']['particles_wind'], d['roofline']['phases']['particles_water'])"
python - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r02aa"
f = glob.glob(out + "/trace/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats_16384.csv", "w") as o:
    o.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for r in rows[:14]:
        n = r["Name"].split("(")[0].replace("void ", "")
        o.write(f"{n},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},{r['Percentage']},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f}\n")
print(open(out + "/kernel_stats_16384.csv").read())
PY
rm -rf $out/trace
