set -u
root="${GRAFT_REPO_ROOT:-/root/repo}"; out=$root/gpurun_out/r02u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
P="python tools/lbm_bench.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $P > $out/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $P > $out/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $P > $out/write.log 2>&1; echo "write rc=$?"
python - <<'PY' > $out/lbm_profile.md 2>&1
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r02u"
def rows(pat):
    f = glob.glob(out + pat)
    return list(csv.DictReader(open(f[0]))) if f else []
print("# rocprofv3 of `python tools/lbm_bench.py` (lattices 64x40x64 x2005 steps, 256x160x256 x205, 512x320x512 x65, 768x480x768 x25)\n")
print("## --kernel-trace --stats\n```")
for r in rows("/trace/*kernel_stats.csv")[:6]:
    print(r["Name"][:60], "calls", r["Calls"], "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 2), "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
print("```\n## per launch, by lattice (grid size identifies it): duration from the kernel trace, FETCH_SIZE / WRITE_SIZE from separate --pmc passes\n")
dur = collections.defaultdict(list)
for r in rows("/trace/*kernel_trace.csv"):
    if "k_lbm_step" in r["Kernel_Name"]: dur[int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pm = {}
for tag, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in rows(f"/{tag}/*counter_collection.csv"):
        if "k_lbm_step" in r["Kernel_Name"] and r["Counter_Name"] == cname: acc[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    pm[tag] = acc
print("| grid threads | launches | avg us | FETCH KiB/launch | WRITE KiB/launch | algorithmic read MB (80 B/cell) | algorithmic write MB (96 B/cell) |")
print("|---:|---:|---:|---:|---:|---:|---:|")
for g in sorted(dur):
    d = dur[g]; f = pm["fetch"].get(g, []); w = pm["write"].get(g, [])
    print(f"| {g} | {len(d)} | {sum(d)/len(d)/1e3:.1f} | {sum(f)/max(len(f),1):.0f} | {sum(w)/max(len(w),1):.0f} | | |")
PY
cat $out/lbm_profile.md | head -40
rm -rf $out/trace/*kernel_trace.csv $out/fetch $out/write
