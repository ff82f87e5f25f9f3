#!/bin/bash
# GPU box: one FheUint64 add and one mul (tools/latency_integer.py, the GPU default set g = 4) under
#   TFHE_HIP_PROFILE=1 rocprofv3 --marker-trace --kernel-trace --stats
# -> gpurun_out/<tag>_marker_*.txt: the radix layer's roctx ranges (csrc/profile.h) by name — count, total, mean — next to the kernels
tag=${1:-r06}; which=${2:-multibit_g4}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_marker
TFHE_HIP_PROFILE=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_marker -- python $R/tools/latency_integer.py $which > $R/gpurun_out/${tag}_marker.log 2>&1
cd $R
python - "$tag" "$which" <<'PY'
import csv, glob, sys, collections
tag, which = sys.argv[1], sys.argv[2]
out = open(f"gpurun_out/{tag}_marker_ranges_{which}.txt", "w")
files = glob.glob(f"gpurun_out/{tag}_marker/**/*marker_api_trace.csv", recursive=True)
print(f"# TFHE_HIP_PROFILE=1 rocprofv3 --marker-trace --kernel-trace --stats -- python tools/latency_integer.py {which}", file=out)
print(f"# roctx ranges of the radix layer (host-side: the interval in which the range's launches were ENQUEUED), 3 x (add, mul, sub, bitand, eq, gt, max, if_then_else) of one FheUint64", file=out)
agg = collections.OrderedDict()
for f in files:
    for row in csv.DictReader(open(f)):
        name = row.get("Function") or row.get("Name") or "?"
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += dur
print(f"{'range':70s} {'count':>7s} {'total ms':>10s} {'mean ms':>9s}", file=out)
for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} {n:7d} {tot:10.3f} {tot / n:9.4f}", file=out)
ks = glob.glob(f"gpurun_out/{tag}_marker/**/*kernel_stats.csv", recursive=True)
if ks:
    print("\n# kernels of the same run (rocprofv3 --stats)", file=out)
    for i, row in enumerate(csv.DictReader(open(ks[0]))):
        if i < 12:
            print(f"{row['Name'][:90]:90s} calls {row['Calls']:>6s} total_ms {int(row['TotalDurationNs']) / 1e6:10.3f} avg_us {float(row['AverageNs']) / 1e3:10.2f}", file=out)
out.close()
print(open(out.name).read())
PY
tail -3 gpurun_out/${tag}_marker.log
