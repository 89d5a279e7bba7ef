#!/bin/bash
# usage (via gpurun): bash scratch/prof_c2.sh [tag]  -- config 2 through the C ABI: per-call time, then the per-kernel averages under rocprofv3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-c2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/scratch/time_c2_abi.py > $O/${TAG}_time.log 2>&1 < /dev/null; tail -n 1 $O/${TAG}_time.log
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_c2_abi.py > $O/${TAG}_kt.log 2>&1 < /dev/null || { echo "profiled run failed"; exit 1; }
python3 - "$O/${TAG}_kt" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
tot = 0.0
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", r["Name"])
    if int(r["Calls"]) >= 200:
        tot += float(r["AverageNs"]) / 1000 * int(r["Calls"]) / 520
        print(f"  {(m.group(1) if m else r['Name'][:40]):34s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1000:8.1f}")
print("  kernel time per call, us:", round(tot, 1))
PY
