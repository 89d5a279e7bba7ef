#!/bin/bash
# usage (via gpurun): bash scratch/prof_c5.sh [tag] -- config 5's per-GPU share: the kernels of the steady-state pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-c5p}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_c5_shard.py --image > $O/${TAG}_kt.log 2>&1 < /dev/null || { echo "profiled run failed"; exit 1; }
tail -n 1 $O/${TAG}_kt.log | cut -c1-400
python3 - "$O/${TAG}_kt" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_rocclr_\w+)", r["Name"])
    print(f"  {(m.group(1) if m else r['Name'][:40]):44s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1000:9.1f}")
PY
