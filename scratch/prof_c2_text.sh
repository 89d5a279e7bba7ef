#!/bin/bash
# usage (via gpurun): bash scratch/prof_c2_text.sh [tag] -- config 2 with the query text (GPU BM25 + blend + MMR): per-call time, then kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-c2t}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/scratch/time_c2_hybrid.py hybrid-only > $O/${TAG}_time.log 2>&1 < /dev/null; tail -n 1 $O/${TAG}_time.log
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/time_c2_hybrid.py hybrid-only > $O/${TAG}_kt.log 2>&1 < /dev/null || { echo "profiled run failed"; exit 1; }
python3 - "$O/${TAG}_kt" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", r["Name"])
    if int(r["Calls"]) >= 200:
        print(f"  {(m.group(1) if m else r['Name'][:40]):34s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1000:8.1f} min_us {float(r['MinNs'])/1000:8.1f} max_us {float(r['MaxNs'])/1000:8.1f}")
PY
