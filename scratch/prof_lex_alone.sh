#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-lexalone}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/scratch/lex_alone.py > $O/${TAG}.log 2>&1 < /dev/null || { echo failed; tail -5 $O/${TAG}.log; exit 1; }
tail -n 1 $O/${TAG}.log
python3 - "$O/${TAG}_kt" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", r["Name"])
    if int(r["Calls"]) >= 100:
        print(f"  {(m.group(1) if m else r['Name'][:40]):34s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1000:8.1f} min_us {float(r['MinNs'])/1000:8.1f} max_us {float(r['MaxNs'])/1000:8.1f}")
PY
rm -rf $O/${TAG}_kt
