#!/bin/bash
# usage (via gpurun): bash scratch/prof_gram_dims.sh -- the single-pool Gram kernel (300 rows) at several row lengths: slope = cost per k-step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for d in 128 384 768 1536; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gramdim_$d -- python3 $R/scratch/time_greedy.py 300 2 $d > $O/gramdim_$d.log 2>&1 < /dev/null || { echo "failed $d"; exit 1; }
done
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for d in (128, 384, 768, 1536):
    f = sorted(glob.glob(f"{O}/gramdim_{d}/*/*kernel_stats.csv"))[-1]
    for r in csv.DictReader(open(f)):
        if "gram" in r["Name"] or "greedy" in r["Name"]:
            print(d, r["Name"][38:80], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1000, 2), "min_us", float(r["MinNs"]) / 1000)
PY
