#!/bin/bash
# usage (via gpurun): bash scratch/prof_greedy_ab.sh  -- greedy-MMR kernel time at P = 300 for k = 100 and k = 2 (everything in front of
# the chain + one pick): the lazy tie-break kernel and the register-position kernel before it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in lazy reg; do
  for k in 100 2; do
    if [ $mode = reg ]; then export RLR_MMR_GREEDY=reg; else unset RLR_MMR_GREEDY; fi
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/greedy_${mode}_$k -- python3 $R/scratch/time_greedy.py 300 $k > $O/greedy_${mode}_$k.log 2>&1 < /dev/null || { echo "failed $mode $k"; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for mode in ("lazy", "reg"):
    for k in (100, 2):
        f = sorted(glob.glob(f"{O}/greedy_{mode}_{k}/*/*kernel_stats.csv"))[-1]
        for r in csv.DictReader(open(f)):
            if "greedy" in r["Name"]:
                print(mode, k, r["Calls"], "avg_us", float(r["AverageNs"]) / 1000, "min_us", float(r["MinNs"]) / 1000)
PY
