#!/bin/bash
# gpurun --timeout 600 -- 'bash scratch/trace_stall.sh'   HIP API trace of 400 single-query steps: which runtime call stalls ~call 206?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/scratch/step_jitter.py 1250000 noprof 400 > $O/stall_plain.log 2>&1 < /dev/null
cat $O/stall_plain.log | tail -3
timeout -k 10 280 rocprofv3 --hip-trace --output-format csv -d $O/stall_hip -- python3 $R/scratch/step_jitter.py 1250000 noprof 400 > $O/stall_hip.log 2>&1 < /dev/null
tail -3 $O/stall_hip.log
F=$(find $O/stall_hip -name '*hip_api_trace.csv' 2>/dev/null | head -1)
[ -z "$F" ] && { echo "no hip trace"; ls -R $O/stall_hip | head; exit 1; }
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("calls", len(rows), "columns", list(rows[0].keys()))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])) for r in rows]
t0 = min(d[2] for d in dur)
big = sorted(dur, reverse=True)[:25]
for d, f, s in big:
    print(f"{d/1e6:9.3f} ms  {f:40s} at {(s - t0)/1e6:10.1f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for d, f, s in dur:
    agg[f][0] += 1; agg[f][1] += d
for f, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{f:40s} n={n:7d} total {t/1e6:9.1f} ms  mean {t/n/1e3:8.1f} us")
PY
