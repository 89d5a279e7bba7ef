#!/bin/bash
# usage (via gpurun): bash scratch/ab_c5_sync.sh "4 2 1"  -- config 5's share: main-pass time and HBM fetch per sibling-meeting interval
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ev in ${1:-4 2 1}; do
  export RLR_GEMM8_SYNC_EVERY=$ev
  if [ -n "$2" ]; then export $2; fi
  echo "== RLR_GEMM8_SYNC_EVERY=$ev $2"
  timeout -k 10 200 python3 $R/scratch/time_c5_shard.py --image 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  search_ms %.2f gemm_ms %.2f other_ms %.2f TFLOPs %.0f same=%s' % (d['batched_search_top308_ms'], d['gemm_ms'], d['other_ms'], d['gemm_TFLOPs'], d['batched_equals_single_path']))"
  rm -rf $O/c5ab_fetch; timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5ab_fetch -- python3 $R/scratch/time_c5_shard.py --image > $O/c5ab_fetch.log 2>&1 < /dev/null
  python3 - $O/c5ab_fetch <<'PY'
import csv, glob, sys, collections
v = []
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm8_kernel<false" in r["Kernel_Name"] or "gemm8_kernelILb0" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]))
v.sort()
# FETCH_SIZE counts 32-byte units x ... : use the repo's convention (profiles/summarize.py): KiB, x2 gfx950 correction
big = [x for x in v if x > 0.5 * v[-1]]
print("  gemm8 main launches %d  FETCH_SIZE median %.4g KiB -> %.2f GB corrected x2 = %.2f x the 12.5 GB image" % (len(big), big[len(big)//2], big[len(big)//2] * 1024 * 2 / 1e9, big[len(big)//2] * 1024 * 2 / 12.5e9))
PY
done
rm -rf $O/c5ab_fetch
