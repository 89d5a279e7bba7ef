#!/bin/bash
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "default" "RLR_LEX_READY_BY_LAUNCH=1"; do
  if [ "$v" = default ]; then e=""; else e="$v"; fi
  echo -n "$v: "; env $e timeout -k 10 120 python3 $R/scratch/time_c2_text_abi.py 2>&1 | tail -n 1
done; done
