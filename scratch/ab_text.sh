#!/bin/bash
# usage (via gpurun): bash scratch/ab_text.sh -- the text search at the C ABI (bench.py's with_query_text leg) under the
# round-4 switches: accumulator form of the BM25 kernel, adaptive cosine fetch
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "default" "RLR_LEX_TERMS=global" "RLR_HYBRID_FETCH=full" "RLR_LEX_TERMS=global RLR_HYBRID_FETCH=full"; do
  if [ "$v" = default ]; then e=""; else e="$v"; fi
  echo -n "$v: "; env $e timeout -k 10 120 python3 $R/scratch/time_c2_text_abi.py 2>&1 | tail -n 1
done; done
