#!/bin/bash
# run1.sh <binary> [env...]: single-pool shapes only
cd $GRAFT_REPO_ROOT/scratch/gram_bench
b=$1; shift
echo "== $b $*"
env "$@" timeout -k 5 60 ./$b 1 308 768 0 100000 300 < /dev/null
env "$@" timeout -k 5 60 ./$b 1 301 770 0 100000 300 < /dev/null
env "$@" timeout -k 5 60 ./$b 1 308 1024 1 100000 300 < /dev/null
env "$@" timeout -k 5 60 ./$b 4 150 384 0 100000 300 < /dev/null
