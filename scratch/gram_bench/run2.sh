#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/gram_bench
b=$1; shift
echo "== $b $*"
for P in 200 450 600 800 1024; do env "$@" timeout -k 5 60 ./$b 1 $P 768 0 100000 100 < /dev/null; done
env "$@" timeout -k 5 60 ./$b 2 308 768 0 100000 100 < /dev/null
env "$@" timeout -k 5 60 ./$b 7 308 768 0 100000 100 < /dev/null
