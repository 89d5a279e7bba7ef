#!/bin/bash
# run.sh <binary> [env assignments...]: C2 single f32, an odd shape single / 16 pools, and the C5 batch over a 12.8 GB corpus
cd $GRAFT_REPO_ROOT/scratch/gram_bench
b=$1; shift
echo "== $b $*"
env "$@" timeout -k 5 60 ./$b 1 308 768 0 100000 300 < /dev/null
env "$@" timeout -k 5 60 ./$b 1 301 770 0 100000 300 < /dev/null
env "$@" timeout -k 5 60 ./$b 16 301 770 0 100000 50 < /dev/null
env "$@" timeout -k 5 120 ./$b 1024 300 1024 1 6250000 10 < /dev/null
