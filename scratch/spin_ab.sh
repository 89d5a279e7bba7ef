#!/bin/bash
for n in 20000 100000 1250000 10000000; do
  for g in 1 ""; do
    echo -n "n=$n spin=$g: "; RLR_SPIN=$g timeout -k 5 200 python scratch/step_jitter.py $n x 600 2>&1 | grep median | cut -c1-90
  done
done
