#!/bin/bash
# conv1 (f16x3) workgroup-count sweep: avg launch time at B = 1, 2, 8 (results under gpurun_out/c1_*.json)
for B in 1 2 8; do for N in 256 512 1024 2048; do
  ACTMI_CONV1_BLOCKS=$N python bench.py --batch $B --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null > gpurun_out/c1_${B}_${N}.json || exit 1
done; done
