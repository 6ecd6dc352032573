#!/bin/bash
# sweep of the forward split-K heuristic (results under gpurun_out/sk_*.json)
for B in 1 2 4 8; do
for T in 1024 1536 2048; do for N in 8 12 16; do for X in 768 1280 2048; do
  ACTMI_FWD_SPLITK_TARGET=$T ACTMI_FWD_SPLITK_MINNK=$N ACTMI_FWD_SPLITK_MAXTILES=$X \
    python bench.py --batch $B --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/sk_${B}_${T}_${N}_${X}.json || exit 1
  echo "B=$B $T $N $X" >> gpurun_out/sk_progress.log
done; done; done
done
