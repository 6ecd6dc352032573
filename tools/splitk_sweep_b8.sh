#!/bin/bash
# sweep of the forward split-K heuristic at the benchmark batch (B = 8): per-shape tables under gpurun_out/sk8_*.json
# triples: ACTMI_FWD_SPLITK_TARGET / _MINNK / _MAXTILES
mkdir -p gpurun_out
for cfg in "1536 12 768" "2432 12 1300" "4864 12 1300" "4800 12 2500" "2432 8 1300" "4800 8 2500"; do
  set -- $cfg
  ACTMI_FWD_SPLITK_TARGET=$1 ACTMI_FWD_SPLITK_MINNK=$2 ACTMI_FWD_SPLITK_MAXTILES=$3 \
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 --shapes \
      2> gpurun_out/sk8_$1_$2_$3.err > gpurun_out/sk8_$1_$2_$3.json || exit 1
  echo "done $cfg" >> gpurun_out/sk8_progress.log
done
