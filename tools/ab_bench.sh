#!/bin/bash
# A/B matrix of the headline bench on ONE box (boxes differ by +-3 %, runs on one box by ~0.5 %): every line of the variant
# file is "name [ENV=value ...] [-- extra bench args]".  usage: tools/ab_bench.sh <variants.txt> <outdir> [repeat]
# Prints name, policy steps/s, ms/step, with_h2d steps/s.  Joins runs with && semantics: stops at the first failing run.
V=$1; O=$2; R=${3:-1}
mkdir -p $O
for rep in $(seq 1 $R); do
while read -r line; do
  [ -z "$line" ] && continue
  name=$(echo "$line" | awk '{print $1}')
  rest=$(echo "$line" | cut -d' ' -f2- -s)
  envs=$(echo "$rest" | sed 's/ -- .*//; s/^-- .*//')
  args=$(echo "$rest" | grep -o -- '-- .*' | sed 's/^-- //')
  env $envs timeout -k 10 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --sustained-s 0 $args > $O/${name}_$rep.json 2> $O/${name}_$rep.err || { echo "$name FAILED"; tail -3 $O/${name}_$rep.err; exit 1; }
  python - "$O/${name}_$rep.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:28s} {d['value']:8.1f} steps/s {d['ms_per_step']:7.3f} ms  h2d {d['with_h2d']['value']:8.1f}  p50 {d['step_latency_ms']['p50']:.3f}", flush=True)
PY
done < $V
done
