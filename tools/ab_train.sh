#!/bin/bash
# training-step A/B on one box: lines "name [ENV=..] [-- bench args]"; prints ms/step.  usage: tools/ab_train.sh <variants> <outdir>
V=$1; O=$2
mkdir -p $O
while read -r line; do
  [ -z "$line" ] && continue
  name=$(echo "$line" | awk '{print $1}')
  rest=$(echo "$line" | cut -d' ' -f2- -s)
  envs=$(echo "$rest" | sed 's/ -- .*//; s/^-- .*//')
  args=$(echo "$rest" | grep -o -- '-- .*' | sed 's/^-- //')
  env $envs timeout -k 10 200 python bench.py --mode train --steps 5 --warmup 2 $args > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; exit 1; }
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:24s} {d['ms_per_step']:8.2f} ms/step  {d['value']:7.1f} samples/s", flush=True)
for k in d['kernels'][:8]:
    print(f"      {k['name'][:52]:52s} x{k['launches_per_step']:5.1f} {k['avg_us']:8.1f} us  {k['share']*100:5.1f}%  {(k['tflops'] or 0):6.1f} TF")
PY
done < $V
