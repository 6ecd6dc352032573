R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "base" "ACTMI_LN_SPLIT_SHORT=2" "ACTMI_LN_SPLIT=1"; do
  for b in 8 1; do
    st=150; [ $b = 1 ] && st=300
    if [ "$cfg" = "base" ]; then
      python3 $R/bench.py --batch $b --steps $st --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
    else
      env $cfg python3 $R/bench.py --batch $b --steps $st --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
    fi
    python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('$cfg', 'B=$b', round(d['ms_per_step'],4), round(d['value'],1))
"
  done
done
done
