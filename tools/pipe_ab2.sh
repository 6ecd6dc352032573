R=$GRAFT_REPO_ROOT
for cfg in "base" "ACTMI_FWD_SPLITK_LONG_NK=100000" "ACTMI_GEMM_CFG=M" "base" "ACTMI_FWD_SPLITK_LONG_NK=100000"; do
  for b in 8; do
    st=60
    if [ "$cfg" = "base" ]; then
      python3 $R/bench.py --batch $b --steps $st --warmup 10 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
    else
      env $cfg python3 $R/bench.py --batch $b --steps $st --warmup 10 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
    fi
    python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('$cfg', 'B=$b', round(d['ms_per_step'],4), round(d['value'],1))
"
  done
done
