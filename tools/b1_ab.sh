# B=1 step under a few environment settings
R=$GRAFT_REPO_ROOT
for cfg in "base" "ACTMI_FWD_SPLITK=0" "ACTMI_FWD_SPLITK=0 ACTMI_GEMM_CFG=S" "ACTMI_GEMM_CFG=S" "ACTMI_GEMM_CFG=M" "$@"; do
  if [ "$cfg" = "base" ]; then
    python3 $R/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
  else
    env $cfg python3 $R/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null > /tmp/o.json
  fi
  python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('$cfg', round(d['ms_per_step'],4))
"
done
