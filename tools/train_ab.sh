# training step B=64 under a few environment settings (one process each), ms per step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_ab
mkdir -p $O
for cfg in "base" "ACTMI_ATTN_BWD_TILE=1" "ACTMI_ATTN_BWD_TILE=2" "$@"; do
  name=$(echo $cfg | tr '= ' '__')
  if [ "$cfg" = "base" ]; then
    python3 $R/bench.py --mode train --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/$name.json 2>/dev/null
  else
    env $cfg python3 $R/bench.py --mode train --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/$name.json 2>/dev/null
  fi
  python3 -c "
import json,sys
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
print('$cfg', round(d['ms_per_step'],2))
for k in d.get('kernels',[]):
    if '64,64,32,32,0,0' in k['name'] or '128,128,64,64,0,0' in k['name'] or '128,64,64,32,0,0' in k['name'] or '2,0>' in k['name']: print('   ',k['name'],k['launches_per_step'],round(k['avg_us'],1))
"
done
