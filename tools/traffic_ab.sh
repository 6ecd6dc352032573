# fabric traffic of the step's kernels for one environment setting: two PMC passes (FETCH_SIZE, WRITE_SIZE) -> <out>.json
# usage: bash tools/traffic_ab.sh <name> [ENV=value ...]
R=$GRAFT_REPO_ROOT
N=$1; shift
O=$R/gpurun_out/traffic_$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ACTMI_CAM_PIPE=0 "$@"
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --no-graph --steps 3 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o pf -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o pw -- $BENCH > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/traffic.json "$N: $*" "${ACTMI_COMMIT:-unknown}"
python3 - $O/traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["per_shape"][:14]:
    print(f"  {r['kernel'][:46]:46s} wgs {r['workgroups']:6d} x{r['launches']:4d}  fetch {r['fetch_bytes_per_launch']/1e6:8.1f} MB  write {r['write_bytes_per_launch']/1e6:8.1f} MB")
PY
