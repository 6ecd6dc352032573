# bench line + rocprofv3 kernel stats + the two PMC traffic passes of one build (one gpurun call); outputs under gpurun_out/prof_r03
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_infer.json 2> $O/bench_infer.err
echo bench done >&2
ACTMI_CAM_PIPE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --steps 30 --warmup 5 > $O/bench_under_rocprof.json 2>/dev/null
echo kt done >&2
# the default launch structure (trunk as two concurrent camera branches): per-kernel durations overlap here
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pipe -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --steps 30 --warmup 5 > $O/bench_under_rocprof_pipe.json 2>/dev/null
echo kt_pipe done >&2
ACTMI_CAM_PIPE=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o pf -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
echo pf done >&2
ACTMI_CAM_PIPE=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o pw -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
echo pw done >&2
python3 $R/bench.py --shapes --no-extras --no-cpu-baseline --sustained-s 0 --steps 30 --warmup 5 > $O/bench_shapes.json 2>/dev/null
python3 $R/tools/pmc_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/traffic.json "round 3: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of ACTMI_CAM_PIPE=0 bench.py --no-graph --steps 3 --warmup 1 (tools/prof_round.sh); FETCH_SIZE x2 per the gfx950 correction" "${ACTMI_COMMIT:-unknown}" >&2
find $O -name "*.csv" | head -20 >&2
