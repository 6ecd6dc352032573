set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_j
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --warmup 5 > $R/gpurun_out/prof_j/bench_infer.json 2> $R/gpurun_out/prof_j/bench_infer.err
echo bench done >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_j/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $R/gpurun_out/prof_j/bench_under_rocprof.json 2>/dev/null
echo kt done >&2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_j/pf -o pf -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
echo pf done >&2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_j/pw -o pw -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
echo pw done >&2
find $R/gpurun_out/prof_j -name "*.csv" | head -20 >&2
