set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_b1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1/kt -o kt -- python3 $R/bench.py --batch 1 --no-cpu-baseline --steps 200 --warmup 20 > $R/gpurun_out/prof_b1/bench_b1_under_rocprof.json 2>/dev/null
echo kt done >&2
