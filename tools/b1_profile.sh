# B=1 query: per-kernel table (library profiler, per-shape classes) + rocprofv3 kernel stats of the graph-replayed run
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/b1_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --batch 1 --shapes --no-extras --no-cpu-baseline --sustained-s 0 --steps 200 --warmup 20 > $O/bench_b1_shapes.json 2> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --batch 1 --no-cpu-baseline --no-extras --sustained-s 0 --steps 200 --warmup 20 > $O/bench_b1_under_rocprof.json 2>/dev/null
