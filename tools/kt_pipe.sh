set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_pipe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --steps 30 --warmup 5 > $O/bench.json 2>/dev/null
