# training step B=64: rocprofv3 kernel stats
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --mode train --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_train.json 2> $O/err.txt
