# SQ / cache counters of the B=8 step's kernels, eager single-branch launches (one gpurun call); outputs under gpurun_out/pmc_r03
# usage: bash tools/conv_pmc.sh [extra env assignments for the bench, e.g. ACTMI_FUSE_DS=0]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ACTMI_CAM_PIPE=0 "$@"
rocprofv3 -L > $O/counters.txt 2>&1
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras --sustained-s 0 --no-graph --steps 3 --warmup 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- $BENCH > $O/p$i.out 2> $O/p$i.err
  echo "pass $i rc $?" >&2
done
python3 $R/tools/pmc_sq.py $O > $O/summary.txt 2>&1
tail -60 $O/summary.txt >&2
