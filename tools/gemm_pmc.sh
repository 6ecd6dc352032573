# SQ counters of the f16x3 GEMM on the ACT shapes (two passes of 8 SQ slots); run through gpurun
R=$GRAFT_REPO_ROOT
export ACTMI_GEMM_PREC=f16x3 GEMM_BENCH_SPLITW=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/gpmc1 -o p -- python3 $R/tools/gemm_bench.py 8 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/gpmc2 -o p -- python3 $R/tools/gemm_bench.py 8 > /dev/null 2>&1
ls $R/gpurun_out/gpmc1 $R/gpurun_out/gpmc2
