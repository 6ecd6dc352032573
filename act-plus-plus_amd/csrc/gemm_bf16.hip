// bf16 instantiations of the GEMM / implicit-GEMM convolution family (gemm.hip): the single-product "speed mode" of the training
// step (ACTMI_PREC_BF16; BASELINE config 3 as written).  A translation unit of its own so that it compiles beside gemm.hip.
#define ACTMI_GEMM_TU_BF16 1
#include "gemm.hip"
