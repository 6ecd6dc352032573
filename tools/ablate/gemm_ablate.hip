// Ablation of the fp32-MFMA GEMM main loop on gfx950: which part of an iteration costs MFMA issue slots?
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_ablate gemm_ablate.hip ; run: ./gemm_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, NPL = 8, PSA = BM + 1, PSB = BN + 1, STAGE = NPL * (PSA + PSB);

template <int V>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int lda) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wrow0 = (wave >> 1) * 64, wcol0 = (wave & 1) * 64;
    const int cidx = t & 7, srow = t >> 3;
    for (int i = t; i < 2 * STAGE; i += 256) smem[i] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 ra[4], rb[4];
    const float* ap = A + ((size_t)blockIdx.x * 128 + srow) * lda + cidx * 4;
    const float* bp = B + ((size_t)(blockIdx.x % 32) * 128 + srow) * lda + cidx * 4;
    for (int i = 0; i < 4; ++i) { ra[i] = f32x4{1, 1, 1, 1}; rb[i] = ra[i]; }
    f32x4 af[2], bf[2];
    af[0] = smem[lh * PSA + wrow0 + li]; af[1] = smem[lh * PSA + wrow0 + 32 + li];
    bf[0] = smem[NPL * PSA + lh * PSB + wcol0 + li]; bf[1] = smem[NPL * PSA + lh * PSB + wcol0 + 32 + li];
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        const f32x4* sa = smem + cur * STAGE;
        const f32x4* sb = sa + NPL * PSA;
        if (V >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)32 * i * lda + it * 32);
                                          rb[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)32 * i * lda + it * 32); }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (V >= 1) {
                af[0] = sa[(2 * kb + lh) * PSA + wrow0 + li]; af[1] = sa[(2 * kb + lh) * PSA + wrow0 + 32 + li];
                bf[0] = sb[(2 * kb + lh) * PSB + wcol0 + li]; bf[1] = sb[(2 * kb + lh) * PSB + wcol0 + 32 + li];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (V >= 3) {
            f32x4* wa = smem + (cur ^ 1) * STAGE;
            f32x4* wb = wa + NPL * PSA;
#pragma unroll
            for (int i = 0; i < 4; ++i) { wa[cidx * PSA + srow + 32 * i] = ra[i]; wb[cidx * PSB + srow + 32 * i] = rb[i]; }
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + t] = s;
}

template <int V>
void run(const char* name, const float* A, const float* B, float* out, int blocks, int iters, int lda) {
    const int smem = 2 * STAGE * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), smem, 0, A, B, out, iters, lda);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), smem, 0, A, B, out, iters, lda);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = (double)blocks * iters * 128.0 * 128 * 32 * 2;
    printf("%-44s blocks=%4d  %8.1f us  %7.1f TF\n", name, blocks, ms * 1e3, flop / ms / 1e9);
}

int main() {
    const int lda = 4096, iters = 128;
    float *A, *B, *out;
    const size_t a_bytes = (size_t)512 * 128 * lda * sizeof(float);      // V4 reads rows [0, blocks*128) x lda, blocks <= 512
    const size_t b_bytes = (size_t)32 * 128 * lda * sizeof(float);
    if (hipMalloc(&A, a_bytes) != hipSuccess || hipMalloc(&B, b_bytes) != hipSuccess ||
        hipMalloc(&out, (size_t)1024 * 256 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 0, a_bytes); hipMemset(B, 0, b_bytes);
    for (int blocks : {256, 512, 1024}) {
        run<0>("V0 mfma only", A, B, out, blocks, iters, lda);
        run<1>("V1 + lds fragment reads", A, B, out, blocks, iters, lda);
        run<2>("V2 + barrier", A, B, out, blocks, iters, lda);
        run<3>("V3 + lds stage writes", A, B, out, blocks, iters, lda);
        if (blocks <= 512) run<4>("V4 + global loads (L2/MALL resident)", A, B, out, blocks, iters, lda);
    }
    return 0;
}
