// Direct 3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (ResNet18 layer1: torchvision BasicBlock conv1 / conv2 at
// 120x160 per camera), f16x3 arithmetic, with the folded FrozenBN scale/bias, optional residual and ReLU in the epilogue
// (backbone.py:47-57; the block structure is torchvision's).
//
// Why not the implicit GEMM (gemm.hip) here: with only 64 output channels a K tile of the im2col GEMM moves as many
// operand bytes through L2 -> L1 as it feeds MFMAs (the 9 taps re-read every input pixel and every tile re-reads the
// 147 KB weight panel): ~9 TB/s of L2 traffic at 160 TF.  This kernel stages a tile's input patch ONCE in LDS (already split
// into fp16 hi / lo pieces) and walks the 9 taps over it, so each input byte is read from global memory once per tile
// (+ halo) and only the 16 KB weight slice of the current tap streams through LDS.
//
//   tile   4 output rows x 32 pixels x 64 channels; 256 threads = 4 waves, wave w owns output row w (32 px x 64 ch, two
//          32x32 MFMA tiles sharing one A fragment).  73 KB of LDS -> two workgroups per CU, so one's patch staging and
//          epilogue overlap the other's MFMAs (8-row tiles with double-buffered weights, one workgroup per CU: 8 % slower).
//   patch  6 x 34 pixels, per pixel 64 hi halfs | 64 lo halfs | 16 B pad (272 B: conflict-free 16-byte fragment reads).
//   taps   weights arrive pre-split and pre-scaled (the engine's split image of [cout][(r,s,c)], x 2^8): per tap a
//          [64 cout][64 c] slice in LDS (single buffer, two barriers per tap); slices are fetched four taps ahead into
//          registers.
//   K      per tap 4 steps of 16 channels (8 per lane half), 6 MFMAs (v_mfma_f32_32x32x16_f16: lo*hi, hi*lo, hi*hi) per
//          step and wave: 216 MFMAs per wave and tile.
//   out    the accumulator tile leaves through the dead patch area as 16-byte accesses (1 KB contiguous per instruction).
#include "common.h"
#include "split16.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int TR = 4, TW = 32, CH = 64;
constexpr int NTHR = TR * 64;
constexpr int PR = TR + 2, PW = TW + 2;
constexpr int PIX = 272;                        // bytes per patch pixel
constexpr int PATCH = PR * PW * PIX;            // 92480
constexpr int WROW = 272;                       // bytes per cout row of a tap slice
constexpr int WBUF = CH * WROW;                 // 17408
constexpr int SMEM = PATCH + WBUF;              // 72896: two workgroups per CU

__global__ __launch_bounds__(NTHR) void conv3x3_c64_f16x3_kernel(Conv3Args p, int tiles_w, int tiles_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_patch = smem;
    unsigned char* s_w = smem + PATCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (as in gemm.hip / attn.hip): consecutive block ids go round-robin over the 8 XCDs, so neighbouring
    // tiles -- which share their halo rows and columns -- used to sit in eight different L2s and every halo pixel crossed the
    // fabric once per neighbour (329 MB fetched per launch against 157 MB of input, PMC round 3); here each XCD takes a
    // contiguous run of tiles (whole strips of an image)
    int tile;
    {
        const int total = (int)gridDim.x, lin = (int)blockIdx.x;
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile = base + (lin >> 3);
    }
    const int tw = tile % tiles_w; tile /= tiles_w;
    const int th = tile % tiles_h; tile /= tiles_h;
    const int64_t img = tile;                              // image index over groups x batch (camera-major)
    const int g = (int)(img / p.B);
    const int h0 = th * TR, w0 = tw * TW;
    const float* xin = p.x + img * (int64_t)p.H * p.W * CH;
    const float* wsplit = p.w16 + (int64_t)g * CH * 9 * CH;       // [cout][(r,s,c)] split image

    // ---- tap slice loader: 64 cout x 16 groups of 16 bytes = 1024 groups, 2 per thread.  Slices run WD taps ahead in
    //      registers (a tap is only ~770 MFMA cycles per wave, a global round trip 2-4x that), one tap ahead in LDS.
    constexpr int WD = 4;
    constexpr int NWG = 1024 / NTHR;
    uint4 wreg[WD][NWG];
    auto fetch_w = [&](int tap, uint4 (&wr)[NWG]) {
#pragma unroll
        for (int i = 0; i < NWG; ++i) {
            const int e = t + NTHR * i;
            const int n = e >> 4, grp = e & 15;           // 4 channels per group
            wr[i] = *reinterpret_cast<const uint4*>(wsplit + ((int64_t)n * 9 + tap) * CH + grp * 4);
        }
    };
    auto commit_w = [&](int buf, const uint4 (&wr)[NWG]) {
        unsigned char* dst = s_w;      // single buffer
#pragma unroll
        for (int i = 0; i < NWG; ++i) {
            const int e = t + NTHR * i;
            const int n = e >> 4, grp = e & 15;
            *reinterpret_cast<uint2*>(dst + n * WROW + grp * 8) = uint2{wr[i].x, wr[i].y};             // 4 hi halfs
            *reinterpret_cast<uint2*>(dst + n * WROW + 128 + grp * 8) = uint2{wr[i].z, wr[i].w};       // 4 lo halfs
        }
    };
#pragma unroll
    for (int k = 0; k < WD; ++k) fetch_w(k, wreg[k]);

    // ---- input patch: 10 x 34 pixels x 16 float4 groups, zero outside the image; split on the way in.  All of a thread's
    //      loads are issued before the first is consumed (a rolled loop serialised ~11 global round trips per tile).
    constexpr int NP = (PR * PW * 16 + NTHR - 1) / NTHR;
    const float xs = p.x_scale_dev ? *p.x_scale_dev : 1.f;        // (gradient maps: operand scale of the data-gradient use)
    // running maximum read at the start (a lower bound that only filters the atomics at the end: common.h amax_commit)
    const unsigned amax_seen = p.amax_out ? __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    f32x4 pv[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = t + NTHR * i;
        const int grp = e & 15, pix = e >> 4;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int hi = h0 - 1 + pr, wi = w0 - 1 + pc;
        const bool ok = e < PR * PW * 16 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xin + (ok ? ((int64_t)hi * p.W + wi) * CH + grp * 4 : 0));
        pv[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = t + NTHR * i;
        if (e < PR * PW * 16) {
            const int grp = e & 15, pix = e >> 4;
            uint2 hv, lv;
            split16(pv[i] * xs, hv, lv);
            *reinterpret_cast<uint2*>(s_patch + pix * PIX + grp * 8) = hv;
            *reinterpret_cast<uint2*>(s_patch + pix * PIX + 128 + grp * 8) = lv;
        }
    }
    commit_w(0, wreg[0]);
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int buf = 0;
        // slot tap % WD held this tap's slice (now in LDS): refill it with tap + WD
        if (tap + WD < 9) fetch_w(tap + WD, wreg[tap % WD]);
        const int r = tap / 3, s = tap - r * 3;
        const unsigned char* ap = s_patch + ((wave + r) * PW + li + s) * PIX + lh * 16;
        const unsigned char* bp = s_w + li * WROW + lh * 16;
        // all fragments of the tap first (24 x 16-byte LDS reads in flight), then its 24 MFMAs
        uint4 fa[4][2], fb[4][2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa[ks][0] = *reinterpret_cast<const uint4*>(ap + ks * 32);
            fa[ks][1] = *reinterpret_cast<const uint4*>(ap + 128 + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                fb[ks][nt][0] = *reinterpret_cast<const uint4*>(bp + nt * 32 * WROW + ks * 32);
                fb[ks][nt][1] = *reinterpret_cast<const uint4*>(bp + nt * 32 * WROW + 128 + ks * 32);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8 xh = __builtin_bit_cast(h16x8, fa[ks][0]), xl = __builtin_bit_cast(h16x8, fa[ks][1]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const h16x8 yh = __builtin_bit_cast(h16x8, fb[ks][nt][0]), yl = __builtin_bit_cast(h16x8, fb[ks][nt][1]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();                                            // every wave is done with this tap's slice
        if (tap + 1 < 9) { commit_w(buf, wreg[(tap + 1) % WD]); __syncthreads(); }
    }

    // ---- epilogue.  C layout: col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel of this wave's row).
    //      Written straight from that layout a wave instruction touches 128-byte pieces of many pixels (measured: 250 us
    //      of a 380 us launch).  The wave's 32 px x 64 ch tile goes through the now dead patch area instead and leaves as
    //      16-byte accesses: 16 lanes cover one pixel's 64 channels, an instruction 4 consecutive pixels = 1 KB contiguous.
    const int ho = h0 + wave;
    constexpr int RS = CH + 4;                                    // scratch row stride in floats
    float* scr = reinterpret_cast<float*>(s_patch) + wave * (TW * RS);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * RS + nt * 32 + li] = acc[nt][e];
    if (ho >= p.H) return;
    const int c4 = lane & 15, pq = lane >> 4;                     // channel group of 4, pixel within a group of 4
    const float inv = 1.f / (p.w_scale * xs);
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sc4 = (p.scale ? *reinterpret_cast<const f32x4*>(p.scale + g * CH + c4 * 4) : one4) * inv;
    const f32x4 bi4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + g * CH + c4 * 4) : zero4;
    const f32x4 ps4 = p.post_scale ? *reinterpret_cast<const f32x4*>(p.post_scale + g * CH + c4 * 4) : one4;
    unsigned amx = 0;
    const int64_t rowbase = ((img * p.H + ho) * (int64_t)p.W) * CH + c4 * 4;
    f32x4 rv[8];
    if (p.res) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int wo = w0 + q * 4 + pq;
            rv[q] = *reinterpret_cast<const f32x4*>(p.res + rowbase + (int64_t)(wo < p.W ? wo : 0) * CH);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int px = q * 4 + pq, wo = w0 + px;
        f32x4 v = *reinterpret_cast<const f32x4*>(scr + px * RS + c4 * 4) * sc4 + bi4;
        if (p.res) v += rv[q];
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (wo < p.W) {
            if (p.mask) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + rowbase + (int64_t)wo * CH);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            }
            v = v * ps4;
            *reinterpret_cast<f32x4*>(p.out + rowbase + (int64_t)wo * CH) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) amx = max(amx, __float_as_uint(v[e]) & 0x7fffffffu);
        }
    }
    if (p.amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (lane == 0 && amx > amax_seen) atomicMax(p.amax_out, amx);
    }
}

}  // namespace

int launch_conv3x3_c64(const Conv3Args& a, hipStream_t st, std::string* err) {
    if (a.H <= 0 || a.W <= 0 || a.B <= 0 || a.G <= 0) return 0;
    if (((uintptr_t)a.x & 15) || ((uintptr_t)a.w16 & 15)) { if (err) *err = "conv3x3_c64: pointers must be 16-byte aligned"; return -2; }
    if (!(a.w_scale > 0.f)) { if (err) *err = "conv3x3_c64: w_scale must be the (positive) scale of the split weight image"; return -2; }
    const int tiles_w = (a.W + TW - 1) / TW, tiles_h = (a.H + TR - 1) / TR;
    const int64_t blocks = (int64_t)a.G * a.B * tiles_h * tiles_w;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_f16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            if (err) *err = "conv3x3_c64: cannot raise the dynamic LDS limit";
            return -3;
        }
        attr_set = true;
    }
    const double px = (double)a.G * a.B * a.H * a.W;
    prof_begin("conv3x3_c64_f16x3_kernel", 2.0 * px * CH * 9 * CH, 4.0 * (px * CH * (a.res ? 3.0 : 2.0) + (double)a.G * CH * 9 * CH), st);
    hipLaunchKernelGGL(conv3x3_c64_f16x3_kernel, dim3((unsigned)blocks), dim3(NTHR), SMEM, st, a, tiles_w, tiles_h);
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("conv3x3_c64 launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
