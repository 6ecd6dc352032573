// Direct 3x3 / stride 1 / pad 1 convolution for Cin, Cout multiples of 64 (ResNet18 layer2-4 second convolutions and every
// BasicBlock.conv2 at 128 / 256 / 512 channels; torchvision BasicBlock behind backbone.py:66-71), f16x3 arithmetic, folded
// FrozenBN scale / bias, optional residual, ReLU (backbone.py:47-57).  conv3.hip's scheme (layer1, 64 -> 64 channels)
// generalised:
//
//   * the implicit GEMM (gemm.hip, A_CONV) re-reads every input pixel once per filter tap through the CU's vector-memory path
//     and moves 341 operand bytes per MFMA at its 128x128 tile -- the layer2-4 launches sat at 208-258 TF against 296 for
//     layer1's direct kernel.  Here a tile's input patch is staged ONCE per 64-channel chunk in LDS (already split into fp16
//     hi / lo pieces) and the 9 taps walk over it; only the 16 KB weight slice of the current (tap, chunk) streams through
//     LDS: ~230 operand bytes per MFMA.
//   * workgroup = 128 output pixels x 64 output channels; 256 threads = 4 waves, wave w owns a WR x WC pixel block
//     (WR * WC = 32; 1x32 for wide maps, 2x16 / 4x8 for the 40- and 20-pixel-wide layer3 / layer4 maps) and both 32-channel
//     halves of the 64 couts (two 32x32 MFMA tiles sharing one A fragment).  73 KB of LDS: two workgroups per CU.
//   * contraction = (Cin / 64 chunks) x (9 taps) x (64 channels): weight slices run WD steps ahead in registers across chunk
//     boundaries; the next chunk's patch is fetched and split into LDS between two chunks (prefetching it
//     under the last taps spilled registers: 52 more live beside the fragments; the CU's second workgroup covers the gap).
#include "common.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int CH = 64;                          // channels per chunk and couts per workgroup
constexpr int NTHR = 256;
constexpr int PIX = 272;                        // bytes per patch pixel: 64 hi halfs | 64 lo halfs | 16 B pad
constexpr int WROW = 272;                       // bytes per cout row of a weight slice
constexpr int WBUF = CH * WROW;                 // 17408

template <int WR, int WC> struct Geo {
    static constexpr int TR = 4 * WR, TW = WC;              // output tile: 4 waves stacked along rows
    static constexpr int PR = TR + 2, PW = TW + 2;
    static constexpr int NPIX = PR * PW;
    static constexpr int PATCH = NPIX * PIX;
    static constexpr int SMEM = PATCH + WBUF;
    static constexpr int NP = (NPIX * 16 + NTHR - 1) / NTHR;      // float4 groups of the patch per thread
};

template <int WR, int WC>
__global__ __launch_bounds__(NTHR, 2) void conv3x3_direct_kernel(Conv3gArgs p, int tiles_w, int tiles_h) {
    using G = Geo<WR, WC>;
    static_assert(WR * WC == 32, "a wave owns 32 output pixels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_patch = smem;
    unsigned char* s_w = smem + G::PATCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // blockIdx.x = ((image * tiles_h + th) * tiles_w + tw) * ncout + cout tile: the cout tiles of one pixel tile run side by
    // side and share the patch through L2
    const int ncout = p.Cout / CH;
    int tile = blockIdx.x;
    const int ct = tile % ncout; tile /= ncout;
    const int tw = tile % tiles_w; tile /= tiles_w;
    const int th = tile % tiles_h; tile /= tiles_h;
    const int64_t img = tile;                               // image index over groups x batch (camera-major)
    const int g = (int)(img / p.B);
    const int h0 = th * G::TR, w0 = tw * G::TW;
    const float* xin = p.x + img * (int64_t)p.H * p.W * p.Cin;
    const int64_t wrow = (int64_t)9 * p.Cin;               // floats per cout row of the split weight image
    const float* wsplit = p.w16 + ((int64_t)g * p.Cout + (int64_t)ct * CH) * wrow;
    const int nchunk = p.Cin / CH;
    const int nstep = 9 * nchunk;                           // (chunk, tap) steps

    // ---- weight slices: 64 cout x 16 groups of 16 bytes = 1024 groups, 4 per thread; WD steps ahead in registers
    constexpr int WD = 2;          // two slices ahead (a step is ~770 matrix-pipe cycles per wave); four spilled beside the patch prefetch
    constexpr int NWG = 1024 / NTHR;
    uint4 wreg[WD][NWG];
    auto fetch_w = [&](int step, uint4 (&wr)[NWG]) {
        const int chunk = step / 9, tap = step - chunk * 9;
#pragma unroll
        for (int i = 0; i < NWG; ++i) {
            const int e = t + NTHR * i;
            const int n = e >> 4, grp = e & 15;             // 4 channels per group
            wr[i] = *reinterpret_cast<const uint4*>(wsplit + (int64_t)n * wrow + (int64_t)tap * p.Cin + chunk * CH + grp * 4);
        }
    };
    auto commit_w = [&](const uint4 (&wr)[NWG]) {
#pragma unroll
        for (int i = 0; i < NWG; ++i) {
            const int e = t + NTHR * i;
            const int n = e >> 4, grp = e & 15;
            *reinterpret_cast<uint2*>(s_w + n * WROW + grp * 8) = uint2{wr[i].x, wr[i].y};             // 4 hi halfs
            *reinterpret_cast<uint2*>(s_w + n * WROW + 128 + grp * 8) = uint2{wr[i].z, wr[i].w};       // 4 lo halfs
        }
    };

    // ---- input patch of one 64-channel chunk: NPIX pixels x 16 float4 groups, zero outside the image
    f32x4 pv[G::NP];
    auto fetch_patch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < G::NP; ++i) {
            const int e = t + NTHR * i;
            const int grp = e & 15, pix = e >> 4;
            const int pr = pix / G::PW, pc = pix - pr * G::PW;
            const int hi = h0 - 1 + pr, wi = w0 - 1 + pc;
            const bool ok = e < G::NPIX * 16 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xin + (ok ? ((int64_t)hi * p.W + wi) * p.Cin + chunk * CH + grp * 4 : 0));
            pv[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit_patch = [&]() {
#pragma unroll
        for (int i = 0; i < G::NP; ++i) {
            const int e = t + NTHR * i;
            if (e < G::NPIX * 16) {
                const int grp = e & 15, pix = e >> 4;
                uint2 hv, lv;
                split16(pv[i], hv, lv);
                *reinterpret_cast<uint2*>(s_patch + pix * PIX + grp * 8) = hv;
                *reinterpret_cast<uint2*>(s_patch + pix * PIX + 128 + grp * 8) = lv;
            }
        }
    };

#pragma unroll
    for (int k = 0; k < WD; ++k)
        if (k < nstep) fetch_w(k, wreg[k]);
    fetch_patch(0);
    commit_patch();
    commit_w(wreg[0]);
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    // this lane's pixel inside the tile (A-fragment row li of the wave's WR x WC block)
    const int prow = wave * WR + li / WC, pcol = li % WC;

    for (int chunk = 0; chunk < nchunk; ++chunk) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = chunk * 9 + tap;
            // slot step % WD held this step's slice (now in LDS): refill it with step + WD.  9 is odd, so the slot
            // index is not a compile-time constant across chunks: select with a small switch on (step % WD)
            const int slot = step & (WD - 1);
            if (step + WD < nstep) {
                if (slot == 0) fetch_w(step + WD, wreg[0]);
                else fetch_w(step + WD, wreg[1]);
            }
            const int r = tap / 3, s = tap - r * 3;
            const unsigned char* ap = s_patch + ((prow + r) * G::PW + pcol + s) * PIX + lh * 16;
            const unsigned char* bp = s_w + li * WROW + lh * 16;
            // fragments two k-steps at a time (48 registers in flight instead of 96: the patch prefetch and four weight slices
            // are live too, and two workgroups per CU need <= 256 registers per lane)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                uint4 fa[2][2], fb[2][2][2];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int ks = kp * 2 + k2;
                    fa[k2][0] = *reinterpret_cast<const uint4*>(ap + ks * 32);
                    fa[k2][1] = *reinterpret_cast<const uint4*>(ap + 128 + ks * 32);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        fb[k2][nt][0] = *reinterpret_cast<const uint4*>(bp + nt * 32 * WROW + ks * 32);
                        fb[k2][nt][1] = *reinterpret_cast<const uint4*>(bp + nt * 32 * WROW + 128 + ks * 32);
                    }
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const h16x8 xh = __builtin_bit_cast(h16x8, fa[k2][0]), xl = __builtin_bit_cast(h16x8, fa[k2][1]);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const h16x8 yh = __builtin_bit_cast(h16x8, fb[k2][nt][0]), yl = __builtin_bit_cast(h16x8, fb[k2][nt][1]);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[nt], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                            // every wave is done with this step's slice (and, at tap 8, with the patch)
            if (step + 1 < nstep) {
                if (tap == 8) { fetch_patch(chunk + 1); commit_patch(); }   // next chunk's patch: the CU's other workgroup computes meanwhile
                if (((step + 1) & (WD - 1)) == 0) commit_w(wreg[0]);
                else commit_w(wreg[1]);
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  C layout: col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel li' of this wave's block).
    //      The wave's 32 px x 64 ch tile goes through the dead patch area and leaves as 16-byte accesses: 16 lanes cover one
    //      pixel's 64 channels (256 contiguous bytes), an instruction 4 pixels.
    constexpr int RS = CH + 4;                                    // scratch row stride in floats
    float* scr = reinterpret_cast<float*>(s_patch) + wave * (32 * RS);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * RS + nt * 32 + li] = acc[nt][e];
    const int c4 = lane & 15, pq = lane >> 4;                     // channel group of 4, pixel within a group of 4
    const float inv = 1.f / p.w_scale;
    const int cbase = g * p.Cout + ct * CH + c4 * 4;
    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.scale + cbase) * inv;
    const f32x4 bi4 = *reinterpret_cast<const f32x4*>(p.bias + cbase);
    const int64_t imgbase = img * (int64_t)p.H * p.W * p.Cout + ct * CH + c4 * 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int px = q * 4 + pq;                                // pixel index inside the wave's block
        const int ho = h0 + wave * WR + px / WC, wo = w0 + px % WC;
        if (ho >= p.H || wo >= p.W) continue;
        const int64_t o = imgbase + ((int64_t)ho * p.W + wo) * p.Cout;
        f32x4 v = *reinterpret_cast<const f32x4*>(scr + px * RS + c4 * 4) * sc4 + bi4;
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(p.out + o) = v;
    }
}

template <int WR, int WC>
int launch_geo(const Conv3gArgs& a, hipStream_t st, std::string* err) {
    using G = Geo<WR, WC>;
    const int tiles_w = (a.W + G::TW - 1) / G::TW, tiles_h = (a.H + G::TR - 1) / G::TR;
    const int64_t blocks = (int64_t)a.G * a.B * tiles_h * tiles_w * (a.Cout / CH);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_direct_kernel<WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::SMEM) != hipSuccess) {
            if (err) *err = "conv3x3_direct: cannot raise the dynamic LDS limit";
            return -3;
        }
        attr_set = true;
    }
    if (prof_enabled()) {
        char nm[128];
        static const bool by_shape = getenv("ACTMI_PROF_SHAPES") && getenv("ACTMI_PROF_SHAPES")[0] == '1';
        if (by_shape) snprintf(nm, sizeof(nm), "conv3x3_direct_f16x3_kernel<%d,%d>[H=%d,W=%d,Cin=%d,Cout=%d,wgs=%lld]", WR, WC, a.H, a.W, a.Cin, a.Cout, (long long)blocks);
        else snprintf(nm, sizeof(nm), "conv3x3_direct_f16x3_kernel");
        const double px = (double)a.G * a.B * a.H * a.W;
        prof_begin(nm, 2.0 * px * a.Cout * 9.0 * a.Cin, 4.0 * (px * (a.Cin + a.Cout * (a.res ? 2.0 : 1.0)) + (double)a.G * a.Cout * 9 * a.Cin), st);
    }
    hipLaunchKernelGGL((conv3x3_direct_kernel<WR, WC>), dim3((unsigned)blocks), dim3(NTHR), G::SMEM, st, a, tiles_w, tiles_h);
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("conv3x3_direct launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}

}  // namespace

int launch_conv3x3_direct(const Conv3gArgs& a, hipStream_t st, std::string* err) {
    if (a.H <= 0 || a.W <= 0 || a.B <= 0 || a.G <= 0) return 0;
    if ((a.Cin % CH) || (a.Cout % CH) || a.Cin < CH || a.Cout < CH) { if (err) *err = "conv3x3_direct: Cin and Cout must be multiples of 64"; return -2; }
    if (((uintptr_t)a.x & 15) || ((uintptr_t)a.w16 & 15) || ((uintptr_t)a.out & 15) || (a.res && ((uintptr_t)a.res & 15))) {
        if (err) *err = "conv3x3_direct: pointers must be 16-byte aligned";
        return -2;
    }
    if (!(a.w_scale > 0.f)) { if (err) *err = "conv3x3_direct: w_scale must be the (positive) scale of the split weight image"; return -2; }
    // wave block shape by map width: the fewest wasted pixel columns, then rows
    auto waste = [&](int tr, int tw) {
        const double th = (a.H + tr - 1) / tr, twn = (a.W + tw - 1) / tw;
        return th * tr * twn * tw / ((double)a.H * a.W);
    };
    const double w32 = waste(4, 32), w16 = waste(8, 16), w8 = waste(16, 8);
    if (w32 <= w16 && w32 <= w8) return launch_geo<1, 32>(a, st, err);
    if (w16 <= w8) return launch_geo<2, 16>(a, st, err);
    return launch_geo<4, 8>(a, st, err);
}
