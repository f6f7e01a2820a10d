// K9g — the 256 -> N 1x1 convolutions of the fp32-class (split-plane) pseudo-label forward with register-resident
// weights: conv3 of every layer3 bottleneck, 256 -> 1024 + BN(eval) + identity + ReLU (reference: Bottleneck.forward,
// sseg/models/modules/resnet.py:91-98, in the eval forward of IASPseudoGenerator.run, workflows/
// pseudo_label_generator.py:190-192).  gfx950 only.
//
// This launch (B = 8: 34.4 GFLOP x 3 split products against 605 MB) is the largest launch group of the step.  On the tile
// kernel (igemm_kernel.h) it runs as lockstep phases on all CUs — 8 k-steps of matrix work with the HBM idle, then an
// epilogue in which every CU reads 256 KiB of residual and writes 256 KiB at the same time — and reaches 3.8-3.9 TB/s.
// Here the structure of xconv.hip (K9e) is carried over to split planes:
//   * the WEIGHTS live in registers: a wave owns 32 output columns = one 128-byte output slab (32 hi | 32 lo), i.e.
//     32 x 256 x (hi, lo) bf16 = 128 VGPRs per lane in MFMA fragment layout; a block 8 x 32 = 256 columns;
//   * the block is PERSISTENT over 32-row panels of X (32 x 1 KiB = 32 KiB per panel: row = 8 slabs of [32 hi | 32 lo]),
//     streamed through three LDS stages by LDS-DMA, two panels in flight per CU;
//   * D^T = W * X^T on v_mfma_f32_16x16x32_bf16 as hi*hi + lo*hi + hi*lo (the dropped lo*lo is 2^-16 relative): a lane
//     ends up with 8 output channels of ONE pixel, so BN, residual (requested one panel ahead), ReLU, the re-split and the
//     two 16-byte stores (hi, lo) come straight from registers — no LDS round trip, no barrier, no block-wide phase:
//     one wave's epilogue runs under the other waves' MFMAs and the DMA of the next panels.
// The N / 256 blocks that walk the same panels sit in neighbouring slots of ONE XCD: re-reads of a panel are L2 hits.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 x2_bf16x8;
typedef __attribute__((address_space(3))) void* x2_lds_ptr;

constexpr int X2_PANEL = 32;            // rows of X per panel
constexpr int X2_STAGES = 3;
constexpr int X2_COLS = 256;            // output columns per block (8 waves x 32)
constexpr int X2_K = 256;               // reduction length (channels)
constexpr int X2_ROWB = X2_K * 4;       // bytes per activation row (hi | lo planes)
constexpr int X2_STAGE = X2_PANEL * X2_ROWB;      // 32 KiB

__device__ __forceinline__ void x2_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (x2_lds_ptr)lds, 16, voff, soff, 0, 0);
}

// LDS image of one slab tile [rows][128 B]: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7) (as igemm / xconv)
__device__ __forceinline__ int x2_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// fragment reads as inline asm + explicit lgkmcnt wait (the compiler orders every LDS load it can see behind ALL pending
// LDS-DMA: s_waitcnt vmcnt(0), which would serialise the two-panel prefetch)
template <int OFF>
__device__ __forceinline__ x2_bf16x8 x2_lds_read(unsigned addr)
{
    x2_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ void x2_lds_wait(x2_bf16x8& a, x2_bf16x8& b, x2_bf16x8& c, x2_bf16x8& d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __forceinline__ void x2_split(float v, unsigned short& h, unsigned short& l)
{
    const __hip_bfloat16 hb = __float2bfloat16(v);
    h = __bfloat16_as_ushort(hb);
    l = __bfloat16_as_ushort(__float2bfloat16(v - __bfloat162float(hb)));
}

// Y[m][n] = act( (sum_k X[m][k] W[n][k]) * scale_n + shift_n (+ R[m][n]) ), everything in split planes.
template <bool RES, bool RELU>
__global__ __launch_bounds__(512) void xconv2_kernel(const unsigned short* __restrict__ X,
                                                     const unsigned short* __restrict__ Wp,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     float eps, const unsigned short* __restrict__ R,
                                                     unsigned short* __restrict__ Y, int M, int N)
{
    constexpr int KS = X2_K / 32;                        // 32-deep MFMA steps = slabs per row (8)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[X2_STAGES * X2_STAGE];
    __shared__ float s_sc[X2_COLS], s_sh[X2_COLS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, px = lane & 15;
    // block -> (column group, panel stream): consecutive block ids go round the 8 XCDs, so the NG blocks that walk the
    // same panels take neighbouring slots of ONE XCD
    const int NG = N / X2_COLS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % NG;
    const int nstream = (int)gridDim.x / NG;
    const int stream = (slot / NG) * 8 + xcd;
    const int n0 = cg * X2_COLS + wave * 32;             // this wave's first output column (one 32-channel slab)
    const int npanel = (M + X2_PANEL - 1) / X2_PANEL;

    // ---- weights -> registers, A-operand layout of v_mfma_f32_16x16x32_bf16 (lane: row px, k = 8 g .. 8 g + 7 of a
    // 32-deep step).  Row px of n-tile b (b = 0, 1) is output channel n0 + (px >> 2) * 8 + b * 4 + (px & 3): lane group g
    // of a pixel then holds channels g * 8 .. g * 8 + 7 (tile 0: + 0..3, tile 1: + 4..7) — one 16-byte piece of the slab.
    // Packed weight row (hiast_pack_conv_weight, split planes): [slab s][32 hi | 32 lo].
    x2_bf16x8 wh[2][KS], wl[2][KS];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned short* wrow = Wp + (size_t)(n0 + (px >> 2) * 8 + b * 4 + (px & 3)) * (2 * X2_K) + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wh[b][s] = *reinterpret_cast<const x2_bf16x8*>(wrow + s * 64);
            wl[b][s] = *reinterpret_cast<const x2_bf16x8*>(wrow + s * 64 + 32);
        }
    }
    // pin the fragments down HERE (first use inside the panel loop would put the compiler's wait for these loads — an
    // s_waitcnt vmcnt(0) — behind every DMA issue)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wh[b][s]), "+v"(wl[b][s]));
    for (int c = tid; c < X2_COLS; c += 512) {
        const int n = cg * X2_COLS + c;
        const float sc = (gamma ? gamma[n] : 1.0f) * (1.0f / sqrtf(var[n] + eps));
        s_sc[c] = sc;
        s_sh[c] = fmaf(-mean[n], sc, beta ? beta[n] : 0.0f);
    }

    // ---- DMA: a panel = 32 rows x 8 slabs = 4 row groups x 8 slabs of (8 rows x 128 B); wave w moves row group w & 3 of
    // the slabs 4 (w >> 2) .. + 3 (four wave-instructions); lane l: row l >> 3, physical chunk l & 7 of the swizzled image
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)M * X2_ROWB), 0x00020000);
    const int rgrp = wave & 3, j0 = (wave >> 2) * 4;
    const int drow = rgrp * 8 + (lane >> 3);
    const int dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    auto issue = [&](int p, int st) {
        const int m = p * X2_PANEL + drow;
        const int voff = (p < npanel && m < M) ? (int)((size_t)m * X2_ROWB) + dchunk : OOB;
        unsigned char* base = smem + st * X2_STAGE + rgrp * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) x2_dma16(xrs, base + (j0 + j) * (X2_PANEL * 128), voff, (j0 + j) * 128);
    };
    const unsigned lds_base = (unsigned)(size_t)smem;

    // residual rows of a panel: pixel (a, px), channels n0 + g * 8 .. + 7 as hi (16 B) and lo (16 B, + 64 B in the slab);
    // requested ONE PANEL AHEAD (ring of two): a panel is only ~0.7 us of matrix work, less than an HBM round trip
    const int nslab = N >> 5;
    auto slab_off = [&](int m) { return ((size_t)(m < M ? m : 0) * nslab + (n0 >> 5)) * 64 + g * 8; };
#ifndef X2_RR
#define X2_RR 2           // depth of the residual register ring: rows are requested X2_RR - 1 panels ahead (-DX2_RR=3, round 5:
                          // 249 VGPRs, 154.5-157.7 against 155.6-161.5 us stand-alone — the request distance is not the limiter)
#endif
    uint4 rh[X2_RR][2], rl[X2_RR][2];
    auto load_res = [&](int p, int buf) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int m = (p < npanel ? p : 0) * X2_PANEL + a * 16 + px;
            const size_t o = slab_off(m);
            rh[buf][a] = h_load16_once(R + o);
            rl[buf][a] = h_load16_once(R + o + 32);
        }
    };

    // panels of this stream: stream, stream + nstream, ... (one contiguous row range per stream measured no better:
    // 164 vs 160 us, profiles/r03_ab_xconv2.txt)
    const int pfirst = stream, pstep = nstream;
    const int plast = npanel;
    auto valid = [&](int p) { return p < plast; };
    issue(valid(pfirst) ? pfirst : npanel, 0);
    issue(valid(pfirst + pstep) ? pfirst + pstep : npanel, 1);
    if (RES) {
#pragma unroll
        for (int r = 0; r + 1 < X2_RR; ++r) load_res(valid(pfirst + r * pstep) ? pfirst + r * pstep : 0, r);
    }
    __syncthreads();                                     // s_sc / s_sh

    // One panel.  CUR (compile time: the residual ring must not be indexed at run time, or the compiler moves it out of
    // the register file) = ring entry holding this panel's residual rows; the next panel's go to CUR ^ 1.
    auto panel = [&](int p, int it, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        const int st = it % X2_STAGES;
        // This wave's share of panel p has landed once everything it issued BEFORE the previous panel's iteration has
        // retired (vector-memory operations of a wave retire in order): younger are that iteration's residual request (4),
        // its DMA of panel p + 1 (4) and its stores (4) — the residual request stays in flight until this panel's epilogue
        // needs it.  (First iteration: only the DMA of the second panel and the first residual request are younger.)
        // No scratch traffic may hide in this count (the Makefile fails the build on spills).
        // A BARE s_barrier: behind __syncthreads() the compiler emits `s_waitcnt vmcnt(4)` here (it wants every LDS-DMA it
        // knows of landed), i.e. the DMA of panel p + 1 would have to land before panel p is touched.
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 + (RES ? 4 * (X2_RR - 1) : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(8 + (RES ? 4 : 0)) : "memory");     // everyone's has landed;
                                                                               // everyone left the stage of panel p - 1
        if (RES)          // residual of the panel X2_RR - 1 ahead on this stream
            load_res(valid(p + (X2_RR - 1) * pstep) ? p + (X2_RR - 1) * pstep : 0, (CUR + X2_RR - 1) % X2_RR);
        issue(valid(p + 2 * pstep) ? p + 2 * pstep : npanel, (it + 2) % X2_STAGES);

        h_f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = (h_f32x4){0.f, 0.f, 0.f, 0.f};
        // Fragments of a step are read at its start (no ring across steps: the 16 registers it takes spill in the residual
        // variant, and scratch traffic would break the counted waits).  The launch is HBM-bound with the matrix pipes
        // < 50 % busy; while one wave of a SIMD waits ~100 cycles for its four reads the other one issues MFMAs.
        // Addresses: everything but the swizzled chunk is an instruction immediate — step s = + s * 4 KiB, row tile 1 =
        // + 2 KiB (16 rows: the swizzle (row >> 1) & 7 is the same for px and px + 16), lo plane = chunk + 4 = address XOR 64.
        const unsigned fa = lds_base + (unsigned)(st * X2_STAGE + x2_lds_off(px, g));
        const unsigned fb = fa ^ 64u;
        // ring of two fragment sets: the reads of step S + 1 are in flight under the 12 MFMAs of step S
        x2_bf16x8 xh[2][2], xl[2][2];                    // [ring][row tile]
#define X2_READ(S, R)                                                                                               \
        xh[R][0] = x2_lds_read<(S) * 4096>(fa);        xl[R][0] = x2_lds_read<(S) * 4096>(fb);                      \
        xh[R][1] = x2_lds_read<(S) * 4096 + 2048>(fa); xl[R][1] = x2_lds_read<(S) * 4096 + 2048>(fb);
#define X2_STEP(S, R)                                                                                               \
        x2_lds_wait(xh[R][0], xl[R][0], xh[R][1], xl[R][1]);                                                        \
        if ((S) + 1 < 8) { X2_READ(((S) + 1) & 7, (R) ^ 1) }                                                         \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                               \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) {          /* lo*hi + hi*lo + hi*hi */                    \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b][S], xh[R][a], acc[a][b], 0, 0, 0);        \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][S], xl[R][a], acc[a][b], 0, 0, 0);        \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][S], xh[R][a], acc[a][b], 0, 0, 0);        \
            }
        X2_READ(0, 0)
        X2_STEP(0, 0) X2_STEP(1, 1) X2_STEP(2, 0) X2_STEP(3, 1) X2_STEP(4, 0) X2_STEP(5, 1) X2_STEP(6, 0) X2_STEP(7, 1)
#undef X2_STEP
#undef X2_READ
        static_assert(KS == 8, "eight 32-deep steps");
        // ---- epilogue: lane = pixel (a, px), channels n0 + g * 8 .. + 7
        float sc[8], sh[8];
        {
            const int c0 = wave * 32 + g * 8;
            const float4 s0 = *reinterpret_cast<const float4*>(&s_sc[c0]), s1 = *reinterpret_cast<const float4*>(&s_sc[c0 + 4]);
            const float4 t0 = *reinterpret_cast<const float4*>(&s_sh[c0]), t1 = *reinterpret_cast<const float4*>(&s_sh[c0 + 4]);
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int m = p * X2_PANEL + a * 16 + px;
            float o[8];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[b * 4 + r] = fmaf(acc[a][b][r], sc[b * 4 + r], sh[b * 4 + r]);
            if (RES) {
                const uint4 vh = rh[CUR][a], vl = rl[CUR][a];
                const unsigned wh4[4] = {vh.x, vh.y, vh.z, vh.w}, wl4[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {            // residual = hi + lo (added in that order, as the tile kernel does)
                    o[2 * q] += __uint_as_float(wh4[q] << 16);
                    o[2 * q + 1] += __uint_as_float(wh4[q] & 0xFFFF0000u);
                    o[2 * q] += __uint_as_float(wl4[q] << 16);
                    o[2 * q + 1] += __uint_as_float(wl4[q] & 0xFFFF0000u);
                }
            }
            if (RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
            }
            unsigned ph[4], pl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned short h0, l0, h1, l1;
                x2_split(o[2 * q], h0, l0);
                x2_split(o[2 * q + 1], h1, l1);
                ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
                pl[q] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
            // (rows beyond M are not stored.  The counted wait at the top of a panel assumes four stores per panel; a
            // ragged tail only occurs in the LAST panel, after which no block takes another counted wait)
            if (m < M) {
                unsigned short* y = Y + slab_off(m);
                h_store16_out(y, ph[0], ph[1], ph[2], ph[3]);        // (plain stores: common.h)
                h_store16_out(y + 32, pl[0], pl[1], pl[2], pl[3]);
            }
        }
    };
    int it = 0;
#if X2_RR == 2
    for (int p = pfirst; valid(p); p += 2 * pstep, it += 2) {            // two panels per trip: ring entries 0, 1
        panel(p, it, std::integral_constant<int, 0>());
        if (valid(p + pstep)) panel(p + pstep, it + 1, std::integral_constant<int, 1>());
    }
#else
    for (int p = pfirst; valid(p); p += 3 * pstep, it += 3) {            // three panels per trip: ring entries 0, 1, 2
        panel(p, it, std::integral_constant<int, 0>());
        if (valid(p + pstep)) panel(p + pstep, it + 1, std::integral_constant<int, 1>());
        if (valid(p + 2 * pstep)) panel(p + 2 * pstep, it + 2, std::integral_constant<int, 2>());
    }
#endif
}

static int x2_blocks(int64_t M, int N)
{
    const int NG = N / X2_COLS;
    const long long npanel = (M + X2_PANEL - 1) / X2_PANEL;
    static const int env_cus = [] { const char* e = getenv("HIAST_XCONV_CUS"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= 256 ? v : 0; }();
    const int cus = env_cus ? env_cus : hiast_grid_cus();   // one block per CU (HIAST_XCONV_CUS: experiment, part of the chip)
    long long streams = cus / NG / 8 * 8;           // the (slot, xcd) numbering wants whole rounds of the 8 XCDs
    if (streams < 8) streams = 8;
    if (streams > npanel) streams = (npanel + 7) / 8 * 8;
    return (int)(streams * NG);
}

}  // namespace hiast

// shapes this kernel takes over from the tile kernel: split planes, 1x1, K = 256, BatchNorm(eval) in the epilogue
int hiast_xconv2_ok(int64_t M, int K, int N, int planes, int taps, int out_f32, int has_bn, int has_res, int relu,
                    int has_gate, int has_stats)
{
    const char* env = getenv("HIAST_XCONV2");        // HIAST_XCONV2=0: A/B switch back to the tile kernel
    if ((env && atoi(env) == 0) || planes != 2 || taps != 1 || out_f32 || has_gate || has_stats) return 0;
    if (K != hiast::X2_K || N % hiast::X2_COLS != 0 || N > 2048 || !has_bn) return 0;
    if (has_res && !relu) return 0;                  // (no caller in the trunk)
    if (M < 4096) return 0;                          // small maps: the tile kernel's grid fills the chip better
    return 1;
}

int hiast_xconv2_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                        const float* var, float eps, const void* res, int relu, void* y, int64_t M, int N, hipStream_t st)
{
    using namespace hiast;
    const dim3 grid((unsigned)x2_blocks(M, N));
#define X2L(RESF, RELUF)                                                                                          \
    hipLaunchKernelGGL((xconv2_kernel<RESF, RELUF>), grid, dim3(512), 0, st, (const unsigned short*)x,             \
                       (const unsigned short*)wp, gamma, beta, mean, var, eps, (const unsigned short*)res,         \
                       (unsigned short*)y, (int)M, N)
    if (res) X2L(true, true);
    else if (relu) X2L(false, true);
    else X2L(false, false);
#undef X2L
    HIAST_CHECK_LAUNCH();
    return 0;
}
