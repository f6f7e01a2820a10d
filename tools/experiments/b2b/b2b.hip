// K9m — the tail of a layer3 bottleneck in the two INFERENCE forwards as ONE kernel:
//     y = relu( bn3( conv3_1x1( relu( bn2( conv2_3x3(x) ) ) ) ) + identity )
// (reference: Bottleneck.forward, sseg/models/modules/resnet.py:84-98, in the eval forward of IASPseudoGenerator.run,
//  workflows/pseudo_label_generator.py:190-192, and of the EMA teacher, trainer/consistency_self_training_trainer.py:92-126.)
// gfx950 only.
//
// Until round 4 these were two launches (igemm_kernel.h 3x3 -> xconv2.hip / xconv.hip 1x1) with the 256-channel
// activation a2 = relu(bn2(conv2(x))) written to HBM and read back (2 x 67 MB in split planes, 2 x 34 MB in 16 bits per
// layer at B = 8) and ~19 us of launch + prologue + epilogue per launch that nothing overlaps.  Here a2 never leaves the
// register file:
//   phase 1  the 3x3 implicit GEMM of igemm_kernel.h (same LDS-DMA tiles, same k order, same counted waits) with the
//            operand ROLES swapped: the weights are the MFMA A operand, the pixels the B operand, and a wave owns
//            32 pixels x ALL 256 output channels (8 waves = a 256-pixel tile).  D^T = W2 * X^T leaves a lane with
//            channels {4g .. 4g + 3} of a 16-channel tile for ONE pixel (g = lane >> 4).  The weight tile is DMA'd into LDS
//            in a permuted row order (bb_chan) so that two neighbouring tiles give the lane 8 CONSECUTIVE channels
//            32 s + 8 g .. + 7 —
//   switch   — which is exactly the B-operand fragment (k = 8 g .. 8 g + 7 of the 32-deep step s) of the 1x1 that follows:
//            BN2 + ReLU + (re-split | encode) turn the 128 accumulator registers into the 128 (64) operand registers of
//            phase 2 in place.  No LDS round trip, no exchange between waves: every wave holds the whole reduction
//            dimension of its pixels.
//   phase 2  y^T = W3 * a2^T: W3 streams through LDS in stages of 32 output channels (one 128-byte output slab per pixel;
//            LDS-DMA ring of three, the operand tiles of phase 1 are dead by then), every wave multiplies each stage with
//            its register-resident a2 and finishes its own 32 pixels x 32 channels straight from the accumulators —
//            BN3, identity (requested a stage ahead), ReLU, re-split, two 16-byte stores per pixel — under the other
//            waves' MFMAs (the structure of xconv2.hip with the operand roles swapped).
// Arithmetic: the same products in the same order as the two-launch form (phase 1: slab outer / tap inner,
// lo*hi + hi*lo + hi*hi per k-step; phase 2: ascending k-steps), BN / residual / ReLU / split as there.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include <type_traits>

#include "igemm_kernel.h"

namespace hiast {

constexpr int BB_K = 256;        // channels of the 3x3 (in = out) = reduction length of the 1x1
constexpr int BB_BM = 256;       // pixels per block (8 waves x 32)

// LDS row q of a weight tile holds output channel bb_chan(q): within a 32-row group, MFMA row i of 16-row tile b
// (q = 16 b + i) is channel 8 (i >> 2) + 4 b + (i & 3) — the accumulators of tiles b = 0, 1 then give lane group g = lane >> 4
// the channels 8 g .. 8 g + 3 and 8 g + 4 .. 8 g + 7 of a pixel
__host__ __device__ __forceinline__ int bb_chan(int q) { return (q & ~31) + (((q & 15) >> 2) << 3) + (((q >> 4) & 1) << 2) + (q & 3); }

__device__ __forceinline__ void bb_read_tile(ig_bf16x8 (&f)[2], unsigned addr, int ct)       // ct is a constant after unrolling
{
#define BB_C(I) case I: f[0] = ig_lds_read<(I) * 2048>(addr); f[1] = ig_lds_read<(I) * 2048>(addr ^ 64u); break;
    switch (ct) {
        BB_C(1) BB_C(2) BB_C(3) BB_C(4) BB_C(5) BB_C(6) BB_C(7) BB_C(8) BB_C(9) BB_C(10) BB_C(11) BB_C(12) BB_C(13) BB_C(14) BB_C(15)
    default: f[0] = ig_lds_read<0>(addr); f[1] = ig_lds_read<0>(addr ^ 64u); break;
    }
#undef BB_C
}

// BatchNorm tables live in the operand buffer: read them with inline asm like the fragments (a C++ LDS load would make the
// compiler drain every pending LDS-DMA first)
template <int OFF>
__device__ __forceinline__ ig_f32x4 bb_lds_f4(unsigned addr)
{
    ig_f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// raw buffer accesses for the identity rows and the output: a row beyond M gets an out-of-range offset (loads return 0,
// stores are dropped) — every wave then issues the SAME number of vector-memory operations per stage whatever its rows,
// which the counted s_waitcnt vmcnt of the stage loop relies on
typedef unsigned int bb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bb_u32x4 bb_buf_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
__device__ __forceinline__ void bb_buf_store(__amdgpu_buffer_rsrc_t rs, bb_u32x4 v, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
}

// PL = 2: split bf16 planes (fp32-class, the pseudo-label forward); PL = 1: plain bf16 / fp16 (F16) rows (the teacher forward).
// X [M = B*H*W][PL*256] channels-last, W2p packed [256][9][PL*256], W3p packed [N3][1][PL*256], R / Y [M][PL*N3].
template <int PL, bool F16>
__global__ __launch_bounds__(512) void b2b_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ W2p, const float* __restrict__ g2,
    const float* __restrict__ b2, const float* __restrict__ m2, const float* __restrict__ v2, float eps2,
    const unsigned short* __restrict__ W3p, const float* __restrict__ g3, const float* __restrict__ b3,
    const float* __restrict__ m3, const float* __restrict__ v3, float eps3, const unsigned short* __restrict__ R,
    unsigned short* __restrict__ Y, int M, int N3, IGeo geo)
{
    static_assert(!F16 || PL == 1, "fp16 rows are a one-plane format");
    using HT = H16<F16>;
    constexpr int KS = BB_K * PL / 64;                  // 128-byte slabs per row (8 | 4) = k-steps per tap
    constexpr int NK = 9 * KS;
    constexpr int A_BYTES = BB_BM * 128, B_BYTES = BB_K * 128;
    constexpr int NSA = 3, NSB = 2;                     // LDS stages of phase 1 (igemm_kernel.h): 160 KiB
    constexpr int LDS_BYTES = NSA * A_BYTES + NSB * B_BYTES;
    // phase 2 (same buffer): three W3 stages of 32 rows x KS slabs, then the BatchNorm tables
    constexpr int W3_STAGE = 32 * KS * 128;             // 32 KiB | 16 KiB
    constexpr int NS3 = 3;
    constexpr int OFF_BN3 = NS3 * W3_STAGE;             // float sc3[N3], sh3[N3]  (N3 <= 1024: 8 KiB)
    constexpr int OFF_BN2 = OFF_BN3 + 2 * 1024 * 4;     // float sc2[256], sh2[256]
    static_assert(OFF_BN2 + 2 * 256 * 4 <= LDS_BYTES, "phase-2 image must fit");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bm;
    {   // XCD k takes the k-th contiguous eighth of the pixel tiles (neighbouring tiles share their halo rows in one L2)
        const int total = gridDim.x;
        bm = blockIdx.x;
        if ((total & 7) == 0) bm = (bm & 7) * (total >> 3) + (bm >> 3);
    }
    const int m0 = bm * BB_BM;

    // ---- BatchNorm(eval) scale / shift: thread t prepares channel t of bn2 and channels t, t + 512 of bn3 now (the loads
    // retire under the first DMA) and parks them in six registers until the operand tiles are dead
    float sc2r = 0.f, sh2r = 0.f, sc3r[2] = {0.f, 0.f}, sh3r[2] = {0.f, 0.f};
    if (tid < BB_K) {
        sc2r = (g2 ? g2[tid] : 1.0f) * (1.0f / sqrtf(v2[tid] + eps2));
        sh2r = fmaf(-m2[tid], sc2r, b2 ? b2[tid] : 0.0f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = tid + 512 * i;
        if (n < N3) {
            sc3r[i] = (g3 ? g3[n] : 1.0f) * (1.0f / sqrtf(v3[n] + eps3));
            sh3r[i] = fmaf(-m3[n], sc3r[i], b3 ? b3[n] : 0.0f);
        }
    }
    asm volatile("" : "+v"(sc2r), "+v"(sh2r), "+v"(sc3r[0]), "+v"(sc3r[1]), "+v"(sh3r[0]), "+v"(sh3r[1]));

    // ================================ phase 1: a2^T = W2 * X^T (3x3, dilated, padding = dilation) ========================
    const int srow = lane >> 3;
    constexpr int OOB = (int)0x80000000;
    const size_t in_pix = (size_t)(M / (geo.H * geo.W)) * geo.H * geo.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(in_pix * KS * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)W2p, 0, (int)((size_t)BB_K * 9 * KS * 128), 0x00020000);
    int an[4], ay[4], ax[4], achunk[4];
    bool aok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rl = (4 * wave + g) * 8 + srow;
        achunk[g] = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        const int m = m0 + rl;
        aok[g] = m < M;
        const int mc = aok[g] ? m : 0;
        const int hw = geo.H * geo.W;
        an[g] = mc / hw;
        const int r = mc - an[g] * hw;
        ay[g] = r / geo.W;
        ax[g] = r - ay[g] * geo.W;
    }
    int bvoff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rl = (4 * wave + g) * 8 + srow;                       // LDS row of the weight tile
        bvoff[g] = (int)((size_t)bb_chan(rl) * 9 * KS * 128) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
    }
    auto dma_a = [&](int kt, int sa, int g, bool on) {
        const int j = kt / 9, tap = kt - j * 9;
        const int yy = ay[g] + (tap / 3 - 1) * geo.dil, xx = ax[g] + (tap % 3 - 1) * geo.dil;
        const bool ok = aok[g] & on & ((unsigned)yy < (unsigned)geo.H) & ((unsigned)xx < (unsigned)geo.W);
        const int pix = (an[g] * geo.H + yy) * geo.W + xx;
        ig_dma16(xrs, smem + sa * A_BYTES + (4 * wave + g) * 1024, ok ? pix * (KS * 128) + achunk[g] : OOB, j * 128);
    };
    auto dma_b = [&](int kt, int sb, int g, bool on) {
        const int j = kt / 9, tap = kt - j * 9;
        ig_dma16(wrs, smem + NSA * A_BYTES + sb * B_BYTES + (4 * wave + g) * 1024, on ? bvoff[g] : OOB, (tap * KS + j) * 128);
    };

    ig_f32x4 acc[16][2];                                 // [16-channel tile][16-pixel tile]
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[c][p] = (ig_f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int g = 0; g < 4; ++g) dma_a(0, 0, g, true);
#pragma unroll
    for (int g = 0; g < 4; ++g) dma_b(0, 0, g, true);
#pragma unroll
    for (int g = 0; g < 4; ++g) dma_a(1, 1, g, true);
    // fragment addressing: lane l holds row (l & 15), chunk 4 j + (l >> 4) of a 16-row fragment (j: first | second half of a
    // 16-bit slab, hi | lo plane of a split slab); chunk j = 1 is the j = 0 address XOR 64, tile t an immediate of t * 2 KiB
    const int r16 = lane & 15, kq = lane >> 4;
    const unsigned fswz = (unsigned)((kq ^ ((r16 >> 1) & 7)) << 4);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const unsigned fx0 = lds_base + (unsigned)((wave * 32 + r16) * 128) + fswz;             // this wave's 32 pixel rows
    const unsigned fw0 = lds_base + (unsigned)(NSA * A_BYTES + r16 * 128) + fswz;           // weight rows (all waves: all 256)
    int sa = 0;
    for (int kt = 0; kt < NK; ++kt) {
        const int sb = kt & 1;
        // A(kt), B(kt) of this wave have landed once all but its four youngest requests — A(kt + 1) — are done; then everyone's
        // has, and everyone has left A stage (kt + 2) % 3 and B stage sb ^ 1 (igemm_kernel.h)
#ifdef BB_DRAIN1      // diagnostic build
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#endif
        const int sa2 = sa == 0 ? 2 : sa - 1;
        const bool more_b = kt + 1 < NK, more_a = kt + 2 < NK;
        const unsigned ca = fx0 + (unsigned)(sa * A_BYTES), cb = fw0 + (unsigned)(sb * B_BYTES);
        sa = sa == 2 ? 0 : sa + 1;
        // pixel fragments of the whole k-step stay in registers (16); weight fragments go through a ring of three 16-channel
        // tiles: tile c + 2 is requested behind the MFMAs of tile c
        ig_bf16x8 xf[2][2], wf[3][2];
        xf[0][0] = ig_lds_read<0>(ca);    xf[0][1] = ig_lds_read<0>(ca ^ 64u);
        xf[1][0] = ig_lds_read<2048>(ca); xf[1][1] = ig_lds_read<2048>(ca ^ 64u);
        bb_read_tile(wf[0], cb, 0);
        bb_read_tile(wf[1], cb, 1);
        if (wave < 4) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int cur = c % 3;
            if (c == 8) {                                // the two waves of a SIMD hand the issue priority over mid-step
                if (wave < 4) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(1);
            }
            // LDS reads return in order: all but the two youngest (tile c + 1) are done -> tile c (and, at c = 0, the pixel
            // fragments) are there
            if (c == 0)
                asm volatile("s_waitcnt lgkmcnt(2)"
                             : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[1][0]), "+v"(xf[1][1]));
            else if (c + 1 < 16)
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[cur][0]), "+v"(wf[cur][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[cur][0]), "+v"(wf[cur][1]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (PL == 2) {                           // lo*hi + hi*lo + hi*hi, in the tile kernel's order (x lo * w hi first)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][0], xf[p][1], acc[c][p], 0, 0, 0);
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][1], xf[p][0], acc[c][p], 0, 0, 0);
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][0], xf[p][0], acc[c][p], 0, 0, 0);
                } else {
                    acc[c][p] = HT::mfma16(wf[cur][0], xf[p][0], acc[c][p]);
                    acc[c][p] = HT::mfma16(wf[cur][1], xf[p][1], acc[c][p]);
                }
                if (p == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + 2 < 16) bb_read_tile(wf[(c + 2) % 3], cb, c + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // DMA of the next tiles, one piece every other channel tile: first B(kt + 1), which must land within this k-step,
            // then A(kt + 2)
            if ((c & 1) == 0) {
                const int p = c >> 1;
                if (p < 4) dma_b(kt + 1, sb ^ 1, p, more_b);
                else dma_a(kt + 2, sa2, p - 4, more_a);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the zero fills past the last k-step)
    __syncthreads();                                    // all waves are done with the operand tiles

#ifdef BB_P1_ONLY      // diagnostic build: the time of phase 1 alone (one never-taken store keeps the accumulators alive)
    {
        float chk = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) chk += acc[c][p][0] + acc[c][p][1] + acc[c][p][2] + acc[c][p][3];
        if (chk == 1234.5678f) Y[tid] = 1;
        return;
    }
#endif
    // ================================ switch: BatchNorm tables -> LDS, first W3 stages, a2 -> operand registers ============
    float* s_sc3 = reinterpret_cast<float*>(smem + OFF_BN3);
    float* s_sh3 = s_sc3 + 1024;
    float* s_sc2 = reinterpret_cast<float*>(smem + OFF_BN2);
    float* s_sh2 = s_sc2 + BB_K;
    if (tid < BB_K) { s_sc2[tid] = sc2r; s_sh2[tid] = sh2r; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (tid + 512 * i < N3) { s_sc3[tid + 512 * i] = sc3r[i]; s_sh3[tid + 512 * i] = sh3r[i]; }

    // W3 stage t = output channels 32 t .. 32 t + 31 (LDS row q <-> channel 32 t + bb_chan(q)), image [slab][32 rows][128 B].
    // A stage is 4 row groups x KS slabs of (8 rows x 128 B): wave w moves row group w & 3 of the slabs DW (w >> 2) .. + DW - 1
    constexpr int DW = KS / 2;                          // DMA instructions per wave and stage (4 | 2)
    const int NT = N3 >> 5;
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc((void*)W3p, 0, (int)((size_t)N3 * KS * 128), 0x00020000);
    const int rgrp = wave & 3, j0 = (wave >> 2) * DW;
    const int q3 = rgrp * 8 + srow;
    const int w3voff = bb_chan(q3) * (KS * 128) + (((lane & 7) ^ ((q3 >> 1) & 7)) << 4);
    auto issue_w3 = [&](int t, int st) {
        const int voff = t < NT ? w3voff : OOB;
        unsigned char* base = smem + st * W3_STAGE + rgrp * 1024;
#pragma unroll
        for (int j = 0; j < DW; ++j) ig_dma16(w3rs, base + (j0 + j) * 4096, voff, t * (32 * KS * 128) + (j0 + j) * 128);
    };
    // identity rows / output rows of this lane: pixel m0 + 32 wave + 16 p + r16, channels 32 t + 8 kq .. + 7
    // (split planes: hi at the slab's byte 16 kq, lo at + 64; one plane: byte 64 t + 16 kq of the row)
    const size_t row_bytes = (size_t)N3 * 2 * PL;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)((size_t)M * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)((size_t)M * row_bytes), 0x00020000);
    int rvoff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int m = m0 + wave * 32 + p * 16 + r16;
        rvoff[p] = m < M ? (int)((size_t)m * row_bytes) + kq * 16 : OOB;
    }
    constexpr int SLAB_OUT = PL == 2 ? 128 : 64;        // bytes of 32 output channels in a row
    constexpr int NR = PL == 2 ? 4 : 2;                 // identity loads ( = output stores) per lane and stage
#ifdef BB_NO_RES       // diagnostic builds: phase 2 without its identity loads / without its stores
    constexpr int NRL = 0;
#else
    constexpr int NRL = NR;
#endif
#ifdef BB_NO_STORE
    constexpr int NRS = 0;
#else
    constexpr int NRS = NR;
#endif
    bb_u32x4 rr[2][NR];                                 // ring of two stages: [ring][p (, plane)]
    auto load_res = [&](int t, int buf) {
        const int so = (t < NT ? t : 0) * SLAB_OUT;
        if (NRL == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) rr[buf][i] = (bb_u32x4){0u, 0u, 0u, 0u};
            return;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            rr[buf][p * (NR / 2)] = bb_buf_load(rrs, rvoff[p], so);
            if (PL == 2) rr[buf][p * 2 + 1] = bb_buf_load(rrs, rvoff[p], so + 64);
        }
    };
    __syncthreads();                                    // the BatchNorm tables are in LDS (nothing is in flight yet: the
                                                        // fence of this barrier would drain it)
    issue_w3(0, 0);
    issue_w3(1, 1);
    load_res(0, 0);

    // a2 = relu(bn2(acc)) as the B-operand fragments of the 1x1: k-step s (channels 32 s .. + 31), pixel tile p
    ig_bf16x8 ah[8][2], al[PL == 2 ? 8 : 1][2];
    const unsigned t2 = lds_base + (unsigned)(OFF_BN2 + kq * 32);
#define BB_CONV(S)                                                                                                         \
    {                                                                                                                      \
        ig_f32x4 c0 = bb_lds_f4<(S) * 128>(t2), c1 = bb_lds_f4<(S) * 128 + 16>(t2);                                        \
        ig_f32x4 d0 = bb_lds_f4<1024 + (S) * 128>(t2), d1 = bb_lds_f4<1024 + (S) * 128 + 16>(t2);                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1));                                     \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                    \
            float v[8];                                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                \
                v[r] = fmaf(acc[2 * (S)][p][r], c0[r], d0[r]);                                                             \
                v[4 + r] = fmaf(acc[2 * (S) + 1][p][r], c1[r], d1[r]);                                                     \
            }                                                                                                              \
            unsigned ph[4], pl_[4];                                                                                        \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
                const float e0 = v[2 * q] > 0.f ? v[2 * q] : 0.f, e1 = v[2 * q + 1] > 0.f ? v[2 * q + 1] : 0.f;            \
                unsigned short h0, l0 = 0, h1, l1 = 0;                                                                     \
                if (PL == 2) {                                                                                             \
                    ig_split(e0, h0, l0);                                                                                  \
                    ig_split(e1, h1, l1);                                                                                  \
                } else {                                                                                                   \
                    h0 = HT::enc(e0);                                                                                      \
                    h1 = HT::enc(e1);                                                                                      \
                }                                                                                                          \
                ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);                                                               \
                pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);                                                              \
            }                                                                                                              \
            ah[S][p] = __builtin_bit_cast(ig_bf16x8, (bb_u32x4){ph[0], ph[1], ph[2], ph[3]});                              \
            if (PL == 2) al[S][p] = __builtin_bit_cast(ig_bf16x8, (bb_u32x4){pl_[0], pl_[1], pl_[2], pl_[3]});            \
        }                                                                                                                  \
        /* pin the results between the volatile table reads (else all 32 reads are issued first: 128 live registers) */    \
        asm volatile("" : "+v"(ah[S][0]), "+v"(ah[S][1]));                                                                 \
        if (PL == 2) asm volatile("" : "+v"(al[(PL == 2) ? (S) : 0][0]), "+v"(al[(PL == 2) ? (S) : 0][1]));                \
    }
    BB_CONV(0) BB_CONV(1) BB_CONV(2) BB_CONV(3) BB_CONV(4) BB_CONV(5) BB_CONV(6) BB_CONV(7)
#undef BB_CONV

    // ================================ phase 2: y^T = W3 * a2^T, 32 output channels per stage ===============================
    // W3 fragment of k-step s, 16-row tile b: split planes: slab s, chunks kq (hi) and 4 + kq (lo = address XOR 64);
    // one plane: slab s >> 1, chunk 4 (s & 1) + kq
    const unsigned f3 = lds_base + (unsigned)(r16 * 128) + fswz;
    const unsigned t3 = lds_base + (unsigned)(OFF_BN3 + kq * 32);
    auto stage = [&](int t, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        const int st = t % NS3;
        // This wave's share of stage t has landed once everything it issued BEFORE the previous iteration has retired: younger
        // are that iteration's identity request (NR), its DMA of stage t + 1 (DW) and its stores (NR).  First iteration: the
        // DMA of stage 1 and the first identity request.  A bare s_barrier (xconv2.hip).
#ifdef BB_DRAIN2      // diagnostic build
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#else
        if (t == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DW + NRL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DW + NRL + NRS) : "memory");
#endif
        load_res(t + 1, CUR ^ 1);
        issue_w3(t + 2, (t + 2) % NS3);
        __builtin_amdgcn_sched_barrier(0);

        ig_f32x4 o4[2][2];                               // [pixel tile][channel tile b]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < 2; ++b) o4[p][b] = (ig_f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned fa = f3 + (unsigned)(st * W3_STAGE);
        ig_bf16x8 wh[2][2], wl[2][2];                    // [ring][b]
#define BB_READ(S, RG)                                                                                                     \
        if (PL == 2) {                                                                                                     \
            wh[RG][0] = ig_lds_read<(S) * 4096>(fa);        wl[RG][0] = ig_lds_read<(S) * 4096>(fa ^ 64u);                 \
            wh[RG][1] = ig_lds_read<(S) * 4096 + 2048>(fa); wl[RG][1] = ig_lds_read<(S) * 4096 + 2048>(fa ^ 64u);         \
        } else {                                                                                                           \
            wh[RG][0] = ig_lds_read<((S) >> 1) * 4096>(fa ^ (((S) & 1) ? 64u : 0u));                                       \
            wh[RG][1] = ig_lds_read<((S) >> 1) * 4096 + 2048>(fa ^ (((S) & 1) ? 64u : 0u));                                \
        }
#define BB_STEP(S, RG)                                                                                                     \
        if (PL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[RG][0]), "+v"(wl[RG][0]), "+v"(wh[RG][1]), "+v"(wl[RG][1])); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh[RG][0]), "+v"(wh[RG][1]));                                      \
        if ((S) + 1 < 8) { BB_READ(((S) + 1) & 7, (RG) ^ 1) }                                                               \
        _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                      \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                \
                if (PL == 2) {                                  /* lo*hi + hi*lo + hi*hi (xconv2.hip's order) */            \
                    o4[p][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[RG][b], ah[S][p], o4[p][b], 0, 0, 0);            \
                    o4[p][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[RG][b], al[(PL == 2) ? (S) : 0][p], o4[p][b], 0, 0, 0); \
                    o4[p][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[RG][b], ah[S][p], o4[p][b], 0, 0, 0);            \
                } else {                                                                                                   \
                    o4[p][b] = HT::mfma16(wh[RG][b], ah[S][p], o4[p][b]);                                                  \
                }                                                                                                          \
            }
        BB_READ(0, 0)
        BB_STEP(0, 0) BB_STEP(1, 1) BB_STEP(2, 0) BB_STEP(3, 1) BB_STEP(4, 0) BB_STEP(5, 1) BB_STEP(6, 0) BB_STEP(7, 1)
#undef BB_STEP
#undef BB_READ
        // ---- epilogue: lane = pixel (p, r16), channels 32 t + 8 kq .. + 7 (tile b = 0: + 0..3, b = 1: + 4..7)
        __builtin_amdgcn_sched_barrier(0);
        float sc[8], sh[8];
        {
            const unsigned ta = t3 + (unsigned)(t * 128);
            ig_f32x4 s0 = bb_lds_f4<0>(ta), s1 = bb_lds_f4<16>(ta), t0 = bb_lds_f4<4096>(ta), t1 = bb_lds_f4<4096 + 16>(ta);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(t0), "+v"(t1));
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = s0[r]; sc[4 + r] = s1[r]; sh[r] = t0[r]; sh[4 + r] = t1[r]; }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float o[8];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[b * 4 + r] = fmaf(o4[p][b][r], sc[b * 4 + r], sh[b * 4 + r]);
            const bb_u32x4 vh = rr[CUR][p * (NR / 2)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[2 * q] += HT::lo(vh[q]);
                o[2 * q + 1] += HT::hi(vh[q]);
            }
            if (PL == 2) {
                const bb_u32x4 vl = rr[CUR][p * (NR / 2) + (PL == 2 ? 1 : 0)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[2 * q] += __uint_as_float(vl[q] << 16);
                    o[2 * q + 1] += __uint_as_float(vl[q] & 0xFFFF0000u);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
            unsigned ph[4], pl_[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned short h0, l0 = 0, h1, l1 = 0;
                if (PL == 2) {
                    ig_split(o[2 * q], h0, l0);
                    ig_split(o[2 * q + 1], h1, l1);
                } else {
                    h0 = HT::enc(o[2 * q]);
                    h1 = HT::enc(o[2 * q + 1]);
                }
                ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
                pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
            if (NRS != 0) {
                bb_buf_store(yrs, (bb_u32x4){ph[0], ph[1], ph[2], ph[3]}, rvoff[p], t * SLAB_OUT);
                if (PL == 2) bb_buf_store(yrs, (bb_u32x4){pl_[0], pl_[1], pl_[2], pl_[3]}, rvoff[p], t * SLAB_OUT + 64);
            } else if (ph[0] == 0x12345678u && pl_[1] == 0x9abcdef0u) {
                bb_buf_store(yrs, (bb_u32x4){ph[0], ph[1], ph[2], ph[3]}, rvoff[p], t * SLAB_OUT);      // (never taken)
            }
            // HARDWARE HAZARD (measured, round 5): a buffer_store_dwordx4 with an SGPR soffset reads its data registers late; the
            // compiler's hazard recogniser assumes the register-soffset form is safe and let the next pixel tile's first
            // v_pk_fma_f32 overwrite them in the following cycle — one dword of lanes (l & 15) >= 12 was stored from the NEXT
            // tile's values in the 16-bit variants (tools/dbg/b2b_race.py).  Wait states behind the stores close it.
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t < NT; t += 2) {                    // two stages per trip: the identity ring is indexed at compile time
        stage(t, std::integral_constant<int, 0>());
        stage(t + 1, std::integral_constant<int, 1>());
    }
}

}  // namespace hiast

/* host: does the fused tail take this bottleneck?  (else: hiast_igemm_bn_act x 2) */
extern "C" int hiast_bottleneck_tail_ok(int B, int H, int W, int Cmid, int Cout, int stride, int fmt)
{
    // OPT-IN (HIAST_B2B=1).  Measured in round 5 (profiles/r05_b2b_*.txt): bit-equal to the two launches, but not faster — all
    // CUs run the MFMA-bound phase 1 and then the HBM-bound phase 2 in lockstep, so the phases add up exactly as the two
    // launches did (the 2 x 67 MB round trip of a2 was never the bound), phase 1 in this wave layout (36 instead of 24
    // fragment reads per k-step) runs 188 us against ~150 for the tile kernel's main loop, and the step loses 0.6 ms.
    const char* env = getenv("HIAST_B2B");
    if (!env || atoi(env) == 0) return 0;
    if (!hiast_fmt_ok(fmt) || B <= 0 || H <= 0 || W <= 0 || stride != 1) return 0;
    if (Cmid != hiast::BB_K || Cout % 64 != 0 || Cout > 1024 || Cout < 64) return 0;
    const int64_t M = (int64_t)B * H * W;
    const int planes = hiast_fmt_planes(fmt);
    if (M < 4096) return 0;                           // small maps: too few 256-pixel tiles to fill the chip
    if ((uint64_t)M * Cout * 2 * planes >= (1ull << 31) || (uint64_t)M * Cmid * 2 * planes >= (1ull << 31)) return 0;
    return 1;
}

extern "C" int hiast_bottleneck_tail(const void* x, const void* w2p, const float* gamma2, const float* beta2,
                                     const float* mean2, const float* var2, float eps2, const void* w3p,
                                     const float* gamma3, const float* beta3, const float* mean3, const float* var3,
                                     float eps3, const void* res, void* y, int B, int H, int W, int Cmid, int Cout, int dil,
                                     int fmt, hiast_stream_t stream)
{
    using namespace hiast;
    if (!x || !w2p || !w3p || !mean2 || !var2 || !mean3 || !var3 || !res || !y) return HIAST_E_ARG;
    if (dil <= 0) return HIAST_E_ARG;
    if (!hiast_bottleneck_tail_ok(B, H, W, Cmid, Cout, 1, fmt)) return HIAST_E_RANGE;
    if ((((uintptr_t)x) | ((uintptr_t)w2p) | ((uintptr_t)w3p) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) return HIAST_E_RANGE;
    const int64_t M = (int64_t)B * H * W;
    const IGeo geo = {H, W, H, W, 1, dil};
    const dim3 grid((unsigned)((M + BB_BM - 1) / BB_BM));
    hipStream_t st = (hipStream_t)stream;
#define BBL(PLV, F16V)                                                                                                   \
    hipLaunchKernelGGL((b2b_kernel<PLV, F16V>), grid, dim3(512), 0, st, (const unsigned short*)x, (const unsigned short*)w2p, \
                       gamma2, beta2, mean2, var2, eps2, (const unsigned short*)w3p, gamma3, beta3, mean3, var3, eps3,    \
                       (const unsigned short*)res, (unsigned short*)y, (int)M, Cout, geo)
    if (fmt == HIAST_FMT_SPLIT_BF16) BBL(2, false);
    else if (fmt == HIAST_FMT_FP16) BBL(1, true);
    else BBL(1, false);
#undef BBL
    HIAST_CHECK_LAUNCH();
    return 0;
}
