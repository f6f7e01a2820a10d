// K9c — trunk convolutions (1x1 and 3x3, dilated / strided) of the inference forwards as an implicit GEMM whose
// operands reach LDS by LDS-DMA (buffer_load ... lds), fused with BatchNorm(eval) + residual + ReLU
// (reference: Bottleneck.forward, sseg/models/modules/resnet.py:78-98: conv -> bn -> relu / += identity as separate
//  cuDNN / ATen passes).
//
//   Y[m][n] = act( (Σ_tap Σ_k X[m + off(tap)][k] * W[n][tap][k]) * scale_n + shift_n (+ R[m][n]) )
//
// Operand formats (channels-last, 16-bit).  Every row is a sequence of 128-byte SLABS, one per k-step:
//   PL = 2  "split planes": an fp32-class value v is stored as hi = bf16(v), lo = bf16(v - hi); slab j of a row
//           holds channels 32j .. 32j+31 as [hi x32 | lo x32] (4 bytes per value, like fp32).  The product is
//           accumulated in fp32 as hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_bf16 (the dropped lo*lo term is
//           2^-16 relative): ~5e-6 of max|Y| against fp64 at 3/16 of the exact-fp32 MFMA cost (gfx950 has no
//           TF32/xf32).  The split is done ONCE, by the producing kernel's epilogue (and once per weight by
//           hiast_pack_conv_weight) — conv1x1.hip re-splits every operand element in every block that stages it.
//   PL = 1  plain bf16 rows; a slab is 64 consecutive channels (mixed-precision teacher forward).
// Weights: Wp[n][tap][slab] (hiast_pack_conv_weight).
//
// Structure: 256 x BN block tile (BN = 256 | 128 | 64), 8 waves, one slab of k per step.  A k-step's tile
// ((256 + BN) rows x 128 B) is written into LDS by the DMA path: each wave-instruction moves 8 whole 128-byte
// lines (one row slab each: full L2 lines) to a wave-uniform LDS address + 16 B x lane; the bank-conflict-free XOR
// swizzle of the image is therefore applied to each lane's SOURCE chunk and again by the fragment reads.  Out-of-image
// taps and tail rows use a buffer offset beyond num_records, for which the DMA writes zeros: zero padding costs no
// instruction.  No VGPR staging and no VALU in the loop besides address selection.  The 256-row tile halves the L2->LDS
// bytes per MFMA of a 128 x 128 tile.
//
// The k-step loop (round 2; measured with the s_memtime stamps of -DIG_STAMP, tools/igemm_stamps.py):
//  * three LDS stages of the activation tile, two of the weight tile (160 KiB), counted `s_waitcnt vmcnt` + a bare
//    `s_barrier` per k-step (behind __syncthreads() the compiler drains vmcnt to 0);
//  * fragment reads are inline asm (`ds_read_b128` with the tile offset as immediate), waits are explicit: the compiler
//    orders every LDS load it can see behind ALL pending LDS-DMAs;
//  * the DMA pieces of the following k-steps are issued one at a time BETWEEN the MFMAs of the 16-row tiles: the CU's
//    texture path takes 64 B per clock (tools/micro/dma_fill.hip: 64 KiB per ~1000 cycles, 35 TB/s chip-wide from L2),
//    so a burst of the eight pieces of every wave stalls the issuing waves for ~1000 cycles per k-step;
//  * no branch in the loop (the last k-steps issue zero fills instead of skipping the DMA);
//  * the two waves of a SIMD hand the issue priority over in the middle of the k-step (s_setprio): the arbiter serves the
//    older wave first, which left waves 4-7 ~900 cycles behind at every barrier.
// Round 1's loop (two stages, compiler-scheduled reads, 20 branches per k-step) held the matrix pipe 54 % busy at the
// clock the chip runs (3780 cycles per 2048-cycle k-step); this one 63-66 %.
#pragma once
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 ig_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ig_f32x16;
typedef __attribute__((ext_vector_type(4))) float ig_f32x4;

constexpr int IG_BM = 256;
// which launches run the early-barrier k-loop (igemm_bn_act_kernel): a compile-time choice per (planes, taps).  Measured
// (profiles/r05_ab_early_barrier.txt, A/B of two builds on one box): split-plane 3x3 -4 % per launch (layer3 192.6 -> 184.2 us,
// layer4 606 -> 582), step 54.7 -> 54.3 ms; 16-bit 3x3 -1 ... -3 % (step: no further gain); 1x1 launches 0 ... +9 % SLOWER (their A
// operand comes from HBM and the early barrier leaves its DMA one step of latency instead of 1.5-2).
#ifndef IG_EARLY_BARRIER
#define IG_EARLY_BARRIER(PL, TAPS) ((TAPS) == 9 && (PL) == 2)
#endif
#ifndef IG_FLIP8
#define IG_FLIP8 4
#endif
// automatic choice of the 128 x 128 / two-blocks-per-CU form (ig_half_tile below); M pixels, K -> N channels, 1x1 only
// Measured (profiles/r05_ab_igemm_half_tile.txt): stand-alone the 128 x 128 form wins wherever the 256-row form leaves half of
// the chip idle (<= 128 blocks: the 1024 -> 256 launch of a 4-image batch 71 -> 53 us in split planes, 38 -> 32 us in fp16) and
// loses 5-12 % where the 256-row form fills the chip once (B = 8: 50 -> 55 us; the operands pass the L2 -> LDS path twice).
// In the step the 4-image launches are the two sub-batches of an inference forward on two streams — two half-chip launches
// side by side already fill the chip, and the step did not move (55.5 / 55.5 against 55.7 / 55.8 ms) — so the automatic rule takes
// the quarter-chip launches only (<= 64 blocks): the pseudo-label generator at the reference's batch size 2 runs 247 -> 271
// images/s end to end with it.
#ifndef IG_HALF_AUTO
#define IG_HALF_BLOCKS(M, N) ((((M) + 255) / 256) * (((N) % 256 == 0) ? (N) / 256 : (N) / 128))
// (half-chip launches too — unless the caller runs two launch sequences side by side: HIAST_IGEMM_COSCHED=1, set by
// functional.eval_forward_split around its sub-batches; two half-chip launches on two streams already fill the chip)
#define IG_HALF_AUTO(M, K, N) (IG_HALF_BLOCKS(M, N) <= 64 || (IG_HALF_BLOCKS(M, N) <= 128 && !ig_cosched()))
#endif

#ifdef IG_STAMP       // diagnostic build (tools/igemm_stamps.py): cycles a wave spends in the parts of a k-step, summed over the loop
__device__ unsigned ig_stamp_buf[1024 * 8 * 8 + 32 * 64];
__device__ __forceinline__ unsigned ig_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (unsigned)t;
}
#define IG_T(var) const unsigned var = ig_now()
#define IG_ACC(i, a, b) stamp_acc[i] += (b) - (a)
#else
#define IG_T(var)
#define IG_ACC(i, a, b)
#endif

struct IGeo {
    int H, W, Ho, Wo, stride, dil;
};

// LDS image of a [rows][128 B] tile: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7).  A ds_read_b128
// fragment read (16 lanes = 16 distinct rows of a 32-row fragment, same logical chunk) then covers all sixteen
// 16-byte slots of the 256-byte bank row: conflict free.
__device__ __forceinline__ int ig_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void ig_split(float v, unsigned short& h, unsigned short& l)
{
    const __hip_bfloat16 hb = __float2bfloat16(v);
    h = __bfloat16_as_ushort(hb);
    l = __bfloat16_as_ushort(__float2bfloat16(v - __bfloat162float(hb)));
}

typedef __attribute__((address_space(3))) void* ig_lds_ptr;

// LDS-DMA: 64 lanes x 16 bytes from buffer offset (voff per lane + soff) to lds + 16*lane (lds wave-uniform).
// Kept in a NON-template function: with the builtin inside a kernel template, hipcc (ROCm 7.2) emits no host
// stub for the instantiations.
__device__ __forceinline__ void ig_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ig_lds_ptr)lds, 16, voff, soff, 0, 0);
}
// ... with the non-temporal cache policy (aux bit 1 = nt): the A operand of a 1x1 launch is read exactly ONCE, by one block
// (IG_A_NT: A/B build, tools/build_variant_igemm.sh ... -DIG_A_NT=1)
#ifndef IG_A_NT
#define IG_A_NT 0
#endif
__device__ __forceinline__ void ig_dma16_nt(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ig_lds_ptr)lds, 16, voff, soff, 0, IG_A_NT ? 2 : 0);
}

// Fragment reads as inline asm (ds_read_b128, tile offset as the instruction's immediate) with explicit lgkmcnt waits that
// tie the destination registers: the compiler orders every LDS load it can see behind ALL pending LDS-DMAs
// (s_waitcnt vmcnt(0)), which would make the DMA of k-step t+1 a wait inside k-step t.
template <int OFF>
__device__ __forceinline__ ig_bf16x8 ig_lds_read(unsigned addr)
{
    ig_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ void ig_lds_read4(ig_bf16x8 (&f)[4][2], unsigned addr)       // B: four 16-row tiles x two chunks
{
    f[0][0] = ig_lds_read<0>(addr);        f[0][1] = ig_lds_read<0>(addr ^ 64u);
    f[1][0] = ig_lds_read<2048>(addr);     f[1][1] = ig_lds_read<2048>(addr ^ 64u);
    f[2][0] = ig_lds_read<4096>(addr);     f[2][1] = ig_lds_read<4096>(addr ^ 64u);
    f[3][0] = ig_lds_read<6144>(addr);     f[3][1] = ig_lds_read<6144>(addr ^ 64u);
}
__device__ __forceinline__ void ig_lds_read_b(ig_bf16x8 (&f)[2], unsigned addr, int b)  // B: 16-row tile b (a constant after unrolling)
{
    switch (b) {
    case 1: f[0] = ig_lds_read<2048>(addr); f[1] = ig_lds_read<2048>(addr ^ 64u); break;
    case 2: f[0] = ig_lds_read<4096>(addr); f[1] = ig_lds_read<4096>(addr ^ 64u); break;
    case 3: f[0] = ig_lds_read<6144>(addr); f[1] = ig_lds_read<6144>(addr ^ 64u); break;
    default: f[0] = ig_lds_read<0>(addr); f[1] = ig_lds_read<0>(addr ^ 64u); break;
    }
}
__device__ __forceinline__ void ig_lds_read_a(ig_bf16x8 (&f)[2], unsigned addr, int a)  // A: 16-row tile a (a is a constant
{                                                                                        // after unrolling)
    switch (a) {
    case 1: f[0] = ig_lds_read<1 * 2048>(addr); f[1] = ig_lds_read<1 * 2048>(addr ^ 64u); break;
    case 2: f[0] = ig_lds_read<2 * 2048>(addr); f[1] = ig_lds_read<2 * 2048>(addr ^ 64u); break;
    case 3: f[0] = ig_lds_read<3 * 2048>(addr); f[1] = ig_lds_read<3 * 2048>(addr ^ 64u); break;
    case 4: f[0] = ig_lds_read<4 * 2048>(addr); f[1] = ig_lds_read<4 * 2048>(addr ^ 64u); break;
    case 5: f[0] = ig_lds_read<5 * 2048>(addr); f[1] = ig_lds_read<5 * 2048>(addr ^ 64u); break;
    case 6: f[0] = ig_lds_read<6 * 2048>(addr); f[1] = ig_lds_read<6 * 2048>(addr ^ 64u); break;
    case 7: f[0] = ig_lds_read<7 * 2048>(addr); f[1] = ig_lds_read<7 * 2048>(addr ^ 64u); break;
    default: f[0] = ig_lds_read<0>(addr); f[1] = ig_lds_read<0>(addr ^ 64u); break;
    }
}
template <int N>
__device__ __forceinline__ void ig_lds_wait_n(ig_bf16x8 (&f)[2], ig_bf16x8 (&g)[2])    // all but the N youngest LDS reads are done
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(g[0]), "+v"(g[1]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ig_lds_wait_n(ig_bf16x8 (&f)[2])
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N));
}
__device__ __forceinline__ void ig_lds_wait_a(ig_bf16x8 (&f)[2])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
}
__device__ __forceinline__ void ig_lds_wait_b(ig_bf16x8 (&f)[4][2], ig_bf16x8 (&g)[2])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]),
                   "+v"(f[3][1]), "+v"(g[0]), "+v"(g[1]));
}

// element offset (in bf16 units) of channel c (a multiple of 8) of row m in a [M][C] activation
template <int PL>
__device__ __forceinline__ size_t ig_elem(size_t m, int c, int C)
{
    if (PL == 2) return (m * (C >> 5) + (c >> 5)) * 64 + (c & 31);     // lo plane: + 32
    return m * C + c;
}

// OUTF32: write fp32 [M][N] — the ASPP tap GEMM; otherwise the output has the input's format.
// Waves: WM x WN = 8; wave tile (256/WM) x (BN/WN) with BN/WN == 64.
// The wave tile is built from v_mfma_f32_16x16x32_bf16 (16 x 16 output tiles, 32-deep): the same LDS bytes and MFMA cycles
// per flop as 32x32x16, but the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back
// item 7; measured 4-10 % on every trunk shape in round 1).
// GATE (PL = 1, RES, no ReLU): 0 = plain residual; 1 = the residual is kept where the gate tensor Rg (values like R)
// is > 0; 2 = where bit (n & 7) of byte Rg[m][n / 8] is set.  A compile-time switch: as a run-time test on Rg the
// gate put ~500 branches and ~300 s_waitcnt into the epilogue of EVERY bf16 residual launch (the teacher forward
// included), which serialised its residual prefetch.
// STATS (plain bf16 launches only): 1 = emit the per-block BatchNorm sums Σy, Σy² of the stored outputs (the student
// forward: the BN that follows needs no pass over Y for its statistics); 2 = data-gradient launch whose output dA is the
// gradient of a BN + ReLU activation A = relu(bn(x)): emit the per-block sums Σg, Σ g*xhat of that BN's backward
// (g = dA where bn(x) > 0; x is passed as R, its batch mean / invstd as mean / var, gamma / beta as themselves) — the
// BatchNorm backward then skips its statistics pass (one read of dA and x per layer).
// F16 (PL = 1 only): the rows are IEEE fp16 instead of bf16 (HIAST_FMT_FP16: the reference's apex-O1 type) — same slabs,
// same DMA / LDS images, v_mfma_f32_16x16x32_f16; decode / encode through H16<F16> (common.h).
// UPS (3x3, plain launches only): the TRANSPOSED stride-2 convolution — the data gradient of a 3x3 / stride-2 / padding-1
// convolution (layer2.0.conv2 of the trunk).  The output map (geo.Ho x geo.Wo) is the convolution's INPUT map, the operand
// (geo.H x geo.W) its output gradient: tap (ty, tx) of output pixel (y, x) reads source pixel ((y + ty - 1) / 2, (x + tx - 1) / 2)
// when both are even and nothing otherwise (the weights are packed in adjoint form as for a stride-1 data gradient).  The
// parity test joins the in-image test of the DMA address; three of four taps fetch zeros (which the DMA writes for free).
// BM (round 5): rows of the block tile = 32 x the number of waves.  256 (8 waves, one block per CU: the form every MFMA-bound launch
// uses) or 128 with BN = 128 (4 waves, 80 KiB of LDS: TWO blocks per CU, one wave of each per SIMD) — the HBM-bound 1x1 launches
// with a long reduction (1024 -> 256, 2048 -> 512) run ONE round of 256-row tiles, so nothing overlapped their ~19 us of
// prologue + epilogue; as 128 x 128 tiles a CU always holds two blocks that are out of phase — one streams its operands while
// the other one stores its tile.  The A tile is then read by two column blocks (neighbours on one XCD: L2).
template <int PL, bool OUTF32, int BN, int TAPS, bool RES, bool RELU, int GATE = 0, int STATS = 0, bool F16 = false,
          bool UPS = false, int BM = IG_BM>
__global__ __launch_bounds__(BM * 2) void igemm_bn_act_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, float eps,
    const unsigned short* __restrict__ R, void* __restrict__ Yv, int M, int K, int N, IGeo geo,
    float* __restrict__ stats,           // optional [gridDim m-blocks][N][2]: per-block Σy, Σy² of the STORED values
    const unsigned short* __restrict__ Rg)   // GATE != 0: o += gate ? R : 0 (the ReLU-masked gradient of an identity branch)
{
    static_assert(GATE == 0 || (PL == 1 && RES && !RELU && !OUTF32), "gated residual: bf16 data-gradient launches only");
    static_assert(!F16 || PL == 1, "fp16 rows are a one-plane format");
    static_assert(!UPS || (TAPS == 9 && PL == 1 && !RES && !RELU && GATE == 0 && STATS == 0 && !OUTF32), "transposed stride 2: plain 3x3");
    using HT = H16<F16>;
    constexpr int NWV = BM / 32;                        // waves per block (8 | 4)
    static_assert((BM == 256 || (BM == 128 && BN <= 128)) && NWV * 64 == BM * 2, "block tile: 256 x BN (8 waves) or 128 x <=128 (4 waves)");
    constexpr int WN = BN / 64, WM = NWV / WN;
    static_assert(WM >= 1 && WM * WN == NWV, "wave grid");
    static_assert(!STATS || (PL == 1 && !OUTF32 && !RES && !RELU && GATE == 0), "statistics epilogue: plain bf16 launches only");
    constexpr bool XROWS = RES || STATS == 2;           // the epilogue reads rows of R (residual | BN input)
    constexpr int TM = BM / WM / 32;                    // 32-row tiles per wave (4 | 2 | 1)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    // LDS: THREE stages of the activation tile and TWO of the weight tile (160 KiB at BN = 256).  The fill rate of a tile
    // is (bytes in flight) / latency, and with one 64 KiB tile in flight per CU the launches were bound by exactly that
    // (removing every MFMA left the kernel time unchanged): the activation rows — first touched in HBM — are requested
    // two k-steps ahead, the weight rows (L2 hits, the same for every block) one.
    constexpr int NSA = 3, NSB = 2;
    constexpr int LDS_BYTES = NSA * A_BYTES + NSB * B_BYTES;
    constexpr bool EB = IG_EARLY_BARRIER(PL, TAPS) != 0;   // the early-barrier k-loop (below)
    constexpr int BG = BN / 8 / NWV;                    // 8-row B groups per wave (4 | 2 | 1)
    constexpr int EP = 68;                              // padded row of a wave's private epilogue tile (floats)
    static_assert(LDS_BYTES >= NWV * 32 * EP * 4, "epilogue staging must fit in the tile buffers");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    IG_T(stamp_kernel);
    int bm, bn_;
    {   // XCD-aware tile order (see conv1x1.hip): channel tile fastest, XCD k takes the k-th contiguous eighth
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
        int lid = blockIdx.x + gx * blockIdx.y;
        if ((total & 7) == 0) lid = (lid & 7) * (total >> 3) + (lid >> 3);
        bn_ = lid % gy;
        bm = lid / gy;
    }
    const int m0 = bm * BM, n0 = bn_ * BN;
    const int KS = (K * PL) >> 6;                       // slabs per row (= k-steps per tap)
    const int nk = TAPS * KS;

    // ---- DMA addressing.  This wave fills A row groups {4*wave .. 4*wave+3} (8 rows each) and B row groups
    // {BG*wave ..}; lane l supplies row (l >> 3) of a group and the logical 16-byte chunk that belongs at
    // physical chunk (l & 7) of that row in the swizzled image.
    const int srow = lane >> 3;
    constexpr int OOB = (int)0x80000000;
    const size_t in_pix = (TAPS == 1) ? (size_t)M : (size_t)(M / (geo.Ho * geo.Wo)) * geo.H * geo.W;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(in_pix * KS * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)((size_t)N * TAPS * KS * 128), 0x00020000);
    int an[4], ay[4], ax[4], achunk[4];   // per A row group: image / row / column of the (stride-scaled) output pixel
    bool aok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rl = (4 * wave + g) * 8 + srow;                      // tile row
        achunk[g] = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
        const int m = m0 + rl;
        aok[g] = m < M;
        const int mc = aok[g] ? m : 0;
        if (TAPS == 1) {
            an[g] = mc; ay[g] = 0; ax[g] = 0;
        } else {
            const int hw = geo.Ho * geo.Wo;
            an[g] = mc / hw;
            const int r = mc - an[g] * hw;
            ay[g] = (r / geo.Wo) * (UPS ? 1 : geo.stride);
            ax[g] = (r - (r / geo.Wo) * geo.Wo) * (UPS ? 1 : geo.stride);
        }
    }
    int bvoff[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) {
        const int rl = (BG * wave + g) * 8 + srow;
        bvoff[g] = (int)((size_t)(n0 + rl) * TAPS * KS * 128) + ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
    }

    // DMA pieces of a wave (one instruction = 8 rows x 128 B): p < BG -> its B row group p, else its A row group p - BG.
    // No branches: an out-of-image tap (or a tail row) selects the out-of-range offset, for which the DMA writes zeros.
    // k-step order: channel slab OUTER, tap INNER — the nine shifted re-reads of a slab follow each other, so
    // the set an XCD's 32 blocks re-read (32 x 256 rows x 128 B = 1 MiB + the weights) stays in its 4 MiB L2
    // (tap-outer order swept the whole 8 MiB image between two uses: 50 % L2 hit rate, 4.8x over-fetch)
    constexpr int NPIECE = BG + 4;
    auto dma_a = [&](int kt, int sa, int g, bool on) {       // !on (past the last k-step): a zero fill nobody reads
        const int j = TAPS == 1 ? kt : kt / TAPS;
        const int tap = TAPS == 1 ? 0 : kt - j * TAPS;
        int voff;
        if (TAPS == 1) {
#ifdef IG_SLABMAJOR_A      // diagnostic build (tools/ab_slabmajor.py): the A operand stored slab-major inside each 256-row tile
                           // ([tile][slab][row][128 B]: a k-step's tile is ONE contiguous 32 KiB run) — timing only
            voff = (aok[g] & on) ? (an[g] / BM) * (BM * KS * 128) + (an[g] % BM) * 128 + achunk[g] : OOB;
#else
            voff = (aok[g] & on) ? an[g] * (KS * 128) + achunk[g] : OOB;
#endif
        } else {
            int yy = ay[g] + (tap / 3 - 1) * geo.dil, xx = ax[g] + (tap % 3 - 1) * geo.dil;
            bool par = true;
            if (UPS) {                                       // (-1 is odd: rejected before the shift)
                par = ((yy | xx) & 1) == 0;
                yy >>= 1; xx >>= 1;
            }
            const bool ok = aok[g] & on & par & ((unsigned)yy < (unsigned)geo.H) & ((unsigned)xx < (unsigned)geo.W);
            const int pix = (an[g] * geo.H + yy) * geo.W + xx;
            voff = ok ? pix * (KS * 128) + achunk[g] : OOB;
        }
#ifdef IG_SLABMAJOR_A
        ig_dma16(xrs, smem + sa * A_BYTES + (4 * wave + g) * 1024, voff, TAPS == 1 ? j * (BM * 128) : j * 128);
#else
        if (TAPS == 1) ig_dma16_nt(xrs, smem + sa * A_BYTES + (4 * wave + g) * 1024, voff, j * 128);
        else ig_dma16(xrs, smem + sa * A_BYTES + (4 * wave + g) * 1024, voff, j * 128);
#endif
    };
    auto dma_b = [&](int kt, int sb, int g, bool on) {
        const int j = TAPS == 1 ? kt : kt / TAPS;
        const int tap = TAPS == 1 ? 0 : kt - j * TAPS;
        ig_dma16(wrs, smem + NSA * A_BYTES + sb * B_BYTES + (BG * wave + g) * 1024, on ? bvoff[g] : OOB,
                 (tap * KS + j) * 128);
    };

    constexpr int NA = 2 * TM;                          // 16-row tiles of the wave tile (8 | 4 | 2); 4 column tiles
    ig_f32x4 acc4[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc4[a][b] = (ig_f32x4){0.f, 0.f, 0.f, 0.f};

    const int erow = lane >> 3, ec8 = lane & 7;             // epilogue: lane -> (row of an 8-row group, 8-channel group)
    const int nc = n0 + wn * 64 + ec8 * 8;
    // residual rows: with one 8-wave block per CU only this wave's own loads hide the HBM latency of the epilogue, so
    // the rows of chunk a + RD are requested while chunk a goes through its LDS round trip (RD chunks = RD x 4 x 16-byte
    // loads per lane in flight; 2 for one plane, 1 for split planes where hi and lo double the registers)
    constexpr int RD = PL == 2 ? 1 : 2;
    uint4 rhA[TM][4], rlA[TM][4];
    // part: 0 = everything, 1 = the hi plane only, 2 = the rest (split planes: hi and lo of a 32-channel slab share one
    // 128-byte line, so an early request for hi also brings lo to the L2)
    auto load_res = [&](int a, int part = 0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = m0 + wm * (TM * 32) + a * 32 + ps * 8 + erow;
            const size_t g = ig_elem<PL>((size_t)(m < M ? m : 0), nc, N);
            if (part != 2) rhA[a][ps] = *reinterpret_cast<const uint4*>(R + g);
            if (part == 1) continue;
            if (PL == 2) rlA[a][ps] = *reinterpret_cast<const uint4*>(R + g + 32);
            if (GATE == 2)                                                  // gate rows ride in rlA
                rlA[a][ps].x = reinterpret_cast<const unsigned char*>(Rg)[(size_t)(m < M ? m : 0) * (N >> 3) + (nc >> 3)];
            else if (GATE == 1)
                rlA[a][ps] = *reinterpret_cast<const uint4*>(Rg + g);
        }
    };
    // The first RD chunks are requested during the LAST k-step of the main loop: the block's epilogue no longer starts
    // with an exposed HBM round trip (four such rounds per CU on the 256->1024 shapes).  One plane only: with split
    // planes (and with a value gate) the 32 extra live registers spill inside the main loop.
    constexpr bool HOIST = XROWS && PL == 1 && GATE != 1 && !((GATE == 2 || STATS == 2) && TAPS == 9);   // (3x3 + bit gate: no registers left)
    constexpr bool HOIST_HI = RES && PL == 2 && TAPS == 1;   // split planes: the hi rows of the first chunk only (16 registers;
                                                            // the 3x3 variants have none to spare)
    // BatchNorm(eval) scale / shift of this lane's accumulator columns: fetched BEFORE the main loop (as the first thing
    // of the epilogue they cost every block an exposed memory round trip)
    constexpr int NCT = 4;                              // 16-wide column tiles of the wave tile
    float sc[NCT], sh[NCT];
#pragma unroll
    for (int b = 0; b < NCT; ++b) {
        const int n = n0 + wn * 64 + b * 16 + (lane & 15);
        sc[b] = 1.0f; sh[b] = 0.0f;
        if (GATE == 0 && !STATS && mean) {              // (the gradient / statistics variants are launched without BN)
            const float invstd = 1.0f / sqrtf(var[n] + eps);
            sc[b] = (gamma ? gamma[n] : 1.0f) * invstd;
            sh[b] = fmaf(-mean[n], sc[b], beta ? beta[n] : 0.0f);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) dma_a(0, 0, g, true);
#pragma unroll
    for (int g = 0; g < BG; ++g) dma_b(0, 0, g, true);
#pragma unroll
    for (int g = 0; g < 4; ++g) dma_a(1, 1, g, nk > 1);
    // Fragment addressing (LDS byte addresses for the asm reads).  Lane l holds row (l & 15), chunk 4*j + (l >> 4) of a
    // 16-row fragment: j = 0 | 1 is the first | second 32-deep half of a bf16 slab, or the hi | lo plane of a split slab.
    // Everything but the swizzled chunk has zero low 7 bits, so chunk j = 1 is the j = 0 address XOR 64, tile a (b) is
    // an instruction offset of a (b) * 2 KiB.
    const int r16 = lane & 15, kq = lane >> 4;
    const unsigned fswz = (unsigned)((kq ^ ((r16 >> 1) & 7)) << 4);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const unsigned fa0 = lds_base + (unsigned)((wm * (TM * 32) + r16) * 128) + fswz;
    const unsigned fb0 = lds_base + (unsigned)(NSA * A_BYTES + (wn * 64 + r16) * 128) + fswz;
    int sa = 0;                                          // A stage of k-step kt (kt % 3)
#ifdef IG_STAMP
    unsigned stamp_acc[5] = {0, 0, 0, 0, 0};
    const unsigned stamp_begin = ig_now();
#endif
    constexpr int NAe = 2 * TM;
    auto mfma_tile = [&](ig_bf16x8 (&fa_)[2], ig_bf16x8 (&fb_)[2], ig_f32x4& c) {
        if (PL == 2) {      // lo*hi + hi*lo + hi*hi
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[1], fb_[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[0], fb_[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[0], fb_[0], c, 0, 0, 0);
        } else {
            c = HT::mfma16(fa_[0], fb_[0], c);
            c = HT::mfma16(fa_[1], fb_[1], c);
        }
    };
    if (EB) {
        // ---- EARLY-BARRIER k-loop (round 5).  In the loop below every k-step starts with a barrier behind which ALL waves issue
        // their first fragment reads and wait an LDS latency: ~540 (16-bit) / ~810 (split planes) cycles per k-step in which no
        // wave of the CU issues an MFMA (in-kernel stamps: profiles/r05_igemm_stamps.txt).  Here the barrier of a k-step sits in
        // the MIDDLE of the previous one: behind it stage kt + 1 is known to have landed, so the LAST tile iteration of step kt
        // reads the first A tile of step kt + 1 through the A ring and refills each B fragment right after its last use — a step
        // begins with its operands in registers.  Stage lifetimes: the B fragments of a step are all in registers before the
        // step starts, so its LDS stage is dead at the next mid-step barrier and two B stages carry a prefetch distance of TWO
        // steps; A(kt + 2) goes into the stage step kt - 1 read, free at mid-step kt.  Everything a wave issues (second half of
        // a step) has to be landed by the next mid-step barrier: vmcnt(0), no counting.
        // The loop is ROTATED: its back edge sits behind the fragment wait of the mid-step tile, where no LDS read is in flight —
        // the fragment registers are loop-carried, and a register copy the compiler places on a back edge must never see a
        // register whose asm-issued load has not landed (the first form of this loop kept the prefetch in flight across the
        // edge: correct in most instantiations, stale fragments in the 64-column one).
        constexpr int NH = NAe / 2;
#pragma unroll
        for (int g = 0; g < BG; ++g) dma_b(1, 1, g, nk > 1);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        ig_bf16x8 fb[4][2], fa[2][2];
        fa[0][0] = ig_lds_read<0>(fa0);
        fa[0][1] = ig_lds_read<0>(fa0 ^ 64u);
        ig_lds_read4(fb, fb0);
        // tiles 0 .. NH - 1 of the step whose A stage is `st` (its operands' first fragments are in registers / in flight)
        auto half1 = [&](int st) __attribute__((always_inline)) {
            const unsigned ca = fa0 + (unsigned)(st * A_BYTES);
            if (NWV == 8) {
                if (wave < 4) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#pragma unroll
            for (int a = 0; a < NH; ++a) {
                const int cur = a & 1;
                IG_T(h0_);
                if (a == 0) ig_lds_wait_n<6>(fa[0], fb[0]);     // outstanding: B tiles 1..3 (issue order: A tile 0, B tiles 0..3)
                else ig_lds_wait_n<0>(fa[cur]);
                IG_T(h1_);
                if (a == 0) { IG_ACC(2, h0_, h1_); }            // (diagnostic build: cycles waiting for the prefetched first fragments)
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[0], acc4[a][0]);
                __builtin_amdgcn_sched_barrier(0);
                ig_lds_read_a(fa[cur ^ 1], ca, a + 1);          // (a + 1 <= NH < NAe)
                if (a == 0) ig_lds_wait_n<6>(fb[1]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[1], acc4[a][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (a == 0) ig_lds_wait_n<4>(fb[2]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[2], acc4[a][2]);
                __builtin_amdgcn_sched_barrier(0);
                if (a == 0) ig_lds_wait_n<2>(fb[3]);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[3], acc4[a][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // mid-step of step kt: this wave's pieces of stage kt + 1 (issued in the second half of step kt - 1) have landed, then
        // everyone's; everyone has finished step kt - 1: its A stage and the B stage of step kt are free.  Then the fragment
        // wait of tile NH: behind it nothing is in flight (the back edge)
        auto mid = [&](int kt) __attribute__((always_inline)) {
            IG_T(m0_);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            IG_T(m1_);
            IG_ACC(0, m0_, m1_);                            // (diagnostic build: cycles in the mid-step wait + barrier)
            if (HOIST && kt == nk - 1) {
#pragma unroll
                for (int r = 0; r < (RD < TM ? RD : TM); ++r) load_res(r);
            }
            if (HOIST_HI && kt == nk - 1) load_res(0, 1);
            if (NH == 0) ig_lds_wait_n<6>(fa[0], fb[0]);
            else ig_lds_wait_n<0>(fa[NH & 1]);
        };
        // tiles NH .. NAe - 1 of step kt (A stage st): the DMA of step kt + 2, and in the last tile the first fragments of step kt + 1
        auto half2 = [&](int kt, int st) __attribute__((always_inline)) {
            const int sb = kt & 1;
            const int st1 = st == 2 ? 0 : st + 1, st2 = st == 0 ? 2 : st - 1;      // A stages of steps kt + 1, kt + 2
            const bool more = kt + 2 < nk;
            const unsigned ca = fa0 + (unsigned)(st * A_BYTES), ca1 = fa0 + (unsigned)(st1 * A_BYTES);
            const unsigned cb1 = fb0 + (unsigned)((sb ^ 1) * B_BYTES);
            if (NWV == 8) {
                if (wave < 4) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(1);
            }
#pragma unroll
            for (int a = NH; a < NAe; ++a) {
                const int cur = a & 1;
                const bool last = a == NAe - 1;
                if (a != NH) ig_lds_wait_n<0>(fa[cur]);         // (tile NH: waited for in mid())
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[0], acc4[a][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (!last) ig_lds_read_a(fa[cur ^ 1], ca, a + 1);
                else {                                          // first A tile and first B tile of the NEXT step
                    ig_lds_read_a(fa[cur ^ 1], ca1, 0);
                    ig_lds_read_b(fb[0], cb1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[1], acc4[a][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (last) ig_lds_read_b(fb[1], cb1, 1);
#ifndef IG_ABL_NODMA
#pragma unroll
                for (int p = 0; p < NPIECE; ++p) {
                    if (NH + p * (NAe - NH) / NPIECE != a) continue;
                    if (p < BG) dma_b(kt + 2, sb, p, more);
                    else dma_a(kt + 2, st2, p - BG, more);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[2], acc4[a][2]);
                __builtin_amdgcn_sched_barrier(0);
                if (last) ig_lds_read_b(fb[2], cb1, 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile(fa[cur], fb[3], acc4[a][3]);
                __builtin_amdgcn_sched_barrier(0);
                if (last) ig_lds_read_b(fb[3], cb1, 3);
            }
        };
        static_assert(NH >= 1, "a wave tile has at least two 16-row tiles");
        half1(0);
        mid(0);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            half2(kt, sa);
            sa = sa == 2 ? 0 : sa + 1;
            half1(sa);
            mid(kt + 1);
        }
        half2(nk - 1, sa);
        // the prefetch of the step behind the last one is in flight into registers the epilogue is about to reuse
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[0][0]), "+v"(fb[0][1]),
                       "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]));
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int sb = kt & 1;
        IG_T(t0);
        // This wave's share of A(kt) and B(kt) has landed once all but its four youngest requests — A(kt+1) — are done
        // (vector-memory operations complete in issue order: a k-step issues B(kt+1) before A(kt+2)); then everyone's
        // has, and everyone has left A stage (kt+2) % 3 and B stage sb ^ 1 (last read in k-step kt-1).  A bare s_barrier:
        // the fragment reads are asm, the DMA waits are counted by hand, and behind __syncthreads() the compiler would
        // drain vmcnt to 0.
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        IG_T(t1);
        const int sa2 = sa == 0 ? 2 : sa - 1;             // (kt + 2) % 3
        const bool more_b = kt + 1 < nk, more_a = kt + 2 < nk;
        IG_T(t2);
        if (HOIST && kt == nk - 1) {
#pragma unroll
            for (int a = 0; a < (RD < TM ? RD : TM); ++a) load_res(a);
        }
        if (HOIST_HI && kt == nk - 1) load_res(0, 1);
        const unsigned ca = fa0 + (unsigned)(sa * A_BYTES), cb = fb0 + (unsigned)(sb * B_BYTES);
        sa = sa == 2 ? 0 : sa + 1;
        // B fragments of the whole k-step stay in registers (32); A fragments go through a ring of two 16-row tiles:
        // tile a + 1 is requested during the MFMAs of tile a and waited for after them.  Everything that is not an MFMA
        // (the next tile's reads, the DMA pieces with their address arithmetic) is issued BETWEEN the MFMAs of a tile, in
        // the shadow of the ones already in the pipe.  The first tile starts as soon as A tile 0 and B tile 0 are there
        // (counted lgkmcnt waits: LDS reads return in order, and the loop holds no other LGKM operation).
        ig_bf16x8 fb[4][2], fa[2][2];
        fa[0][0] = ig_lds_read<0>(ca);
        fa[0][1] = ig_lds_read<0>(ca ^ 64u);
        ig_lds_read4(fb, cb);
        // The two waves of a SIMD (w and w + 4) share its matrix pipe and the arbiter serves the older one first: left
        // alone, waves 0-3 ran ahead and waves 4-7 finished each k-step ~900 cycles later with the pipe 45 % busy.  The
        // priority is handed over in the middle of the k-step, so that both finish together.
        if (NWV == 8) {                                  // (4-wave blocks: one wave of the block per SIMD, nothing to hand over)
            if (wave < 4) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        IG_T(t3);
#ifdef IG_STAMP
        unsigned tile_t[NA + 1];
#endif
        auto mfma_col = [&](int a, int cur, int b) {
#ifndef IG_ABL_NOMFMA
            if (PL == 2) {      // lo*hi + hi*lo + hi*hi
                acc4[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][1], fb[b][0], acc4[a][b], 0, 0, 0);
                acc4[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][0], fb[b][1], acc4[a][b], 0, 0, 0);
                acc4[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][0], fb[b][0], acc4[a][b], 0, 0, 0);
            } else {
                acc4[a][b] = HT::mfma16(fa[cur][0], fb[b][0], acc4[a][b]);
                acc4[a][b] = HT::mfma16(fa[cur][1], fb[b][1], acc4[a][b]);
            }
#endif
        };
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int cur = a & 1;
#ifndef IG_FLIP8
#define IG_FLIP8 4
#endif
            if (NWV == 8 && a == NA * IG_FLIP8 / 8) {
                if (wave < 4) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(1);
            }
            if (a == 0) ig_lds_wait_n<6>(fa[0], fb[0]);         // outstanding: B tiles 1..3
            else ig_lds_wait_n<0>(fa[cur]);
#ifdef IG_STAMP
            tile_t[a] = ig_now();
#endif
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler otherwise sinks MFMA groups below the next wait)
            mfma_col(a, cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (a + 1 < NA) ig_lds_read_a(fa[cur ^ 1], ca, a + 1);
            if (a == 0) ig_lds_wait_n<6>(fb[1]);                // (+ A tile 1 behind them)
            __builtin_amdgcn_sched_barrier(0);
            mfma_col(a, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            // DMA of the next tiles, piece by piece (the CU's texture path takes 64 B per clock: issued in one burst the
            // eight pieces of a wave cost it ~1000 cycles without an MFMA, and the waves left the burst up to 1500
            // cycles apart, which the barrier then waited for): first B(kt+1), which must land within this k-step,
            // then A(kt+2).
#ifndef IG_ABL_NODMA
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                if (p * NA / NPIECE != a) continue;
                if (p < BG) dma_b(kt + 1, sb ^ 1, p, more_b);
                else dma_a(kt + 2, sa2, p - BG, more_a);
            }
#endif
            if (a == 0) ig_lds_wait_n<4>(fb[2]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_col(a, cur, 2);
            if (a == 0) ig_lds_wait_n<2>(fb[3]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_col(a, cur, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        IG_T(t4);
#ifdef IG_STAMP
        if (kt == 10 && lane == 0 && blockIdx.x + gridDim.x * blockIdx.y == 5) {      // timeline of one k-step, one block
            unsigned* o = ig_stamp_buf + 1023 * 64 + wave * 0;                          // (block 1023's slot is unused here)
            o = ig_stamp_buf + (1024 + wave) * 64;
            o[0] = t0; o[1] = t1; o[2] = t3;
            for (int a = 0; a < NA; ++a) o[3 + a] = tile_t[a];
            o[3 + NA] = t4;
        }
#endif
        IG_ACC(0, t0, t1); IG_ACC(1, t1, t2); IG_ACC(2, t2, t3); IG_ACC(3, t3, t4);
    }
#ifdef IG_STAMP
    {
        const unsigned stamp_end = ig_now();
        const int blk = blockIdx.x + gridDim.x * blockIdx.y;
        if (lane == 0 && blk < 1024) {
            unsigned* o = ig_stamp_buf + (blk * 8 + wave) * 8;
            o[0] = stamp_acc[0]; o[1] = stamp_acc[1]; o[2] = stamp_acc[2]; o[3] = stamp_acc[3];
            o[4] = stamp_end - stamp_begin; o[5] = (unsigned)nk; o[6] = stamp_begin;
        }
    }
#endif

    // ---- epilogue: each wave moves its 32 x 64 sub-tiles through a PRIVATE LDS tile (BN scale/shift applied on
    // the way in), then every lane owns 8 consecutive channels of a row: residual (+), ReLU, conversion and
    // 16-byte stores (hi and lo of a 32-channel slab together fill one 128-byte line).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the zero fills past the last k-step)
    __syncthreads();                                    // all waves are done with the operand tiles
    IG_T(stamp_epi);
#ifdef IG_STAMP
    unsigned chunk_t[TM];
#endif
    float* sW = reinterpret_cast<float*>(smem) + wave * (32 * EP);
    float st1[8], st2[8];                               // BatchNorm statistics of this lane's 8 channels (if asked for)
#pragma unroll
    for (int q = 0; q < 8; ++q) { st1[q] = 0.f; st2[q] = 0.f; }
    if (XROWS && !HOIST) {
#pragma unroll
        for (int a = 0; a < (RD < TM ? RD : TM); ++a) load_res(a, (HOIST_HI && a == 0) ? 2 : 0);
    }
    float bmean[8], binv[8], bgsc[8], bgsh[8];          // STATS == 2: this lane's 8 channels of the BN whose gradient this is
    if (STATS == 2) {
        const float4 m0_ = *reinterpret_cast<const float4*>(mean + nc), m1_ = *reinterpret_cast<const float4*>(mean + nc + 4);
        const float4 i0_ = *reinterpret_cast<const float4*>(var + nc), i1_ = *reinterpret_cast<const float4*>(var + nc + 4);
        const float* gp = gamma ? gamma : mean;
        const float* bp = beta ? beta : mean;
        const float4 g0_ = *reinterpret_cast<const float4*>(gp + nc), g1_ = *reinterpret_cast<const float4*>(gp + nc + 4);
        const float4 b0_ = *reinterpret_cast<const float4*>(bp + nc), b1_ = *reinterpret_cast<const float4*>(bp + nc + 4);
        const float mm[8] = {m0_.x, m0_.y, m0_.z, m0_.w, m1_.x, m1_.y, m1_.z, m1_.w};
        const float ii[8] = {i0_.x, i0_.y, i0_.z, i0_.w, i1_.x, i1_.y, i1_.z, i1_.w};
        const float gg[8] = {g0_.x, g0_.y, g0_.z, g0_.w, g1_.x, g1_.y, g1_.z, g1_.w};
        const float bb[8] = {b0_.x, b0_.y, b0_.z, b0_.w, b1_.x, b1_.y, b1_.z, b1_.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bmean[q] = mm[q]; binv[q] = ii[q];
            bgsc[q] = (gamma ? gg[q] : 1.0f) * ii[q];
            bgsh[q] = fmaf(-mm[q], bgsc[q], beta ? bb[q] : 0.0f);
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if (XROWS && a + RD < TM) load_res(a + RD);
        uint4 (&rh)[4] = rhA[a];
        uint4 (&rl4)[4] = rlA[a];
        // 16x16 tiles: column = lane & 15, rows 4*(lane >> 4) + r
#pragma unroll
        for (int ta2 = 0; ta2 < 2; ++ta2)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sW[(ta2 * 16 + 4 * (lane >> 4) + r) * EP + b * 16 + (lane & 15)] =
                        fmaf(acc4[2 * a + ta2][b][r], sc[b], sh[b]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int rl = ps * 8 + erow;
            const int m = m0 + wm * (TM * 32) + a * 32 + rl;
            float o[8];
            const float4 o0 = *reinterpret_cast<const float4*>(sW + rl * EP + ec8 * 8);
            const float4 o1 = *reinterpret_cast<const float4*>(sW + rl * EP + ec8 * 8 + 4);
            o[0] = o0.x; o[1] = o0.y; o[2] = o0.z; o[3] = o0.w; o[4] = o1.x; o[5] = o1.y; o[6] = o1.z; o[7] = o1.w;
            // plain variants (no residual / ReLU / gate / statistics) may be launched with a last column tile that hangs over
            // N (N % BN != 0: the weight rows behind N arrive as zeros by the buffer rule, nothing is stored for them)
            constexpr bool RAGGED_N = !RES && !RELU && GATE == 0 && STATS == 0;
            if (m < M && (!RAGGED_N || nc < N)) {
                if (RES) {
                    unsigned wh[4] = {rh[ps].x, rh[ps].y, rh[ps].z, rh[ps].w};
                    if (GATE != 0) {                     // keep a residual element only where its gate value is > 0
                        const unsigned wg[4] = {rl4[ps].x, rl4[ps].y, rl4[ps].z, rl4[ps].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool g0 = GATE == 2 ? ((wg[0] >> (2 * q)) & 1u) != 0u : HT::lo(wg[q]) > 0.f;
                            const bool g1 = GATE == 2 ? ((wg[0] >> (2 * q + 1)) & 1u) != 0u : HT::hi(wg[q]) > 0.f;
                            wh[q] = (g0 ? wh[q] & 0x0000FFFFu : 0u) | (g1 ? wh[q] & 0xFFFF0000u : 0u);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[2 * q] += HT::lo(wh[q]);
                        o[2 * q + 1] += HT::hi(wh[q]);
                    }
                    if (PL == 2) {
                        const unsigned wl[4] = {rl4[ps].x, rl4[ps].y, rl4[ps].z, rl4[ps].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            o[2 * q] += __uint_as_float(wl[q] << 16);
                            o[2 * q + 1] += __uint_as_float(wl[q] & 0xFFFF0000u);
                        }
                    }
                }
                if (RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
                }
                if (OUTF32) {
                    float* Y = reinterpret_cast<float*>(Yv) + (size_t)m * N + nc;
                    *reinterpret_cast<float4*>(Y) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(Y + 4) = make_float4(o[4], o[5], o[6], o[7]);
                } else {
                    unsigned short* Y = reinterpret_cast<unsigned short*>(Yv) + ig_elem<PL>((size_t)m, nc, N);
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
                        if (STATS == 1) {                // statistics of what is stored (the 16-bit roundings)
                            const float v0 = HT::dec(h0), v1 = HT::dec(h1);
                            st1[2 * q] += v0; st2[2 * q] = fmaf(v0, v0, st2[2 * q]);
                            st1[2 * q + 1] += v1; st2[2 * q + 1] = fmaf(v1, v1, st2[2 * q + 1]);
                        }
                        if (STATS == 2) {                // Σg, Σ g*xhat with g = stored gradient where bn(x) > 0
                            const unsigned xw = q == 0 ? rh[ps].x : (q == 1 ? rh[ps].y : (q == 2 ? rh[ps].z : rh[ps].w));
                            const float x0 = HT::lo(xw), x1 = HT::hi(xw);
                            const float v0 = HT::dec(h0), v1 = HT::dec(h1);
                            const float g0 = fmaf(x0, bgsc[2 * q], bgsh[2 * q]) > 0.f ? v0 : 0.f;
                            const float g1 = fmaf(x1, bgsc[2 * q + 1], bgsh[2 * q + 1]) > 0.f ? v1 : 0.f;
                            st1[2 * q] += g0; st2[2 * q] = fmaf(g0, (x0 - bmean[2 * q]) * binv[2 * q], st2[2 * q]);
                            st1[2 * q + 1] += g1;
                            st2[2 * q + 1] = fmaf(g1, (x1 - bmean[2 * q + 1]) * binv[2 * q + 1], st2[2 * q + 1]);
                        }
                    }
                    // streaming stores (common.h) — except in the split-plane 3x3 launches, which measured 3 % slower with them
#ifndef IG_NT_PL2_9
#define IG_NT_PL2_9 0
#endif
                    constexpr bool NT = !(PL == 2 && TAPS == 9) || IG_NT_PL2_9 != 0;
                    h_store16(Y, ph[0], ph[1], ph[2], ph[3], NT);
                    if (PL == 2) h_store16(Y + 32, pl_[0], pl_[1], pl_[2], pl_[3], NT);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef IG_STAMP
        chunk_t[a] = ig_now();
#endif
    }
#ifdef IG_STAMP
    {
        const int blk = blockIdx.x + gridDim.x * blockIdx.y;
        if (lane == 0 && (blk == 5 || blk == 600)) {
            unsigned* o = ig_stamp_buf + (1032 + (blk == 600 ? 8 : 0) + wave) * 64;
            o[0] = stamp_kernel; o[1] = stamp_begin; o[2] = stamp_epi;
            for (int a = 0; a < TM; ++a) o[3 + a] = chunk_t[a];
            o[3 + TM] = ig_now();
        }
    }
#endif
    if (STATS) {
        // fold the 8 row-lanes of each channel group (lane bits 3..5), then the WM waves that share these columns
        // (fixed order), and store this block's partial sums: the BN forward then needs no pass over Y for them
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                st1[q] += __shfl_xor(st1[q], o, 64);
                st2[q] += __shfl_xor(st2[q], o, 64);
            }
        }
        __syncthreads();                                // the private epilogue tiles are free now
        float* sS = reinterpret_cast<float*>(smem);     // [waves][64 channels][2]
        if (erow == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                sS[(wave * 64 + ec8 * 8 + q) * 2] = st1[q];
                sS[(wave * 64 + ec8 * 8 + q) * 2 + 1] = st2[q];
            }
        }
        __syncthreads();
        for (int e = tid; e < BN * 2; e += NWV * 64) {
            const int c = e >> 1, which = e & 1;        // column of the block tile
            const int wnc = c >> 6, cl = c & 63;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += sS[((w * WN + wnc) * 64 + cl) * 2 + which];
            stats[((size_t)bm * N + n0 + c) * 2 + which] = t;
        }
    }
}

}  // namespace hiast

// host: does this launch run on 128 x 128 tiles, two 4-wave blocks per CU (BM = 128 above)?  ONE rule for the launch and for the
// callers that size the per-block statistics buffers (hiast_igemm_stats_rows, hiast_igemm_dgrad_bn_stats_rows); the launch checks
// the caller's row count against the form it takes (HIAST_E_ARG on a mismatch).
// HIAST_IGEMM_HALF=0 / 1: never / every 1x1 launch with N % 128 == 0 (A/B and tests); unset: the HBM-bound shapes measured faster
// (profiles/r05_ab_igemm_half_tile.txt).  The tuning variables are read ONCE per process (ig_env below): a value that changes
// between the row-count call and the launch would size `partial` for one tile form and run the other.
struct IgEnv {
    int half;        // HIAST_IGEMM_HALF: -1 unset | 0 | 1
    int bn;          // HIAST_IGEMM_BN: 0 unset | 64 | 128 | 256
    int ragged_off;  // HIAST_IGEMM_RAGGED set: no ragged 256-column tiles for the plain N = 640 GEMM
    int cosched0;    // HIAST_IGEMM_COSCHED: the process default of the co-scheduling hint
};
const IgEnv& hiast_ig_env();                             // igemm.hip
// The co-scheduling hint (hiast_igemm_set_cosched, include/hiast_hip.h): "this thread runs two launch sequences side by side" —
// two half-chip launches on two streams already fill the chip, so they keep the 256-row form.  THREAD-LOCAL state of the calling
// thread (-1 = the process default above), never the environment: setenv / unsetenv around every forward raced with getenv in
// other threads.
int hiast_ig_cosched_tls_get();                          // igemm.hip
int hiast_ig_half_tls_get();                             // igemm.hip: hiast_igemm_set_half of the calling thread (-1 = HIAST_IGEMM_HALF / automatic)
static inline int ig_half_forced()
{
    const int t = hiast_ig_half_tls_get();
    return t >= 0 ? t : hiast_ig_env().half;
}
static inline int ig_cosched()
{
    const int t = hiast_ig_cosched_tls_get();
    return t >= 0 ? t : hiast_ig_env().cosched0;
}
static inline int ig_half_tile(int64_t M, int K, int N, int taps, int out_f32)
{
    // (1x1 only: on the 3x3 launches the form was measured 20-25 % slower wherever the 256-row form fills the chip — twice the
    // L2 -> LDS bytes and a third more fragment reads per flop in an MFMA-bound loop — profiles/r05_ab_igemm_half_tile.txt)
    if (taps != 1 || out_f32 || N % 128 != 0 || M < 4096) return 0;
    if (ig_half_forced() >= 0) return ig_half_forced();
    return IG_HALF_AUTO(M, K, N);
}
// ... and the split-plane 3x3 + BN + ReLU launch of the pseudo-label forward in the same form when its 256-row form fills at most a
// quarter of the chip (the generator at the reference's batch size 2: 64 tiles): stand-alone 144 -> 103 us already at 4 images
// (profiles/r05_ab_igemm_half_tile.txt); on full-chip launches the form is 20-25 % slower, so never there.
static inline int ig_half_tile9(int64_t M, int N, int PL, bool bn_relu_only)
{
    if (PL != 2 || !bn_relu_only || N % 128 != 0 || M < 4096) return 0;
    if (ig_half_forced() >= 0) return ig_half_forced();
    return IG_HALF_AUTO(M, 0, N);
}
static inline int ig_block_rows(int64_t M, int K, int N, int taps, int out_f32)
{
    return ig_half_tile(M, K, N, taps, out_f32) ? 128 : hiast::IG_BM;
}

template <int PL, bool OUTF32, bool F16 = false>
static int launch_igemm_t(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                          const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                          int taps, hiast::IGeo geo, float* stats, const void* res_gate, int gate_mask, hipStream_t st,
                          int stats_mode, int stats_rows)
{
    int BN = (N % 256 == 0) ? 256 : ((N % 128 == 0) ? 128 : 64);
    // a plain GEMM (no BN / residual / ReLU / gate / statistics) of N = 2.5, 3.5, ... tiles of 256 runs on 256-column tiles
    // with the last one hanging over N (zero weight rows from the buffer rule, no stores): the ASPP tap GEMM, N = 640, 16-bit
    // output — 0.170 against 0.181 ms on five 128-column tiles although a fifth of the last tile's MFMAs multiply zeros (with
    // fp32 output the 128-column tiles win: 0.187 against 0.197)
    const bool plain = !mean && !gamma && !res && !relu && !stats && !res_gate && stats_mode == 0;
    if (plain && !OUTF32 && BN == 128 && N > 512 && !hiast_ig_env().ragged_off) BN = 256;
    if (const int v = hiast_ig_env().bn) {                      // tuning override (HIAST_IGEMM_BN)
        if (N % v == 0) BN = v;
    }
    const bool half9 = taps == 9 && !OUTF32 && geo.stride == 1 &&
                       ig_half_tile9(M, N, PL, mean && relu && !res && !stats && !res_gate && stats_mode == 0) != 0;
    const bool half = half9 || (geo.stride >= 0 && ig_half_tile(M, K, N, taps, OUTF32 ? 1 : 0) != 0);
    if (half) BN = 128;
    const int BMr = half ? 128 : hiast::IG_BM;
    dim3 grid((unsigned)((M + BMr - 1) / BMr), (N + BN - 1) / BN);
    const dim3 block((unsigned)(BMr * 2));
    // the statistics epilogues write one row of sums per block row of THIS tile form: the caller's buffer has to be that long
    if (stats && stats_rows != (int)grid.x) return HIAST_E_ARG;
    const int gate = !res_gate ? 0 : (gate_mask ? 2 : 1);
    if (geo.stride < 0) {                                   // transposed stride-2 3x3 (UPS): plain 16-bit launches only
        if constexpr (PL == 1 && !OUTF32) {
            if (taps != 9 || !plain || N % BN != 0) return HIAST_E_RANGE;
#define LU(BNV)                                                                                                        \
    hipLaunchKernelGGL((hiast::igemm_bn_act_kernel<PL, OUTF32, BNV, 9, false, false, 0, 0, F16, true>), grid, dim3(512), 0, st, \
                       (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,                \
                       (const unsigned short*)res, y, (int)M, K, N, geo, stats, (const unsigned short*)res_gate)
            if (BN == 256) LU(256); else if (BN == 128) LU(128); else LU(64);
#undef LU
            HIAST_CHECK_LAUNCH();
            return 0;
        } else {
            return HIAST_E_RANGE;
        }
    }
#define LK(BNV, T, RES, RELU, G, S, BMV)                                                                              \
    hipLaunchKernelGGL((hiast::igemm_bn_act_kernel<PL, OUTF32, BNV, T, RES, RELU, G, S, F16, false, BMV>), grid, block, 0, st, \
                       (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,              \
                       (const unsigned short*)res, y, (int)M, K, N, geo, stats, (const unsigned short*)res_gate)
#define LG(BNV, T, BMV)                                                                 \
    if constexpr (PL == 1 && !OUTF32) {                                                 \
        if (gate == 1) LK(BNV, T, true, false, 1, 0, BMV); else LK(BNV, T, true, false, 2, 0, BMV); \
    }
#define LS(BNV, T, BMV)                                                                 \
    if constexpr (PL == 1 && !OUTF32) {                                                 \
        if (stats_mode == 2) LK(BNV, T, false, false, 0, 2, BMV);                       \
        else LK(BNV, T, false, false, 0, 1, BMV);                                       \
    }
#define LL(BNV, T, BMV)                                                                 \
    if (stats_mode == 2) { LS(BNV, T, BMV) }                                            \
    else if (res) {                                                                     \
        if (relu) LK(BNV, T, true, true, 0, 0, BMV);                                    \
        else if (gate == 0) LK(BNV, T, true, false, 0, 0, BMV);                         \
        else { LG(BNV, T, BMV) }                                                        \
    } else if (relu) LK(BNV, T, false, true, 0, 0, BMV);                                \
    else if (!stats) LK(BNV, T, false, false, 0, 0, BMV);                               \
    else { LS(BNV, T, BMV) }
    if (half9) {
        if constexpr (!OUTF32 && PL == 2) LK(128, 9, false, true, 0, 0, 128);
    } else if (half) {
        if constexpr (!OUTF32) { LL(128, 1, 128) }
    } else if (taps == 1) {
        if (BN == 256) { LL(256, 1, 256) } else if (BN == 128) { LL(128, 1, 256) } else { LL(64, 1, 256) }
    } else {
        if (BN == 256) { LL(256, 9, 256) } else if (BN == 128) { LL(128, 9, 256) } else { LL(64, 9, 256) }
    }
#undef LL
#undef LS
#undef LG
#undef LK
    HIAST_CHECK_LAUNCH();
    return 0;
}
