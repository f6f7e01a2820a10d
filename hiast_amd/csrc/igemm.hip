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
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 ig_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ig_f32x16;
typedef __attribute__((ext_vector_type(4))) float ig_f32x4;

constexpr int IG_BM = 256;

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
template <int PL, bool OUTF32, int BN, int TAPS, bool RES, bool RELU, int GATE = 0, int STATS = 0>
__global__ __launch_bounds__(512) void igemm_bn_act_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, float eps,
    const unsigned short* __restrict__ R, void* __restrict__ Yv, int M, int K, int N, IGeo geo,
    float* __restrict__ stats,           // optional [gridDim m-blocks][N][2]: per-block Σy, Σy² of the STORED values
    const unsigned short* __restrict__ Rg)   // GATE != 0: o += gate ? R : 0 (the ReLU-masked gradient of an identity branch)
{
    static_assert(GATE == 0 || (PL == 1 && RES && !RELU && !OUTF32), "gated residual: bf16 data-gradient launches only");
    constexpr int WN = BN / 64, WM = 8 / WN;
    static_assert(!STATS || (PL == 1 && !OUTF32 && !RES && !RELU && GATE == 0), "statistics epilogue: plain bf16 launches only");
    constexpr bool XROWS = RES || STATS == 2;           // the epilogue reads rows of R (residual | BN input)
    constexpr int TM = IG_BM / WM / 32;                 // 32-row tiles per wave (4 | 2 | 1)
    constexpr int A_BYTES = IG_BM * 128, B_BYTES = BN * 128;
    // LDS: THREE stages of the activation tile and TWO of the weight tile (160 KiB at BN = 256).  The fill rate of a tile
    // is (bytes in flight) / latency, and with one 64 KiB tile in flight per CU the launches were bound by exactly that
    // (removing every MFMA left the kernel time unchanged): the activation rows — first touched in HBM — are requested
    // two k-steps ahead, the weight rows (L2 hits, the same for every block) one.
    constexpr int NSA = 3, NSB = 2;
    constexpr int LDS_BYTES = NSA * A_BYTES + NSB * B_BYTES;
    constexpr int BG = BN / 64;                         // 8-row B groups per wave (4 | 2 | 1)
    constexpr int EP = 68;                              // padded row of a wave's private epilogue tile (floats)
    static_assert(LDS_BYTES >= 8 * 32 * EP * 4, "epilogue staging must fit in the tile buffers");
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
    const int m0 = bm * IG_BM, n0 = bn_ * BN;
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
            ay[g] = (r / geo.Wo) * geo.stride;
            ax[g] = (r - (r / geo.Wo) * geo.Wo) * geo.stride;
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
            voff = (aok[g] & on) ? an[g] * (KS * 128) + achunk[g] : OOB;
        } else {
            const int yy = ay[g] + (tap / 3 - 1) * geo.dil, xx = ax[g] + (tap % 3 - 1) * geo.dil;
            const bool ok = aok[g] & on & ((unsigned)yy < (unsigned)geo.H) & ((unsigned)xx < (unsigned)geo.W);
            const int pix = (an[g] * geo.H + yy) * geo.W + xx;
            voff = ok ? pix * (KS * 128) + achunk[g] : OOB;
        }
        ig_dma16(xrs, smem + sa * A_BYTES + (4 * wave + g) * 1024, voff, j * 128);
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
        if (wave < 4) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
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
                acc4[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][0], fb[b][0], acc4[a][b], 0, 0, 0);
                acc4[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][1], fb[b][1], acc4[a][b], 0, 0, 0);
            }
#endif
        };
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int cur = a & 1;
#ifndef IG_FLIP8
#define IG_FLIP8 4
#endif
            if (a == NA * IG_FLIP8 / 8) {
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
            if (m < M) {
                if (RES) {
                    unsigned wh[4] = {rh[ps].x, rh[ps].y, rh[ps].z, rh[ps].w};
                    if (GATE != 0) {                     // keep a residual element only where its gate value is > 0
                        const unsigned wg[4] = {rl4[ps].x, rl4[ps].y, rl4[ps].z, rl4[ps].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool g0 = GATE == 2 ? ((wg[0] >> (2 * q)) & 1u) != 0u : __uint_as_float(wg[q] << 16) > 0.f;
                            const bool g1 = GATE == 2 ? ((wg[0] >> (2 * q + 1)) & 1u) != 0u
                                                      : __uint_as_float(wg[q] & 0xFFFF0000u) > 0.f;
                            wh[q] = (g0 ? wh[q] & 0x0000FFFFu : 0u) | (g1 ? wh[q] & 0xFFFF0000u : 0u);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[2 * q] += __uint_as_float(wh[q] << 16);
                        o[2 * q + 1] += __uint_as_float(wh[q] & 0xFFFF0000u);
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
                        unsigned short h0, l0, h1, l1;
                        ig_split(o[2 * q], h0, l0);
                        ig_split(o[2 * q + 1], h1, l1);
                        ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
                        pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
                        if (STATS == 1) {                // statistics of what is stored (the bf16 roundings)
                            const float v0 = __uint_as_float((unsigned)h0 << 16), v1 = __uint_as_float((unsigned)h1 << 16);
                            st1[2 * q] += v0; st2[2 * q] = fmaf(v0, v0, st2[2 * q]);
                            st1[2 * q + 1] += v1; st2[2 * q + 1] = fmaf(v1, v1, st2[2 * q + 1]);
                        }
                        if (STATS == 2) {                // Σg, Σ g*xhat with g = stored gradient where bn(x) > 0
                            const unsigned xw = q == 0 ? rh[ps].x : (q == 1 ? rh[ps].y : (q == 2 ? rh[ps].z : rh[ps].w));
                            const float x0 = __uint_as_float(xw << 16), x1 = __uint_as_float(xw & 0xFFFF0000u);
                            const float v0 = __uint_as_float((unsigned)h0 << 16), v1 = __uint_as_float((unsigned)h1 << 16);
                            const float g0 = fmaf(x0, bgsc[2 * q], bgsh[2 * q]) > 0.f ? v0 : 0.f;
                            const float g1 = fmaf(x1, bgsc[2 * q + 1], bgsh[2 * q + 1]) > 0.f ? v1 : 0.f;
                            st1[2 * q] += g0; st2[2 * q] = fmaf(g0, (x0 - bmean[2 * q]) * binv[2 * q], st2[2 * q]);
                            st1[2 * q + 1] += g1;
                            st2[2 * q + 1] = fmaf(g1, (x1 - bmean[2 * q + 1]) * binv[2 * q + 1], st2[2 * q + 1]);
                        }
                    }
                    *reinterpret_cast<uint4*>(Y) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                    if (PL == 2) *reinterpret_cast<uint4*>(Y + 32) = make_uint4(pl_[0], pl_[1], pl_[2], pl_[3]);
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
        float* sS = reinterpret_cast<float*>(smem);     // [8 waves][64 channels][2]
        if (erow == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                sS[(wave * 64 + ec8 * 8 + q) * 2] = st1[q];
                sS[(wave * 64 + ec8 * 8 + q) * 2 + 1] = st2[q];
            }
        }
        __syncthreads();
        for (int e = tid; e < BN * 2; e += 512) {
            const int c = e >> 1, which = e & 1;        // column of the block tile
            const int wnc = c >> 6, cl = c & 63;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += sS[((w * WN + wnc) * 64 + cl) * 2 + which];
            stats[((size_t)bm * N + n0 + c) * 2 + which] = t;
        }
    }
}

// fp32 conv weight [N][K][taps] (torch layout; taps = kh*kw) -> Wp[N][taps][slabs of 128 B] bf16.
// transpose: pack the weight of the ADJOINT convolution instead (the data-gradient of a stride-1 conv is a conv of
// dY with the channel-transposed, spatially flipped kernel): Wp[k][taps-1-t][.. n ..] = w[n][k][t]; the output then
// has N_out = K rows and K_out = N reduction channels.
// element offset of (row n, tap t, reduction channel k) in a packed weight with KO reduction channels
template <int PL>
__device__ __forceinline__ size_t ig_wp_elem(int n, int t, int k, int taps, int KO)
{
    const size_t row = ((size_t)n * taps + t) * PL * KO;
    return PL == 2 ? row + (size_t)(k >> 5) * 64 + (k & 31) : row + k;      // lo plane: + 32
}

// mode 0: wp = forward weight; 1: wp = adjoint weight; 2: wp = forward, wpt = adjoint (one pass over w)
template <int PL>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w,
                                                               unsigned short* __restrict__ wp,
                                                               unsigned short* __restrict__ wpt, int N, int K, int taps,
                                                               int mode)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * taps * K) return;
    const int k = (int)(idx % K);
    const int t = (int)((idx / K) % taps);
    const int n = (int)(idx / ((long long)K * taps));
    const float v = w[((size_t)n * K + k) * taps + t];
    unsigned short h, l;
    ig_split(v, h, l);
    if (mode != 1) {
        unsigned short* d = wp + ig_wp_elem<PL>(n, t, k, taps, K);
        d[0] = h;
        if (PL == 2) d[32] = l;
    }
    if (mode != 0) {                                     // adjoint: rows = k, reduction = n, taps flipped
        unsigned short* d = (mode == 1 ? wp : wpt) + ig_wp_elem<PL>(k, taps - 1 - t, n, taps, N);
        d[0] = h;
        if (PL == 2) d[32] = l;
    }
}

// Tiled form for channel counts that are multiples of 64 (every trunk convolution, the ASPP tap matrices): one block
// takes the 64 (n) x 64 (k) x taps tile of one tensor through LDS, so that the source is read in contiguous runs of
// 64*taps floats per n and BOTH packed forms leave as whole 128-byte lines (forward: 64 k of one (n, tap); adjoint:
// 64 n of one (k, flipped tap), written as two 64-byte halves by consecutive passes of the same block).  The elementwise kernel above issued 2-byte stores 2*N (adjoint) bytes apart and
// reached 0.47 TB/s — 0.73 ms per trunk, three trunks per training step once the packs follow every optimiser / EMA
// step.  PL = 2 walks the n range in two halves (a 32-channel slab holds hi|lo: the same 128-byte line).
constexpr int PK_PITCH = 72;        // ushorts per (n, tap) row of the LDS tile: 64 + 8 (keeps 16-byte alignment)

template <int PL>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, unsigned short* __restrict__ wp,
                                          unsigned short* __restrict__ wpt, int N, int K, int taps, int mode, int n0,
                                          int k0, unsigned short* lds)
{
    constexpr int TN = PL == 1 ? 32 : 16;               // rows of n per pass: LDS = TN x taps x 72 x 2 B x PL (41 KiB at 9 taps,
    unsigned short* hi = lds;                           // 4.6 KiB at 1: several blocks per CU instead of one)
    unsigned short* lo = lds + (PL == 2 ? TN * taps * PK_PITCH : 0);
    unsigned short* fwd = (mode == 1) ? nullptr : wp;
    unsigned short* adj = (mode == 0) ? nullptr : (mode == 1 ? wp : wpt);
    const int run = 64 * taps;                                   // contiguous floats per n in the source
    for (int nh = 0; nh < 64; nh += TN) {
        if (nh) __syncthreads();
        for (int e = threadIdx.x; e < TN * run; e += 256) {
            const int n = e / run, r = e - n * run;
            const int k = r / taps, t = r - k * taps;
            const float v = w[((size_t)(n0 + nh + n) * K + k0) * taps + r];
            unsigned short h, l;
            ig_split(v, h, l);
            hi[(n * taps + t) * PK_PITCH + k] = h;
            if (PL == 2) lo[(n * taps + t) * PK_PITCH + k] = l;
        }
        __syncthreads();
        if (fwd) {
            for (int e = threadIdx.x; e < TN * taps * 8; e += 256) {       // (n, t, 8-channel chunk)
                const int c = e & 7, nt = e >> 3;
                const int n = nt / taps, t = nt - n * taps;
                const uint4 vh = *reinterpret_cast<const uint4*>(&hi[nt * PK_PITCH + c * 8]);
                unsigned short* d = fwd + ig_wp_elem<PL>(n0 + nh + n, t, k0 + c * 8, taps, K);
                *reinterpret_cast<uint4*>(d) = vh;
                if (PL == 2) *reinterpret_cast<uint4*>(d + 32) = *reinterpret_cast<const uint4*>(&lo[nt * PK_PITCH + c * 8]);
            }
        }
        if (adj) {
            for (int e = threadIdx.x; e < 64 * taps * (TN / 8); e += 256) {     // (k, t, 8-row chunk of n)
                const int c = e % (TN / 8), kt = e / (TN / 8);
                const int k = kt / taps, t = kt - k * taps;
                unsigned short vh[8], vl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vh[j] = hi[((c * 8 + j) * taps + t) * PK_PITCH + k];
                    if (PL == 2) vl[j] = lo[((c * 8 + j) * taps + t) * PK_PITCH + k];
                }
                unsigned short* d = adj + ig_wp_elem<PL>(k0 + k, taps - 1 - t, n0 + nh + c * 8, taps, N);
                *reinterpret_cast<uint4*>(d) = make_uint4(vh[0] | (unsigned)vh[1] << 16, vh[2] | (unsigned)vh[3] << 16,
                                                          vh[4] | (unsigned)vh[5] << 16, vh[6] | (unsigned)vh[7] << 16);
                if (PL == 2)
                    *reinterpret_cast<uint4*>(d + 32) = make_uint4(vl[0] | (unsigned)vl[1] << 16, vl[2] | (unsigned)vl[3] << 16,
                                                                   vl[4] | (unsigned)vl[5] << 16, vl[6] | (unsigned)vl[7] << 16);
            }
        }
    }
}

template <int PL>
__global__ __launch_bounds__(256) void pack_conv_weight_tiled_kernel(const float* __restrict__ w,
                                                                     unsigned short* __restrict__ wp,
                                                                     unsigned short* __restrict__ wpt, int N, int K,
                                                                     int taps, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];       // 32 * taps * PK_PITCH
    const int kt = K / 64;
    pack_tile<PL>(w, wp, wpt, N, K, taps, mode, (int)(blockIdx.x / kt) * 64, (int)(blockIdx.x % kt) * 64, lds);
}

// the same for a LIST of weights in one launch (one ResNet trunk = 104 convolutions whose packed forms are rebuilt
// after every optimiser / EMA step): block = one 64 x 64 (n, k) tile of one tensor; chunk_start = the tile's index
// within its tensor, row-major over (N/64, K/64)
__global__ __launch_bounds__(256) void pack_conv_weight_multi_kernel(const hiast_pack_rec* __restrict__ table,
                                                                     const int32_t* __restrict__ chunk_tensor,
                                                                     const int64_t* __restrict__ chunk_start)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];       // 32 * max taps of the launch * PK_PITCH
    const hiast_pack_rec r = table[chunk_tensor[blockIdx.x]];
    const int kt = r.K / 64;
    const int tile = (int)chunk_start[blockIdx.x];
    const int n0 = (tile / kt) * 64, k0 = (tile % kt) * 64;
    if (r.planes == 2)
        pack_tile<2>(r.w, (unsigned short*)r.wp, (unsigned short*)r.wpt, r.N, r.K, r.taps, r.mode, n0, k0, lds);
    else
        pack_tile<1>(r.w, (unsigned short*)r.wp, (unsigned short*)r.wpt, r.N, r.K, r.taps, r.mode, n0, k0, lds);
}

// fp32 [M][C] -> split planes (and back: v = hi + lo, exact in fp32)
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ p,
                                                        long long M, int C)
{
    const long long total8 = M * C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const long long m = (i * 8) / C;
        const int c = (int)((i * 8) - m * C);
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned ph[4], pl_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h0, l0, h1, l1;
            ig_split(v[2 * q], h0, l0);
            ig_split(v[2 * q + 1], h1, l1);
            ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
            pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* dst = p + ig_elem<2>((size_t)m, c, C);
        *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        *reinterpret_cast<uint4*>(dst + 32) = make_uint4(pl_[0], pl_[1], pl_[2], pl_[3]);
    }
}

__global__ __launch_bounds__(256) void from_planes_kernel(const unsigned short* __restrict__ p, float* __restrict__ x,
                                                          long long M, int C)
{
    const long long total8 = M * C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const long long m = (i * 8) / C;
        const int c = (int)((i * 8) - m * C);
        const unsigned short* src = p + ig_elem<2>((size_t)m, c, C);
        const uint4 h = *reinterpret_cast<const uint4*>(src), l = *reinterpret_cast<const uint4*>(src + 32);
        const unsigned wh[4] = {h.x, h.y, h.z, h.w}, wl[4] = {l.x, l.y, l.z, l.w};
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[2 * q] = __uint_as_float(wh[q] << 16) + __uint_as_float(wl[q] << 16);
            v[2 * q + 1] = __uint_as_float(wh[q] & 0xFFFF0000u) + __uint_as_float(wl[q] & 0xFFFF0000u);
        }
        reinterpret_cast<float4*>(x)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(x)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace hiast

template <int PL, bool OUTF32>
static int launch_igemm_t(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                          const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                          int taps, hiast::IGeo geo, float* stats, const void* res_gate, int gate_mask, hipStream_t st,
                          int stats_mode)
{
    int BN = (N % 256 == 0) ? 256 : ((N % 128 == 0) ? 128 : 64);
    if (const char* env = getenv("HIAST_IGEMM_BN")) {          // tuning override
        const int v = atoi(env);
        if ((v == 64 || v == 128 || v == 256) && N % v == 0) BN = v;
    }
    dim3 grid((unsigned)((M + hiast::IG_BM - 1) / hiast::IG_BM), N / BN);
    const int gate = !res_gate ? 0 : (gate_mask ? 2 : 1);
#define L(BNV, T, RES, RELU, G)                                                                                      \
    hipLaunchKernelGGL((hiast::igemm_bn_act_kernel<PL, OUTF32, BNV, T, RES, RELU, G>), grid, dim3(512), 0, st,    \
                       (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,              \
                       (const unsigned short*)res, y, (int)M, K, N, geo, stats, (const unsigned short*)res_gate)
#define LG(BNV, T)                                                                      \
    if constexpr (PL == 1 && !OUTF32) {                                                 \
        if (gate == 1) L(BNV, T, true, false, 1); else L(BNV, T, true, false, 2);       \
    }
#define LS(BNV, T)                                                                      \
    if constexpr (PL == 1 && !OUTF32) {                                                 \
        if (stats_mode == 2)                                                            \
        hipLaunchKernelGGL((hiast::igemm_bn_act_kernel<PL, OUTF32, BNV, T, false, false, 0, 2>), grid, dim3(512), 0, st, \
                           (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,                     \
                           (const unsigned short*)res, y, (int)M, K, N, geo, stats, (const unsigned short*)res_gate);       \
        else                                                                            \
        hipLaunchKernelGGL((hiast::igemm_bn_act_kernel<PL, OUTF32, BNV, T, false, false, 0, 1>), grid, dim3(512), 0, st, \
                           (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,                     \
                           (const unsigned short*)res, y, (int)M, K, N, geo, stats, (const unsigned short*)res_gate);       \
    }
#define LL(BNV, T)                                                                      \
    if (stats_mode == 2) { LS(BNV, T) }                                                 \
    else if (res) {                                                                     \
        if (relu) L(BNV, T, true, true, 0);                                             \
        else if (gate == 0) L(BNV, T, true, false, 0);                                  \
        else { LG(BNV, T) }                                                             \
    } else if (relu) L(BNV, T, false, true, 0);                                         \
    else if (!stats) L(BNV, T, false, false, 0);                                        \
    else { LS(BNV, T) }
#define LLL                                                                                                 \
    if (taps == 1) { if (BN == 256) { LL(256, 1) } else if (BN == 128) { LL(128, 1) } else { LL(64, 1) } }  \
    else { if (BN == 256) { LL(256, 9) } else if (BN == 128) { LL(128, 9) } else { LL(64, 9) } }
    LLL
#undef LLL
#undef LL
#undef LS
#undef LG
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}

// K9e (xconv.hip): register-resident-weight kernel for the HBM-bound expanding 1x1 shapes
int hiast_xconv_ok(int64_t M, int K, int N, int planes, int taps, int out_f32, int has_bn, int has_res, int relu,
                   int has_gate, int gate_mask, int has_stats);
int hiast_xconv_stats_rows(int64_t M, int N);
int hiast_xconv_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       float* stats, const void* res_gate, hipStream_t st);

// stats_mode: 0 none, 1 forward BatchNorm sums of the output (stats), 2 backward BatchNorm sums of the activation whose
// gradient the output is (data-gradient launches: res = that BN's input x, mean / var = its batch mean / invstd)
static int igemm_launch_mode(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                             const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                             int taps, int H, int W, int stride, int dil, int planes, int out_f32, hipStream_t st,
                             float* stats, const void* res_gate, int gate_mask, int stats_mode)
{
    if (stats_mode == 2) {
        if (!stats || !res || !mean || !var || planes != 1 || out_f32 || relu || res_gate || N % 8) return HIAST_E_RANGE;
    } else {
        if (stats && (planes != 1 || out_f32 || res || relu)) return HIAST_E_RANGE;
        if ((stats || res_gate) && mean) return HIAST_E_RANGE;   // the statistics / gradient variants carry no BatchNorm
    }
    if (res_gate && (planes != 1 || !res || relu || out_f32 || (!gate_mask && (((uintptr_t)res_gate) & 15)))) return HIAST_E_RANGE;
    if (!x || !wp || !y || (mean && !var)) return HIAST_E_ARG;
    if (M <= 0 || K <= 0 || N <= 0) return HIAST_E_ARG;
    if ((planes != 1 && planes != 2) || (taps != 1 && taps != 9)) return HIAST_E_RANGE;
    if ((K * planes) % 64 != 0 || N % 64 != 0 || M > (1ll << 31) - 512) return HIAST_E_RANGE;
    if (out_f32 && res) return HIAST_E_RANGE;
    if ((((uintptr_t)x) | ((uintptr_t)wp) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) return HIAST_E_RANGE;
    hiast::IGeo geo = {0, 0, 0, 0, 1, 1};
    size_t in_pix = (size_t)M;
    if (taps != 1) {
        if (H <= 0 || W <= 0 || stride <= 0 || dil <= 0) return HIAST_E_ARG;
        const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
        if (M % ((int64_t)Ho * Wo) != 0) return HIAST_E_ARG;
        geo = {H, W, Ho, Wo, stride, dil};
        in_pix = (size_t)(M / ((int64_t)Ho * Wo)) * H * W;
    }
    // buffer-descriptor addressing: byte offsets and the out-of-range marker need 31 bits
    if (in_pix * planes * K * 2 >= (1ull << 31) || (size_t)N * taps * planes * K * 2 >= (1ull << 31)) return HIAST_E_RANGE;
    if (stats_mode != 2 &&
        hiast_xconv_ok(M, K, N, planes, taps, out_f32, mean != nullptr, res != nullptr, relu, res_gate != nullptr, gate_mask,
                       stats != nullptr))
        return hiast_xconv_launch(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, stats, res_gate, st);
    if (planes == 2) {
        if (out_f32) return launch_igemm_t<2, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode);
        return launch_igemm_t<2, false>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode);
    }
    if (out_f32) return launch_igemm_t<1, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode);
    return launch_igemm_t<1, false>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode);
}

// shared launcher (also used by aspp2.hip for the ASPP tap GEMM)
int hiast_igemm_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       int taps, int H, int W, int stride, int dil, int planes, int out_f32, hipStream_t st,
                       float* stats, const void* res_gate, int gate_mask)
{
    return igemm_launch_mode(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, H, W, stride, dil, planes,
                             out_f32, st, stats, res_gate, gate_mask, stats ? 1 : 0);
}

// Data gradient of a stride-1 trunk convolution (dy x the adjoint weight, hiast_pack_conv_weight transpose) whose output
// dA is the gradient of A = relu(bn(x)): also emits the per-block sums (Σg, Σ g*xhat) of that BatchNorm's backward,
// partial[hiast_igemm_stats_rows][Cout][2] — hiast_bn_nhwc_stats_from_partial turns them into the sums that
// hiast_bn_nhwc_bwd_apply takes (reference: autograd of conv -> bn -> relu in Bottleneck.forward, resnet.py:78-98).
extern "C" int hiast_igemm_dgrad_bn_stats(const void* dy, const void* wpt, void* da, int B, int H, int W, int Cin, int Cout,
                                          int taps, int dil, const void* bn_x, const float* gamma, const float* beta,
                                          const float* save_mean, const float* save_invstd, float* partial,
                                          hiast_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (!bn_x || !save_mean || !save_invstd || !partial) return HIAST_E_ARG;
    if ((((uintptr_t)save_mean) | ((uintptr_t)save_invstd) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) return HIAST_E_RANGE;
    return igemm_launch_mode(dy, wpt, gamma, beta, save_mean, save_invstd, 0.0f, bn_x, 0, da, (int64_t)B * H * W, Cin, Cout,
                             taps, H, W, 1, dil, 1, 0, (hipStream_t)stream, partial, nullptr, 0, 2);
}

extern "C" int hiast_igemm_dgrad_bn_stats_rows(int64_t M) { return M <= 0 ? 0 : (int)((M + hiast::IG_BM - 1) / hiast::IG_BM); }

extern "C" int hiast_igemm_bn_act(const void* x, const void* wp, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, const void* res, int relu, void* y,
                                  int B, int H, int W, int Cin, int Cout, int taps, int stride, int dil, int planes,
                                  int out_f32, float* stats, const void* res_gate, int gate_mask, hiast_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (taps == 1 && stride != 1) return HIAST_E_RANGE;       // strided 1x1: subsample the input first
    const int Ho = taps == 1 ? H : (H - 1) / stride + 1, Wo = taps == 1 ? W : (W - 1) / stride + 1;
    return hiast_igemm_launch(x, wp, gamma, beta, mean, var, eps, res, relu, y, (int64_t)B * Ho * Wo, Cin, Cout, taps, H,
                              W, stride, dil, planes, out_f32, (hipStream_t)stream, stats, res_gate, gate_mask);
}

#ifdef IG_STAMP
namespace hiast {
__global__ void ig_stamp_copy_kernel(unsigned* dst)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 1024 * 8 * 8 + 32 * 64; i += gridDim.x * blockDim.x) dst[i] = ig_stamp_buf[i];
}
}  // namespace hiast
extern "C" int hiast_igemm_debug_stamps(unsigned* dst_device)
{
    hipLaunchKernelGGL(hiast::ig_stamp_copy_kernel, dim3(64), dim3(256), 0, 0, dst_device);
    return (int)hipDeviceSynchronize();
}
#endif

// rows of the partial-sum buffer of a want_stats launch.  Such a launch is always the PLAIN convolution: igemm_launch_mode
// rejects stats together with a BatchNorm, a residual or a ReLU (HIAST_E_RANGE), so the kernel choice below (xconv for the
// expanding 1x1 shapes, the tile kernel otherwise) is the one the launch makes.
extern "C" int hiast_igemm_stats_rows(int64_t M, int Cin, int Cout, int taps, int planes)
{
    if (M <= 0 || Cin <= 0 || Cout <= 0) return 0;
    if (hiast_xconv_ok(M, Cin, Cout, planes, taps, 0, 0, 0, 0, 0, 0, 1)) return hiast_xconv_stats_rows(M, Cout);
    return (int)((M + hiast::IG_BM - 1) / hiast::IG_BM);
}

extern "C" int hiast_pack_conv_weight(const float* w, int N, int K, int taps, int planes, int transpose, void* wp,
                                      void* wpt, hiast_stream_t stream)
{
    if (!w || !wp) return HIAST_E_ARG;
    if (N <= 0 || K <= 0 || taps <= 0) return HIAST_E_ARG;
    if ((planes != 1 && planes != 2) || transpose < 0 || transpose > 2) return HIAST_E_RANGE;
    if (transpose == 2 && !wpt) return HIAST_E_ARG;
    const long long total = (long long)N * K * taps;
    if (N % 64 == 0 && K % 64 == 0 && taps <= 9 && !((((uintptr_t)wp) | ((uintptr_t)wpt)) & 15)) {
        const dim3 tg((unsigned)((N / 64) * (K / 64)));
        const size_t lds_bytes = (size_t)32 * taps * hiast::PK_PITCH * sizeof(unsigned short);
        if (planes == 2)
            hipLaunchKernelGGL(hiast::pack_conv_weight_tiled_kernel<2>, tg, dim3(256), lds_bytes, (hipStream_t)stream, w,
                               (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
        else
            hipLaunchKernelGGL(hiast::pack_conv_weight_tiled_kernel<1>, tg, dim3(256), lds_bytes, (hipStream_t)stream, w,
                               (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
        HIAST_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (planes == 2)
        hipLaunchKernelGGL(hiast::pack_conv_weight_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
    else
        hipLaunchKernelGGL(hiast::pack_conv_weight_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
    HIAST_CHECK_LAUNCH();
    return 0;
}

namespace hiast {

// K9f: BN(eval) + ReLU + MaxPool2d(3, stride 2, padding 1) of the library stem's output, written in the operand format of
// the trunk kernels — one pass instead of three (BN + ReLU, torch's pooling kernel, split / cast).  Thread = 8 channels of an
// output pixel (its nine inputs: the 8 lanes of a pixel read one contiguous run); scale / shift are derived once per thread
// (the channel group of a thread is fixed along its grid-stride walk: the stride is a multiple of C / 8).
template <bool IN_BF16, int PL>
__global__ __launch_bounds__(256) void stem_tail_kernel(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean,
                                                        const float* __restrict__ var, float eps,
                                                        unsigned short* __restrict__ out, int B, int H, int W, int C, int Ho,
                                                        int Wo)
{
    const int G = C >> 3;
    const long long total = (long long)B * Ho * Wo * G;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i % G);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cg * 8 + k;
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        sc[k] = (gamma ? gamma[c] : 1.0f) * invstd;
        sh[k] = fmaf(-mean[c], sc[k], beta ? beta[c] : 0.0f);
    }
    for (; i < total; i += stride) {
        const long long pix = i / G;
        const int xo = (int)(pix % Wo);
        const int yo = (int)((pix / Wo) % Ho);
        const int b = (int)(pix / ((long long)Wo * Ho));
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * yo - 1 + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo - 1 + dx;
                if (xx < 0 || xx >= W) continue;
                const size_t off = (((size_t)b * H + yy) * W + xx) * C + cg * 8;
                float v[8];
                if (IN_BF16) {
                    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(xv) + off);
                    const unsigned w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = __uint_as_float(w4[q] << 16);
                        v[2 * q + 1] = __uint_as_float(w4[q] & 0xFFFF0000u);
                    }
                } else {
                    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + off);
                    const float4 c4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + off + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float t = fmaf(v[k], sc[k], sh[k]);
                    t = t > 0.f ? t : 0.f;
                    if (IN_BF16) t = __bfloat162float(__float2bfloat16(t));     // the bf16 activation the pooling kernel saw
                    m[k] = t > m[k] ? t : m[k];
                }
            }
        }
        unsigned ph[4], pl_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h0, l0, h1, l1;
            ig_split(m[2 * q], h0, l0);
            ig_split(m[2 * q + 1], h1, l1);
            ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
            pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* dst = out + ig_elem<PL>((size_t)pix, cg * 8, C);
        *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        if (PL == 2) *reinterpret_cast<uint4*>(dst + 32) = make_uint4(pl_[0], pl_[1], pl_[2], pl_[3]);
    }
}

}  // namespace hiast

extern "C" int hiast_stem_tail(const void* x, int dtype, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, void* out, int planes, int B, int H, int W, int C,
                               hiast_stream_t stream)
{
    if (!x || !mean || !var || !out) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return HIAST_E_ARG;
    if ((dtype != 0 && dtype != 1) || (planes != 1 && planes != 2) || C % 8 != 0 || (planes == 2 && C % 32 != 0) ||
        ((((uintptr_t)x) | ((uintptr_t)out)) & 15))
        return HIAST_E_RANGE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, G = C / 8;
    const long long total = (long long)B * Ho * Wo * G;
    if ((long long)B * H * W * C >= (1ll << 40)) return HIAST_E_RANGE;
    long long nb = (total + 255) / 256;
    int grid = (int)(nb > 16384 ? 16384 : nb);
    {   // grid-stride = multiple of G, so that a thread keeps its channel group
        long long a = 256, g = G;
        while (a) { const long long t = g % a; g = a; a = t; }
        const int need = (int)(G / g);
        if (grid >= need) grid -= grid % need;
        else return HIAST_E_RANGE;
    }
    hipStream_t st = (hipStream_t)stream;
#define L(BF, PL)                                                                                                  \
    hipLaunchKernelGGL((hiast::stem_tail_kernel<BF, PL>), dim3(grid), dim3(256), 0, st, x, gamma, beta, mean, var, eps, \
                       (unsigned short*)out, B, H, W, C, Ho, Wo)
    if (dtype == 1) { if (planes == 2) L(true, 2); else L(true, 1); }
    else { if (planes == 2) L(false, 2); else L(false, 1); }
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_split_planes(float* x, void* planes, int64_t M, int C, int inverse, hiast_stream_t stream)
{
    if (!x || !planes) return HIAST_E_ARG;
    if (M <= 0 || C <= 0) return HIAST_E_ARG;
    if (C % 32 != 0 || ((((uintptr_t)x) | ((uintptr_t)planes)) & 15)) return HIAST_E_RANGE;
    const long long total8 = (long long)M * C / 8;
    long long nb = (total8 + 255) / 256;
    const int grid = (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
    if (inverse)
        hipLaunchKernelGGL(hiast::from_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)planes, x, (long long)M, C);
    else
        hipLaunchKernelGGL(hiast::to_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (unsigned short*)planes, (long long)M, C);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_pack_conv_weight_multi(const hiast_pack_rec* table, const int32_t* chunk_tensor,
                                            const int64_t* chunk_start, int n_chunks, int max_taps, hiast_stream_t stream)
{
    if (!table || !chunk_tensor || !chunk_start) return HIAST_E_ARG;
    if (n_chunks <= 0 || max_taps <= 0) return HIAST_E_ARG;
    if (max_taps > 9) return HIAST_E_RANGE;
    const size_t lds_bytes = (size_t)32 * max_taps * hiast::PK_PITCH * sizeof(unsigned short);
    hipLaunchKernelGGL(hiast::pack_conv_weight_multi_kernel, dim3(n_chunks), dim3(256), lds_bytes, (hipStream_t)stream, table,
                       chunk_tensor, chunk_start);
    HIAST_CHECK_LAUNCH();
    return 0;
}
