// K9e' (round 4) — the data gradient of conv1 of an identity bottleneck (256 -> 1024, + the gated identity gradient) that
// ALSO delivers the backward sums of the BatchNorm BEHIND its output: its output G is the gradient of the previous block's
// output y = relu(bn3(x3) + identity), and bn3's backward needs Σg and Σ g*xhat3 over g = G * (y > 0) before it can write a
// single dx3 (reference: autograd of `out = self.relu(self.bn3(self.conv3(out)) + identity)` followed by the next block's
// `self.conv1(x)`, sseg/models/modules/resnet.py:78-98, under apex O1).
//
// Until round 4 those sums cost a pass of their own over (G, x3, mask): 289 MB per layer3 block, 62 us, 22 times per step.
// Here the epilogue that WRITES G takes them: it reads x3 and the gate bits of y for the rows it finishes (142 MB more through
// this launch, 289 MB less in all; measured at B = 8: 135 us (fp16) / 123 us (bf16) against 85 + 65 us for the two launches).  The kernel is xconv.hip's — weights in registers for the whole launch, a block persistent
// over 64-row panels of the small operand that stream through three LDS stages by LDS-DMA, the MFMA computing D^T = W * X^T so
// that a lane finishes consecutive channels of ONE pixel straight from its accumulators — with HALF the columns per wave
// (32: 64 VGPRs of weights instead of 128, a block covers 256 output columns): xconv's gated variant sits at 256 VGPRs and
// has no room for a second row operand, its parameters and 2 x 16 statistics accumulators.  Four column groups walk the same
// panels (neighbouring blocks of one XCD: the re-reads are L2 hits).
//   Y[m][n]  = Σ_k X[m][k] * W[n][k] + (bit n of Rg[m] ? R[m][n] : 0)                          (16-bit rows, fp32 accumulate)
//   stats[stream][n] = ( Σ_m g, Σ_m g * (BX[m][n] - mean[n]) * invstd[n] ),  g = stored Y[m][n] where bit n of BM[m] is set
// gfx950 only.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 xb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float xb_f32x4;
typedef __attribute__((address_space(3))) void* xb_lds_ptr;

constexpr int XB_PANEL = 64;            // rows of X per panel
constexpr int XB_STAGES = 3;
constexpr int XB_COLS = 256;            // output columns per block (8 waves x 32)
constexpr int XB_KC = 256;              // reduction length (conv1's output channels)

__device__ __forceinline__ void xb_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (xb_lds_ptr)lds, 16, voff, soff, 0, 0);
}
// same LDS image as igemm.hip / xconv.hip: 16-byte chunk c of row r of a [rows][128 B] slab tile lives at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ int xb_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// fragment read with the tile offset as the instruction's immediate (idx * 2 KiB): two base registers per panel serve all 32
// reads of a half (computed per read, the addresses were 32 more live VGPRs and the kernel spilled)
template <int IDX>
__device__ __forceinline__ xb_bf16x8 xb_lds_read_i(unsigned addr)
{
    xb_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IDX * 2048));
    return v;
}
__device__ __forceinline__ xb_bf16x8 xb_lds_read(unsigned addr, int idx)     // idx is a constant after unrolling
{
    switch (idx) {
    case 1: return xb_lds_read_i<1>(addr);
    case 2: return xb_lds_read_i<2>(addr);
    case 3: return xb_lds_read_i<3>(addr);
    case 4: return xb_lds_read_i<4>(addr);
    case 5: return xb_lds_read_i<5>(addr);
    case 6: return xb_lds_read_i<6>(addr);
    case 7: return xb_lds_read_i<7>(addr);
    case 8: return xb_lds_read_i<8>(addr);
    case 9: return xb_lds_read_i<9>(addr);
    case 10: return xb_lds_read_i<10>(addr);
    case 11: return xb_lds_read_i<11>(addr);
    case 12: return xb_lds_read_i<12>(addr);
    case 13: return xb_lds_read_i<13>(addr);
    case 14: return xb_lds_read_i<14>(addr);
    case 15: return xb_lds_read_i<15>(addr);
    default: return xb_lds_read_i<0>(addr);
    }
}
__device__ __forceinline__ void xb_lds_wait(xb_bf16x8& a, xb_bf16x8& b)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

template <bool F16>
__global__ __launch_bounds__(512) void xconv_gated_bnstat_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp, const unsigned short* __restrict__ R,
    const unsigned char* __restrict__ Rg, const unsigned short* __restrict__ BX, const unsigned char* __restrict__ BM,
    const float* __restrict__ bmean, const float* __restrict__ binvstd, unsigned short* __restrict__ Y, int M, int N,
    float* __restrict__ stats)
{
    constexpr int SL = XB_KC / 64;                       // 128-byte slabs per row (4)
    constexpr int KS = XB_KC / 32;                       // 32-deep MFMA steps (8)
    constexpr int STAGE = XB_PANEL * XB_KC * 2;          // bytes per panel (32 KiB)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XB_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, px = lane & 15;
    // block -> (column group, panel stream): the NG blocks that walk the same panels take neighbouring slots of ONE XCD
    const int NG = N / XB_COLS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % NG;
    const int nstream = (int)gridDim.x / NG;
    const int stream = (slot / NG) * 8 + xcd;
    const int n0 = cg * XB_COLS + wave * 32;             // this wave's first output column
    const int npanel = (M + XB_PANEL - 1) / XB_PANEL;
    const int c0 = n0 + g * 8;                           // this lane's 8 consecutive channels

    // weights -> registers in the A-operand layout of v_mfma_f32_16x16x32 (lane: row px, k = 8 g .. 8 g + 7 of a step); row px of
    // n-tile b (0 | 1) is output channel n0 + (px >> 2) * 8 + b * 4 + (px & 3): lane group g' of D then holds channels 8 g' + 4 b + r
    xb_bf16x8 wr[2][KS];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned short* wrow = Wp + (size_t)(n0 + (px >> 2) * 8 + b * 4 + (px & 3)) * XB_KC + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) wr[b][s] = *reinterpret_cast<const xb_bf16x8*>(wrow + s * 32);
    }
    // pin the weights down HERE (xconv.hip: with the first use inside the panel loop the compiler's wait for these loads lands in
    // the loop as an s_waitcnt vmcnt(0) behind every DMA issue)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wr[b][s]));

    // DMA: a panel = 64 rows x SL slabs; wave w moves row group w of every slab (SL wave-instructions, 8 rows x 128 B each)
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)M * XB_KC * 2), 0x00020000);
    const int drow = wave * 8 + (lane >> 3);
    const int dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    auto issue = [&](int p, int st) {
        const int m = p * XB_PANEL + drow;
        const int voff = (p < npanel && m < M) ? (int)((size_t)m * XB_KC * 2) + dchunk : OOB;
        unsigned char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < SL; ++j) xb_dma16(xrs, base + j * (XB_PANEL * 128), voff, j * 128);
    };
    const unsigned lds_base = (unsigned)(size_t)smem;
    // Σg and Σ g*x per channel (RAW x: the BatchNorm's mean / invstd are applied once, at the end — Σ g*xhat = invstd * (Σ g*x -
    // mean * Σg) — which keeps 16 registers free for the row ring below)
    float st1[8], st2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { st1[q] = 0.f; st2[q] = 0.f; }

    // The rows this block finishes (gated residual row, its gate byte, BatchNorm input row, the BatchNorm's gate byte: 4 loads per
    // 16-pixel tile, 16 per panel and lane) are requested ONE PANEL AHEAD into a two-deep register ring (set index = compile-time
    // constant: the panel loop is unrolled by two).  With the rows of a half requested just before its MFMAs (xconv.hip's
    // arrangement, the first form of this kernel) every half waited out a full memory latency — ~10 k cycles per 32 rows,
    // 2.9 TB/s; a lane has 64 B in flight then, 64 KiB per CU with the ring.
    struct Rows {
        uint4 res[4], bx[4];          // [2 * half + a]
        unsigned gate[4], bm[4];
    };
    Rows ring[2];
    auto load_rows = [&](Rows& r, int p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = p * XB_PANEL + t * 16 + px;
            const size_t mr = (size_t)((p < npanel && m < M) ? m : 0);
            r.res[t] = h_load16_once(R + mr * N + c0);
            r.gate[t] = Rg[mr * (N >> 3) + (c0 >> 3)];
            r.bx[t] = h_load16_once(BX + mr * N + c0);
            r.bm[t] = BM[mr * (N >> 3) + (c0 >> 3)];
        }
    };
    // one panel: counted wait + barrier, the next panel's rows and the DMA of the panel after next, then the two 32-row halves
    auto panel = [&](int p, int it, const Rows& cur, Rows& nxt) {
        const int st = it % XB_STAGES;
        // this wave's share of panel p has landed once everything older than the DMA of panel p + 1 has retired: younger are
        // that DMA (SL) and the 4 stores of the previous panel (its row loads, older still, have to be there anyway).  First
        // panel: the second panel's DMA and the 16 row loads of the prologue may be in flight.
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(SL + 16) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(SL + 4) : "memory");
        load_rows(nxt, p + nstream);
        issue(p + 2 * nstream, (it + 2) % XB_STAGES);
        const unsigned fbase = lds_base + (unsigned)(st * STAGE) + (unsigned)xb_lds_off(px, g);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            xb_f32x4 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = (xb_f32x4){0.f, 0.f, 0.f, 0.f};
            // row = 32 half + 16 a + px: its swizzle ((row >> 1) & 7) depends on px only, and the two 32-deep halves of a slab
            // are chunks g and g ^ 4: address = fbase ^ (64 if s odd) + 2 KiB * (4 (s >> 1) + 2 half + a)
            auto frag = [&](int s, int a) {
                return xb_lds_read((s & 1) ? fbase ^ 64u : fbase, 4 * (s >> 1) + 2 * half + a);
            };
            xb_bf16x8 xa[2][2];
            xa[0][0] = frag(0, 0);
            xa[0][1] = frag(0, 1);
            xb_lds_wait(xa[0][0], xa[0][1]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    xa[(s + 1) & 1][0] = frag(s + 1, 0);
                    xa[(s + 1) & 1][1] = frag(s + 1, 1);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = H16<F16>::mfma16(wr[b][s], xa[s & 1][a], acc[a][b]);
                if (s + 1 < KS) xb_lds_wait(xa[(s + 1) & 1][0], xa[(s + 1) & 1][1]);
            }
            // epilogue of this half: lane = pixel (a, px), channels c0 .. c0 + 7 (tile b holds 4 b .. 4 b + 3)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int t = 2 * half + a;
                const int m = p * XB_PANEL + t * 16 + px;
                const bool ok = m < M;
                const unsigned w0[4] = {cur.res[t].x, cur.res[t].y, cur.res[t].z, cur.res[t].w};
                const unsigned xw[4] = {cur.bx[t].x, cur.bx[t].y, cur.bx[t].z, cur.bx[t].w};
                const unsigned gate = cur.gate[t], bm = cur.bm[t];
                unsigned pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o0 = acc[a][q >> 1][2 * (q & 1)], o1 = acc[a][q >> 1][2 * (q & 1) + 1];       // channels 2q, 2q + 1
                    o0 += ((gate >> (2 * q)) & 1u) ? H16<F16>::lo(w0[q]) : 0.f;
                    o1 += ((gate >> (2 * q + 1)) & 1u) ? H16<F16>::hi(w0[q]) : 0.f;
                    pk[q] = H16<F16>::pack(o0, o1);
                    if (ok) {       // sums of the STORED gradient where the BatchNorm's ReLU gate is open
                        const float g0 = ((bm >> (2 * q)) & 1u) ? H16<F16>::lo(pk[q]) : 0.f;
                        const float g1 = ((bm >> (2 * q + 1)) & 1u) ? H16<F16>::hi(pk[q]) : 0.f;
                        st1[2 * q] += g0;
                        st2[2 * q] = fmaf(g0, H16<F16>::lo(xw[q]), st2[2 * q]);
                        st1[2 * q + 1] += g1;
                        st2[2 * q + 1] = fmaf(g1, H16<F16>::hi(xw[q]), st2[2 * q + 1]);
                    }
                }
                // (rows beyond M are not stored: a ragged tail only occurs in a block's LAST panel, after which it takes no
                // counted wait)
                if (ok) h_store16_out(Y + (size_t)m * N + c0, pk[0], pk[1], pk[2], pk[3]);   // (streaming store: no change stand-alone)
            }
        }
    };

    issue(stream, 0);
    issue(stream + nstream, 1);
    load_rows(ring[0], stream);
    for (int p = stream, it = 0; p < npanel; p += 2 * nstream, it += 2) {
        panel(p, it, ring[0], ring[1]);
        if (p + nstream >= npanel) break;
        panel(p + nstream, it + 1, ring[1], ring[0]);
    }
    // fold the 16 pixel-lanes of each channel group; lane px == 0 of every g then holds the wave's column sums
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            st1[q] += __shfl_xor(st1[q], o, 64);
            st2[q] += __shfl_xor(st2[q], o, 64);
        }
    }
    if (px == 0) {
        float* d = stats + ((size_t)stream * N + c0) * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float mu = bmean[c0 + q], is = binvstd[c0 + q];
            d[2 * q] = st1[q];
            d[2 * q + 1] = is * (st2[q] - mu * st1[q]);
        }
    }
}

static int xb_blocks(int64_t M, int N)
{
    const int NG = N / XB_COLS;
    const long long npanel = (M + XB_PANEL - 1) / XB_PANEL;
    long long streams = hiast_grid_cus() / NG / 8 * 8;   // one block per CU; the (slot, xcd) numbering wants whole rounds of the 8 XCDs
    if (streams < 8) streams = 8;
    if (streams > npanel) streams = (npanel + 7) / 8 * 8;
    return (int)(streams * NG);
}

}  // namespace hiast

// rows of the `partial` output of hiast_xconv_dgrad_gated_bn_stats (0: shape not taken)
extern "C" int hiast_xconv_dgrad_gated_bn_stats_rows(int64_t M, int K, int N)
{
    if (M < 4096 || K != hiast::XB_KC || N % hiast::XB_COLS != 0 || N > 2048) return 0;
    return hiast::xb_blocks(M, N) / (N / hiast::XB_COLS);
}

// dy [M][K = 256] (gradient of conv1's output), wpt: conv1's adjoint packed weight [N][1][K] (hiast_pack_conv_weight, transpose),
// res [M][N] + res_gate [M][N/8]: the gradient of THIS block's output and the gate bits of its ReLU (the identity branch's
// gradient, added where the gate is open); bn_x [M][N], bn_mask [M][N/8], save_mean / save_invstd [N]: input, ReLU gate bits and
// batch statistics of the BatchNorm whose output (+ identity, ReLU) conv1 consumed — the PREVIOUS block's bn3.
// -> dx [M][N] (the gradient of the previous block's output) and partial fp32 [rows][N][2] = per-block (Σg, Σ g*xhat) of that
// BatchNorm's backward (hiast_bn_nhwc_stats_from_partial reduces them).  16-bit channels-last rows of format fmt.
extern "C" int hiast_xconv_dgrad_gated_bn_stats(const void* dy, const void* wpt, const void* res, const void* res_gate,
                                                const void* bn_x, const void* bn_mask, const float* save_mean,
                                                const float* save_invstd, void* dx, float* partial, int64_t M, int K, int N,
                                                int fmt, hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if (!dy || !wpt || !res || !res_gate || !bn_x || !bn_mask || !save_mean || !save_invstd || !dx || !partial) return HIAST_E_ARG;
    if (hiast_xconv_dgrad_gated_bn_stats_rows(M, K, N) == 0) return HIAST_E_RANGE;
    if ((size_t)M * N * 2 >= (1ull << 40) || (size_t)M * K * 2 >= (1ull << 31)) return HIAST_E_RANGE;
    if ((((uintptr_t)dy) | ((uintptr_t)wpt) | ((uintptr_t)res) | ((uintptr_t)bn_x) | ((uintptr_t)dx) | ((uintptr_t)save_mean) |
         ((uintptr_t)save_invstd)) & 15)
        return HIAST_E_RANGE;
    const dim3 grid((unsigned)hiast::xb_blocks(M, N));
    hipStream_t st = (hipStream_t)stream;
    if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::xconv_gated_bnstat_kernel<true>, grid, dim3(512), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)wpt, (const unsigned short*)res, (const unsigned char*)res_gate,
                           (const unsigned short*)bn_x, (const unsigned char*)bn_mask, save_mean, save_invstd,
                           (unsigned short*)dx, (int)M, N, partial);
    else
        hipLaunchKernelGGL(hiast::xconv_gated_bnstat_kernel<false>, grid, dim3(512), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)wpt, (const unsigned short*)res, (const unsigned char*)res_gate,
                           (const unsigned short*)bn_x, (const unsigned char*)bn_mask, save_mean, save_invstd,
                           (unsigned short*)dx, (int)M, N, partial);
    HIAST_CHECK_LAUNCH();
    return 0;
}
