// K9e — "expanding" 1x1 convolutions of the trunk on channels-last bf16 rows: Y[M,N] = X[M,K] * W[N,K]^T with a short
// reduction (K = 256: conv3 of every layer3 bottleneck, 256 -> 1024; the data gradient of conv1, 256 -> 1024 with the
// gated identity gradient).  gfx950 only.
//
// These launches are HBM-bound (34 GFLOP against 168-310 MB at B = 8): the tile kernel of igemm.hip spends them as
// load phase -> 4 k-steps -> long epilogue, one 128 KiB block per CU with nothing to overlap the phases (3.7 TB/s alone,
// 2-2.5 TB/s beside another stream).  Here the operand roles are turned round:
//   * the WEIGHTS live in registers for the whole kernel: a block owns 512 output columns, each of its 8 waves 64 of
//     them = 64 x 256 bf16 = 128 VGPRs per lane, loaded once, already in MFMA fragment layout;
//   * the block is PERSISTENT over 64-row panels of X, which stream through three LDS stages by LDS-DMA
//     (buffer_load ... lds, 32 KiB per panel, two panels in flight per CU);
//   * the MFMA computes D^T = W * X^T (v_mfma_f32_16x16x32_bf16 with W as the A operand), so a lane ends up with 16
//     output channels of ONE pixel: residual loads and output stores are 16-byte accesses straight from registers —
//     the epilogue needs no LDS round trip and no barrier, and one wave's epilogue overlaps the other waves' MFMAs and
//     the DMA of the next panels.
// The two blocks that share a panel stream (N = 1024: two column groups) sit in the same XCD, so the second read of a
// panel is an L2 hit.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 xc_bf16x8;
typedef __attribute__((ext_vector_type(4))) float xc_f32x4;
typedef __attribute__((address_space(3))) void* xc_lds_ptr;

constexpr int XC_PANEL = 64;            // rows of X per panel
constexpr int XC_STAGES = 3;
constexpr int XC_COLS = 512;            // output columns per block (8 waves x 64)

__device__ __forceinline__ void xc_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (xc_lds_ptr)lds, 16, voff, soff, 0, 0);
}

// same image as igemm.hip: 16-byte chunk c of row r of a [rows][128 B] slab tile lives at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ int xc_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }


// LDS fragment read as inline asm: the compiler orders every LDS load it knows about behind ALL pending LDS-DMA
// (s_waitcnt vmcnt(0) in front of the first ds_read after a buffer_load ... lds — it cannot tell the stage being read
// from the stage being filled), which would turn the two-panel prefetch into none.  The reads of one MFMA step are
// issued together and waited for by xc_lds_wait (lgkmcnt), which also ties the registers so nothing is moved across it.
__device__ __forceinline__ xc_bf16x8 xc_lds_read(unsigned addr)
{
    xc_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void xc_lds_wait(xc_bf16x8& a, xc_bf16x8& b)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

// channel (relative to the wave's first column) of accumulator element r of n-tile b held by lane group g = lane >> 4:
// lanes of one pixel (g = 0..3) cover 32 consecutive channels with the tiles (0,1), the next 32 with the tiles (2,3)
__device__ __forceinline__ int xc_chan(int g, int b, int r) { return (b >> 1) * 32 + g * 8 + (b & 1) * 4 + r; }

// EPI: what the epilogue does with o = acc (fp32):
//   BN      o = o * scale[n] + shift[n]  (eval-mode BatchNorm folded into two vectors, LDS)
//   RES     o += R[m][n]                 (GATE: only where bit (n & 7) of Rg[m][n / 8] is set)
//   RELU    o = max(o, 0)
//   STATS   per-block sums Σy, Σy² of the STORED (bf16-rounded) values -> stats[stream][N][2]
// F16: the rows are IEEE fp16 (HIAST_FMT_FP16) instead of bf16 — H16<F16> decodes / encodes / multiplies (common.h)
template <int KC, bool BN, bool RES, bool RELU, bool GATE, bool STATS, bool F16 = false>
__global__ __launch_bounds__(512) void xconv_kernel(const unsigned short* __restrict__ X,
                                                    const unsigned short* __restrict__ Wp,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    float eps, const unsigned short* __restrict__ R,
                                                    const unsigned char* __restrict__ Rg,
                                                    unsigned short* __restrict__ Y, int M, int N,
                                                    float* __restrict__ stats)
{
    constexpr int SL = KC / 64;                          // 128-byte slabs per row
    constexpr int KS = KC / 32;                          // 32-deep MFMA steps
    constexpr int STAGE = XC_PANEL * KC * 2;             // bytes per panel
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XC_STAGES * STAGE];
    __shared__ float s_sc[XC_COLS], s_sh[XC_COLS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, px = lane & 15;
    // block -> (column group, panel stream): consecutive block ids go round the 8 XCDs, so the NG blocks that walk
    // the same panels take neighbouring slots of ONE XCD
    const int NG = N / XC_COLS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % NG;
    const int nstream = (int)gridDim.x / NG;
    const int stream = (slot / NG) * 8 + xcd;
    const int n0 = cg * XC_COLS + wave * 64;             // this wave's first output column
    const int npanel = (M + XC_PANEL - 1) / XC_PANEL;

    // ---- weights -> registers, in the A-operand layout of v_mfma_f32_16x16x32_bf16 (lane: row px, k = 8 g .. 8 g + 7
    // of a 32-deep step); row px of n-tile b is output channel n0 + xc_chan(px >> 2, b, px & 3)
    xc_bf16x8 wr[4][KS];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const unsigned short* wrow = Wp + (size_t)(n0 + xc_chan(px >> 2, b, px & 3)) * KC + g * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) wr[b][s] = *reinterpret_cast<const xc_bf16x8*>(wrow + s * 32);
    }
    // pin the fragments down HERE: with their first use inside the panel loop the compiler's wait for these loads
    // lands in the loop as an s_waitcnt vmcnt(0) behind every DMA issue (measured: the prefetch was dead)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wr[b][s]));
    if (BN) {
        for (int c = tid; c < XC_COLS; c += 512) {
            const int n = cg * XC_COLS + c;
            const float sc = (gamma ? gamma[n] : 1.0f) * (1.0f / sqrtf(var[n] + eps));
            s_sc[c] = sc;
            s_sh[c] = fmaf(-mean[n], sc, beta ? beta[n] : 0.0f);
        }
        __syncthreads();                                 // (the loop's barriers are bare: they publish no LDS stores)
    }

    // ---- DMA: a panel = 64 rows x SL slabs = 8 row groups x SL; wave w moves row group w of every slab
    // (SL wave-instructions, 8 rows x 128 B each); lane l: row l >> 3, physical chunk l & 7 of the swizzled image
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)M * KC * 2), 0x00020000);
    const int drow = wave * 8 + (lane >> 3);
    const int dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    auto issue = [&](int p, int st) {
        const int m = p * XC_PANEL + drow;
        const int voff = (p < npanel && m < M) ? (int)((size_t)m * KC * 2) + dchunk : OOB;
        unsigned char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < SL; ++j) xc_dma16(xrs, base + j * (XC_PANEL * 128), voff, j * 128);
    };

    const unsigned lds_base = (unsigned)(size_t)smem;       // LDS byte address of the stage buffers (for the asm reads)
    float st1[STATS ? 16 : 1], st2[STATS ? 16 : 1];
#pragma unroll
    for (int q = 0; q < (STATS ? 16 : 1); ++q) { st1[q] = 0.f; st2[q] = 0.f; }

    issue(stream, 0);
    issue(stream + nstream, 1);
    int it = 0;
    for (int p = stream; p < npanel; p += nstream, ++it) {
        const int st = it % XC_STAGES;
        // This wave's share of panel p has landed once everything it issued BEFORE the DMA of panel p + 1 has retired
        // (memory operations of one wave retire in order): younger are that DMA and, after it, the previous
        // iteration's stores of half 0 (4), residual / gate loads of half 1 and stores of half 1 (4).
        // (First iteration: only the DMA of the second panel is younger.)  No scratch traffic may hide in this count:
        // the variants are built without spills (checked in the build log: private_segment_fixed_size == 0).
        constexpr int NRES = RES ? (GATE ? 8 : 4) : 0;          // residual (+ gate) loads per 32-row half
        // A bare s_barrier behind the counted wait (round 3): behind __syncthreads() the compiler (ROCm 7.2) emits
        // `s_waitcnt vmcnt(0) lgkmcnt(0)` — every store of the previous panel had to be acknowledged by the memory side and
        // the DMA of the next panel had to land before any wave went on, once per panel (found in the ISA of xconv2.hip;
        // this loop had the same drain, which is what held the residual variants at 4.3 TB/s).  The fragment reads are
        // asm with their own lgkmcnt waits, so nothing else needs the implied wait.
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(SL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(SL + 8 + NRES) : "memory");      // everyone's has landed;
                                                                 // everyone left the stage of panel p - 1
        // residual rows (and gate bytes) of a 32-row half are requested before its MFMAs
        uint4 rres[RES ? 2 : 1][2];
        unsigned rgate[GATE ? 2 : 1][2];
        auto load_res = [&](int half) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = p * XC_PANEL + half * 32 + t * 16 + px;
                const size_t row = (size_t)(m < M ? m : 0) * N + n0 + g * 8;
                rres[t][0] = h_load16_once(R + row);
                rres[t][1] = h_load16_once(R + row + 32);
                if (GATE) {
                    const unsigned char* gp = Rg + (size_t)(m < M ? m : 0) * (N >> 3) + ((n0 + g * 8) >> 3);
                    rgate[t][0] = gp[0];
                    rgate[t][1] = gp[4];
                }
            }
        };
        if (RES) load_res(0);
        issue(p + 2 * nstream, (it + 2) % XC_STAGES);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (RES && half == 1) load_res(1);
            xc_f32x4 acc[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (xc_f32x4){0.f, 0.f, 0.f, 0.f};
            // fragments of step s + 1 are requested before the MFMAs of step s (LDS latency hidden behind 8 MFMAs)
            auto frag = [&](int s, int a) {
                return xc_lds_read(lds_base + (unsigned)(st * STAGE + (s >> 1) * (XC_PANEL * 128) +
                                                         xc_lds_off(half * 32 + a * 16 + px, (s & 1) * 4 + g)));
            };
            // (double-buffered where the registers allow it: the statistics accumulators / BN + residual temporaries
            // of those variants would otherwise spill, and scratch traffic would break the vmcnt bookkeeping above)
            constexpr bool DB = !(STATS || (BN && RES));
            xc_bf16x8 xa[DB ? 2 : 1][2];
            xa[0][0] = frag(0, 0);
            xa[0][1] = frag(0, 1);
            xc_lds_wait(xa[0][0], xa[0][1]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                constexpr int one = DB ? 1 : 0;
                if (DB && s + 1 < KS) {
                    xa[(s + 1) & one][0] = frag(s + 1, 0);
                    xa[(s + 1) & one][1] = frag(s + 1, 1);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = H16<F16>::mfma16(wr[b][s], xa[s & one][a], acc[a][b]);
                if (s + 1 < KS) {
                    if (!DB) {
                        xa[0][0] = frag(s + 1, 0);
                        xa[0][1] = frag(s + 1, 1);
                    }
                    xc_lds_wait(xa[(s + 1) & one][0], xa[(s + 1) & one][1]);
                }
            }
            // ---- epilogue of this 32-row half: lane = pixel (a, px), channels n0 + {g*8 .. g*8+7} and + 32
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int m = p * XC_PANEL + half * 32 + a * 16 + px;
                const bool ok = m < M;
                const size_t row = (size_t)(ok ? m : 0) * N + n0 + g * 8;
                const uint4 r0 = rres[RES ? a : 0][0], r1 = rres[RES ? a : 0][1];
                const unsigned gate0 = GATE ? rgate[GATE ? a : 0][0] : 0xFFu;
                const unsigned gate1 = GATE ? rgate[GATE ? a : 0][1] : 0xFFu;
                float o[16];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(b >> 1) * 8 + (b & 1) * 4 + r] = acc[a][b][r];
                if (BN) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int c = wave * 64 + hh * 32 + g * 8;
#pragma unroll
                        for (int q = 0; q < 8; q += 4) {
                            const float4 sc = *reinterpret_cast<const float4*>(&s_sc[c + q]);
                            const float4 sh = *reinterpret_cast<const float4*>(&s_sh[c + q]);
                            o[hh * 8 + q] = fmaf(o[hh * 8 + q], sc.x, sh.x);
                            o[hh * 8 + q + 1] = fmaf(o[hh * 8 + q + 1], sc.y, sh.y);
                            o[hh * 8 + q + 2] = fmaf(o[hh * 8 + q + 2], sc.z, sh.z);
                            o[hh * 8 + q + 3] = fmaf(o[hh * 8 + q + 3], sc.w, sh.w);
                        }
                    }
                }
                if (RES) {
                    const unsigned w0[4] = {r0.x, r0.y, r0.z, r0.w}, w1[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float a0 = H16<F16>::lo(w0[q]), a1 = H16<F16>::hi(w0[q]);
                        float b0 = H16<F16>::lo(w1[q]), b1 = H16<F16>::hi(w1[q]);
                        if (GATE) {
                            a0 = ((gate0 >> (2 * q)) & 1u) ? a0 : 0.f;
                            a1 = ((gate0 >> (2 * q + 1)) & 1u) ? a1 : 0.f;
                            b0 = ((gate1 >> (2 * q)) & 1u) ? b0 : 0.f;
                            b1 = ((gate1 >> (2 * q + 1)) & 1u) ? b1 : 0.f;
                        }
                        o[2 * q] += a0;
                        o[2 * q + 1] += a1;
                        o[8 + 2 * q] += b0;
                        o[8 + 2 * q + 1] += b1;
                    }
                }
                if (RELU) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
                }
                unsigned pk[8];
                if (STATS && F16) {
                    // fp16 statistics variant: pair by pair, each pair finished before the next is converted (the cvt
                    // temporaries of eight pairs in flight at once pushed this variant 3 registers over the file)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        pk[q] = H16<F16>::pack(o[2 * q], o[2 * q + 1]);
                        if (ok) {
                            const float v0 = H16<F16>::lo(pk[q]), v1 = H16<F16>::hi(pk[q]);
                            st1[2 * q] += v0; st2[2 * q] = fmaf(v0, v0, st2[2 * q]);
                            st1[2 * q + 1] += v1; st2[2 * q + 1] = fmaf(v1, v1, st2[2 * q + 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) pk[q] = H16<F16>::pack(o[2 * q], o[2 * q + 1]);
                    if (STATS && ok) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float v0 = H16<F16>::lo(pk[q]), v1 = H16<F16>::hi(pk[q]);
                            st1[2 * q] += v0; st2[2 * q] = fmaf(v0, v0, st2[2 * q]);
                            st1[2 * q + 1] += v1; st2[2 * q + 1] = fmaf(v1, v1, st2[2 * q + 1]);
                        }
                    }
                }
                if (ok) {
                    h_store16_out(Y + row, pk[0], pk[1], pk[2], pk[3]);        // (plain: common.h)
                    h_store16_out(Y + row + 32, pk[4], pk[5], pk[6], pk[7]);
                }
            }
        }
    }
    if (STATS) {
        // fold the 16 pixel-lanes of each channel group; lane px == 0 of every g then holds the wave's column sums
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                st1[q] += __shfl_xor(st1[q], o, 64);
                st2[q] += __shfl_xor(st2[q], o, 64);
            }
        }
        if (px == 0) {
            float* d = stats + ((size_t)stream * N + n0 + g * 8) * 2;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                d[2 * q] = st1[q];
                d[2 * q + 1] = st2[q];
                d[64 + 2 * q] = st1[8 + q];
                d[64 + 2 * q + 1] = st2[8 + q];
            }
        }
    }
}

static int xc_blocks(int64_t M, int N)
{
    const int NG = N / XC_COLS;
    const long long npanel = (M + XC_PANEL - 1) / XC_PANEL;
    static const int env_cus = [] { const char* e = getenv("HIAST_XCONV_CUS"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= 256 ? v : 0; }();
    const int cus = env_cus ? env_cus : hiast_grid_cus();   // one block per CU (HIAST_XCONV_CUS: experiment, part of the chip)
    long long streams = cus / NG / 8 * 8;           // the (slot, xcd) numbering wants whole rounds of the 8 XCDs
    if (streams < 8) streams = 8;
    if (streams > npanel) streams = (npanel + 7) / 8 * 8;
    return (int)(streams * NG);
}

}  // namespace hiast

// shapes this kernel takes over from the tile kernel (plain bf16, 1x1)
int hiast_xconv_ok(int64_t M, int K, int N, int planes, int taps, int out_f32, int has_bn, int has_res, int relu,
                   int has_gate, int gate_mask, int has_stats)
{
    const char* env = getenv("HIAST_XCONV");          // HIAST_XCONV=0: A/B switch back to the tile kernel
    if ((env && atoi(env) == 0) || planes != 1 || taps != 1 || out_f32) return 0;
    if (K != 256 || N % hiast::XC_COLS != 0 || N > 2048) return 0;
    if (has_gate && (!gate_mask || !has_res || relu)) return 0;
    if (has_bn && has_res && !relu) return 0;        // (no caller in the trunk; that variant would need scratch)
    if (has_stats && (has_res || relu || has_bn)) return 0;      // the statistics variant is the plain GEMM only
    if (M < 4096) return 0;                          // small maps: the tile kernel's grid fills the chip better
    return 1;
}

int hiast_xconv_stats_rows(int64_t M, int N)
{
    return hiast::xc_blocks(M, N) / (N / hiast::XC_COLS);
}

int hiast_xconv_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       float* stats, const void* res_gate, int f16, hipStream_t st)
{
    using namespace hiast;
    const dim3 grid((unsigned)xc_blocks(M, N));
#define XL(BNF, RESF, RELUF, GATEF, STATSF)                                                                          \
    do {                                                                                                             \
        if (f16)                                                                                                     \
            hipLaunchKernelGGL((xconv_kernel<256, BNF, RESF, RELUF, GATEF, STATSF, true>), grid, dim3(512), 0, st,    \
                               (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,      \
                               (const unsigned short*)res, (const unsigned char*)res_gate, (unsigned short*)y, (int)M, N, \
                               stats);                                                                               \
        else                                                                                                         \
            hipLaunchKernelGGL((xconv_kernel<256, BNF, RESF, RELUF, GATEF, STATSF>), grid, dim3(512), 0, st,          \
                               (const unsigned short*)x, (const unsigned short*)wp, gamma, beta, mean, var, eps,      \
                               (const unsigned short*)res, (const unsigned char*)res_gate, (unsigned short*)y, (int)M, N, \
                               stats);                                                                               \
    } while (0)
    const bool bn = mean != nullptr;
    if (stats) { XL(false, false, false, false, true); }
    else if (res_gate) { XL(false, true, false, true, false); }
    else if (res) {
        if (bn) { if (relu) XL(true, true, true, false, false); else return HIAST_E_RANGE; }
        else { if (relu) XL(false, true, true, false, false); else XL(false, true, false, false, false); }
    } else {
        if (bn) { if (relu) XL(true, false, true, false, false); else XL(true, false, false, false, false); }
        else { if (relu) XL(false, false, true, false, false); else XL(false, false, false, false, false); }
    }
#undef XL
    HIAST_CHECK_LAUNCH();
    return 0;
}
