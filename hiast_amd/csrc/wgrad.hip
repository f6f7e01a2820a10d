// K9d — weight gradient of the trunk convolutions (1x1 and 3x3, dilated / strided) on channels-last bf16
// activations (reference: autograd of nn.Conv2d in Bottleneck.forward, sseg/models/modules/resnet.py:78-98, under
// apex O1: half-precision operands, fp32 accumulation):
//
//     dW[n][tap][k] = Σ_m dY[m][n] * X[m + off(tap)][k]          m over all B*Ho*Wo output pixels
//
// A GEMM that reduces over the PIXEL index: both operands are pixel-major ([M][N] and [M][K] rows), the opposite
// of what an MFMA fragment wants (8 consecutive reduction elements per lane).  The tiles are therefore written into
// LDS exactly as they lie in memory — by LDS-DMA, 4 rows x 256 B per wave-instruction, zero rows for out-of-image
// taps via the buffer out-of-range rule — and every fragment is built by two transposing ds_read_b64_tr_b16 reads
// (swizzle (b) of the CDNA4 guide on 256-byte rows: conflict free for the transposed reads).
// Block = one (256 n) x (256 k) x tap tile of dW over one pixel range; 8 waves (wave tile 128 x 64); k-step = 64
// pixels; two LDS stages (the DMA of step t+1 flies during the MFMAs of step t).  Pixel ranges are reduced in a fixed
// order by wgrad_reduce_kernel (bitwise reproducible, no float atomics), which also writes torch's [N][K][kh][kw]
// layout.
#include <hip/hip_bf16.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float wg_f32x16;
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* wg_lds_ptr;

struct WGeo {
    int H, W, Ho, Wo, stride, dil;
};

__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wg_lds_ptr)lds, 16, voff, soff, 0, 0);
}

// transposing LDS read, as inline asm on the LDS byte address.  Through the builtin the compiler sees an LDS load and
// orders it behind ALL pending LDS-DMA: it put an s_waitcnt vmcnt(0) in front of the fragment reads that follow each
// pair of DMA pieces — four full HBM round trips per 64-pixel step during which the wave did nothing (the prefetch of
// step t+1 was waited for inside step t).  The asm reads are invisible to that rule; wg_wait ties their results to an
// explicit lgkmcnt wait.
__device__ __forceinline__ wg_s16x4 wg_tr_read(unsigned addr)
{
    wg_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
struct WgFrags {
    wg_s16x4 a[4][2], b[2][2];          // two transposed halves per fragment
};
__device__ __forceinline__ void wg_wait(WgFrags& f)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]),
                   "+v"(f.a[3][0]), "+v"(f.a[3][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]));
}
__device__ __forceinline__ wg_bf16x8 wg_join(const wg_s16x4& v0, const wg_s16x4& v1)
{
    wg_s16x8 v;
    v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
    v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
    return __builtin_bit_cast(wg_bf16x8, v);
}

// [rows][128 x 16-bit] sub-tile with 256-byte rows: 16-byte chunk ch of row r lives at chunk ch ^ swz(r)
__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int wg_off(int row, int ch) { return 256 * row + 16 * (ch ^ wg_swz(row)); }

constexpr int WG_ROWS = 64;                       // pixels per k-step
constexpr int WG_SUB = WG_ROWS * 256;             // one [64][128] sub-tile: 16 KiB
constexpr int WG_STAGE = 4 * WG_SUB;              // dY (2 sub-tiles) + X (2 sub-tiles)

template <int TAPS>
__global__ __launch_bounds__(512) void wgrad_tn_kernel(const unsigned short* __restrict__ dY,
                                                       const unsigned short* __restrict__ X, float* __restrict__ P,
                                                       int M, int N, int K, WGeo geo, int m_per_split)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // wave tile: n rows [128*wm, +128), k cols [64*wn, +64)
    const int kt_tiles = K / 256;
    // XCD-aware block order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (private L2
    // each).  The logical order is pixel range OUTER / (n tile, k tile, tap) INNER, and XCD k takes the k-th contiguous
    // share of it: the blocks that re-read one pixel range's dY and X rows (all taps, all tiles) sit on ONE XCD at the
    // same time, so those rows leave HBM once instead of once per XCD they were dealt to.
    int lid = blockIdx.x + gridDim.x * blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y, base = total >> 3, rem = total & 7;
        const int xcd = lid & 7, slot = lid >> 3;
        lid = xcd * base + (xcd < rem ? xcd : rem) + slot;
    }
    int t = lid % (int)gridDim.x;                         // (n tile, k tile, tap)
    const int tap = t % TAPS; t /= TAPS;
    const int k0 = (t % kt_tiles) * 256, n0 = (t / kt_tiles) * 256;
    const int split = lid / (int)gridDim.x;
    const int m_begin = split * m_per_split;
    const int m_end = (m_begin + m_per_split < M) ? m_begin + m_per_split : M;
    const int nk = (m_end - m_begin + WG_ROWS - 1) / WG_ROWS;
    const int oy = TAPS == 1 ? 0 : (tap / 3 - 1) * geo.dil, ox = TAPS == 1 ? 0 : (tap % 3 - 1) * geo.dil;

    constexpr int OOB = (int)0x80000000;
    const float inv_hw = 1.0f / (float)(geo.Ho * geo.Wo), inv_wo = 1.0f / (float)geo.Wo;
    const size_t in_pix = (TAPS == 1) ? (size_t)M : (size_t)(M / (geo.Ho * geo.Wo)) * geo.H * geo.W;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)((size_t)M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(in_pix * K * 2), 0x00020000);

    // DMA pieces of one k-step: 64 wave-instructions (4 sub-tiles x 16 groups of 4 rows); wave w issues pieces
    // 8w .. 8w+7: piece = sub*16 + grp.  Lane l supplies row 4*grp + (l >> 4), logical chunk (l & 15) ^ swz(row).
    // 3x3: the (image, row, column) of a lane's output pixel is decoded ONCE per k-step (piece 0: float reciprocal +
    // one-step correction, exact for m < 2^24, no integer division) and then walked: consecutive pieces of a wave are
    // 4 pixels apart.  Decoding every piece cost the X-loading waves ~40 VALU instructions x 8 pieces per k-step —
    // more than their 32 MFMAs.
    int d_img = 0, d_yo = 0, d_xo = 0;
    auto piece = [&](int kt, int buf, int pc) {
        const int idx = wave * 8 + pc;
        const int sub = idx >> 4, grp = idx & 15;
        const int row = 4 * grp + (lane >> 4);
        const int ch = (lane & 15) ^ wg_swz(row);
        const int m = m_begin + kt * WG_ROWS + row;
        unsigned char* dst = smem + buf * WG_STAGE + sub * WG_SUB + grp * 1024;
        if (sub < 2) {
            const int voff = m < m_end ? (int)(((size_t)m * N + n0 + sub * 128) * 2) + ch * 16 : OOB;
            wg_dma16(yrs, dst, voff, 0);
        } else {
            int voff = OOB;
            if (TAPS == 1) {
                if (m < m_end) voff = (int)(((size_t)m * K + k0 + (sub - 2) * 128) * 2) + ch * 16;
            } else {
                if (pc == 0) {
                    const int hw = geo.Ho * geo.Wo;
                    int img = (int)(((float)m + 0.5f) * inv_hw);
                    int r = m - img * hw;
                    if (r < 0) { --img; r += hw; } else if (r >= hw) { ++img; r -= hw; }
                    int yo = (int)(((float)r + 0.5f) * inv_wo);
                    int xo = r - yo * geo.Wo;
                    if (xo < 0) { --yo; xo += geo.Wo; } else if (xo >= geo.Wo) { ++yo; xo -= geo.Wo; }
                    d_img = img; d_yo = yo; d_xo = xo;
                }
                if (m < m_end) {
                    const int yy = d_yo * geo.stride + oy, xx = d_xo * geo.stride + ox;
                    if (yy >= 0 && yy < geo.H && xx >= 0 && xx < geo.W)
                        voff = (int)((((size_t)(d_img * geo.H + yy) * geo.W + xx) * K + k0 + (sub - 2) * 128) * 2) + ch * 16;
                }
                d_xo += 4;                                   // the next piece of this wave: 4 pixels on
                while (d_xo >= geo.Wo) {
                    d_xo -= geo.Wo;
                    if (++d_yo >= geo.Ho) { d_yo = 0; ++d_img; }
                }
            }
            wg_dma16(xrs, dst, voff, 0);
        }
    };

    wg_f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed fragment: 32 columns starting at c0 of a sub-tile, reduction rows kb .. kb+7 for this lane half
    const int grp4 = lane >> 4, t16 = lane & 15;
    const int fq = t16 >> 2, fp = t16 & 3;
    const unsigned lds0 = (unsigned)(size_t)smem;                // LDS byte address of the stage buffers
    auto frag = [&](unsigned subtile, int c0, int kk, wg_s16x4& v0, wg_s16x4& v1) {
        const int kb = kk * 16 + 8 * (grp4 >> 1);
        const int ch = ((c0 + 16 * (grp4 & 1)) >> 3) + (fp >> 1);
        v0 = wg_tr_read(subtile + (unsigned)(wg_off(kb + fq, ch) + 8 * (fp & 1)));
        v1 = wg_tr_read(subtile + (unsigned)(wg_off(kb + 4 + fq, ch) + 8 * (fp & 1)));
    };

    if (nk > 0) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(0, 0, pc);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = kt + 1 < nk;
        const unsigned st = lds0 + buf * WG_STAGE;
        const unsigned ta = st + wm * WG_SUB;                             // dY sub-tile of this wave's 128 n rows
        const unsigned tb = st + 2 * WG_SUB + (wn >> 1) * WG_SUB;         // X sub-tile holding this wave's 64 k cols
        auto read_all = [&](WgFrags& f, int kk) {
#pragma unroll
            for (int a = 0; a < 4; ++a) frag(ta, a * 32, kk, f.a[a][0], f.a[a][1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) frag(tb, (wn & 1) * 64 + b * 32, kk, f.b[b][0], f.b[b][1]);
        };
        // One fragment set in flight: with the set of sub-step kk + 1 prefetched as well the kernel needs 254 VGPRs,
        // i.e. two of its waves fill a SIMD's register file and NOTHING else fits on the CU — the short BatchNorm
        // kernels of the backward chain on the main stream then queue behind whole wgrad blocks (bnh_finalize: 7 -> 59 us
        // per launch, x 105 launches per step).  At ~220 VGPRs a small wave still fits beside it.
        WgFrags fr;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            read_all(fr, kk);
            if (more) {
                piece(kt + 1, buf ^ 1, 2 * kk);
                piece(kt + 1, buf ^ 1, 2 * kk + 1);
            }
            wg_wait(fr);
            wg_bf16x8 fa[4], fb[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = wg_join(fr.a[a][0], fr.a[a][1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = wg_join(fr.b[b][0], fr.b[b][1]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }

    // partial P[split][n][tap][k]
    float* Ps = P + (size_t)split * N * TAPS * K;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = k0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Ps[((size_t)n * TAPS + tap) * K + k] = acc[a][b][r];
            }
        }
}

// dW[n][k][tap] (torch [N][K][kh][kw]) = Σ_s P[s][n][tap][k], ascending s (fixed order: bitwise reproducible).
// thread = (n, tap, 4 consecutive k): the partials are read as they lie — 16 bytes per lane, eight splits in flight
// (the first form, one float per thread and four loads in flight, streamed the 67 MB of a layer3 1x1 at 1.8 TB/s:
// 38 us x 81 launches per training step) — only the small result is written with the tap stride.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int N,
                                                           int K, int taps, int nsplit)
{
    const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;       // index of a float4 along [n][tap][k]
    const size_t per = (size_t)N * taps * K;
    if (idx4 * 4 >= (long long)per) return;
    const long long idx = idx4 * 4;
    const int k = (int)(idx % K);
    const long long nt = idx / K;
    const int t = (int)(nt % taps), n = (int)(nt / taps);
    const float4* p = reinterpret_cast<const float4*>(P + idx);
    const size_t per4 = per / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + u) * per4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {           // the additions keep their order
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
    }
    for (; s < nsplit; ++s) {
        const float4 v = p[(size_t)s * per4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = dw + ((size_t)n * K + k) * taps + t;
    if (taps == 1) {
        *reinterpret_cast<float4*>(d) = acc;
    } else {
        d[0] = acc.x; d[taps] = acc.y; d[2 * taps] = acc.z; d[3 * taps] = acc.w;
    }
}

static int wgrad_nsplit(long long M, int tiles, int taps)
{
    // one block per CU at a time (128 KiB of LDS each) and every block costs a 256 KiB partial tile written here and read
    // again by the reduction: ONE round of <= 256 blocks (two rounds doubled the partial traffic — 129 MB against 67 MB
    // of operands on the layer3 3x3 — for the same MFMA time per CU); at least 8 k-steps (512 pixels) per block
    long long s = 256 / tiles;
    const long long smax = M / 512 > 0 ? M / 512 : 1;
    s = s < 1 ? 1 : (s > smax ? smax : s);
    s = s > 64 ? 64 : s;
    return (int)s;
}

}  // namespace hiast

extern "C" size_t hiast_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int taps)
{
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin % 256 != 0 || Cout % 256 != 0 || (taps != 1 && taps != 9)) return 0;
    const long long M = (long long)B * Ho * Wo;
    const int tiles = (Cout / 256) * (Cin / 256) * taps;
    return (size_t)hiast::wgrad_nsplit(M, tiles, taps) * Cout * taps * Cin * sizeof(float);
}

extern "C" int hiast_conv_wgrad_nhwc(const void* dy, const void* x, float* dw, int B, int H, int W, int Cin, int Cout,
                                     int taps, int stride, int dil, void* workspace, size_t workspace_bytes,
                                     hiast_stream_t stream)
{
    if (!dy || !x || !dw || !workspace) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || stride <= 0 || dil <= 0) return HIAST_E_ARG;
    if (Cin % 256 != 0 || Cout % 256 != 0 || (taps != 1 && taps != 9) || (taps == 1 && stride != 1)) return HIAST_E_RANGE;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace)) & 15) return HIAST_E_RANGE;
    const int Ho = taps == 1 ? H : (H - 1) / stride + 1, Wo = taps == 1 ? W : (W - 1) / stride + 1;
    const long long M = (long long)B * Ho * Wo;
    if ((size_t)M * Cout * 2 >= (1ull << 31) || (size_t)B * H * W * Cin * 2 >= (1ull << 31) || M >= (1ll << 24))
        return HIAST_E_RANGE;
    const int tiles = (Cout / 256) * (Cin / 256) * taps;
    const int nsplit = hiast::wgrad_nsplit(M, tiles, taps);
    if (workspace_bytes < (size_t)nsplit * Cout * taps * Cin * sizeof(float)) return HIAST_E_WS;
    int mps = (int)((M + nsplit - 1) / nsplit);
    mps = ((mps + hiast::WG_ROWS - 1) / hiast::WG_ROWS) * hiast::WG_ROWS;
    hiast::WGeo geo = {H, W, Ho, Wo, stride, dil};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(tiles, nsplit);
    if (taps == 1)
        hipLaunchKernelGGL(hiast::wgrad_tn_kernel<1>, grid, dim3(512), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)x, (float*)workspace, (int)M, Cout, Cin, geo, mps);
    else
        hipLaunchKernelGGL(hiast::wgrad_tn_kernel<9>, grid, dim3(512), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)x, (float*)workspace, (int)M, Cout, Cin, geo, mps);
    HIAST_CHECK_LAUNCH();
    const long long total = (long long)Cout * Cin * taps / 4;           // float4 per thread (Cin % 256 == 0)
    hipLaunchKernelGGL(hiast::wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)workspace, dw, Cout, Cin, taps, nsplit);
    HIAST_CHECK_LAUNCH();
    return 0;
}
