// K9h — weight gradient of the trunk convolutions BELOW 256 channels (layer1 / layer2 of the ResNet: 64 / 128 / 256 / 512
// channels, 1x1 and 3x3 stride 1; reference: autograd of nn.Conv2d in Bottleneck.forward, sseg/models/modules/resnet.py:78-98,
// under apex O1: half-precision operands, fp32 accumulation) — the shapes wgrad.hip (256 x 256 tiles) does not take and
// round 2 left to the library (27 launches, 1.5 ms per step + its cast kernels).
//
//     dW[n][tap][k] = Σ_m dY[m][n] * X[m + off(tap)][k]          m over all B*H*W pixels (stride 1, 'same' padding)
//
// These launches are HBM-bound (2 x 33.5 MB of operands for 19 GFLOP on the layer1 3x3), so the kernel is the plain
// register-staged form of aspp2.hip's transposed-read GEMM: block tile TI x 128 over a pixel range, 4 waves, k-step 64
// pixels, tiles stored in LDS as [64 pixels][256 B] rows with the swizzle that makes the transposing ds_read_b64_tr_b16
// fragment reads conflict free, several blocks per CU.  What is new is how 64-channel operands fill 256-byte tile rows:
//   * the i operand (tile rows of dW: dY, or X for the transposed 1x1 form) with 64 channels uses TI = 64 — the left half
//     of the LDS rows, the right half is neither written nor read;
//   * a 64-channel X as j operand packs TWO TAPS side by side (3x3: tap pairs (0,1) .. (8,-)): every lane fetches its own
//     16 bytes, so the two halves of a row may come from different input pixels; no multiply is wasted.
// Pixel ranges are reduced in ascending order by wgrad_small_reduce_kernel (bitwise reproducible, no float atomics), which
// also undoes the tile / tap-pair / transposed layouts and writes torch's [N][K][kh][kw].
#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 ws_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ws_f32x16;
typedef short ws_s16x4 __attribute__((ext_vector_type(4)));
typedef short ws_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int ws_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

struct WSArgs {
    const unsigned short* I;      // i operand [M][CI]
    const unsigned short* J;      // j operand [M][CJ] (X when taps = 9)
    float* P;                     // partial tiles [split][it][jt][TI][128]
    int M, CI, CJ;
    int H, W, dil;                // 3x3: OUTPUT map geometry ('same' padding = dil)
    int stride, Hin, Win;         // 3x3: input map (stride 1: the same map)
    int m_per_split, ni, nj;
};

template <int TI, int TAPS, bool F16>
__global__ __launch_bounds__(256) void wgrad_small_kernel(WSArgs g)
{
    constexpr int KS = 64;                               // pixels per k-step (two tiles of [KS][256 B] per LDS stage)
    constexpr int TILE_BYTES = KS * 256;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // [buf][i tile | j tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order (as wgrad.hip): logical order pixel range OUTER / tile INNER, XCD k takes the k-th contiguous
    // share — the tiles that re-read one pixel range's rows sit on one XCD at the same time
    int lid = blockIdx.x + gridDim.x * blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y, base = total >> 3, rem = total & 7;
        const int xcd = lid & 7, slot = lid >> 3;
        lid = xcd * base + (xcd < rem ? xcd : rem) + slot;
    }
    const int tile_id = lid % (int)gridDim.x, split = lid / (int)gridDim.x;
    const int it = tile_id % g.ni, jt = tile_id / g.ni;
    const int m_begin = split * g.m_per_split;
    const int m_end = (m_begin + g.m_per_split < g.M) ? m_begin + g.m_per_split : g.M;
    const int nk = (m_end - m_begin + KS - 1) / KS;

    // j tile: which tap(s) and which 128-channel window of the j operand
    const int jwin = g.CJ >= 128 ? g.CJ / 128 : 1;
    const int tap0 = TAPS == 1 ? 0 : (g.CJ >= 128 ? jt / jwin : 2 * jt);
    const int jcol = g.CJ >= 128 ? (jt % jwin) * 128 : 0;
    constexpr int NB = TI == 128 ? 2 : 1;                 // 32-column fragments per wave along j
    const int wm = TI == 128 ? wave >> 1 : 0, wn = TI == 128 ? wave & 1 : wave;

    ws_f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging: the j tile is 512 chunks of 16 B (2 per thread: row = (tid >> 4) + 16 u, chunk = tid & 15), the i tile 512
    // (TI = 128) or 256 (TI = 64: chunk = tid & 7, row = tid >> 3).
    // 3x3: (row, column) of this thread's two output pixels are WALKED 32 pixels per k-step (a decode per chunk and step was
    // ~100 VALU instructions per wave and step against 4 MFMAs); tap offsets and the chunk's column are constants.
    const int jch = tid & 15;
    int jtap = tap0, jc8 = jcol + jch * 8;
    if (g.CJ == 64) { jtap = tap0 + (jch >> 3); jc8 = (jch & 7) * 8; }
    const bool jlive = jtap < TAPS && (g.CJ >= 128 || TAPS == 9 || jch < 8);
    const int oy = TAPS == 9 ? (jtap / 3 - 1) * g.dil : 0, ox = TAPS == 9 ? (jtap % 3 - 1) * g.dil : 0;
    const long long jshift = (long long)(oy * g.W + ox) * g.CJ + jc8;       // elements from X[m][0] to this thread's chunk
    constexpr int NU = KS / 16;                          // 16-byte chunks per thread and tile (j tile; i tile for TI = 128)
    int wy[NU], wx[NU], wi[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        wy[u] = 0; wx[u] = 0; wi[u] = 0;
        if (TAPS == 9) {
            const int m = m_begin + (tid >> 4) + 16 * u;
            const int r = m % (g.H * g.W);
            wy[u] = r / g.W + (g.stride == 1 ? oy : 0);                      // stride 1: input row / column incl. the tap offset;
            wx[u] = r % g.W + (g.stride == 1 ? ox : 0);                      // otherwise the OUTPUT row / column
            wi[u] = m / (g.H * g.W);
        }
    }
    const int wx_end = g.W + (g.stride == 1 ? ox : 0), wy_end = g.H + (g.stride == 1 ? oy : 0);
    // (two register sets — the rows of step t + 2 requested while step t is multiplied — bought nothing at TI = 64 and cost the
    // second block per CU at TI = 128: 194 VGPRs, 36.8 -> 60.8 us on the layer2 3x3)
    uint4 ri[1][NU], rj[1][NU];
    auto gload = [&](int kt) {
        constexpr int SET = 0;
        const int mrow = m_begin + kt * KS;
        if (TI == 128) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int m = mrow + (tid >> 4) + 16 * u, ch = tid & 15;
                ri[SET][u] = m < m_end ? *reinterpret_cast<const uint4*>(g.I + (size_t)m * g.CI + it * 128 + ch * 8) : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NU / 2; ++u) {
                const int m = mrow + (tid >> 3) + 32 * u, ch = tid & 7;
                ri[SET][u] = m < m_end ? *reinterpret_cast<const uint4*>(g.I + (size_t)m * g.CI + ch * 8) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int m = mrow + (tid >> 4) + 16 * u;
            bool ok = jlive && m < m_end;
            long long at = (long long)m * g.CJ + jshift;                     // (1x1, and 3x3 stride 1: input pixel = m + tap shift)
            if (TAPS == 9) {
                if (g.stride == 1) {
                    ok = ok && (unsigned)wy[u] < (unsigned)g.H && (unsigned)wx[u] < (unsigned)g.W;
                } else {                                                     // strided: input pixel from the output position
                    const int iy = wy[u] * g.stride + oy, ix = wx[u] * g.stride + ox;
                    ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                    at = ((long long)(wi[u] * g.Hin + iy) * g.Win + ix) * g.CJ + jc8;
                }
                wx[u] += KS;                                                 // the next k-step: KS output pixels on
                while (wx[u] >= wx_end) { wx[u] -= g.W; wy[u] += 1; }
                while (wy[u] >= wy_end) { wy[u] -= g.H; wi[u] += 1; }
            }
            rj[SET][u] = ok ? *reinterpret_cast<const uint4*>(g.J + at) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lds_store = [&](int buf) {
        constexpr int SET = 0;
        unsigned char* base = smem + buf * 2 * TILE_BYTES;
        if (TI == 128) {
#pragma unroll
            for (int u = 0; u < NU; ++u) *reinterpret_cast<uint4*>(base + ws_off((tid >> 4) + 16 * u, tid & 15)) = ri[SET][u];
        } else {
#pragma unroll
            for (int u = 0; u < NU / 2; ++u) *reinterpret_cast<uint4*>(base + ws_off((tid >> 3) + 32 * u, tid & 7)) = ri[SET][u];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) *reinterpret_cast<uint4*>(base + TILE_BYTES + ws_off((tid >> 4) + 16 * u, tid & 15)) = rj[SET][u];
    };
    // transposed fragment: 32 columns starting at c0 (a multiple of 32), reduction rows kk*16 + 8*(lane half) .. +7
    const int grp = lane >> 4, t16 = lane & 15;
    const int q = t16 >> 2, p = t16 & 3;
    auto frag = [&](const unsigned char* tile, int c0, int kk) -> ws_bf16x8 {
        const int kb = kk * 16 + 8 * (grp >> 1);
        const int ch = ((c0 + 16 * (grp & 1)) >> 3) + (p >> 1);
        typedef ws_s16x4 __attribute__((address_space(3))) * lds_p;
        const ws_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + ws_off(kb + q, ch) + 8 * (p & 1)));
        const ws_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + ws_off(kb + 4 + q, ch) + 8 * (p & 1)));
        return __builtin_bit_cast(ws_bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    if (nk > 0) {
        gload(0);
        lds_store(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* ta = smem + buf * 2 * TILE_BYTES;
        const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS / 16; ++kk) {
            ws_bf16x8 fa[2], fb[NB];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = frag(ta, wm * 64 + a * 32, kk);
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b] = frag(tb, (TI == 128 ? wn * 64 : wn * 32) + b * 32, kk);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = H16<F16>::mfma32(fa[a], fb[b], acc[a][b]);
        }
        if (kt + 1 < nk) lds_store(buf ^ 1);
        __syncthreads();
    }

    float* Pt = g.P + (((size_t)split * g.ni + it) * g.nj + jt) * (size_t)(TI * 128);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = (TI == 128 ? wn * 64 : wn * 32) + b * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Pt[(size_t)i * 128 + j] = acc[a][b][r];
            }
        }
}

// dW[n][k][tap] = Σ_s P[s][tile of (n, k, tap)].  Few outputs (4096 for a 64 x 64 1x1), many pixel ranges (up to ~1000):
// a block of 256 threads = 16 consecutive k x 16 range groups; group g adds the ranges s = g, g + 16, ... in ascending order
// (16 lanes read 64 contiguous bytes of a partial tile), the 16 group sums are then added in ascending g — a fixed order, so
// the result is bitwise reproducible.  (One thread per output walking all ranges: 262 us for the 64 -> 64 1x1.)
__global__ __launch_bounds__(256) void wgrad_small_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int N,
                                                                 int K, int taps, int transposed, int TI, int ni, int nj,
                                                                 int nsplit)
{
    __shared__ float s_part[16][17];
    const int kl = threadIdx.x & 15, gsel = threadIdx.x >> 4;
    const long long idx = (long long)blockIdx.x * 16 + kl;              // element along [n][tap][k], k fastest (K % 16 == 0)
    const int k = (int)(idx % K);
    const int t = (int)((idx / K) % taps), n = (int)(idx / ((long long)K * taps));
    const int ci = transposed ? k : n, cj = transposed ? n : k;       // channel of the i / j operand
    const int CI = transposed ? K : N, CJ = transposed ? N : K;
    const int it = CI >= 128 ? ci / 128 : 0, ic = CI >= 128 ? ci % 128 : ci;
    int jt, jc;
    if (CJ >= 128) { jt = t * (CJ / 128) + cj / 128; jc = cj % 128; }
    else if (taps == 9) { jt = t >> 1; jc = (t & 1) * 64 + cj; }
    else { jt = 0; jc = cj; }
    const size_t tile = (size_t)TI * 128, per = (size_t)ni * nj * tile;
    // transposed: consecutive k are consecutive ROWS of a tile (stride 128 floats) — still one thread per element, the
    // partial reads are then 16 x 4 bytes 512 B apart (the transposed shapes have 64 x N outputs only)
    const float* p = P + ((size_t)it * nj + jt) * tile + (size_t)ic * 128 + jc;
    float acc = 0.f;
    for (int s = gsel; s < nsplit; s += 16) acc += p[(size_t)s * per];
    s_part[gsel][kl] = acc;
    __syncthreads();
    if (gsel == 0) {
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += s_part[g2][kl];
        dw[((size_t)n * K + k) * taps + t] = v;
    }
}

struct WSPlan {
    int ok, transposed, TI, ni, nj, nsplit, mps;
    size_t bytes;
};

static WSPlan ws_plan(int B, int H, int W, int Cin, int Cout, int taps)   // H, W: OUTPUT map
{
    WSPlan p = {};
    auto chan = [](int c) { return c == 64 || (c >= 128 && c <= 2048 && c % 128 == 0); };
    if (B <= 0 || H <= 0 || W <= 0 || !chan(Cin) || !chan(Cout) || (taps != 1 && taps != 9)) return p;
    const long long M = (long long)B * H * W;
    if (M >= (1ll << 24) || (size_t)M * (Cin > Cout ? Cin : Cout) * 2 >= (1ull << 40)) return p;
    // i operand = dY (tile rows = n) unless that wastes half of every tile: a 1x1 with 64 input and >= 128 output channels
    // is computed transposed (i operand = X, TI = 64)
    p.transposed = taps == 1 && Cin == 64 && Cout >= 128;
    const int CI = p.transposed ? Cin : Cout, CJ = p.transposed ? Cout : Cin;
    p.TI = CI == 64 ? 64 : 128;
    p.ni = CI == 64 ? 1 : CI / 128;
    p.nj = CJ >= 128 ? taps * (CJ / 128) : (taps == 9 ? 5 : 1);
    const long long tiles = (long long)p.ni * p.nj;
    const size_t tile_bytes = (size_t)p.TI * 128 * 4;
    long long s = 2 * hiast_grid_cus() / tiles;                                             // 2 blocks of 256 threads per CU (64 KiB of LDS each)
    const long long cap = (long long)((48ull << 20) / (tiles * tile_bytes));   // partial tiles: at most 48 MiB
    s = s > cap ? cap : s;
    const long long smax = M / 512 > 0 ? M / 512 : 1;                     // at least 8 k-steps per block
    s = s < 1 ? 1 : (s > smax ? smax : s);
    p.nsplit = (int)s;
    int mps = (int)((M + s - 1) / s);
    p.mps = (mps + 63) / 64 * 64;
    p.nsplit = (int)((M + p.mps - 1) / p.mps);
    p.bytes = (size_t)p.nsplit * tiles * tile_bytes;
    p.ok = 1;
    return p;
}

}  // namespace hiast

extern "C" size_t hiast_conv_wgrad_small_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps)
{
    const hiast::WSPlan p = hiast::ws_plan(B, H, W, Cin, Cout, taps);
    return p.ok ? p.bytes : 0;
}

extern "C" int hiast_conv_wgrad_small_nhwc(const void* dy, const void* x, float* dw, int B, int H, int W, int Cin, int Cout,
                                           int taps, int stride, int dil, int fmt, void* workspace, size_t workspace_bytes,
                                           hiast_stream_t stream)
{
    // H, W: the INPUT map; a strided 3x3 ('same' padding = dil) is walked inside the kernel, a strided 1x1 is this call on
    // the subsampled input (stride must be 1 for taps = 1)
    if (stride <= 0 || (taps == 1 && stride != 1)) return HIAST_E_RANGE;
    const int Hin = H, Win = W;
    H = (H - 1) / stride + 1;
    W = (W - 1) / stride + 1;
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if (!dy || !x || !dw || !workspace) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || dil <= 0) return HIAST_E_ARG;
    const hiast::WSPlan p = hiast::ws_plan(B, H, W, Cin, Cout, taps);
    if (!p.ok) return HIAST_E_RANGE;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace)) & 15) return HIAST_E_RANGE;
    if (workspace_bytes < p.bytes) return HIAST_E_WS;
    hiast::WSArgs a;
    a.I = (const unsigned short*)(p.transposed ? x : dy);
    a.J = (const unsigned short*)(p.transposed ? dy : x);
    a.P = (float*)workspace;
    a.M = B * H * W;
    a.CI = p.transposed ? Cin : Cout;
    a.CJ = p.transposed ? Cout : Cin;
    a.H = H; a.W = W; a.dil = dil;
    a.stride = stride; a.Hin = Hin; a.Win = Win;
    if ((long long)B * Hin * Win * Cin * 2 >= (1ll << 40)) return HIAST_E_RANGE;
    a.m_per_split = p.mps; a.ni = p.ni; a.nj = p.nj;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(p.ni * p.nj), (unsigned)p.nsplit);
    const bool f16 = fmt == HIAST_FMT_FP16;
#define WS_L(TIV, T, F) hipLaunchKernelGGL((hiast::wgrad_small_kernel<TIV, T, F>), grid, dim3(256), 0, st, a)
    if (p.TI == 64) {
        if (taps == 1) { if (f16) WS_L(64, 1, true); else WS_L(64, 1, false); }
        else { if (f16) WS_L(64, 9, true); else WS_L(64, 9, false); }
    } else {
        if (taps == 1) { if (f16) WS_L(128, 1, true); else WS_L(128, 1, false); }
        else { if (f16) WS_L(128, 9, true); else WS_L(128, 9, false); }
    }
#undef WS_L
    HIAST_CHECK_LAUNCH();
    const long long total = (long long)Cout * taps * Cin;                 // a multiple of 16 (Cin % 64 == 0)
    hipLaunchKernelGGL(hiast::wgrad_small_reduce_kernel, dim3((unsigned)(total / 16)), dim3(256), 0, st,
                       (const float*)workspace, dw, Cout, Cin, taps, p.transposed, p.TI, p.ni, p.nj, p.nsplit);
    HIAST_CHECK_LAUNCH();
    return 0;
}
