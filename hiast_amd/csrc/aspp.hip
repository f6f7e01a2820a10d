// K1 — ASPP head of DeepLab-V2: y = Σ_{d in dil} conv3x3(x; W_d, b_d, dilation=d, padding=d)
// (reference: ASPP_V2.forward, sseg/models/modules/seg_models/deeplab_v2.py:20-24, 4 cuDNN convs
// + 3 in-place adds).  gfx950 only: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf
// chain), no reduced-precision path, so logits stay fp32-exact-class (<=1e-3 rel is the
// contract; measured ~1e-6).
//
// One implicit GEMM over K = 33 taps x Cin (the four centre taps read the same input pixel and
// are pre-summed into tap 0):   Y[co][p] = Σ_tap Σ_ci Wp[tap][ci][co] * X[ci][p + off(tap)]
// MFMA orientation: rows = output channel (19 padded to 32), columns = 32 consecutive pixels, so
// accumulator columns are contiguous pixels and stores coalesce.
//   A (weights)  : staged per (tap, 32-channel chunk) through LDS, shared by the 4 waves
//   B (input)    : read straight from L2/MALL (each lane one dword per MFMA; a 64-cycle fp32 MFMA
//                  hides it), zero-filled outside the image
// Loop order is channel-chunk OUTER / tap INNER so the 33 shifted re-reads of a channel plane hit
// L2 while it is resident.  Split-K over channel ranges fills the chip at small batch; partials
// are reduced in a fixed order (bitwise reproducible).
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NTAP = 33;
constexpr int KC = 32;        // input channels per staged weight chunk
constexpr int COP = 32;       // padded output channels

struct Taps {
    int dy[NTAP];
    int dx[NTAP];
};

static Taps make_taps(const int* dil)
{
    Taps t;
    t.dy[0] = 0; t.dx[0] = 0;
    int k = 1;
    for (int d = 0; d < 4; ++d)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                if (ky == 1 && kx == 1) continue;
                t.dy[k] = (ky - 1) * dil[d];
                t.dx[k] = (kx - 1) * dil[d];
                ++k;
            }
    return t;
}

// wpack[tap][ci][32] (co >= Cout zero) followed by bias_sum[32]
__global__ __launch_bounds__(256) void aspp_pack_kernel(const float* __restrict__ w0,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ w2,
                                                        const float* __restrict__ w3,
                                                        const float* __restrict__ b0,
                                                        const float* __restrict__ b1,
                                                        const float* __restrict__ b2,
                                                        const float* __restrict__ b3, int Cin,
                                                        int Cout, float* __restrict__ wpack)
{
    const long long total = (long long)NTAP * Cin * COP;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < COP) {
        const int co = (int)idx;
        wpack[total + co] = co < Cout ? ((b0[co] + b1[co]) + b2[co]) + b3[co] : 0.f;
    }
    if (idx >= total) return;
    const int co = (int)(idx % COP);
    const int ci = (int)((idx / COP) % Cin);
    const int tap = (int)(idx / ((long long)COP * Cin));
    float v = 0.f;
    if (co < Cout) {
        const size_t base = ((size_t)co * Cin + ci) * 9;
        if (tap == 0) {
            v = ((w0[base + 4] + w1[base + 4]) + w2[base + 4]) + w3[base + 4];
        } else {
            const int d = (tap - 1) >> 3;
            int k = (tap - 1) & 7;
            k += (k >= 4) ? 1 : 0;                       // skip the centre position
            const float* wd = d == 0 ? w0 : (d == 1 ? w1 : (d == 2 ? w2 : w3));
            v = wd[base + k];
        }
    }
    wpack[idx] = v;
}

// Block: 256 threads = 4 waves; wave handles MT tiles of 32 consecutive pixels.
// grid = (ceil(hw / (128*MT)), B, SPLITK)
template <int MT>
__global__ __launch_bounds__(256) void aspp_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ wpack,
                                                       float* __restrict__ out, int Cin, int h,
                                                       int w, int Cout, Taps taps, int ci_per_split,
                                                       int add_bias)
{
    __shared__ float s_w[2][KC * COP];
    const int hw = h * w;
    const int n = blockIdx.y;
    const int split = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, ksub = lane >> 5;
    const int ci0 = split * ci_per_split, ci1 = ci0 + ci_per_split;

    int pix[MT], py[MT], px[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        pix[m] = (blockIdx.x * 4 + wave) * (32 * MT) + m * 32 + col;
        const int pc = pix[m] < hw ? pix[m] : hw - 1;
        py[m] = pc / w;
        px[m] = pc - py[m] * w;
    }
    const float* xn = x + (size_t)n * Cin * hw;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // prologue: stage weights of (chunk ci0, tap 0)
    const int nchunk = ci_per_split / KC;
    const int niter = nchunk * NTAP;
    {
        const float4 v = reinterpret_cast<const float4*>(wpack + ((size_t)0 * Cin + ci0) * COP)[threadIdx.x];
        reinterpret_cast<float4*>(s_w[0])[threadIdx.x] = v;
    }
    __syncthreads();

    // B operand: the 16 k-steps x MT pixel tiles of one (chunk, tap) are fetched as ONE batch of
    // unconditional loads (addresses of out-of-image lanes are redirected to a valid element and the
    // value is zeroed by a select afterwards: a conditional load would be lowered to a branch plus
    // s_waitcnt vmcnt(0) per element), one iteration AHEAD of the MFMAs that consume them.
    auto tap_setup = [&](int it, const float* (&xp)[MT], bool (&ok)[MT]) {
        const int chunk = it / NTAP, tap = it - chunk * NTAP;
        const int cib = ci0 + chunk * KC;
        const int dy = taps.dy[tap], dx = taps.dx[tap];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int yy = py[m] + dy, xx = px[m] + dx;
            ok[m] = pix[m] < hw && yy >= 0 && yy < h && xx >= 0 && xx < w;
            xp[m] = xn + (size_t)(cib + ksub) * hw + (ok[m] ? yy * w + xx : 0);
        }
    };
    // One batch of KC/2 x MT unconditional loads per (chunk, tap); the MFMAs start as the first
    // values land (counted vmcnt).  Latency is hidden across the 4-5 waves per SIMD that the register
    // budget admits (the host picks the split-K factor so that the grid provides them).
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < niter;
        float4 nxt = make_float4(0, 0, 0, 0);
        if (more) {
            const int chunk2 = (it + 1) / NTAP, tap2 = (it + 1) - chunk2 * NTAP;
            nxt = reinterpret_cast<const float4*>(wpack + ((size_t)tap2 * Cin + ci0 + chunk2 * KC) * COP)[threadIdx.x];
        }
        const float* xp[MT];
        bool ok[MT];
        tap_setup(it, xp, ok);
        float bs[MT][KC / 2];
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) bs[m][ks] = xp[m][(size_t)ks * 2 * hw];
        const float* sw = s_w[buf] + ksub * COP + col;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const float a = sw[ks * 2 * COP];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float bv = ok[m] ? bs[m][ks] : 0.f;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[m], 0, 0, 0);
            }
        }
        if (more) reinterpret_cast<float4*>(s_w[buf ^ 1])[threadIdx.x] = nxt;
        __syncthreads();
    }
    (void)ci1;

    // epilogue: rows = co = (r&3) + 8*(r>>2) + 4*(lane>>5), column = pixel
    const float* bias = wpack + (size_t)NTAP * Cin * COP;
    float* on = out + ((size_t)split * gridDim.y + n) * Cout * hw;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (pix[m] >= hw) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * ksub;
            if (co < Cout) on[(size_t)co * hw + pix[m]] = acc[m][r] + (add_bias ? bias[co] : 0.f);
        }
    }
}

// y[i] = bias + Σ_s partial[s][i], ascending s
__global__ __launch_bounds__(256) void aspp_reduce_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, long long per_split,
                                                          int nsplit, int Cout, int hw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_split) return;
    const int co = (int)((i / hw) % Cout);
    float acc = partial[i];
    for (int s = 1; s < nsplit; ++s) acc += partial[(size_t)s * per_split + i];
    y[i] = acc + bias[co];
}

// ------------------------------------------------------------------------------------ bwd data
// dX[ci][p] = Σ_tap Σ_co Wp[tap][ci][co] * dY[co][p - off(tap)]   (K = 33 taps x Cout, Cout padded
// to an even 20).  MFMA rows = input channel, columns = 32 consecutive pixels -> dX (67 MB/img, the
// only large tensor here) is written once with coalesced 128-B row segments.
// Block tile = 256 channels x 64 pixels; wave = 64 channels x 64 pixels (4 accumulators).
constexpr int DG_CI = 256;
constexpr int DG_K = 20;      // Cout padded to a multiple of 2 (k-step of the 32x32x2 MFMA)

__global__ __launch_bounds__(256) void aspp_bwd_data_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ wpack,
                                                            float* __restrict__ dx, int Cin, int h,
                                                            int w, int Cout, Taps taps)
{
    __shared__ float s_w[2][DG_K * DG_CI];
    const int hw = h * w;
    const int n = blockIdx.z;
    const int cib = blockIdx.y * DG_CI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, ksub = lane >> 5;

    int pix[2], py[2], px[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        pix[m] = blockIdx.x * 64 + m * 32 + col;
        const int pc = pix[m] < hw ? pix[m] : hw - 1;
        py[m] = pc / w;
        px[m] = pc - py[m] * w;
    }
    const float* dyn = dy + (size_t)n * Cout * hw;

    f32x16 acc[2][2];   // [ci tile][pixel tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][m][r] = 0.f;

    // weights of one tap for this thread's channel: wpack[tap][cib + t][0..19]
    const float* wsrc = wpack + (size_t)(cib + threadIdx.x) * COP;
    float4 reg[DG_K / 4];
#pragma unroll
    for (int q = 0; q < DG_K / 4; ++q) reg[q] = reinterpret_cast<const float4*>(wsrc)[q];
#pragma unroll
    for (int q = 0; q < DG_K / 4; ++q) {
        s_w[0][(4 * q + 0) * DG_CI + threadIdx.x] = reg[q].x;
        s_w[0][(4 * q + 1) * DG_CI + threadIdx.x] = reg[q].y;
        s_w[0][(4 * q + 2) * DG_CI + threadIdx.x] = reg[q].z;
        s_w[0][(4 * q + 3) * DG_CI + threadIdx.x] = reg[q].w;
    }
    __syncthreads();

    for (int tap = 0; tap < NTAP; ++tap) {
        const int buf = tap & 1;
        if (tap + 1 < NTAP) {
            const float* nsrc = wsrc + (size_t)(tap + 1) * Cin * COP;
#pragma unroll
            for (int q = 0; q < DG_K / 4; ++q) reg[q] = reinterpret_cast<const float4*>(nsrc)[q];
        }
        // dX[p] gathers dY[p - off]
        const int oy = taps.dy[tap], ox = taps.dx[tap];
        const float* bp[2];
        bool ok[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int yy = py[m] - oy, xx = px[m] - ox;
            ok[m] = pix[m] < hw && yy >= 0 && yy < h && xx >= 0 && xx < w;
            bp[m] = dyn + (ok[m] ? yy * w + xx : 0);
        }
        const float* sa = s_w[buf] + wave * 64 + col;
#pragma unroll
        for (int ks = 0; ks < DG_K / 2; ++ks) {
            const int co = 2 * ks + ksub;
            const float a0 = sa[co * DG_CI];
            const float a1 = sa[co * DG_CI + 32];
            const int coc = co < Cout ? co : Cout - 1;          // keep the address inside dY
            float b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float v = bp[m][(size_t)coc * hw];            // unconditional load, select after
                b[m] = (ok[m] && co < Cout) ? v : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[m], acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[m], acc[1][m], 0, 0, 0);
            }
        }
        if (tap + 1 < NTAP) {
#pragma unroll
            for (int q = 0; q < DG_K / 4; ++q) {
                s_w[buf ^ 1][(4 * q + 0) * DG_CI + threadIdx.x] = reg[q].x;
                s_w[buf ^ 1][(4 * q + 1) * DG_CI + threadIdx.x] = reg[q].y;
                s_w[buf ^ 1][(4 * q + 2) * DG_CI + threadIdx.x] = reg[q].z;
                s_w[buf ^ 1][(4 * q + 3) * DG_CI + threadIdx.x] = reg[q].w;
            }
        }
        __syncthreads();
    }

    float* dxn = dx + ((size_t)n * Cin + cib + wave * 64) * hw;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (pix[m] >= hw) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * ksub;
                dxn[(size_t)ci * hw + pix[m]] = acc[a][m][r];
            }
        }
}

// ---------------------------------------------------------------------------------- bwd weight
// dWp[tap][co][ci] = Σ_n Σ_p dY[n][co][p] * X[n][ci][p + off(tap)]   (K = B*h*w pixels)
// MFMA rows = co (padded 32), columns = 32 input channels; both operands are needed
// "pixel-minor", so a 64-pixel piece of a row is staged through LDS: the dY slab [32][64] and, for
// the block's dilation d, the three X slabs rows y-d, y, y+d x [32 ch][64 + 2d] (zero-filled
// outside the image), which serve the 8 ring taps of that dilation (+ the centre tap).
// Block = (32-channel tile, dilation, pixel-range split); its 4 waves share the slabs and own
// 2 taps each (wave 0 of dilation 0 also owns the centre tap).  Split partials are reduced in a
// fixed order by aspp_wgrad_reduce_kernel (bitwise reproducible, no float atomics).
constexpr int WG_KP = 64;
constexpr int WG_LDY = WG_KP + 1;

__global__ __launch_bounds__(256) void aspp_bwd_weight_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dy,
                                                              float* __restrict__ partial, int B,
                                                              int Cin, int h, int w, int Cout,
                                                              int dil0, int dil1, int dil2, int dil3,
                                                              int nsplit, int lxw)
{
    extern __shared__ float s_mem[];
    float* s_dy = s_mem;                       // [32][WG_LDY]
    float* s_x = s_mem + 32 * WG_LDY;          // [3][32][lxw]
    const int hw = h * w;
    const int cib = blockIdx.x * 32;
    const int g = blockIdx.y;
    const int split = blockIdx.z;
    const int d = g == 0 ? dil0 : (g == 1 ? dil1 : (g == 2 ? dil2 : dil3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, ksub = lane >> 5;

    // this wave's taps: ring positions 2*wave, 2*wave+1 of dilation g (k -> (ky,kx) skipping centre)
    int t_row[3], t_dx[3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int k = 2 * wave + t;
        k += (k >= 4) ? 1 : 0;
        t_row[t] = k / 3;                      // 0: y-d, 1: y, 2: y+d
        t_dx[t] = (k % 3 - 1) * d;
    }
    t_row[2] = 1; t_dx[2] = 0;                 // centre tap
    const bool has_centre = (g == 0 && wave == 0);

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int pieces_per_row = (w + WG_KP - 1) / WG_KP;
    const int total = B * h * pieces_per_row;
    const int per = (total + nsplit - 1) / nsplit;
    const int u0 = split * per, u1 = (u0 + per < total) ? u0 + per : total;

    for (int u = u0; u < u1; ++u) {
        const int piece = u % pieces_per_row;
        const int row = (u / pieces_per_row) % h;
        const int n = u / (pieces_per_row * h);
        const int x0 = piece * WG_KP;
        const float* xn = x + ((size_t)n * Cin + cib) * hw;
        const float* dyn = dy + (size_t)n * Cout * hw;
        __syncthreads();                       // previous piece fully consumed
        for (int idx = threadIdx.x; idx < 32 * WG_KP; idx += 256) {
            const int co = idx / WG_KP, c = idx - co * WG_KP;
            const int xx = x0 + c;
            s_dy[co * WG_LDY + c] = (co < Cout && xx < w) ? dyn[(size_t)co * hw + row * w + xx] : 0.f;
        }
        const int span = WG_KP + 2 * d;
        for (int idx = threadIdx.x; idx < 96 * span; idx += 256) {
            const int pr = idx / span, c = idx - pr * span;
            const int slab = pr >> 5, ci = pr & 31;
            const int yy = row + (slab - 1) * d;
            const int xx = x0 - d + c;
            const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
            s_x[(slab * 32 + ci) * lxw + c] = in ? xn[(size_t)ci * hw + yy * w + xx] : 0.f;
        }
        __syncthreads();
        const float* pa = s_dy + col * WG_LDY + ksub;
        const float* pb0 = s_x + (t_row[0] * 32 + col) * lxw + d + t_dx[0] + ksub;
        const float* pb1 = s_x + (t_row[1] * 32 + col) * lxw + d + t_dx[1] + ksub;
        const float* pb2 = s_x + (32 + col) * lxw + d + ksub;
#pragma unroll 8
        for (int ks = 0; ks < WG_KP / 2; ++ks) {
            const float a = pa[2 * ks];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb0[2 * ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb1[2 * ks], acc[1], 0, 0, 0);
            if (has_centre) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb2[2 * ks], acc[2], 0, 0, 0);
        }
    }

    // partial[split][tap][co 32][Cin]
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t == 2 && !has_centre) continue;
        const int tap = t == 2 ? 0 : 1 + 8 * g + 2 * wave + t;
        float* dst = partial + (((size_t)split * NTAP + tap) * COP) * Cin + cib + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * ksub;
            dst[(size_t)co * Cin] = acc[t][r];
        }
    }
}

// dW_d[co][ci][ky][kx] = Σ_split partial[split][tap(d,ky,kx)][co][ci], ascending split
__global__ __launch_bounds__(256) void aspp_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ dw0,
                                                                float* __restrict__ dw1,
                                                                float* __restrict__ dw2,
                                                                float* __restrict__ dw3, int Cin,
                                                                int Cout, int nsplit)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over [4][Cout][Cin][9]
    const long long total = 4ll * Cout * Cin * 9;
    if (idx >= total) return;
    const int k = (int)(idx % 9);
    const int ci = (int)((idx / 9) % Cin);
    const int co = (int)((idx / (9ll * Cin)) % Cout);
    const int d = (int)(idx / (9ll * Cin * Cout));
    const int tap = (k == 4) ? 0 : 1 + 8 * d + (k > 4 ? k - 1 : k);
    float acc = 0.f;
    for (int s = 0; s < nsplit; ++s) acc += partial[(((size_t)s * NTAP + tap) * COP + co) * Cin + ci];
    float* dw = d == 0 ? dw0 : (d == 1 ? dw1 : (d == 2 ? dw2 : dw3));
    dw[((size_t)co * Cin + ci) * 9 + k] = acc;
}

// db[co] = Σ_n Σ_p dY[n][co][p]  (one block per output channel, fixed order)
__global__ __launch_bounds__(256) void aspp_db_kernel(const float* __restrict__ dy,
                                                      float* __restrict__ db, int B, int Cout, int hw)
{
    __shared__ double s[256];
    const int co = blockIdx.x;
    double acc = 0.0;
    for (int n = 0; n < B; ++n) {
        const float* p = dy + ((size_t)n * Cout + co) * hw;
        for (int i = threadIdx.x; i < hw; i += 256) acc += (double)p[i];
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = (float)s[0];
}

static int pick_splitk(int B, int hw, int Cin, int MT)
{
    const int blocks = ((hw + 128 * MT - 1) / (128 * MT)) * B;
    const char* env = getenv("HIAST_ASPP_SPLITK");          // tuning override
    if (env && atoi(env) > 0) {
        int s = atoi(env);
        while (s > 1 && ((Cin / s) % KC != 0 || Cin % s != 0)) s >>= 1;
        return s > 8 ? 8 : s;
    }
    // >= 4 blocks (16 waves) per CU so that other waves' MFMAs cover a wave's load latency
    int s = 1;
    while (blocks * s < 1024 && s < 8 && (Cin / (s * 2)) % KC == 0 && Cin / (s * 2) >= 64) s *= 2;
    return s;
}

}  // namespace hiast

extern "C" size_t hiast_aspp_wpack_bytes(int Cin, int Cout)
{
    (void)Cout;
    return ((size_t)hiast::NTAP * Cin * hiast::COP + hiast::COP) * sizeof(float);
}

static int wgrad_nsplit(int B, int h, int w)
{
    const int total = B * h * ((w + hiast::WG_KP - 1) / hiast::WG_KP);
    int s = 8;
    while (s > 1 && total / s < 16) s >>= 1;
    return s;
}

extern "C" size_t hiast_aspp_workspace_bytes(int B, int Cin, int h, int w, int Cout)
{
    const size_t fwd = (size_t)8 * B * Cout * h * w * sizeof(float);          // <= 8 split-K partials
    const size_t wg = (size_t)wgrad_nsplit(B, h, w) * hiast::NTAP * hiast::COP * Cin * sizeof(float);
    return (fwd > wg ? fwd : wg) + 256;
}

static int aspp_check(int B, int Cin, int h, int w, int Cout)
{
    if (B <= 0 || Cin <= 0 || h <= 0 || w <= 0 || Cout <= 0) return HIAST_E_ARG;
    if (Cin % 64 != 0 || Cout > hiast::COP || B > 65535) return HIAST_E_RANGE;
    if ((long long)Cin * h * w >= (1ll << 31)) return HIAST_E_RANGE;
    return 0;
}

extern "C" int hiast_aspp_pack_weights(const float* w0, const float* w1, const float* w2,
                                       const float* w3, const float* b0, const float* b1,
                                       const float* b2, const float* b3, int Cin, int Cout,
                                       float* wpack, hiast_stream_t stream)
{
    if (!w0 || !w1 || !w2 || !w3 || !b0 || !b1 || !b2 || !b3 || !wpack) return HIAST_E_ARG;
    int e = aspp_check(1, Cin, 1, 1, Cout);
    if (e) return e;
    const long long total = (long long)hiast::NTAP * Cin * hiast::COP;
    hipLaunchKernelGGL(hiast::aspp_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w0, w1, w2, w3, b0, b1, b2, b3, Cin, Cout, wpack);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_aspp_fwd(const float* x, const float* wpack, float* y, int B, int Cin, int h,
                              int w, int Cout, const int* dil, void* workspace,
                              size_t workspace_bytes, hiast_stream_t stream)
{
    if (!x || !wpack || !y || !dil) return HIAST_E_ARG;
    int e = aspp_check(B, Cin, h, w, Cout);
    if (e) return e;
    const int hw = h * w;
    constexpr int MT = 2;
    const int splitk = hiast::pick_splitk(B, hw, Cin, MT);
    const hiast::Taps taps = hiast::make_taps(dil);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((hw + 128 * MT - 1) / (128 * MT), B, splitk);
    if (splitk == 1) {
        hipLaunchKernelGGL(hiast::aspp_fwd_kernel<MT>, grid, dim3(256), 0, st, x, wpack, y, Cin, h, w,
                           Cout, taps, Cin, 1);
        HIAST_CHECK_LAUNCH();
        return 0;
    }
    const long long per_split = (long long)B * Cout * hw;
    if (!workspace || workspace_bytes < (size_t)splitk * per_split * sizeof(float)) return HIAST_E_WS;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(hiast::aspp_fwd_kernel<MT>, grid, dim3(256), 0, st, x, wpack, partial, Cin, h, w,
                       Cout, taps, Cin / splitk, 0);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::aspp_reduce_kernel, dim3((unsigned)((per_split + 255) / 256)), dim3(256), 0,
                       st, partial, wpack + (size_t)hiast::NTAP * Cin * hiast::COP, y, per_split, splitk,
                       Cout, hw);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_aspp_bwd_data(const float* dy, const float* wpack, float* dx, int B, int Cin,
                                   int h, int w, int Cout, const int* dil, hiast_stream_t stream)
{
    if (!dy || !wpack || !dx || !dil) return HIAST_E_ARG;
    int e = aspp_check(B, Cin, h, w, Cout);
    if (e) return e;
    if (Cin % hiast::DG_CI != 0 || Cout > hiast::DG_K) return HIAST_E_RANGE;
    const int hw = h * w;
    const hiast::Taps taps = hiast::make_taps(dil);
    dim3 grid((hw + 63) / 64, Cin / hiast::DG_CI, B);
    hipLaunchKernelGGL(hiast::aspp_bwd_data_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, wpack, dx,
                       Cin, h, w, Cout, taps);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_aspp_bwd_weight(const float* x, const float* dy, float* dw0, float* dw1, float* dw2,
                                     float* dw3, float* db, int B, int Cin, int h, int w, int Cout,
                                     const int* dil, void* workspace, size_t workspace_bytes,
                                     hiast_stream_t stream)
{
    if (!x || !dy || !dw0 || !dw1 || !dw2 || !dw3 || !db || !dil || !workspace) return HIAST_E_ARG;
    int e = aspp_check(B, Cin, h, w, Cout);
    if (e) return e;
    int dmax = 0;
    for (int i = 0; i < 4; ++i) {
        if (dil[i] <= 0) return HIAST_E_ARG;
        dmax = dil[i] > dmax ? dil[i] : dmax;
    }
    if (dmax > 64) return HIAST_E_RANGE;
    const int nsplit = wgrad_nsplit(B, h, w);
    if (workspace_bytes < (size_t)nsplit * hiast::NTAP * hiast::COP * Cin * sizeof(float)) return HIAST_E_WS;
    int lxw = hiast::WG_KP + 2 * dmax;
    lxw |= 1;                                             // odd row pitch: conflict-free column reads
    const size_t lds = (size_t)(32 * hiast::WG_LDY + 96 * lxw) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    dim3 grid(Cin / 32, 4, nsplit);
    hipLaunchKernelGGL(hiast::aspp_bwd_weight_kernel, grid, dim3(256), lds, st, x, dy, partial, B, Cin, h, w,
                       Cout, dil[0], dil[1], dil[2], dil[3], nsplit, lxw);
    HIAST_CHECK_LAUNCH();
    const long long total = 4ll * Cout * Cin * 9;
    hipLaunchKernelGGL(hiast::aspp_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       st, partial, dw0, dw1, dw2, dw3, Cin, Cout, nsplit);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::aspp_db_kernel, dim3(Cout), dim3(256), 0, st, dy, db, B, Cout, h * w);
    HIAST_CHECK_LAUNCH();
    return 0;
}
