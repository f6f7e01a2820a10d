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

// XCD-aware work assignment: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs
// (each with a private 4 MB L2).  Work items are ordered image-major, so giving XCD k the k-th contiguous
// eighth of them keeps all tiles of one image (whose shifted re-reads share a channel slab) on one L2.
// Placement only changes speed, never results.
__device__ __forceinline__ void xcd_remap(int& bx, int& by, int& bz)
{
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int total = gx * gy * gz;
    int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((total & 7) == 0) lid = (lid & 7) * (total >> 3) + (lid >> 3);
    // item order: image (y) major, then split (z), then pixel tile (x)
    bx = lid % gx;
    bz = (lid / gx) % gz;
    by = lid / (gx * gz);
}

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
    int bx, n, split;
    xcd_remap(bx, n, split);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, ksub = lane >> 5;
    const int ci0 = split * ci_per_split, ci1 = ci0 + ci_per_split;

    int pix[MT], py[MT], px[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        pix[m] = (bx * 4 + wave) * (32 * MT) + m * 32 + col;
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

// ---- forward, no row padding: output channels 0..15 on v_mfma_f32_16x16x4_f32 (rows = channel,
// columns = 16 consecutive pixels, k = 4 input channels), channels 16..16+NV-1 (NV <= 3) on the
// otherwise idle VALU: every lane already holds x[ci = k0 + (lane>>4)][pixel = lane&15] as its B
// operand, so it adds w[c][ci] * x to NV private partial sums that are folded over the four k-lanes
// at the end.  For C = 19 this removes the 19 -> 32 padding of the 32x32 variant (41 % of its MFMA work).
// Wave = NP groups of 16 pixels; block = 4 waves.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC16 = 64;      // input channels per staged chunk of the 16-row kernel

template <int NP, int NV>
__global__ __launch_bounds__(256) void aspp_fwd16_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ wpack,
                                                         float* __restrict__ out, int Cin, int h, int w,
                                                         int Cout, Taps taps, int ci_per_split, int add_bias)
{
    __shared__ float s_w[2][KC16 * COP];
    const int hw = h * w;
    int bx, n, split;
    xcd_remap(bx, n, split);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int ci0 = split * ci_per_split;

    int pix[NP], py[NP], px[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        pix[g] = (bx * 4 + wave) * (16 * NP) + g * 16 + col;
        const int pc = pix[g] < hw ? pix[g] : hw - 1;
        py[g] = pc / w;
        px[g] = pc - py[g] * w;
    }
    // Input operand through a raw buffer descriptor over image n: per-lane byte offset in a VGPR (fixed
    // for a whole (chunk, tap)), channel advance in the SCALAR offset (no per-load VALU address math),
    // and out-of-image lanes get an offset beyond num_records, for which the hardware returns 0:
    // the conv's zero padding costs neither a branch nor a select.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (size_t)n * Cin * hw), 0, (int)((size_t)Cin * hw * sizeof(float)), 0x00020000);

    f32x4 acc[NP];
    float accv[NV > 0 ? NV : 1][NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < (NV > 0 ? NV : 1); ++v) accv[v][g] = 0.f;
    }

    const int nchunk = ci_per_split / KC16;
    const int niter = nchunk * NTAP;
    auto wsrc = [&](int it) {
        const int chunk2 = it / NTAP, tap2 = it - chunk2 * NTAP;
        return reinterpret_cast<const float4*>(wpack + ((size_t)tap2 * Cin + ci0 + chunk2 * KC16) * COP);
    };
    {
        const float4* src = wsrc(0);
        reinterpret_cast<float4*>(s_w[0])[threadIdx.x] = src[threadIdx.x];
        reinterpret_cast<float4*>(s_w[0])[threadIdx.x + 256] = src[threadIdx.x + 256];
    }
    __syncthreads();

    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < niter;
        float4 nx0 = make_float4(0, 0, 0, 0), nx1 = nx0;
        if (more) {
            const float4* src = wsrc(it + 1);
            nx0 = src[threadIdx.x];
            nx1 = src[threadIdx.x + 256];
        }
        const int chunk = it / NTAP, tap = it - chunk * NTAP;
        const int cib = ci0 + chunk * KC16;
        const int dy = taps.dy[tap], dx = taps.dx[tap];
        int voff[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const int yy = py[g] + dy, xx = px[g] + dx;
            const bool ok = pix[g] < hw && yy >= 0 && yy < h && xx >= 0 && xx < w;
            voff[g] = ok ? (kq * hw + yy * w + xx) * 4 : (int)0x80000000;
        }
        const int sbase = cib * hw * 4;
        float bs[KC16 / 4][NP];
#pragma unroll
        for (int ks = 0; ks < KC16 / 4; ++ks)
#pragma unroll
            for (int g = 0; g < NP; ++g)
                bs[ks][g] = __int_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[g], sbase + ks * 4 * hw * 4, 0));
        const float* sw = s_w[buf] + kq * COP;
#pragma unroll
        for (int ks = 0; ks < KC16 / 4; ++ks) {
            const float a = sw[ks * 4 * COP + col];                          // A[co = col][k = kq]
            float wv[4] = {0.f, 0.f, 0.f, 0.f};
            if (NV > 0) {
                const float4 t = *reinterpret_cast<const float4*>(sw + ks * 4 * COP + 16);   // co 16..19
                wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w;
            }
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                const float bv = bs[ks][g];
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[g], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) accv[v][g] = fmaf(wv[v], bv, accv[v][g]);
            }
        }
        if (more) {
            reinterpret_cast<float4*>(s_w[buf ^ 1])[threadIdx.x] = nx0;
            reinterpret_cast<float4*>(s_w[buf ^ 1])[threadIdx.x + 256] = nx1;
        }
        __syncthreads();
    }

    const float* bias = wpack + (size_t)NTAP * Cin * COP;
    float* on = out + ((size_t)split * gridDim.y + n) * Cout * hw;
#pragma unroll
    for (int g = 0; g < NP; ++g) {
        if (pix[g] < hw) {          // MFMA rows: co = 4*(lane>>4) + r, column = pixel
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * kq + r;
                if (co < Cout) on[(size_t)co * hw + pix[g]] = acc[g][r] + (add_bias ? bias[co] : 0.f);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float t = accv[v][g];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            const int co = 16 + v;
            if (kq == 0 && pix[g] < hw && co < Cout)
                on[(size_t)co * hw + pix[g]] = t + (add_bias ? bias[co] : 0.f);
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
// v_mfma_f32_16x16x4_f32: rows = co 0..15, columns = 16 input channels, k = 4 pixels; channels
// co 16..18 on the VALU (as in the forward kernel).  Both operands are pixel-contiguous in NCHW, so each
// lane fetches FOUR consecutive pixels with one 16-byte load and feeds them to four MFMAs: the k index of
// MFMA t is pixel 4*kq + t of a 16-pixel step (the same permutation on both operands, so the sum over k
// is unchanged).  No LDS: dY rows are shared through L1, X planes stream from L2.
// Wave = 2 channel tiles (32 channels) x the 8 ring taps of one dilation (+ the centre tap for
// dilation 0); block = 4 waves = 128 channels; pixel-range split-K over blockIdx.z, partials reduced
// in fixed order by aspp_wgrad_reduce_kernel.  Row shifts that leave the image are skipped (wave
// uniform); column shifts are masked per element only in the steps that touch a row end.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int NV, int NTILE>
__global__ __launch_bounds__(256) void aspp_bwd_weight_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dy,
                                                              float* __restrict__ partial, int B,
                                                              int Cin, int h, int w, int Cout,
                                                              int dil0, int dil1, int dil2, int dil3,
                                                              int nsplit)
{
    const int hw = h * w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int cib = (blockIdx.x * 4 + wave) * (16 * NTILE);
    const int g = blockIdx.y;
    const int split = blockIdx.z;
    const int d = g == 0 ? dil0 : (g == 1 ? dil1 : (g == 2 ? dil2 : dil3));
    const bool has_centre = (g == 0);
    constexpr int NT = 9;                      // 8 ring taps + centre (only accumulated when g == 0)

    f32x4 acc[NTILE][NT];
    float accv[NV > 0 ? NV : 1][NTILE][NT];
#pragma unroll
    for (int tt = 0; tt < NTILE; ++tt)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            acc[tt][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < (NV > 0 ? NV : 1); ++v) accv[v][tt][k] = 0.f;
        }

    const int steps_per_row = (w + 15) / 16;
    const int total = B * h * steps_per_row;
    const int per = (total + nsplit - 1) / nsplit;
    const int u0 = split * per, u1 = (u0 + per < total) ? u0 + per : total;

    // The loop walks INPUT rows r: the X vector x[ci][r][c0+ox..] (3 column shifts) is fetched once and
    // meets dY of the three output rows y = r - oy that use it, so each X row leaves L2 once per dilation
    // instead of three times (an output-row loop re-fetched every row 3x per dilation: 10x HBM over-fetch
    // in the PMC profile).
    for (int u = u0; u < u1; ++u) {
        const int st = u % steps_per_row;
        const int row = (u / steps_per_row) % h;             // input row r
        const int n = u / (steps_per_row * h);
        const int c0 = st * 16 + 4 * kq;                     // this lane's first pixel column
        const float* dyn = dy + (size_t)n * Cout * hw;
        const float* xr = x + ((size_t)n * Cin + cib + j) * hw + row * w;
        const bool full = c0 + 3 < w;

        float bv[NTILE][3][4];
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int ox = (sx - 1) * d;
            const int stc = st * 16 + ox;                    // first source column of the step
            const bool edge = stc < 0 || stc + 15 >= w;      // wave-uniform
            const int sc = c0 + ox;
#pragma unroll
            for (int tt = 0; tt < NTILE; ++tt) {
                const float* p = xr + (size_t)tt * 16 * hw + sc;
                if (!edge) {
                    const f32x4u t = *reinterpret_cast<const f32x4u*>(p);
                    bv[tt][sx][0] = t.x; bv[tt][sx][1] = t.y; bv[tt][sx][2] = t.z; bv[tt][sx][3] = t.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int cc = sc + t;
                        const bool in = cc >= 0 && cc < w;
                        const float vload = p[in ? t : -sc];  // redirect to column 0 of the row: valid address
                        bv[tt][sx][t] = in ? vload : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int qy = 0; qy < 3; ++qy) {
            const int y = row - (qy - 1) * d;                // output row whose tap (qy, *) reads input row r
            if (y < 0 || y >= h) continue;                   // wave-uniform
            float av[4] = {0.f, 0.f, 0.f, 0.f};
            float dv[NV > 0 ? NV : 1][4];
            if (j < Cout) {
                const float* p = dyn + (size_t)j * hw + y * w + c0;
                if (full) { const f32x4u t = *reinterpret_cast<const f32x4u*>(p); av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w; }
                else { for (int t = 0; t < 4; ++t) av[t] = (c0 + t < w) ? p[t] : 0.f; }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float* p = dyn + (size_t)(16 + v) * hw + y * w + c0;
                if (16 + v < Cout) {
                    if (full) { const f32x4u t = *reinterpret_cast<const f32x4u*>(p); dv[v][0] = t.x; dv[v][1] = t.y; dv[v][2] = t.z; dv[v][3] = t.w; }
                    else { for (int t = 0; t < 4; ++t) dv[v][t] = (c0 + t < w) ? p[t] : 0.f; }
                } else { dv[v][0] = dv[v][1] = dv[v][2] = dv[v][3] = 0.f; }
            }
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int pos = qy * 3 + sx;
                if (pos == 4 && !has_centre) continue;
                const int k = pos == 4 ? 8 : (pos < 4 ? pos : pos - 1);
#pragma unroll
                for (int tt = 0; tt < NTILE; ++tt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[tt][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[tt][sx][t], acc[tt][k], 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < NV; ++v) accv[v][tt][k] = fmaf(dv[v][t], bv[tt][sx][t], accv[v][tt][k]);
                    }
            }
        }
    }

    // partial[split][tap][co 32][Cin]; MFMA rows co = 4*kq + r, column ci = cib + 16*tt + j
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (k == 8 && !has_centre) continue;
        const int tap = k == 8 ? 0 : 1 + 8 * g + k;
        float* dst = partial + (((size_t)split * NTAP + tap) * COP) * Cin + cib + j;
#pragma unroll
        for (int tt = 0; tt < NTILE; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(4 * kq + r) * Cin + tt * 16] = acc[tt][k][r];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float t = accv[v][tt][k];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (kq == 0) dst[(size_t)(16 + v) * Cin + tt * 16] = t;
            }
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

static int pick_splitk(int B, int hw, int Cin, int px_per_block)
{
    const int blocks = ((hw + px_per_block - 1) / px_per_block) * B;
    const char* env = getenv("HIAST_ASPP_SPLITK");          // tuning override
    if (env && atoi(env) > 0) {
        int s = atoi(env);
        while (s > 1 && ((Cin / s) % KC16 != 0 || Cin % s != 0)) s >>= 1;
        return s > 8 ? 8 : s;
    }
    // >= 4 blocks (16 waves) per CU so that other waves' MFMAs cover a wave's load latency
    int s = 1;
    while (blocks * s < 1024 && s < 8 && (Cin / (s * 2)) % KC16 == 0 && Cin / (s * 2) >= 64) s *= 2;
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
    const int total = B * h * ((w + 15) / 16);      // 16-pixel steps
    int s = 16;
    while (s > 1 && total / s < 32) s >>= 1;
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
    const hiast::Taps taps = hiast::make_taps(dil);
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = Cout <= 19;                 // 16 MFMA rows + up to 3 VALU channels
    constexpr int MT = 2, NP = 4;
    const int px_per_block = narrow ? 4 * 16 * NP : 128 * MT;
    const int splitk = hiast::pick_splitk(B, hw, Cin, px_per_block);
    dim3 grid((hw + px_per_block - 1) / px_per_block, B, splitk);
    const long long per_split = (long long)B * Cout * hw;
    float* dst = y;
    if (splitk > 1) {
        if (!workspace || workspace_bytes < (size_t)splitk * per_split * sizeof(float)) return HIAST_E_WS;
        dst = (float*)workspace;
    }
    const int cps = Cin / splitk, ab = splitk == 1 ? 1 : 0;
#define F16(NV) hipLaunchKernelGGL((hiast::aspp_fwd16_kernel<NP, NV>), grid, dim3(256), 0, st, x, wpack, dst, Cin, \
                                   h, w, Cout, taps, cps, ab)
    if (narrow) {
        switch (Cout > 16 ? Cout - 16 : 0) {
            case 0: F16(0); break;
            case 1: F16(1); break;
            case 2: F16(2); break;
            default: F16(3); break;
        }
    } else {
        hipLaunchKernelGGL(hiast::aspp_fwd_kernel<MT>, grid, dim3(256), 0, st, x, wpack, dst, Cin, h, w, Cout,
                           taps, cps, ab);
    }
#undef F16
    HIAST_CHECK_LAUNCH();
    if (splitk > 1) {
        hipLaunchKernelGGL(hiast::aspp_reduce_kernel, dim3((unsigned)((per_split + 255) / 256)), dim3(256), 0,
                           st, dst, wpack + (size_t)hiast::NTAP * Cin * hiast::COP, y, per_split, splitk, Cout, hw);
        HIAST_CHECK_LAUNCH();
    }
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
    const char* envt = getenv("HIAST_ASPP_WG_NTILE");      // tuning override
    const int ntile = (envt && atoi(envt) == 2) ? 2 : 1;
    if (Cin % (64 * ntile) != 0 || Cout > 19) return HIAST_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    // every (split, tap, co < Cout, ci) element of `partial` is written exactly once by the kernel
    dim3 grid(Cin / (64 * ntile), 4, nsplit);
#define WG(NV)                                                                                                  \
    if (ntile == 2)                                                                                             \
        hipLaunchKernelGGL((hiast::aspp_bwd_weight_kernel<NV, 2>), grid, dim3(256), 0, st, x, dy, partial, B, Cin, \
                           h, w, Cout, dil[0], dil[1], dil[2], dil[3], nsplit);                                  \
    else                                                                                                        \
        hipLaunchKernelGGL((hiast::aspp_bwd_weight_kernel<NV, 1>), grid, dim3(256), 0, st, x, dy, partial, B, Cin, \
                           h, w, Cout, dil[0], dil[1], dil[2], dil[3], nsplit)
    switch (Cout > 16 ? Cout - 16 : 0) {
        case 0: WG(0); break;
        case 1: WG(1); break;
        case 2: WG(2); break;
        default: WG(3); break;
    }
#undef WG
    HIAST_CHECK_LAUNCH();
    const long long total = 4ll * Cout * Cin * 9;
    hipLaunchKernelGGL(hiast::aspp_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       st, partial, dw0, dw1, dw2, dw3, Cin, Cout, nsplit);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::aspp_db_kernel, dim3(Cout), dim3(256), 0, st, dy, db, B, Cout, h * w);
    HIAST_CHECK_LAUNCH();
    return 0;
}
