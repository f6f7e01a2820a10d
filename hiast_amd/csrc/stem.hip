// K9j — the stem of an INFERENCE forward in one kernel: conv 7x7 stride 2 pad 3 (3 -> 64) -> BatchNorm (eval) -> ReLU ->
// MaxPool2d(3, stride 2, padding 1), written in the operand format of the trunk kernels (reference: ResNet.forward,
// sseg/models/modules/resnet.py:180-184, `x = self.maxpool(self.relu(self.bn1(self.conv1(x))))`, in the eval forwards of the
// pseudo-label pass and of the EMA teacher).  gfx950 only.
//
// Round 2 ran the library's convolution (0.125 ms in 16 bits, 0.24 ms in fp32 at B = 8, 1024x512) and then hiast_stem_tail
// over its output (0.07 - 0.11 ms): the full-resolution conv output — 134 MB in 16 bits, 268 MB in fp32, the largest
// tensor of the whole forward — was written and read once for nothing.  Here a block owns a tile of 4 x 6 POOLED pixels:
//   * the 23 x 31 input pixels under it go to LDS as [row][pixel][4 channels] 16-bit (3 + a zero; two planes hi | lo for
//     the fp32-class format), double-buffered: the next tile's pixels are in flight during the MFMAs;
//   * the 9 x 13 convolution outputs the tile's pooling windows touch are an implicit GEMM on v_mfma_f32_16x16x32: K = 7 rows
//     x 32 (8 input pixels x 4 channels; 7 x 3 real), so a B fragment is ONE aligned ds_read_b128 of two neighbouring input
//     pixels; a wave owns 2 of the 4 output-channel tiles and 2 of the 8 pixel tiles and keeps its weight fragments (2 x 7,
//     x 2 planes) in registers for the whole kernel; split planes multiply hi*hi + lo*hi + hi*lo as the trunk does;
//   * BN scale / shift and ReLU on the accumulators, the 117 x 64 results to LDS (fp32), the 3 x 3 maximum from there
//     (a NaN wins, as in ATen), 8-byte stores of 4 channels in the output format.
// Blocks are persistent over tiles.  22 % of the convolution outputs are computed twice (halo of the pooling windows).
#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 st_bf16x8;
typedef __attribute__((ext_vector_type(4))) float st_f32x4;

constexpr int ST_PH = 4, ST_PW = 6;                   // pooled tile (9 x 13 = 117 conv outputs: 8 pixel tiles, one per wave)
constexpr int ST_CR = 2 * ST_PH + 1, ST_CC = 2 * ST_PW + 1;   // 9 x 17 convolution outputs
constexpr int ST_NPX = ST_CR * ST_CC;                 // 117
constexpr int ST_PT = (ST_NPX + 15) / 16;             // 8 pixel tiles of 16
constexpr int ST_IR = 2 * ST_CR + 5;                  // 23 input rows
constexpr int ST_IC = 32;                             // 2 * 13 + 5 = 31 input columns, pitch 32
constexpr int ST_OPITCH = 68;                         // floats per conv pixel in LDS (64 + 4: 16-byte aligned rows)

struct StemGeo {
    int B, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y;
};

template <bool F16>
__device__ __forceinline__ unsigned short st_enc(float v) { return H16<F16>::enc(v); }

// FMT: HIAST_FMT_BF16 (1) | HIAST_FMT_SPLIT_BF16 (2) | HIAST_FMT_FP16 (3)
template <int FMT>
__global__ __launch_bounds__(512) void stem_eval_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        float eps, unsigned short* __restrict__ out, StemGeo g)
{
    constexpr bool F16 = FMT == HIAST_FMT_FP16;
    constexpr int PL = FMT == HIAST_FMT_SPLIT_BF16 ? 2 : 1;
    constexpr int IN_BYTES = ST_IR * ST_IC * 8;          // one plane of one input tile
    constexpr int W_BYTES = 7 * 64 * 32 * 2;             // one plane of the weights
    __shared__ __attribute__((aligned(16))) unsigned char s_in[2 * PL * IN_BYTES];      // [buffer][plane]
    // one buffer, two lives: the packed weights [plane][ky][oc][32] during the prologue (every wave then keeps ITS fragments
    // in registers), the convolution outputs of a tile afterwards
    constexpr int U_BYTES = PL * W_BYTES > ST_PT * 16 * ST_OPITCH * 4 ? PL * W_BYTES : ST_PT * 16 * ST_OPITCH * 4;
    __shared__ __attribute__((aligned(16))) unsigned char s_u[U_BYTES];
    unsigned char* const s_w = s_u;
    float* const s_out = reinterpret_cast<float*>(s_u);
    __shared__ float s_sc[64], s_sh[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- once per block: weights -> LDS in A-fragment order, BN -> scale / shift
    // (all 28 loads of a thread first: as a loop of dependent load -> store rounds this prologue took ~40 us per block)
    float wv[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        const int e = tid + 512 * i;
        const int kq = e & 31, oc = (e >> 5) & 63, ky = e >> 11;
        const int kx = kq >> 2, ci = kq & 3;
        wv[i] = (kx < 7 && ci < 3) ? w[((oc * 3 + ci) * 7 + ky) * 7 + kx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        const int e = tid + 512 * i;
        const float v = wv[i];
        if (PL == 2) {
            unsigned short h, l;
            const unsigned u = __float_as_uint(v);
            // round to nearest even bf16, remainder again (the split of hiast_split_planes)
            const unsigned hb = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
            h = (unsigned short)hb;
            const float r = v - __uint_as_float(hb << 16);
            const unsigned ur = __float_as_uint(r);
            l = (unsigned short)((ur + 0x7FFFu + ((ur >> 16) & 1u)) >> 16);
            reinterpret_cast<unsigned short*>(s_w)[e] = h;
            reinterpret_cast<unsigned short*>(s_w + W_BYTES)[e] = l;
        } else {
            reinterpret_cast<unsigned short*>(s_w)[e] = st_enc<F16>(v);
        }
    }
    if (tid < 64) {
        const float sc = (gamma ? gamma[tid] : 1.0f) * (1.0f / sqrtf(var[tid] + eps));
        s_sc[tid] = sc;
        s_sh[tid] = fmaf(-mean[tid], sc, beta ? beta[tid] : 0.0f);
    }

    const int tiles_per_img = g.tiles_x * g.tiles_y;
    const int ntiles = g.B * tiles_per_img;

    // input staging: 23 x 32 positions, 2 per thread (tid, tid + 512 < 736)
    float rin[2][3];
    auto load_tile = [&](int t) {
        const int b = t / tiles_per_img, r = t % tiles_per_img;
        const int iy0 = 4 * (r / g.tiles_x) * ST_PH - 5, ix0 = 4 * (r % g.tiles_x) * ST_PW - 5;
        const float* xb = x + (size_t)b * 3 * g.H * g.W;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = tid + 512 * u;
            const int iy = iy0 + pos / ST_IC, ix = ix0 + pos % ST_IC;
            const bool ok = t < ntiles && pos < ST_IR * ST_IC && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const size_t o = (size_t)(ok ? iy : 0) * g.W + (ok ? ix : 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) rin[u][c] = ok ? xb[(size_t)c * g.H * g.W + o] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = tid + 512 * u;
            if (pos >= ST_IR * ST_IC) continue;
            unsigned short h[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = rin[u][c];
                if (PL == 2) {
                    const unsigned uu = __float_as_uint(v);
                    const unsigned hb = (uu + 0x7FFFu + ((uu >> 16) & 1u)) >> 16;
                    h[c] = (unsigned short)hb;
                    const unsigned ur = __float_as_uint(v - __uint_as_float(hb << 16));
                    l[c] = (unsigned short)((ur + 0x7FFFu + ((ur >> 16) & 1u)) >> 16);
                } else {
                    h[c] = st_enc<F16>(v);
                }
            }
            unsigned char* d = s_in + (size_t)buf * PL * IN_BYTES + pos * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            if (PL == 2)
                *reinterpret_cast<uint2*>(d + IN_BYTES) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
        }
    };

    int t = blockIdx.x;
    load_tile(t);
    store_tile(0);
    __syncthreads();                                     // weights, BN vectors and the first input tile are in LDS
    const int kg = lane >> 4, l16 = lane & 15;
    // wave -> (two of the four output-channel tiles, two of the eight pixel tiles): its weight fragments (2 x 7 rows, x 2
    // planes) stay in registers for the whole kernel; an activation fragment serves two MFMAs (x 3 with split planes).
    // (Weights read from LDS per pixel tile: 448 KB of LDS reads per tile with split planes — 1.6 us of the LDS pipe.)
    const int og = wave & 1, pg = wave >> 1;
    st_bf16x8 wh[2][7], wl[PL == 2 ? 2 : 1][7];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const unsigned char* wp = s_w + (((ky * 64 + (2 * og + o) * 16 + l16) * 32 + 8 * kg) * 2);
            wh[o][ky] = *reinterpret_cast<const st_bf16x8*>(wp);
            if (PL == 2) wl[o][ky] = *reinterpret_cast<const st_bf16x8*>(wp + W_BYTES);
        }
    __syncthreads();                                     // everyone has its weights: the buffer becomes the output tile
    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        load_tile(t + gridDim.x);                        // the next tile's pixels fly during the MFMAs
        const int b = t / tiles_per_img, r = t % tiles_per_img;
        const int ph0 = (r / g.tiles_x) * ST_PH, pw0 = (r % g.tiles_x) * ST_PW;
        const unsigned char* in_h = s_in + (size_t)buf * PL * IN_BYTES;
        // ---- implicit GEMM: this wave's 2 pixel tiles x 2 channel tiles
        {
            int pcl[2], crl[2], ccl[2];
            const unsigned char* xrow[2];
            st_f32x4 acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (2 * pg + j) * 16 + l16;
                pcl[j] = p;
                const int pc = p < ST_NPX ? p : ST_NPX - 1;
                crl[j] = pc / ST_CC; ccl[j] = pc % ST_CC;
                xrow[j] = in_h + ((2 * crl[j]) * ST_IC + 2 * ccl[j] + 2 * kg) * 8;
#pragma unroll
                for (int o = 0; o < 2; ++o) acc[j][o] = (st_f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // activation fragments of kernel row ky + 1 are requested before the MFMAs of row ky (two register sets)
            st_bf16x8 xh[2][2], xl[2][2];
            auto fetch = [&](int set, int ky) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xh[set][j] = *reinterpret_cast<const st_bf16x8*>(xrow[j] + ky * ST_IC * 8);
                    if (PL == 2) xl[set][j] = *reinterpret_cast<const st_bf16x8*>(xrow[j] + IN_BYTES + ky * ST_IC * 8);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int cur = ky & 1;
                if (ky + 1 < 7) fetch(cur ^ 1, ky + 1);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        acc[j][o] = H16<F16>::mfma16(wh[o][ky], xh[cur][j], acc[j][o]);
                        if (PL == 2) {
                            acc[j][o] = H16<F16>::mfma16(wl[o][ky], xh[cur][j], acc[j][o]);
                            acc[j][o] = H16<F16>::mfma16(wh[o][ky], xl[cur][j], acc[j][o]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);       // (keeps the requests of rows ky + 2 .. 6 out of this iteration)
            }
            // D[oc][pixel]: lane holds oc = 16 (2 og + o) + 4 (lane >> 4) + r of pixel l16.  BN, ReLU; conv outputs outside the
            // map are 0 (every pooling window holds a real output >= 0, or a NaN that wins anyway)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = pcl[j];
                const int gy = 2 * ph0 - 1 + crl[j], gx = 2 * pw0 - 1 + ccl[j];
                const bool inside = p < ST_NPX && (unsigned)gy < (unsigned)g.Hc && (unsigned)gx < (unsigned)g.Wc;
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int oc = (2 * og + o) * 16 + 4 * kg;
                    const float4 sc = *reinterpret_cast<const float4*>(&s_sc[oc]);
                    const float4 sh = *reinterpret_cast<const float4*>(&s_sh[oc]);
                    float4 v;
                    v.x = fmaf(acc[j][o][0], sc.x, sh.x); v.y = fmaf(acc[j][o][1], sc.y, sh.y);
                    v.z = fmaf(acc[j][o][2], sc.z, sh.z); v.w = fmaf(acc[j][o][3], sc.w, sh.w);
                    v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;       // (NaN < 0 is false: a NaN stays)
                    v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                    if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(&s_out[p * ST_OPITCH + oc]) = v;
                }
            }
        }
        // (bare barrier: behind __syncthreads() the compiler drains vmcnt — the next tile's pixels, requested above, would
        // have to arrive before the pooling of this tile could start)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the conv tile is complete
        // ---- 3 x 3 / stride 2 maximum: thread = (pooled pixel, 4 channels)
        if ((tid >> 4) < ST_PH * ST_PW) {
            const int q = tid >> 4, c4 = (tid & 15) * 4;
            const int py = q / ST_PW, pxl = q % ST_PW;
            const int ph = ph0 + py, pw = pw0 + pxl;
            float4 m = *reinterpret_cast<const float4*>(&s_out[((2 * py) * ST_CC + 2 * pxl) * ST_OPITCH + c4]);
#pragma unroll
            for (int k = 1; k < 9; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(&s_out[((2 * py + k / 3) * ST_CC + 2 * pxl + k % 3) * ST_OPITCH + c4]);
                m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x;
                m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
                m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z;
                m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
            }
            if (ph < g.Hp && pw < g.Wp) {
                const size_t pix = ((size_t)b * g.Hp + ph) * g.Wp + pw;
                if (PL == 2) {
                    unsigned short* d = out + pix * 128 + (c4 >> 5) * 64 + (c4 & 31);
                    const float vv[4] = {m.x, m.y, m.z, m.w};
                    unsigned short h[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned uu = __float_as_uint(vv[e]);
                        const unsigned hb = (uu + 0x7FFFu + ((uu >> 16) & 1u)) >> 16;
                        h[e] = (unsigned short)hb;
                        const unsigned ur = __float_as_uint(vv[e] - __uint_as_float(hb << 16));
                        l[e] = (unsigned short)((ur + 0x7FFFu + ((ur >> 16) & 1u)) >> 16);
                    }
                    *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                    *reinterpret_cast<uint2*>(d + 32) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                } else {
                    unsigned short* d = out + pix * 64 + c4;
                    *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)st_enc<F16>(m.x) | ((unsigned)st_enc<F16>(m.y) << 16),
                                                              (unsigned)st_enc<F16>(m.z) | ((unsigned)st_enc<F16>(m.w) << 16));
                }
            }
        }
        store_tile(buf ^ 1);                             // the next tile's pixels -> the other input buffer
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // ... visible; the conv tile may be overwritten
    }
}

}  // namespace hiast

// x: fp32 [B,3,H,W] contiguous (NCHW: the normalised image batch), w: fp32 [64,3,7,7] (nn.Conv2d layout), BN of 64 channels in
// eval mode (gamma / beta may be NULL: 1 / 0) -> out [B,Hp,Wp,planes*64] in operand format fmt,
// Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W)
extern "C" int hiast_stem_eval(const float* x, const float* w, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, void* out, int fmt, int B, int H, int W, hiast_stream_t stream)
{
    if (!x || !w || !mean || !var || !out) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (!hiast_fmt_ok(fmt)) return HIAST_E_RANGE;
    if ((size_t)B * 3 * H * W >= (1ull << 31)) return HIAST_E_RANGE;
    if ((((uintptr_t)out) & 7)) return HIAST_E_RANGE;
    hiast::StemGeo g;
    g.B = B; g.H = H; g.W = W;
    g.Hc = (H - 1) / 2 + 1; g.Wc = (W - 1) / 2 + 1;
    g.Hp = (g.Hc - 1) / 2 + 1; g.Wp = (g.Wc - 1) / 2 + 1;
    g.tiles_y = (g.Hp + hiast::ST_PH - 1) / hiast::ST_PH;
    g.tiles_x = (g.Wp + hiast::ST_PW - 1) / hiast::ST_PW;
    const long long ntiles = (long long)B * g.tiles_x * g.tiles_y;
    if (ntiles >= (1ll << 30)) return HIAST_E_RANGE;
    // persistent blocks: two per CU in the 16-bit formats (76 KiB of LDS each), one with split planes (116 KiB)
    const long long cap = (fmt == HIAST_FMT_SPLIT_BF16 ? 1 : 2) * hiast_grid_cus();
    const unsigned blocks = (unsigned)(ntiles < cap ? ntiles : cap);
    hipStream_t st = (hipStream_t)stream;
    if (fmt == HIAST_FMT_SPLIT_BF16)
        hipLaunchKernelGGL(hiast::stem_eval_kernel<HIAST_FMT_SPLIT_BF16>, dim3(blocks), dim3(512), 0, st, x, w, gamma, beta, mean,
                           var, eps, (unsigned short*)out, g);
    else if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::stem_eval_kernel<HIAST_FMT_FP16>, dim3(blocks), dim3(512), 0, st, x, w, gamma, beta, mean, var,
                           eps, (unsigned short*)out, g);
    else
        hipLaunchKernelGGL(hiast::stem_eval_kernel<HIAST_FMT_BF16>, dim3(blocks), dim3(512), 0, st, x, w, gamma, beta, mean, var,
                           eps, (unsigned short*)out, g);
    HIAST_CHECK_LAUNCH();
    return 0;
}
