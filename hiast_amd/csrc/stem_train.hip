// K9k — the stem convolution of the TRAINING forward and its weight gradient (round 4): conv 7x7 stride 2 pad 3 (3 -> 64) on
// the normalised fp32 image batch, 16-bit channels-last output + the per-block sums Σy, Σy² the batch-statistics BatchNorm
// behind it needs (reference: ResNet.forward, sseg/models/modules/resnet.py:177-180, `x = self.conv1(x)` in model.train()
// under apex O1 — a half-precision cuDNN convolution — and its autograd weight gradient; the image needs no data gradient).
// gfx950 only.
//
// Until round 3 this was the last library convolution of the forward: a channels-last copy of the fp32 batch (50 MB), a cast
// to 16 bits, MIOpen's igemm (0.13 ms) and a statistics pass over its 134 MB output; backward MIOpen's wrw kernel (0.13 ms).
// Both kernels here share the geometry of stem_eval (stem.hip): a block owns a tile of 8 x 16 convolution outputs, the
// 21 x 37 input pixels under it sit in LDS as [row][pixel][4 channels] 16-bit (3 + a zero) — read straight from the fp32
// NCHW batch, no channels-last copy, no cast kernel — and the convolution is an implicit GEMM on v_mfma_f32_16x16x32 with
// K = 7 kernel rows x 32 (8 input pixels x 4 channels; 7 x 3 real).
//   forward:  D[oc][pixel] = W[oc][ky][kx,ci] * X[pixel][ky][kx,ci]; a B fragment is ONE aligned ds_read_b128 of two
//             neighbouring input pixels, the weights stay in registers; the tile goes to LDS in 16 bits (where the statistics
//             are taken: of the STORED values, as the trunk's statistics epilogues do) and from there to HBM in whole
//             128-byte rows.
//   backward: dW[oc][ky][kx,ci] = Σ_pixel dY[pixel][oc] * X[pixel][ky][kx,ci] reduces over the PIXEL index; both operands
//             are pixel-major in LDS, so every fragment is two transposing ds_read_b64_tr_b16 reads (as in wgrad.hip).  The
//             "im2col row" of a pixel and a kernel row is 64 contiguous bytes of the input tile — the transposing read takes
//             per-lane addresses, so the im2col matrix is never built.  Blocks are persistent over tiles and keep their
//             64 x 7 x 32 accumulators in registers; per-block partials are added in a fixed order by a second launch.
#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 tt_bf16x8;
typedef __attribute__((ext_vector_type(4))) float tt_f32x4;
typedef short tt_s16x4 __attribute__((ext_vector_type(4)));

constexpr int TT_CR = 8, TT_CC = 16;                  // convolution outputs per tile: 8 rows x 16 columns
constexpr int TT_NPX = TT_CR * TT_CC;                 // 128 = 8 pixel tiles of 16
constexpr int TT_IR = 2 * TT_CR + 5;                  // 21 input rows
constexpr int TT_IC = 40;                             // 2 * 16 + 5 = 37 input columns (+ the 8th tap column), pitch 40
constexpr int TT_IN_BYTES = TT_IR * TT_IC * 8;        // 6720
constexpr int TT_OP = 68;                             // 16-bit elements per pixel of the output tile in LDS (64 + 4: banks)
constexpr int TT_DP = 144;                            // bytes per pixel of the dY tile in LDS (128 + 16)
constexpr int TT_W_BYTES = 7 * 64 * 32 * 2;           // packed weights [ky][oc][32]

struct StemTGeo {
    int B, H, W, Hc, Wc, tiles_x, tiles_y;
};

// 23 x 40 positions of the input tile, 2 per thread: fp32 NCHW -> registers (load) -> 16-bit [row][pixel][4] in LDS (store)
template <bool F16>
struct StemTile {
    float rin[2][3];
    __device__ __forceinline__ void load(const float* __restrict__ x, const StemTGeo& g, int t, int ntiles)
    {
        const int tiles_per_img = g.tiles_x * g.tiles_y;
        const int b = t / tiles_per_img, r = t % tiles_per_img;
        const int iy0 = 2 * (r / g.tiles_x) * TT_CR - 3, ix0 = 2 * (r % g.tiles_x) * TT_CC - 3;
        const float* xb = x + (size_t)(t < ntiles ? b : 0) * 3 * g.H * g.W;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = (int)threadIdx.x + 512 * u;
            const int iy = iy0 + pos / TT_IC, ix = ix0 + pos % TT_IC;
            const bool ok = t < ntiles && pos < TT_IR * TT_IC && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const size_t o = (size_t)(ok ? iy : 0) * g.W + (ok ? ix : 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) rin[u][c] = ok ? xb[(size_t)c * g.H * g.W + o] : 0.f;
        }
    }
    __device__ __forceinline__ void store(unsigned char* s_in) const
    {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = (int)threadIdx.x + 512 * u;
            if (pos >= TT_IR * TT_IC) continue;
            const unsigned h0 = H16<F16>::enc(rin[u][0]), h1 = H16<F16>::enc(rin[u][1]), h2 = H16<F16>::enc(rin[u][2]);
            *reinterpret_cast<uint2*>(s_in + pos * 8) = make_uint2(h0 | (h1 << 16), h2);
        }
    }
};

// weights fp32 [64][3][7][7] -> LDS [ky][oc][kq = kx * 4 + ci] 16-bit (kx = 7 and ci = 3: zeros)
template <bool F16>
__device__ __forceinline__ void stem_pack_weights(const float* __restrict__ w, unsigned char* s_w)
{
    float wv[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        const int e = (int)threadIdx.x + 512 * i;
        const int kq = e & 31, oc = (e >> 5) & 63, ky = e >> 11;
        const int kx = kq >> 2, ci = kq & 3;
        wv[i] = (kx < 7 && ci < 3) ? w[((oc * 3 + ci) * 7 + ky) * 7 + kx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 28; ++i) reinterpret_cast<unsigned short*>(s_w)[(int)threadIdx.x + 512 * i] = H16<F16>::enc(wv[i]);
}

template <bool F16>
__global__ __launch_bounds__(512) void stem_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             unsigned short* __restrict__ y, float* __restrict__ partial,
                                                             StemTGeo g)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_in[2 * TT_IN_BYTES];
    // one buffer, two lives: the packed weights during the prologue, the 16-bit output tile afterwards
    constexpr int U_BYTES = TT_W_BYTES > TT_NPX * TT_OP * 2 ? TT_W_BYTES : TT_NPX * TT_OP * 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_u[U_BYTES];
    __shared__ float s_red[8][32][2];
    unsigned short* const s_out = reinterpret_cast<unsigned short*>(s_u);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, l16 = lane & 15;
    const int tiles_per_img = g.tiles_x * g.tiles_y;
    const int ntiles = g.B * tiles_per_img;

    stem_pack_weights<F16>(w, s_u);
    StemTile<F16> tile;
    int t = blockIdx.x;
    tile.load(x, g, t, ntiles);
    tile.store(s_in);
    __syncthreads();
    // wave -> (two of the four output-channel tiles, two of the eight pixel tiles = two tile rows)
    const int og = wave & 1, pg = wave >> 1;
    tt_bf16x8 wh[2][7];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            wh[o][ky] = *reinterpret_cast<const tt_bf16x8*>(s_u + (((ky * 64 + (2 * og + o) * 16 + l16) * 32 + 8 * kg) * 2));
    __syncthreads();                                     // everyone has its weights: the buffer becomes the output tile
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[o][r] = 0.f; st2[o][r] = 0.f; }

    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        tile.load(x, g, t + gridDim.x, ntiles);          // the next tile's pixels fly during the MFMAs
        const int b = t / tiles_per_img, r = t % tiles_per_img;
        const int cr0 = (r / g.tiles_x) * TT_CR, cc0 = (r % g.tiles_x) * TT_CC;
        const unsigned char* in_h = s_in + buf * TT_IN_BYTES;
        tt_f32x4 acc[2][2];
        const unsigned char* xrow[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            xrow[j] = in_h + ((2 * (2 * pg + j)) * TT_IC + 2 * l16 + 2 * kg) * 8;      // pixel (row 2 pg + j, column l16)
#pragma unroll
            for (int o = 0; o < 2; ++o) acc[j][o] = (tt_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        tt_bf16x8 xh[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) xh[0][j] = *reinterpret_cast<const tt_bf16x8*>(xrow[j]);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int cur = ky & 1;
            if (ky + 1 < 7) {
#pragma unroll
                for (int j = 0; j < 2; ++j) xh[cur ^ 1][j] = *reinterpret_cast<const tt_bf16x8*>(xrow[j] + (ky + 1) * TT_IC * 8);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int o = 0; o < 2; ++o) acc[j][o] = H16<F16>::mfma16(wh[o][ky], xh[cur][j], acc[j][o]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // D[oc][pixel]: lane holds oc = 16 (2 og + o) + 4 kg + r of pixel (2 pg + j, l16): round, keep the statistics of the
        // rounded values of real pixels, 8-byte store into the output tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = (2 * pg + j) * 16 + l16;
            const bool inside = cr0 + 2 * pg + j < g.Hc && cc0 + l16 < g.Wc;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const unsigned p0 = H16<F16>::pack(acc[j][o][0], acc[j][o][1]), p1 = H16<F16>::pack(acc[j][o][2], acc[j][o][3]);
                *reinterpret_cast<uint2*>(s_out + p * TT_OP + (2 * og + o) * 16 + 4 * kg) = make_uint2(p0, p1);
                if (inside) {
                    const float v[4] = {H16<F16>::lo(p0), H16<F16>::hi(p0), H16<F16>::lo(p1), H16<F16>::hi(p1)};
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st1[o][q] += v[q]; st2[o][q] = fmaf(v[q], v[q], st2[o][q]); }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the output tile is complete
        {   // whole 128-byte rows to HBM: thread = (pixel, 32-byte quarter)
            const int p = tid >> 2, ch = tid & 3;
            const int gy = cr0 + (p >> 4), gx = cc0 + (p & 15);
            const uint2* s = reinterpret_cast<const uint2*>(s_out + p * TT_OP + ch * 16);
            const uint2 a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3];
            if (gy < g.Hc && gx < g.Wc) {
                unsigned short* d = y + (((size_t)b * g.Hc + gy) * g.Wc + gx) * 64 + ch * 16;
                h_store16(d, a0.x, a0.y, a1.x, a1.y);
                h_store16(d + 8, a2.x, a2.y, a3.x, a3.y);
            }
        }
        tile.store(s_in + (buf ^ 1) * TT_IN_BYTES);     // the next tile's pixels -> the other input buffer
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // ... visible; the output tile may be overwritten
    }
    // per-block sums: fold the 16 pixel-lanes, then the four waves that share an output-channel half (fixed order)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                st1[o][q] += __shfl_xor(st1[o][q], sft, 64);
                st2[o][q] += __shfl_xor(st2[o][q], sft, 64);
            }
        }
    if (l16 == 0) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s_red[wave][o * 16 + 4 * kg + q][0] = st1[o][q];
                s_red[wave][o * 16 + 4 * kg + q][1] = st2[o][q];
            }
    }
    __syncthreads();
    if (tid < 128) {
        const int oc = tid >> 1, which = tid & 1;
        const int ogc = oc >> 5, c32 = oc & 31;
        float s = 0.f;
#pragma unroll
        for (int pgi = 0; pgi < 4; ++pgi) s += s_red[ogc + 2 * pgi][c32][which];
        partial[((size_t)blockIdx.x * 64 + oc) * 2 + which] = s;
    }
}

__device__ __forceinline__ tt_s16x4 tt_tr_read(const unsigned char* p)
{
    typedef tt_s16x4 __attribute__((address_space(3))) * lds_p;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p);
}
__device__ __forceinline__ tt_bf16x8 tt_join(const tt_s16x4& v0, const tt_s16x4& v1)
{
    return __builtin_bit_cast(tt_bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// P[block][oc][ky][kq = kx * 4 + ci] fp32: this block's share of dW
template <bool F16>
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const float* __restrict__ x, const unsigned short* __restrict__ dy,
                                                         float* __restrict__ P, StemTGeo g)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_in[2 * TT_IN_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char s_dy[2 * TT_NPX * TT_DP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, l16 = lane & 15;
    const int q4 = l16 >> 2, p4 = l16 & 3;                 // this lane's (row, 4-column group) of a transposed-read block
    const int tiles_per_img = g.tiles_x * g.tiles_y;
    const int ntiles = g.B * tiles_per_img;
    const int oct = wave >> 1, ch = wave & 1;              // wave -> output-channel tile (16 of 64), kq half (16 of 32)

    StemTile<F16> tile;
    uint4 rdy[2];
    auto load_dy = [&](int t) {
        const int b = t / tiles_per_img, r = t % tiles_per_img;
        const int cr0 = (r / g.tiles_x) * TT_CR, cc0 = (r % g.tiles_x) * TT_CC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 512 * u;
            const int p = idx >> 3, c16 = idx & 7;
            const int gy = cr0 + (p >> 4), gx = cc0 + (p & 15);
            const bool ok = t < ntiles && gy < g.Hc && gx < g.Wc;
            rdy[u] = ok ? *reinterpret_cast<const uint4*>(dy + (((size_t)b * g.Hc + gy) * g.Wc + gx) * 64 + c16 * 8)
                        : make_uint4(0u, 0u, 0u, 0u);      // pixels outside the map add nothing
        }
    };
    auto store_dy = [&](unsigned char* s) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 512 * u;
            *reinterpret_cast<uint4*>(s + (idx >> 3) * TT_DP + (idx & 7) * 16) = rdy[u];
        }
    };
    int t = blockIdx.x;
    tile.load(x, g, t, ntiles);
    load_dy(t);
    tile.store(s_in);
    store_dy(s_dy);
    __syncthreads();

    tt_f32x4 acc[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) acc[ky] = (tt_f32x4){0.f, 0.f, 0.f, 0.f};

    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        tile.load(x, g, t + gridDim.x, ntiles);
        load_dy(t + gridDim.x);
        const unsigned char* in_h = s_in + buf * TT_IN_BYTES;
        const unsigned char* dy_h = s_dy + buf * (TT_NPX * TT_DP);
        // four k-steps of 32 pixels (two tile rows each); lane group kg owns pixels 8 kg .. 8 kg + 7 of the step:
        // tile row 2 s + (kg >> 1), columns 8 (kg & 1) + 4 h + q
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cr = 2 * s + (kg >> 1);
            tt_bf16x8 fa;
            {
                const unsigned char* a0 = dy_h + (cr * 16 + 8 * (kg & 1) + q4) * TT_DP + (16 * oct + 4 * p4) * 2;
                fa = tt_join(tt_tr_read(a0), tt_tr_read(a0 + 4 * TT_DP));
            }
            // the 16 columns kq = 16 ch .. + 15 of a pixel and a kernel row are 32 contiguous bytes of the input tile
            const unsigned char* b0 = in_h + ((2 * cr) * TT_IC + 2 * (8 * (kg & 1) + q4) + 4 * ch + p4) * 8;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const unsigned char* bp = b0 + ky * TT_IC * 8;
                const tt_bf16x8 fb = tt_join(tt_tr_read(bp), tt_tr_read(bp + 4 * 16));      // pixels + 4: 4 x 2 input columns on
                acc[ky] = H16<F16>::mfma16(fa, fb, acc[ky]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // everyone has read this tile
        tile.store(s_in + (buf ^ 1) * TT_IN_BYTES);
        store_dy(s_dy + (buf ^ 1) * (TT_NPX * TT_DP));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // D[oc][kq]: lane holds oc = 16 oct + 4 kg + r, kq = 16 ch + l16
    float* Pb = P + (size_t)blockIdx.x * 64 * 7 * 32;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pb[((16 * oct + 4 * kg + r) * 7 + ky) * 32 + 16 * ch + l16] = acc[ky][r];
}

// dW[oc][ci][ky][kx] = Σ_blocks P[block][oc][ky][kx * 4 + ci], ascending block order (bitwise reproducible)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int nblk)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // over [oc][ky][kq]
    if (i >= 64 * 7 * 32) return;
    const int kq = i & 31, ky = (i >> 5) % 7, oc = i / (7 * 32);
    const int kx = kq >> 2, ci = kq & 3;
    if (kx >= 7 || ci >= 3) return;
    float acc = 0.f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = P[(size_t)(b + u) * (64 * 7 * 32) + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblk; ++b) acc += P[(size_t)b * (64 * 7 * 32) + i];
    dw[((oc * 3 + ci) * 7 + ky) * 7 + kx] = acc;
}

static int stem_train_geo(int B, int H, int W, StemTGeo& g, long long& ntiles)
{
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if ((size_t)B * 3 * H * W >= (1ull << 31)) return HIAST_E_RANGE;
    g.B = B; g.H = H; g.W = W;
    g.Hc = (H - 1) / 2 + 1; g.Wc = (W - 1) / 2 + 1;
    g.tiles_y = (g.Hc + TT_CR - 1) / TT_CR;
    g.tiles_x = (g.Wc + TT_CC - 1) / TT_CC;
    ntiles = (long long)B * g.tiles_x * g.tiles_y;
    if (ntiles >= (1ll << 30) || (size_t)B * g.Hc * g.Wc * 64 >= (1ull << 31)) return HIAST_E_RANGE;
    return 0;
}

}  // namespace hiast

// number of blocks of hiast_stem_train_fwd = rows of its `partial` output [blocks][64][2] (0: unsupported geometry)
extern "C" int hiast_stem_train_blocks(int B, int H, int W)
{
    hiast::StemTGeo g;
    long long ntiles;
    if (hiast::stem_train_geo(B, H, W, g, ntiles)) return 0;
    const long long cap = 3ll * hiast_grid_cus();          // persistent: three blocks per CU (42 KiB of LDS each)
    return (int)(ntiles < cap ? ntiles : cap);
}

// x fp32 [B,3,H,W] (NCHW, contiguous), w fp32 [64,3,7,7] -> y [B,Hc,Wc,64] rows in format fmt (HIAST_FMT_BF16 | _FP16),
// Hc = (H-1)/2+1, Wc = (W-1)/2+1; partial fp32 [hiast_stem_train_blocks][64][2]: Σy, Σy² of the stored values per block
extern "C" int hiast_stem_train_fwd(const float* x, const float* w, void* y, float* partial, int fmt, int B, int H, int W,
                                    hiast_stream_t stream)
{
    if (!x || !w || !y || !partial) return HIAST_E_ARG;
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if ((((uintptr_t)y) & 15) || (((uintptr_t)x) & 3)) return HIAST_E_RANGE;
    hiast::StemTGeo g;
    long long ntiles;
    const int e = hiast::stem_train_geo(B, H, W, g, ntiles);
    if (e) return e;
    const unsigned blocks = (unsigned)hiast_stem_train_blocks(B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::stem_train_fwd_kernel<true>, dim3(blocks), dim3(512), 0, st, x, w, (unsigned short*)y, partial, g);
    else
        hipLaunchKernelGGL(hiast::stem_train_fwd_kernel<false>, dim3(blocks), dim3(512), 0, st, x, w, (unsigned short*)y, partial, g);
    HIAST_CHECK_LAUNCH();
    return 0;
}

static int stem_wgrad_blocks(long long ntiles)            // two per CU (50 KiB of LDS)
{
    const long long cap = 2ll * hiast_grid_cus();
    return (int)(ntiles < cap ? ntiles : cap);
}

extern "C" size_t hiast_stem_wgrad_workspace_bytes(int B, int H, int W)
{
    hiast::StemTGeo g;
    long long ntiles;
    if (hiast::stem_train_geo(B, H, W, g, ntiles)) return 0;
    return (size_t)stem_wgrad_blocks(ntiles) * 64 * 7 * 32 * sizeof(float);
}

// dW fp32 [64,3,7,7] = Σ dy[b,gy,gx,oc] * x[b,ci,2gy+ky-3,2gx+kx-3]; dy [B,Hc,Wc,64] rows in format fmt, x as in the forward
extern "C" int hiast_stem_wgrad(const float* x, const void* dy, float* dw, int fmt, int B, int H, int W, void* workspace,
                                size_t workspace_bytes, hiast_stream_t stream)
{
    if (!x || !dy || !dw || !workspace) return HIAST_E_ARG;
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if ((((uintptr_t)dy) & 15) || (((uintptr_t)workspace) & 15)) return HIAST_E_RANGE;
    hiast::StemTGeo g;
    long long ntiles;
    const int e = hiast::stem_train_geo(B, H, W, g, ntiles);
    if (e) return e;
    const int blocks = stem_wgrad_blocks(ntiles);
    if (workspace_bytes < (size_t)blocks * 64 * 7 * 32 * sizeof(float)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::stem_wgrad_kernel<true>, dim3(blocks), dim3(512), 0, st, x, (const unsigned short*)dy,
                           (float*)workspace, g);
    else
        hipLaunchKernelGGL(hiast::stem_wgrad_kernel<false>, dim3(blocks), dim3(512), 0, st, x, (const unsigned short*)dy,
                           (float*)workspace, g);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::stem_wgrad_reduce_kernel, dim3((64 * 7 * 32 + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, dw, blocks);
    HIAST_CHECK_LAUNCH();
    return 0;
}
