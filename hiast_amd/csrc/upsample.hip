// K2 — standalone bilinear upsample, align_corners=True (F.interpolate call sites listed in
// include/hiast_hip.h).  Only the API path that must hand full-resolution logits to a caller uses
// it; the generator and the trainers consume the low-res logits through the fused kernels.
// fwd: write-bound (64x more bytes out than in): 4 output pixels per thread, float4 stores.
// bwd: exact adjoint as a GATHER (one thread per low-res cell, fixed summation order ->
//      bitwise reproducible, no float atomics).
#include "common.h"

namespace hiast {

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int h, int w,
                                                           int H, int W, float sh, float sw)
{
    const int bc = blockIdx.z;
    const int Y = blockIdx.y;
    const int X4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (X4 >= W) return;
    const Src sy = src_of(sh, Y, h);
    const float* p0 = in + ((size_t)bc * h + sy.i0) * w;
    const float* p1 = in + ((size_t)bc * h + sy.i1) * w;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = X4 + k < W ? X4 + k : W - 1;
        const Src sx = src_of(sw, X, w);
        const float top = lerp_h(p0[sx.i0], p0[sx.i1], sx.l0, sx.l1);
        const float bot = lerp_h(p1[sx.i0], p1[sx.i1], sx.l0, sx.l1);
        v[k] = lerp_v(top, bot, sy.l0, sy.l1);
    }
    float* o = out + ((size_t)bc * H + Y) * W + X4;
    if (X4 + 3 < W && ((((size_t)bc * H + Y) * W + X4) & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int k = 0; k < 4 && X4 + k < W; ++k) o[k] = v[k];
    }
}

// gin[bc][j][i] = Σ_Y Σ_X gout[bc][Y][X] * wy(Y, j) * wx(X, i)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gout,
                                                           float* __restrict__ gin, int h, int w,
                                                           int H, int W, float sh, float sw)
{
    const int bc = blockIdx.z;
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w) return;
    // rows with floor(src) in {j-1, j} can touch source row j; same for columns
    const int Ya = band_start(sh, j - 1, h, H), Yb = band_start(sh, j + 1, h, H);
    const int Xa = band_start(sw, i - 1, w, W), Xb = band_start(sw, i + 1, w, W);
    const float* g = gout + (size_t)bc * H * W;
    float acc = 0.0f;
    for (int Y = Ya; Y < Yb; ++Y) {
        const Src sy = src_of(sh, Y, h);
        const float wy = (sy.i0 == j ? sy.l0 : 0.0f) + (sy.i1 == j ? sy.l1 : 0.0f);
        if (wy == 0.0f) continue;
        float row = 0.0f;
        for (int X = Xa; X < Xb; ++X) {
            const Src sx = src_of(sw, X, w);
            const float wx = (sx.i0 == i ? sx.l0 : 0.0f) + (sx.i1 == i ? sx.l1 : 0.0f);
            row = fmaf(g[(size_t)Y * W + X], wx, row);
        }
        acc = fmaf(row, wy, acc);
    }
    gin[((size_t)bc * h + j) * w + i] = acc;
}

}  // namespace hiast

static int check_up(const void* a, const void* b, int B, int C, int h, int w, int H, int W)
{
    if (!a || !b) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if ((long long)B * C > 65535 || H > 65535 || h > 65535) return HIAST_E_RANGE;
    return 0;
}

extern "C" int hiast_upsample_bilinear_ac_fwd(const float* in, float* out, int B, int C, int h, int w,
                                              int H, int W, hiast_stream_t stream)
{
    int e = check_up(in, out, B, C, h, w, H, W);
    if (e) return e;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    dim3 grid(((W + 3) / 4 + 255) / 256, H, B * C);
    hipLaunchKernelGGL(hiast::upsample_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, h,
                       w, H, W, sh, sw);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_upsample_bilinear_ac_bwd(const float* gout, float* gin, int B, int C, int h, int w,
                                              int H, int W, hiast_stream_t stream)
{
    int e = check_up(gout, gin, B, C, h, w, H, W);
    if (e) return e;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    dim3 grid((w + 255) / 256, h, B * C);
    hipLaunchKernelGGL(hiast::upsample_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, gin,
                       h, w, H, W, sh, sw);
    HIAST_CHECK_LAUNCH();
    return 0;
}
