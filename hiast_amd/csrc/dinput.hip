// K15 — discriminator input map of the adversarial warm-up stage
// (reference: AdversarialWarmupSegmentor, sseg/models/segmentors/adversarial_warmup_segmentor.py:
//  F.interpolate(logits, bilinear, align_corners=True) :36,:41 -> D_preprocess_fun :26-29
//  = softmax(dim=1) [AdaptSegNet]  or  prob_2_entropy(softmax) :71-76 [AdvEnt]).
//
// fwd: one thread per full-res pixel; the C low-res logits are interpolated in registers, the
//      softmax / weighted self-information map is written once (write-bound: C*4 B per pixel,
//      39.8 MB/img at 512x1024, C=19).  The full-resolution LOGITS are never stored.
// bwd: given g = dL/d(map) the same thread recomputes p and writes d(full-res logits)
//          softmax:  dz_c = p_c (g_c - Σ_k p_k g_k)
//          entropy:  same with g_c <- g_c * d e_c/d p_c,
//                    e_c = -p_c log2(p_c + 1e-30)/log2(C),
//                    d e_c/d p_c = -(log2(p_c + 1e-30) + p_c / ((p_c + 1e-30) ln 2)) / log2(C)
//      into a caller-provided scratch map, which the exact-adjoint gather of K2
//      (upsample_bwd_kernel, fixed summation order) folds into the low-res gradient.
#include "common.h"

namespace hiast {

template <int C>
__device__ __forceinline__ void dinput_probs(const float* __restrict__ base, int h, int w, const Src& sy,
                                             const Src& sx, float (&p)[C])
{
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* q = base + (size_t)c * h * w;
        const float top = lerp_h(q[sy.i0 * w + sx.i0], q[sy.i0 * w + sx.i1], sx.l0, sx.l1);
        const float bot = lerp_h(q[sy.i1 * w + sx.i0], q[sy.i1 * w + sx.i1], sx.l0, sx.l1);
        p[c] = lerp_v(top, bot, sy.l0, sy.l1);
        m = (c == 0 || p[c] > m) ? p[c] : m;
    }
    float S = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        p[c] = expf(p[c] - m);
        S += p[c];
    }
    const float inv = 1.0f / S;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
}

template <int C, int MODE>
__global__ __launch_bounds__(256) void dinput_fwd_kernel(const float* __restrict__ z_lr, float* __restrict__ out,
                                                         int h, int w, int H, int W, float sh, float sw)
{
    const int b = blockIdx.z, Y = blockIdx.y;
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= W) return;
    const Src sy = src_of(sh, Y, h), sx = src_of(sw, X, w);
    float p[C];
    dinput_probs<C>(z_lr + (size_t)b * C * h * w, h, w, sy, sx, p);
    const float invL = 1.0f / log2f((float)C);
    float* o = out + ((size_t)b * C * H + Y) * W + X;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float v = p[c];
        if (MODE == 1) v = -(p[c] * log2f(p[c] + 1e-30f)) * invL;
        o[(size_t)c * H * W] = v;
    }
}

template <int C, int MODE>
__global__ __launch_bounds__(256) void dinput_bwd_kernel(const float* __restrict__ z_lr, const float* __restrict__ g,
                                                         float* __restrict__ dz, int h, int w, int H, int W,
                                                         float sh, float sw)
{
    const int b = blockIdx.z, Y = blockIdx.y;
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= W) return;
    const Src sy = src_of(sh, Y, h), sx = src_of(sw, X, w);
    float p[C];
    dinput_probs<C>(z_lr + (size_t)b * C * h * w, h, w, sy, sx, p);
    const size_t off = ((size_t)b * C * H + Y) * W + X;
    const float invL = 1.0f / log2f((float)C);
    const float INV_LN2 = 1.44269504088896340736f;
    float t[C];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float gc = g[off + (size_t)c * H * W];
        if (MODE == 1) {
            const float q = p[c] + 1e-30f;
            gc *= -(log2f(q) + (p[c] / q) * INV_LN2) * invL;
        }
        t[c] = gc;
        dot = fmaf(p[c], gc, dot);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dz[off + (size_t)c * H * W] = p[c] * (t[c] - dot);
}

}  // namespace hiast

static int dinput_check(const void* a, const void* b, int mode, int B, int C, int h, int w, int H, int W)
{
    if (!a || !b) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 1) return HIAST_E_ARG;
    if ((long long)B * C > 65535 || H > 65535 || h > 65535) return HIAST_E_RANGE;
    return 0;
}

#define HIAST_DIN_DISPATCH(KERNEL_CALL)                                           \
    switch (C) {                                                                  \
        case 19: { constexpr int CC = 19; KERNEL_CALL; } break;                   \
        case 16: { constexpr int CC = 16; KERNEL_CALL; } break;                   \
        case 9:  { constexpr int CC = 9;  KERNEL_CALL; } break;                   \
        case 2:  { constexpr int CC = 2;  KERNEL_CALL; } break;                   \
        default: return HIAST_E_RANGE;                                            \
    }

extern "C" int hiast_dinput_fwd(const float* logits_lr, int mode, float* out, int B, int C, int h, int w, int H,
                                int W, hiast_stream_t stream)
{
    int e = dinput_check(logits_lr, out, mode, B, C, h, w, H, W);
    if (e) return e;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    dim3 grid((W + 255) / 256, H, B);
#define FWD(M) hipLaunchKernelGGL((hiast::dinput_fwd_kernel<CC, M>), grid, dim3(256), 0, (hipStream_t)stream, \
                                  logits_lr, out, h, w, H, W, sh, sw)
    if (mode == 0) { HIAST_DIN_DISPATCH(FWD(0)) } else { HIAST_DIN_DISPATCH(FWD(1)) }
#undef FWD
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_dinput_bwd(const float* logits_lr, int mode, const float* gout, float* scratch,
                                float* dlogits_lr, int B, int C, int h, int w, int H, int W, hiast_stream_t stream)
{
    int e = dinput_check(logits_lr, gout, mode, B, C, h, w, H, W);
    if (e) return e;
    if (!scratch || !dlogits_lr) return HIAST_E_ARG;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    dim3 grid((W + 255) / 256, H, B);
#define BWD(M) hipLaunchKernelGGL((hiast::dinput_bwd_kernel<CC, M>), grid, dim3(256), 0, (hipStream_t)stream, \
                                  logits_lr, gout, scratch, h, w, H, W, sh, sw)
    if (mode == 0) { HIAST_DIN_DISPATCH(BWD(0)) } else { HIAST_DIN_DISPATCH(BWD(1)) }
#undef BWD
    HIAST_CHECK_LAUNCH();
    return hiast_upsample_bilinear_ac_bwd(scratch, dlogits_lr, B, C, h, w, H, W, stream);
}
