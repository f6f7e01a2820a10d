// Shared device helpers for libhiast_hip.so (gfx950 / CDNA4 only, wave64).
//
// "HIAST-A arithmetic": the fp32 sequences below (explicit fmaf / mul / add in a fixed order,
// the library is built with -ffp-contract=off) define the bilinear upsample and the softmax
// max-probability bit-for-bit, so integer outputs (argmax / pseudo-label maps / histograms)
// are reproducible against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/hiast_hip.h"

#define HIAST_WAVE 64

#define HIAST_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

namespace hiast {

// exp(x) for x <= 0: Cody-Waite reduction + degree-7 Taylor (Horner, fmaf); x < -87 -> 0.
__device__ __forceinline__ float a_expf(float x)
{
    const float LOG2E = 1.44269502162933349609375f;
    const float LN2_HI = 0.693145751953125f;
    const float LN2_LO = 1.428606765330187045037746429443359375e-06f;
    float n = rintf(x * LOG2E);
    float r = fmaf(-n, LN2_HI, x);
    r = fmaf(-n, LN2_LO, r);
    float p = 1.984127011382952332496643066406250e-04f;
    p = fmaf(p, r, 1.388888922519981861114501953125e-03f);
    p = fmaf(p, r, 8.33333376795053482055664062500e-03f);
    p = fmaf(p, r, 4.16666679084300994873046875000e-02f);
    p = fmaf(p, r, 1.66666671633720397949218750000e-01f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    float s = __int_as_float(((int)n + 127) << 23);
    float v = p * s;
    return x < -87.0f ? 0.0f : v;
}

// align_corners=True source coordinate of destination index `dst`
struct Src {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Src src_of(float scale, int dst, int in)
{
    Src s;
    float f = scale * (float)dst;
    int a = (int)f;
    a = a > in - 1 ? in - 1 : a;
    s.i0 = a;
    s.i1 = a + (a < in - 1 ? 1 : 0);
    s.l1 = f - (float)a;
    s.l0 = 1.0f - s.l1;
    return s;
}

__device__ __forceinline__ float lerp_h(float a, float b, float wl0, float wl1)
{
    return fmaf(wl1, b, wl0 * a);   // horizontal pair first (ATen order)
}

__device__ __forceinline__ float lerp_v(float top, float bot, float hl0, float hl1)
{
    return fmaf(hl1, bot, hl0 * top);
}

// First destination row whose source row index floor(scale*Y) is >= j (rows are grouped into
// "bands" that share the same pair of source rows).  Wave-uniform scalar code.
__device__ __forceinline__ int band_start(float scale, int j, int in, int out)
{
    if (j <= 0) return 0;
    if (j > in - 1) return out;
    if (scale <= 0.0f) return out;   // out == 1: everything is band 0
    int Y = (int)((float)j / scale);
    Y = Y < 0 ? 0 : (Y > out ? out : Y);
    while (Y > 0 && (int)(scale * (float)(Y - 1)) >= j) --Y;
    while (Y < out && (int)(scale * (float)Y) < j) ++Y;
    return Y;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// 64-bit wave reductions (all 64 lanes participate)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // fixed order: deterministic
    return v;
}

}  // namespace hiast
