// Shared device helpers for libhiast_hip.so (gfx950 / CDNA4 only, wave64).
//
// "HIAST-A arithmetic": the fp32 sequences below (explicit fmaf / mul / add in a fixed order,
// the library is built with -ffp-contract=off) define the bilinear upsample and the softmax
// max-probability bit-for-bit, so integer outputs (argmax / pseudo-label maps / histograms)
// are reproducible against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/hiast_hip.h"

#define HIAST_WAVE 64

#define HIAST_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

// operand format argument of the C ABI -> (planes per value, fp16?)
static inline int hiast_fmt_planes(int fmt) { return fmt == HIAST_FMT_SPLIT_BF16 ? 2 : 1; }
static inline int hiast_fmt_ok(int fmt) { return fmt == HIAST_FMT_BF16 || fmt == HIAST_FMT_SPLIT_BF16 || fmt == HIAST_FMT_FP16; }

// ---- device geometry (host; misc.hip).  hiast_cu_count(): compute units of the CURRENT device (hipDeviceGetAttribute, cached per
// device; 256 on MI355X).  hiast_grid_cus(): what the one-block-per-CU and persistent launches size their grids (and their
// one-round work splits) to = CU count - hiast_set_reserve_cus(): with n CUs reserved a collective's kernel (RCCL) always finds a
// CU that no block of ours holds for 60-160 us.  Both never return less than 8.
int hiast_cu_count();
int hiast_grid_cus();

namespace hiast {

// ---- the two 16-bit storage types of the mixed-precision path (operand format HIAST_FMT_BF16 / HIAST_FMT_FP16) ----------
// The reference trains under apex O1 = IEEE fp16 (code/utils/default_config.py:109, utils/utils.py:126-132); bf16 is
// the type the round-1/2 kernels were written for.  Both feed the same matrix cores at the same rate
// (v_mfma_f32_16x16x32_{bf16,f16}, v_mfma_f32_32x32x16_{bf16,f16}: fp32 accumulate) and the slab format / LDS images /
// DMA patterns are type-agnostic, so every kernel of that path takes the type as a template parameter and touches
// it only where a value is decoded, encoded (round to nearest even) or multiplied.
typedef __attribute__((ext_vector_type(8))) __bf16 h_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 h_f16x8;
typedef __attribute__((ext_vector_type(4))) float h_f32x4;
typedef __attribute__((ext_vector_type(16))) float h_f32x16;

// 16-byte store of an output row segment.  HIAST_NT (default 1; -DHIAST_NT=0 for an A/B build of a translation unit): as a
// NON-TEMPORAL (streaming) store.  Every activation tensor of the trunk is larger than the 32 MiB of L2 and is read next by
// another kernel (from HBM / the Infinity Cache either way), so keeping its lines dirty in the L2 only (a) evicts the weight /
// residual lines the kernel re-reads and (b) leaves up to 32 MiB to the end-of-kernel write-back, during which nothing runs
// (round 4: step 55.5 -> 54.6 ms with the igemm epilogue alone).
#ifndef HIAST_NT
#define HIAST_NT 1
#endif
__device__ __forceinline__ void h_store_f32(float* p, float v)
{
#if HIAST_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
typedef unsigned int h_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h_store16(void* p, unsigned a, unsigned b, unsigned c, unsigned d)
{
#if HIAST_NT
    __builtin_nontemporal_store((h_u32x4){a, b, c, d}, reinterpret_cast<h_u32x4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#endif
}
// the same with a run-time choice (wave-uniform): tensors that fit in the L2 + Infinity Cache are better left to the caches —
// measured per kernel in the step: 1024-channel BatchNorm apply 78 -> 70 us with streaming stores, the 256-channel one
// (33 MB, read by the next launch) 13.3 -> 15.1 us, the persistent xconv / xconv2 kernels (stores spread over the whole
// launch) 4-10 % slower
__device__ __forceinline__ void h_store16(void* p, unsigned a, unsigned b, unsigned c, unsigned d, bool nt)
{
#if HIAST_NT
    if (nt) {
        __builtin_nontemporal_store((h_u32x4){a, b, c, d}, reinterpret_cast<h_u32x4*>(p));
        return;
    }
#endif
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
}
#ifndef H_NT_MIN_MIB
#define H_NT_MIN_MIB 64
#endif
constexpr long long H_NT_MIN_BYTES = (long long)H_NT_MIN_MIB << 20;      // outputs from 64 MiB on are streamed
// output store of the persistent 1x1 kernels / the weight-gradient partials: plain by default (stand-alone the streaming form
// measured 3-10 % slower there); HIAST_NT_OUT=1 is the in-step A/B build
#ifndef HIAST_NT_OUT
#define HIAST_NT_OUT 0
#endif
__device__ __forceinline__ void h_store16_out(void* p, unsigned a, unsigned b, unsigned c, unsigned d)
{
    h_store16(p, a, b, c, d, HIAST_NT_OUT != 0);
}
__device__ __forceinline__ void h_store_f32_out(float* p, float v)
{
#if HIAST_NT_OUT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// 16-byte load of a row segment that this launch reads exactly once (residual / BatchNorm-input rows of the persistent 1x1 kernels):
// HIAST_NT_RES=1 (A/B build of a translation unit) makes it a non-temporal load
#ifndef HIAST_NT_RES
#define HIAST_NT_RES 0
#endif
__device__ __forceinline__ uint4 h_load16_once(const void* p)
{
#if HIAST_NT_RES
    const h_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const h_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

template <bool F16>
struct H16;

template <>
struct H16<false> {                                   // bf16: the upper half of an fp32
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }           // low 16 bits of w
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }   // high 16 bits
    static __device__ __forceinline__ float dec(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
    static __device__ __forceinline__ unsigned short enc(float v) { return __bfloat16_as_ushort(__float2bfloat16(v)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) { return (unsigned)enc(a) | ((unsigned)enc(b) << 16); }
    template <class V>
    static __device__ __forceinline__ h_f32x4 mfma16(const V& a, const V& b, h_f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(h_bf16x8, a), __builtin_bit_cast(h_bf16x8, b), c, 0, 0, 0);
    }
    template <class V>
    static __device__ __forceinline__ h_f32x16 mfma32(const V& a, const V& b, h_f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h_bf16x8, a), __builtin_bit_cast(h_bf16x8, b), c, 0, 0, 0);
    }
};

template <>
struct H16<true> {                                    // IEEE binary16 (v_cvt_f32_f16 / v_cvt_f16_f32, round to nearest even)
    static __device__ __forceinline__ float dec(unsigned short h) { return __half2float(__ushort_as_half(h)); }
    static __device__ __forceinline__ float lo(unsigned w) { return dec((unsigned short)(w & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return dec((unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned short enc(float v) { return __half_as_ushort(__float2half_rn(v)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) { return (unsigned)enc(a) | ((unsigned)enc(b) << 16); }
    template <class V>
    static __device__ __forceinline__ h_f32x4 mfma16(const V& a, const V& b, h_f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h_f16x8, a), __builtin_bit_cast(h_f16x8, b), c, 0, 0, 0);
    }
    template <class V>
    static __device__ __forceinline__ h_f32x16 mfma32(const V& a, const V& b, h_f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h_f16x8, a), __builtin_bit_cast(h_f16x8, b), c, 0, 0, 0);
    }
};

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
