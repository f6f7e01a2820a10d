// K10 — BatchNorm2d (+ residual add) (+ ReLU) as fused streaming kernels for the ResNet bottlenecks
// (reference: Bottleneck.forward, sseg/models/modules/resnet.py:78-98 — conv → BN → ReLU ... conv → BN →
// (+identity) → ReLU, executed there as separate cuDNN/ATen passes; "frozen" BN still normalises with
// BATCH statistics in train() mode, utils/utils.py:60-65).  gfx950, HBM-bound.
//
// Layout: x, res, y, dy, dx: [B][C][HW] contiguous, fp32 or bf16; parameters/statistics fp32 [C];
// per-plane partial sums in double (no float atomics, fixed summation order => bitwise reproducible).
//
//   stats      : per (n,c) plane Σx, Σx²                          -> part[C][B][2]   (1 read)
//   apply      : y = relu(x*scale_c + shift_c (+ res))            (1 read (+1), 1 write)
//                scale/shift derive in-kernel from (running stats | the plane partials); the n==0 block of
//                a channel also writes save_mean / save_invstd and updates the running statistics.
//   bwd_stats  : g = dy * (y > 0);  Σg, Σ g*xhat                  -> part[C][B][2]
//   bwd_apply  : dx = gamma*invstd*(g - Σg/n - xhat*Σ(g*xhat)/n);  dres = g
// Between stats and apply the caller may all-reduce `part` sums across ranks (SyncBN).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace hiast {

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    using raw = float4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4])
    {
        const float4 r = *reinterpret_cast<const float4*>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4])
    {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Vec<__hip_bfloat16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ float up(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
    static __device__ __forceinline__ unsigned short down(float f)
    {
        return __bfloat16_as_ushort(__float2bfloat16(f));      // RNE, NaN-preserving cast
    }
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8])
    {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8])
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)down(v[2 * i]) | ((unsigned)down(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ float ld1(const __hip_bfloat16* p)
    {
        return up(*reinterpret_cast<const unsigned short*>(p));
    }
    static __device__ __forceinline__ void st1(__hip_bfloat16* p, float v)
    {
        *reinterpret_cast<unsigned short*>(p) = down(v);
    }
};

// fp16 activations (dtype 2): the reference's apex O1 arithmetic (half-precision convolutions; BatchNorm itself in fp32)
template <> struct Vec<__half> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const __half* p, float (&v)[8])
    {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __half2float(__ushort_as_half((unsigned short)(w[i] & 0xFFFFu)));
            v[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(w[i] >> 16)));
        }
    }
    static __device__ __forceinline__ void store(__half* p, const float (&v)[8])
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)__half_as_ushort(__float2half_rn(v[2 * i])) |
                   ((unsigned)__half_as_ushort(__float2half_rn(v[2 * i + 1])) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ float ld1(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ void st1(__half* p, float v) { *p = __float2half_rn(v); }
};

__device__ __forceinline__ double block_sum(double v, double* s_red)
{
    v = wave_sum_f64(v);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += s_red[i];
    return r;
}

// ---------------------------------------------------------------------------------------- fwd stats
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long long HW, int C,
                                                       double* __restrict__ part)
{
    __shared__ double s_red[4];
    const int c = blockIdx.x, n = blockIdx.y, B = gridDim.y;
    const T* p = x + ((size_t)n * C + c) * HW;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        constexpr int N = Vec<T>::N;
        const long long nv = HW / N;
        for (long long i = threadIdx.x; i < nv; i += 256) {
            float v[N];
            Vec<T>::load(p + i * N, v);
#pragma unroll
            for (int k = 0; k < N; ++k) { s1 += v[k]; s2 = fmaf(v[k], v[k], s2); }
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) {
            const float v = Vec<T>::ld1(p + i);
            s1 += v; s2 = fmaf(v, v, s2);
        }
    }
    const double a = block_sum((double)s1, s_red);
    const double b = block_sum((double)s2, s_red);
    if (threadIdx.x == 0) {
        part[((size_t)c * B + n) * 2 + 0] = a;
        part[((size_t)c * B + n) * 2 + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------- fwd apply
// MODE 0: inference (running statistics); MODE 1: training (plane partials, count = nglobal elements)
template <typename T, bool VEC, bool RES, bool RELU, int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
    const double* __restrict__ part, int npart, double count, float momentum, float eps,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, long long HW, int C)
{
    const int c = blockIdx.y, n = blockIdx.z;
    float mean, invstd;
    if (MODE == 0) {
        mean = run_mean[c];
        invstd = 1.0f / sqrtf(run_var[c] + eps);
    } else {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < npart; ++i) {
            s1 += part[((size_t)c * npart + i) * 2 + 0];
            s2 += part[((size_t)c * npart + i) * 2 + 1];
        }
        const double m = s1 / count;
        double var = s2 / count - m * m;
        var = var < 0.0 ? 0.0 : var;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (run_mean) {           // torch: running = (1-m)*running + m*batch; unbiased variance
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
                run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
            }
        }
    }
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const float scale = g * invstd, shift = fmaf(-mean, scale, b);
    const size_t base = ((size_t)n * C + c) * HW;
    const T* px = x + base;
    const T* pr = RES ? res + base : nullptr;
    T* py = y + base;
    if (VEC) {
        constexpr int N = Vec<T>::N;
        const long long nv = HW / N;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
            float v[N], r[N];
            Vec<T>::load(px + i * N, v);
            if (RES) Vec<T>::load(pr + i * N, r);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                float o = fmaf(v[k], scale, shift);
                if (RES) o += r[k];
                if (RELU) o = o > 0.f ? o : 0.f;
                v[k] = o;
            }
            Vec<T>::store(py + i * N, v);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
            float o = fmaf(Vec<T>::ld1(px + i), scale, shift);
            if (RES) o += Vec<T>::ld1(pr + i);
            if (RELU) o = o > 0.f ? o : 0.f;
            Vec<T>::st1(py + i, o);
        }
    }
}

// ---------------------------------------------------------------------------------------- bwd stats
template <typename T, bool VEC, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ x,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, long long HW,
                                                           int C, double* __restrict__ part)
{
    __shared__ double s_red[4];
    const int c = blockIdx.x, n = blockIdx.y, B = gridDim.y;
    const size_t base = ((size_t)n * C + c) * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        constexpr int N = Vec<T>::N;
        const long long nv = HW / N;
        for (long long i = threadIdx.x; i < nv; i += 256) {
            float g[N], yy[N], xx[N];
            Vec<T>::load(dy + base + i * N, g);
            if (RELU) Vec<T>::load(y + base + i * N, yy);
            Vec<T>::load(x + base + i * N, xx);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float gg = (!RELU || yy[k] > 0.f) ? g[k] : 0.f;
                s1 += gg;
                s2 = fmaf(gg, (xx[k] - mean) * invstd, s2);
            }
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) {
            float gg = Vec<T>::ld1(dy + base + i);
            if (RELU && !(Vec<T>::ld1(y + base + i) > 0.f)) gg = 0.f;
            s1 += gg;
            s2 = fmaf(gg, (Vec<T>::ld1(x + base + i) - mean) * invstd, s2);
        }
    }
    const double a = block_sum((double)s1, s_red);
    const double b = block_sum((double)s2, s_red);
    if (threadIdx.x == 0) {
        part[((size_t)c * B + n) * 2 + 0] = a;
        part[((size_t)c * B + n) * 2 + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------- bwd apply
template <typename T, bool VEC, bool RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, const double* __restrict__ part,
    int npart, double count, T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ dgamma,
    float* __restrict__ dbeta, long long HW, int C)
{
    const int c = blockIdx.y, n = blockIdx.z;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < npart; ++i) {
        s1 += part[((size_t)c * npart + i) * 2 + 0];
        s2 += part[((size_t)c * npart + i) * 2 + 1];
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
    }
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float k0 = (gamma ? gamma[c] : 1.0f) * invstd;
    const float mg = (float)(s1 / count), mgx = (float)(s2 / count);
    const size_t base = ((size_t)n * C + c) * HW;
    if (VEC) {
        constexpr int N = Vec<T>::N;
        const long long nv = HW / N;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
            float g[N], yy[N], xx[N];
            Vec<T>::load(dy + base + i * N, g);
            if (RELU) Vec<T>::load(y + base + i * N, yy);
            Vec<T>::load(x + base + i * N, xx);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float gg = (!RELU || yy[k] > 0.f) ? g[k] : 0.f;
                g[k] = gg;
                xx[k] = k0 * (gg - mg - (xx[k] - mean) * invstd * mgx);
            }
            Vec<T>::store(dx + base + i * N, xx);
            if (DRES) Vec<T>::store(dres + base + i * N, g);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
            float gg = Vec<T>::ld1(dy + base + i);
            if (RELU && !(Vec<T>::ld1(y + base + i) > 0.f)) gg = 0.f;
            const float xh = (Vec<T>::ld1(x + base + i) - mean) * invstd;
            Vec<T>::st1(dx + base + i, k0 * (gg - mg - xh * mgx));
            if (DRES) Vec<T>::st1(dres + base + i, gg);
        }
    }
}

static bool vec_ok(const void* a, const void* b, const void* c, const void* d, long long HW, int esz)
{
    const int n = esz == 2 ? 8 : 4;
    auto al = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
    return HW % n == 0 && al(a) && al(b) && al(c) && al(d);
}

static dim3 apply_grid(long long HW, int C, int B, int vecn)
{
    long long chunks = (HW / vecn + 256 * 4 - 1) / (256 * 4);
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    return dim3((unsigned)chunks, C, B);
}

}  // namespace hiast

static int bn_check(const void* x, int B, int C, long long HW, int dtype)
{
    if (!x) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || HW <= 0) return HIAST_E_ARG;
    if (B > 65535 || C > 65535 || (dtype < 0 || dtype > 2)) return HIAST_E_RANGE;
    return 0;
}

extern "C" size_t hiast_bn_workspace_bytes(int B, int C) { return (size_t)B * C * 2 * sizeof(double); }

extern "C" int hiast_bn_stats(const void* x, int B, int C, int64_t HW, int dtype, double* part,
                              hiast_stream_t stream)
{
    int e = bn_check(x, B, C, HW, dtype);
    if (e) return e;
    if (!part) return HIAST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(C, B);
    const bool v = hiast::vec_ok(x, nullptr, nullptr, nullptr, HW, dtype ? 2 : 4);
#define L(T, V) hipLaunchKernelGGL((hiast::bn_stats_kernel<T, V>), grid, dim3(256), 0, st, (const T*)x, (long long)HW, C, part)
    if (dtype == 0) { if (v) L(float, true); else L(float, false); }
    else if (dtype == 1) { if (v) L(__hip_bfloat16, true); else L(__hip_bfloat16, false); }
    else { if (v) L(__half, true); else L(__half, false); }
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}

template <typename T, bool V, int MODE>
static void launch_apply(dim3 grid, hipStream_t st, const void* x, const void* res, void* y, const float* gamma,
                         const float* beta, float* rm, float* rv, const double* part, int npart, double count,
                         float momentum, float eps, float* sm, float* si, long long HW, int C, int relu)
{
#define L(RES, RELU)                                                                                       \
    hipLaunchKernelGGL((hiast::bn_apply_kernel<T, V, RES, RELU, MODE>), grid, dim3(256), 0, st, (const T*)x, \
                       (const T*)res, (T*)y, gamma, beta, rm, rv, part, npart, count, momentum, eps, sm, si, HW, C)
    if (res) { if (relu) L(true, true); else L(true, false); }
    else { if (relu) L(false, true); else L(false, false); }
#undef L
}

extern "C" int hiast_bn_act_apply(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, const double* part, int npart,
                                  double count, float momentum, float eps, int relu, float* save_mean,
                                  float* save_invstd, int B, int C, int64_t HW, int dtype, hiast_stream_t stream)
{
    int e = bn_check(x, B, C, HW, dtype);
    if (e) return e;
    if (!y) return HIAST_E_ARG;
    const bool train = part != nullptr;
    if (train && (!save_mean || !save_invstd || npart <= 0 || count <= 0)) return HIAST_E_ARG;
    if (!train && (!running_mean || !running_var)) return HIAST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype ? 2 : 4;
    const bool v = hiast::vec_ok(x, res, y, nullptr, HW, esz);
    dim3 grid = hiast::apply_grid(HW, C, B, v ? (dtype ? 8 : 4) : 1);
#define A(T, V, M) launch_apply<T, V, M>(grid, st, x, res, y, gamma, beta, running_mean, running_var, part, npart, \
                                         count, momentum, eps, save_mean, save_invstd, (long long)HW, C, relu)
    if (dtype == 0) {
        if (train) { if (v) A(float, true, 1); else A(float, false, 1); }
        else { if (v) A(float, true, 0); else A(float, false, 0); }
    } else if (dtype == 1) {
        if (train) { if (v) A(__hip_bfloat16, true, 1); else A(__hip_bfloat16, false, 1); }
        else { if (v) A(__hip_bfloat16, true, 0); else A(__hip_bfloat16, false, 0); }
    } else {
        if (train) { if (v) A(__half, true, 1); else A(__half, false, 1); }
        else { if (v) A(__half, true, 0); else A(__half, false, 0); }
    }
#undef A
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_bn_act_bwd_stats(const void* dy, const void* y, const void* x, const float* save_mean,
                                      const float* save_invstd, int relu, int B, int C, int64_t HW, int dtype,
                                      double* part, hiast_stream_t stream)
{
    int e = bn_check(x, B, C, HW, dtype);
    if (e) return e;
    if (!dy || !save_mean || !save_invstd || !part || (relu && !y)) return HIAST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(C, B);
    const bool v = hiast::vec_ok(dy, y, x, nullptr, HW, dtype ? 2 : 4);
#define L(T, V, R)                                                                                             \
    hipLaunchKernelGGL((hiast::bn_bwd_stats_kernel<T, V, R>), grid, dim3(256), 0, st, (const T*)dy, (const T*)y, \
                       (const T*)x, save_mean, save_invstd, (long long)HW, C, part)
    if (dtype == 0) {
        if (v) { if (relu) L(float, true, true); else L(float, true, false); }
        else { if (relu) L(float, false, true); else L(float, false, false); }
    } else if (dtype == 1) {
        if (v) { if (relu) L(__hip_bfloat16, true, true); else L(__hip_bfloat16, true, false); }
        else { if (relu) L(__hip_bfloat16, false, true); else L(__hip_bfloat16, false, false); }
    } else {
        if (v) { if (relu) L(__half, true, true); else L(__half, true, false); }
        else { if (relu) L(__half, false, true); else L(__half, false, false); }
    }
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_bn_act_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma,
                                      const float* save_mean, const float* save_invstd, const double* part,
                                      int npart, double count, int relu, void* dx, void* dres, float* dgamma,
                                      float* dbeta, int B, int C, int64_t HW, int dtype, hiast_stream_t stream)
{
    int e = bn_check(x, B, C, HW, dtype);
    if (e) return e;
    if (!dy || !save_mean || !save_invstd || !part || !dx || (relu && !y) || npart <= 0 || count <= 0)
        return HIAST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool v = hiast::vec_ok(dy, y, x, dx, HW, dtype ? 2 : 4) && hiast::vec_ok(dres, nullptr, nullptr, nullptr, HW, dtype ? 2 : 4);
    dim3 grid = hiast::apply_grid(HW, C, B, v ? (dtype ? 8 : 4) : 1);
#define L(T, V, R, D)                                                                                          \
    hipLaunchKernelGGL((hiast::bn_bwd_apply_kernel<T, V, R, D>), grid, dim3(256), 0, st, (const T*)dy,           \
                       (const T*)y, (const T*)x, gamma, save_mean, save_invstd, part, npart, count, (T*)dx,      \
                       (T*)dres, dgamma, dbeta, (long long)HW, C)
#define LL(T, V)                                                           \
    if (relu) { if (dres) L(T, V, true, true); else L(T, V, true, false); } \
    else { if (dres) L(T, V, false, true); else L(T, V, false, false); }
    if (dtype == 0) { if (v) { LL(float, true) } else { LL(float, false) } }
    else if (dtype == 1) { if (v) { LL(__hip_bfloat16, true) } else { LL(__hip_bfloat16, false) } }
    else { if (v) { LL(__half, true) } else { LL(__half, false) } }
#undef LL
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}
