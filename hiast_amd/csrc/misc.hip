// Version / error strings, K11 (EMA teacher update) and K12 (IoU area histograms).
#include <hip/hip_bf16.h>

#include <atomic>

#include "common.h"

namespace hiast {

// utils/utils.py:115-123: ema = ema*gamma + p*(1-gamma) — mul, mul, add (no fma: the library is
// built with -ffp-contract=off), one launch for the whole parameter list.
// 64Ki-element chunks; 256 threads x float4 x 64 iterations per chunk.
__global__ __launch_bounds__(256) void ema_kernel(const hiast_ema_rec* __restrict__ table,
                                                  const int32_t* __restrict__ chunk_tensor,
                                                  const int64_t* __restrict__ chunk_start,
                                                  float gamma, float omg)
{
    const hiast_ema_rec r = table[chunk_tensor[blockIdx.x]];
    const int64_t s = chunk_start[blockIdx.x];
    const int64_t e = (s + 65536 < r.n) ? s + 65536 : r.n;
    float* __restrict__ d = r.ema;
    const float* __restrict__ p = r.p;
    const bool vec = ((((uintptr_t)d) | ((uintptr_t)p)) & 15) == 0;   // s is a multiple of 64Ki
    if (vec) {
        const int64_t nv = (e - s) / 4;
        float4* d4 = reinterpret_cast<float4*>(d + s);
        const float4* p4 = reinterpret_cast<const float4*>(p + s);
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
            float4 a = d4[i], b = p4[i];
            a.x = a.x * gamma + b.x * omg;
            a.y = a.y * gamma + b.y * omg;
            a.z = a.z * gamma + b.z * omg;
            a.w = a.w * gamma + b.w * omg;
            d4[i] = a;
        }
        for (int64_t i = s + nv * 4 + threadIdx.x; i < e; i += 256) d[i] = d[i] * gamma + p[i] * omg;
    } else {
        for (int64_t i = s + threadIdx.x; i < e; i += 256) d[i] = d[i] * gamma + p[i] * omg;
    }
}

// K14 — ToTensor + Normalize on the device (reference: transform, sseg/datasets/utils.py:37-55 = torchvision ToTensor
// then Normalize, executed in the DataLoader workers on float tensors): the workers hand over the uint8 HWC image (4x
// fewer bytes over PCIe), this kernel writes the normalised float32 CHW tensor.  Arithmetic = torch's, operation for
// operation (IEEE float division twice), so the result is bit-identical: v = float(u8) / 255;  out = (v - mean) / std.
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                           long long HW, float m0, float m1, float m2, float s0, float s1,
                                                           float s2)
{
    const int b = blockIdx.y;
    const unsigned char* src = img + (size_t)b * HW * 3;
    float* dst = out + (size_t)b * 3 * HW;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const float r = (float)src[p * 3] / 255.0f, g = (float)src[p * 3 + 1] / 255.0f, bl = (float)src[p * 3 + 2] / 255.0f;
        dst[p] = (r - m0) / s0;
        dst[HW + p] = (g - m1) / s1;
        dst[2 * HW + p] = (bl - m2) / s2;
    }
}

// copy a list of small tensors (the BatchNorm buffers the EMA teacher takes over from the student,
// utils/utils.py:120-123: ~300 tensors of <= 8 KB) in ONE launch: block b copies tensor b.
__global__ __launch_bounds__(256) void multi_copy_kernel(const hiast_copy_rec* __restrict__ table)
{
    const hiast_copy_rec r = table[blockIdx.x];
    unsigned char* d = (unsigned char*)r.dst;
    const unsigned char* s = (const unsigned char*)r.src;
    if (((((uintptr_t)d) | ((uintptr_t)s)) & 15) == 0) {
        const int64_t nv = r.nbytes / 16;
        for (int64_t i = threadIdx.x; i < nv; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (int64_t i = nv * 16 + threadIdx.x; i < r.nbytes; i += 256) d[i] = s[i];
    } else {
        for (int64_t i = threadIdx.x; i < r.nbytes; i += 256) d[i] = s[i];
    }
}

// K13 — Adam step over the whole parameter list in ONE launch (reference: torch.optim.Adam(weight_decay=5e-4) built
// in utils/utils.py:135-154 and stepped by BaseTrainer.update_model, workflows/trainer/base_trainer.py:127-141;
// there a per-tensor loop of ~10 elementwise kernels).  torch's single-tensor formulas, in its operation order:
//   g' = g + wd*p;  m = m + (g' - m)*(1 - b1);  v = v*b2 + (1 - b2)*g'*g';
//   p  = p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// 64Ki-element chunks like the EMA kernel; lr and the bias corrections travel per tensor in the record.
// Mixed precision with dynamic loss scaling (apex O1 / torch GradScaler; base_trainer.py:129-131 of the reference):
// the gradients arrive multiplied by the loss scale and a step whose gradients hold an inf / NaN must be skipped
// ENTIRELY (no moment update, no step count).  Both decisions are taken on the device: adam_prepare_kernel turns the
// scaler's device scalars (scale, found_inf) into a control block — skip flag, 1 / scale, the count of APPLIED steps and
// the bias corrections that count implies — and adam_kernel reads it; the host never waits for found_inf (GradScaler's
// own `if not found_inf.item(): optimizer.step()` drains the launch queue once per iteration).  The host keeps counting
// ATTEMPTED steps per tensor (record field `step`); the device counts the skipped ones; a tensor's applied-step count is
// the difference.  The skipped count is ONE number for all tensors: the host (utils.FusedAdam.step) folds it into the
// per-tensor counts whenever the set of participating tensors changes (a tensor joins later or sits out a step with grad
// None), so between two folds every counted tensor took part in every step and keeps its own count, as in torch.
__global__ void adam_prepare_kernel(hiast_adam_ctl* __restrict__ ctl, const float* __restrict__ grad_scale,
                                    const float* __restrict__ found_inf)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const bool skip = found_inf && *found_inf != 0.f;
    if (skip) ctl->skipped += 1.0f;
    ctl->skip = skip ? 1.0f : 0.0f;
    ctl->inv_scale = grad_scale ? (float)(1.0 / (double)*grad_scale) : 1.0f;
}

__global__ __launch_bounds__(256) void adam_kernel(const hiast_adam_rec* __restrict__ table,
                                                   const int32_t* __restrict__ chunk_tensor,
                                                   const int64_t* __restrict__ chunk_start, float beta1, float beta2,
                                                   float omb1, float omb2, float eps, float wd,
                                                   const hiast_adam_ctl* __restrict__ ctl, double beta1d, double beta2d)
{
    const hiast_adam_rec r = table[chunk_tensor[blockIdx.x]];
    const int64_t s = chunk_start[blockIdx.x];
    const int64_t e = (s + 65536 < r.n) ? s + 65536 : r.n;
    float bc1 = r.bc1, bc2_sqrt = r.bc2_sqrt, inv_scale = 1.0f;
    if (ctl) {                                // device-side control block: skipped step / loss scale / skipped-step count
        if (ctl->skip != 0.f) return;
        inv_scale = ctl->inv_scale;
        double t = (double)(r.step - ctl->skipped);                  // applied steps of THIS tensor, this one included
        t = t < 1.0 ? 1.0 : t;                // (the host folds the counts when the set of tensors changes; never 1 - beta^0)
        bc1 = (float)(1.0 - pow(beta1d, t));  // torch: 1 - beta1 ** step and sqrt(1 - beta2 ** step), formed in double
        bc2_sqrt = (float)sqrt(1.0 - pow(beta2d, t));
    }
    const float step_size = r.lr / bc1;       // omb1 / omb2 = float(1 - beta) formed in double on the host, as torch does
    auto upd = [&](float& p, float g, float& m, float& v) {
        g = g * inv_scale;                    // (x 1.0f is exact: the unscaled path keeps its bits)
        if (wd != 0.f) g = g + wd * p;
        m = m + (g - m) * omb1;
        v = v * beta2 + omb2 * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p = p - step_size * (m / denom);
    };
    const bool vec = ((((uintptr_t)r.p) | ((uintptr_t)r.g) | ((uintptr_t)r.m) | ((uintptr_t)r.v)) & 15) == 0;
    if (vec) {
        const int64_t nv = (e - s) / 4;
        float4* p4 = reinterpret_cast<float4*>(r.p + s);
        const float4* g4 = reinterpret_cast<const float4*>(r.g + s);
        float4* m4 = reinterpret_cast<float4*>(r.m + s);
        float4* v4 = reinterpret_cast<float4*>(r.v + s);
        // MISC_NT=1 (A/B build): the moments and the gradient are touched once per step — non-temporal, so that the parameters
        // (read next by the EMA update and the weight re-packing) stay in the Infinity Cache
#ifndef MISC_NT
#define MISC_NT 0
#endif
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
            float4 p = p4[i], m, v, g;
            if (MISC_NT) {
                const f32x4_ mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(m4 + i));
                const f32x4_ vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(v4 + i));
                const f32x4_ gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(g4 + i));
                m = make_float4(mm.x, mm.y, mm.z, mm.w); v = make_float4(vv.x, vv.y, vv.z, vv.w); g = make_float4(gg.x, gg.y, gg.z, gg.w);
            } else {
                m = m4[i]; v = v4[i]; g = g4[i];
            }
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            p4[i] = p;
            if (MISC_NT) {
                __builtin_nontemporal_store((f32x4_){m.x, m.y, m.z, m.w}, reinterpret_cast<f32x4_*>(m4 + i));
                __builtin_nontemporal_store((f32x4_){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_*>(v4 + i));
            } else {
                m4[i] = m; v4[i] = v;
            }
        }
        for (int64_t i = s + nv * 4 + threadIdx.x; i < e; i += 256) upd(r.p[i], r.g[i], r.m[i], r.v[i]);
    } else {
        for (int64_t i = s + threadIdx.x; i < e; i += 256) upd(r.p[i], r.g[i], r.m[i], r.v[i]);
    }
}

// utils/metrics.py:6-19: pred[target==255] = 255; inter = hist(pred[pred==target]);
// area_pred = hist(pred); area_tgt = hist(target) over K bins.  Per-thread pixels -> LDS integer
// histograms (3*K counters) -> one global integer atomic per non-empty counter per block.
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred,
                                                        const int64_t* __restrict__ target,
                                                        int64_t N, int K,
                                                        unsigned long long* __restrict__ inter,
                                                        unsigned long long* __restrict__ ap,
                                                        unsigned long long* __restrict__ at)
{
    extern __shared__ unsigned s_h[];   // [3][K]
    for (int i = threadIdx.x; i < 3 * K; i += 256) s_h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int64_t t = target[i];
        const int64_t p = (t == HIAST_IGNORE) ? HIAST_IGNORE : pred[i];
        if (p >= 0 && p < K) atomicAdd(&s_h[K + (int)p], 1u);
        if (t >= 0 && t < K) atomicAdd(&s_h[2 * K + (int)t], 1u);
        if (p == t && p >= 0 && p < K) atomicAdd(&s_h[(int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * K; i += 256) {
        const unsigned v = s_h[i];
        if (!v) continue;
        unsigned long long* dst = i < K ? inter + i : (i < 2 * K ? ap + (i - K) : at + (i - 2 * K));
        atomicAdd(dst, (unsigned long long)v);
    }
}

}  // namespace hiast

extern "C" int hiast_version(void) { return HIAST_ABI_VERSION; }

// ---- device geometry and the CU reserve (common.h) --------------------------------------------------------------------
static std::atomic<int> g_reserve_cus{0};
static int query_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    static std::atomic<int> cache[64];
    if (dev >= 0 && dev < 64 && cache[dev].load(std::memory_order_relaxed) > 0) return cache[dev].load(std::memory_order_relaxed);
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64 && n > 0) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}
int hiast_cu_count()
{
    const int n = query_cus();
    return n >= 8 ? n : 256;            // (no device visible: the gfx950 figure, so that the host-side planners stay usable)
}
int hiast_grid_cus()
{
    const int n = hiast_cu_count() - g_reserve_cus.load(std::memory_order_relaxed);
    return n >= 8 ? n : 8;
}
extern "C" int hiast_device_cus(void) { return query_cus(); }
extern "C" int hiast_get_reserve_cus(void) { return g_reserve_cus.load(std::memory_order_relaxed); }
extern "C" int hiast_set_reserve_cus(int n)
{
    if (n < 0) return HIAST_E_ARG;
    const int cus = hiast_cu_count();
    n = (n + 7) / 8 * 8;                // whole rounds of the 8 XCDs: every XCD gives up n / 8 CUs
    if (n > cus / 2) return HIAST_E_RANGE;
    return g_reserve_cus.exchange(n);
}
// A stream whose kernels cannot be placed on the `reserve` highest-numbered CUs of the queue's CU mask.  KFD maps mask bit i of
// a multi-XCC device to XCC (i % 8), then to shader engines and CUs inside it, so clearing the top `reserve` bits (a multiple
// of 8) takes reserve / 8 CUs from every XCD.  Unlike a smaller grid this also holds for kernels that are not ours.
extern "C" int hiast_stream_create_reserved(hiast_stream_t* out, int reserve)
{
    if (!out || reserve < 0) return HIAST_E_ARG;
    const int cus = query_cus();
    if (cus <= 0) return (int)hipErrorNoDevice;
    reserve = (reserve + 7) / 8 * 8;
    if (reserve > cus / 2) return HIAST_E_RANGE;
    const int words = (cus + 31) / 32;
    uint32_t mask[32] = {0};
    if (words > 32) return HIAST_E_RANGE;
    for (int i = 0; i < cus - reserve; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    *out = (hiast_stream_t)st;
    return 0;
}
extern "C" int hiast_stream_destroy(hiast_stream_t s)
{
    if (!s) return HIAST_E_ARG;
    const hipError_t e = hipStreamDestroy((hipStream_t)s);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return 0;
}

extern "C" const char* hiast_error_string(int code)
{
    switch (code) {
        case 0: return "ok";
        case HIAST_E_ARG: return "hiast: null pointer or non-positive extent";
        case HIAST_E_RANGE: return "hiast: extent outside the range the kernels are built for";
        case HIAST_E_WS: return "hiast: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "hiast: unknown error";
    }
}

extern "C" int hiast_ema_update(const hiast_ema_rec* table, const int32_t* chunk_tensor,
                                const int64_t* chunk_start, int n_chunks, float gamma,
                                float one_minus_gamma, hiast_stream_t stream)
{
    if (!table || !chunk_tensor || !chunk_start) return HIAST_E_ARG;
    if (n_chunks <= 0) return HIAST_E_ARG;
    hipLaunchKernelGGL(hiast::ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table,
                       chunk_tensor, chunk_start, gamma, one_minus_gamma);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_confusion_hist(const int64_t* pred, const int64_t* target, int64_t N, int K,
                                    int64_t* inter, int64_t* area_pred, int64_t* area_tgt,
                                    hiast_stream_t stream)
{
    if (!pred || !target || !inter || !area_pred || !area_tgt) return HIAST_E_ARG;
    if (N <= 0 || K <= 0) return HIAST_E_ARG;
    if (K > 1024) return HIAST_E_RANGE;
    int64_t nb = (N + 256 * 16 - 1) / (256 * 16);
    int grid = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
    hipLaunchKernelGGL(hiast::confusion_kernel, dim3(grid), dim3(256), 3 * K * sizeof(unsigned),
                       (hipStream_t)stream, pred, target, N, K, (unsigned long long*)inter,
                       (unsigned long long*)area_pred, (unsigned long long*)area_tgt);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_adam_step(const hiast_adam_rec* table, const int32_t* chunk_tensor, const int64_t* chunk_start,
                               int n_chunks, double beta1, double beta2, float eps, float weight_decay,
                               const hiast_adam_ctl* ctl, hiast_stream_t stream)
{
    if (!table || !chunk_tensor || !chunk_start) return HIAST_E_ARG;
    if (n_chunks <= 0) return HIAST_E_ARG;
    hipLaunchKernelGGL(hiast::adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor,
                       chunk_start, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                       weight_decay, ctl, beta1, beta2);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_adam_prepare(hiast_adam_ctl* ctl, const float* grad_scale, const float* found_inf,
                                  hiast_stream_t stream)
{
    if (!ctl) return HIAST_E_ARG;
    hipLaunchKernelGGL(hiast::adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, grad_scale, found_inf);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_multi_copy(const hiast_copy_rec* table, int n_tensors, hiast_stream_t stream)
{
    if (!table) return HIAST_E_ARG;
    if (n_tensors <= 0) return HIAST_E_ARG;
    hipLaunchKernelGGL(hiast::multi_copy_kernel, dim3(n_tensors), dim3(256), 0, (hipStream_t)stream, table);
    HIAST_CHECK_LAUNCH();
    return 0;
}

namespace hiast {

// K18: MaxPool2d(3, stride 2, padding 1) on channels-last bf16 activations (the stem of the mixed-precision training
// forward), forward with a one-byte window position per element instead of the library's 8-byte flat index, and its
// backward as a gather over the <= 4 windows that cover an input pixel.  Ties go to the first element in row-major
// window order and a NaN takes the maximum, as in ATen's kernel; the backward adds in fp32 and rounds once.
// F16: fp16 rows (HIAST_FMT_FP16) instead of bf16
template <bool F16>
__global__ __launch_bounds__(256) void maxpool_cl_fwd_kernel(const unsigned short* __restrict__ x,
                                                             unsigned short* __restrict__ y, unsigned char* __restrict__ idx,
                                                             int B, int H, int W, int C, int Ho, int Wo)
{
    const int G = C >> 3;
    const long long total = (long long)B * Ho * Wo * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const long long pix = i / G;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((long long)Wo * Ho));
        float m[8];
        unsigned code[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; code[k] = 255u; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * yo - 1 + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo - 1 + dx;
                if (xx < 0 || xx >= W) continue;
                const uint4 r = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + yy) * W + xx) * C + cg * 8);
                const unsigned w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v0 = H16<F16>::lo(w4[q]), v1 = H16<F16>::hi(w4[q]);
                    // ATen: maxidx starts at the first element of the window, then (val > maxval || isnan(val)) takes over
                    if (code[2 * q] == 255u) code[2 * q] = dy * 3 + dx;
                    if (v0 > m[2 * q] || v0 != v0) { m[2 * q] = v0; code[2 * q] = dy * 3 + dx; }
                    if (code[2 * q + 1] == 255u) code[2 * q + 1] = dy * 3 + dx;
                    if (v1 > m[2 * q + 1] || v1 != v1) { m[2 * q + 1] = v1; code[2 * q + 1] = dy * 3 + dx; }
                }
            }
        }
        unsigned pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            pk[q] = H16<F16>::pack(m[2 * q], m[2 * q + 1]);       // exact: the maximum is one of the 16-bit inputs
        *reinterpret_cast<uint4*>(y + (size_t)pix * C + cg * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint2*>(idx + (size_t)pix * C + cg * 8) =
            make_uint2(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24),
                       code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24));
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void maxpool_cl_bwd_kernel(const unsigned short* __restrict__ dy,
                                                             const unsigned char* __restrict__ idx,
                                                             unsigned short* __restrict__ dx, int B, int H, int W, int C,
                                                             int Ho, int Wo)
{
    const int G = C >> 3;
    const long long total = (long long)B * H * W * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const long long pix = i / G;
        const int xi = (int)(pix % W), yi = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        // windows (yo, xo) with 2*yo - 1 <= yi <= 2*yo + 1: yo = yi / 2 and, for odd yi, also (yi + 1) / 2 (ascending order)
        const int y0 = yi >> 1, ny = (yi & 1) ? 2 : 1, x0 = xi >> 1, nx = (xi & 1) ? 2 : 1;
        for (int a = 0; a < ny; ++a) {
            const int yo = y0 + a;
            if (yo >= Ho) continue;
            const unsigned pdy = (unsigned)(yi - (2 * yo - 1));
            for (int c = 0; c < nx; ++c) {
                const int xo = x0 + c;
                if (xo >= Wo) continue;
                const unsigned pos = pdy * 3u + (unsigned)(xi - (2 * xo - 1));
                const size_t o = (((size_t)b * Ho + yo) * Wo + xo) * C + cg * 8;
                const uint2 cd = *reinterpret_cast<const uint2*>(idx + o);
                const uint4 g = *reinterpret_cast<const uint4*>(dy + o);
                const unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned code = ((k < 4 ? cd.x : cd.y) >> (8 * (k & 3))) & 255u;
                    const float gv = (k & 1) ? H16<F16>::hi(gw[k >> 1]) : H16<F16>::lo(gw[k >> 1]);
                    if (code == pos) acc[k] += gv;
                }
            }
        }
        unsigned pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            pk[q] = H16<F16>::pack(acc[2 * q], acc[2 * q + 1]);
        *reinterpret_cast<uint4*>(dx + (size_t)pix * C + cg * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

}  // namespace hiast

static int maxpool_cl_check(const void* a, const void* b, const void* c, int B, int H, int W, int C)
{
    if (!a || !b || !c) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return HIAST_E_ARG;
    if (C % 8 != 0 || ((((uintptr_t)a) | ((uintptr_t)b)) & 15) || (((uintptr_t)c) & 7) ||
        (long long)B * H * W * C >= (1ll << 40))
        return HIAST_E_RANGE;
    return 0;
}

extern "C" int hiast_maxpool3x3s2_nhwc_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int fmt,
                                           hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    int e = maxpool_cl_check(x, y, idx, B, H, W, C);
    if (e) return e;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long long nb = ((long long)B * Ho * Wo * (C / 8) + 255) / 256;
    nb = nb > 16384 ? 16384 : nb;
    if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::maxpool_cl_fwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (unsigned short*)y, idx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(hiast::maxpool_cl_fwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (unsigned short*)y, idx, B, H, W, C, Ho, Wo);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_maxpool3x3s2_nhwc_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C,
                                           int fmt, hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    int e = maxpool_cl_check(dy, dx, idx, B, H, W, C);
    if (e) return e;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long long nb = ((long long)B * H * W * (C / 8) + 255) / 256;
    nb = nb > 32768 ? 32768 : nb;
    if (fmt == HIAST_FMT_FP16)
        hipLaunchKernelGGL(hiast::maxpool_cl_bwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)dy, idx, (unsigned short*)dx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(hiast::maxpool_cl_bwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)dy, idx, (unsigned short*)dx, B, H, W, C, Ho, Wo);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_normalize_u8(const uint8_t* img, float* out, int B, int64_t HW, const float* mean, const float* std,
                                  hiast_stream_t stream)
{
    if (!img || !out || !mean || !std) return HIAST_E_ARG;
    if (B <= 0 || HW <= 0) return HIAST_E_ARG;
    if (B > 65535) return HIAST_E_RANGE;
    long long nb = (HW + 256 * 4 - 1) / (256 * 4);
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(hiast::normalize_u8_kernel, dim3((unsigned)nb, B), dim3(256), 0, (hipStream_t)stream, img, out,
                       (long long)HW, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    HIAST_CHECK_LAUNCH();
    return 0;
}
