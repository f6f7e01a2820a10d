// micro-bench: which part of the BatchNorm apply (+ residual + ReLU) pass keeps it at 4.4 TB/s when a float4 add of the same
// three streams runs at 6.2 (tools/micro/mall_order.hip)?  The kernel below is bnh_apply_kernel<true, true> of bn_nhwc.hip on
// a [65536 x 1024] bf16 tensor, with its ingredients switchable:
//   MATH 0: 16-byte words passed through (integer add), 1: bf16 unpack -> fma + residual + ReLU -> bf16 pack
//   MASK: the 1-byte-per-lane ReLU bit mask store;  PARAMS: per-channel scale / shift from memory (else constants)
//   WALK 0: grid-stride chunks (as bn_nhwc.hip), 1: one contiguous range per block
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bn_apply_bisect.hip -o tools/micro/bn_apply_bisect && tools/micro/bn_apply_bisect
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>

template <int MATH, bool MASK, bool PARAMS, int WALK, int UNR>
__global__ __launch_bounds__(256) void apply(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                             unsigned short* __restrict__ y, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, long long M, int C, unsigned char* __restrict__ mask)
{
    const int G = C >> 3, RPP = 256 / G;
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    const long long chunk = RPP * UNR;
    long long step, lim, r0;
    if (WALK == 0) { step = (long long)gridDim.x * chunk; lim = M; r0 = (long long)blockIdx.x * chunk + rsub; }
    else {
        const long long per = ((M + gridDim.x - 1) / gridDim.x + chunk - 1) / chunk * chunk;
        step = chunk; lim = (blockIdx.x + 1) * per < M ? (blockIdx.x + 1) * per : M; r0 = blockIdx.x * per + rsub;
    }
    uint4 rx[UNR], rr[UNR];
    auto load_rows = [&](long long rb) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + u * RPP;
            const size_t off = (size_t)(r < lim ? r : (rb < lim ? rb : 0)) * C + cg * 8;
            rx[u] = *reinterpret_cast<const uint4*>(x + off);
            rr[u] = *reinterpret_cast<const uint4*>(res + off);
        }
    };
    load_rows(r0);
    float scale[8], shift[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { scale[k] = PARAMS ? gamma[cg * 8 + k] : 1.25f; shift[k] = PARAMS ? beta[cg * 8 + k] : -0.125f; }
    while (r0 < lim) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = r0 + u * RPP;
            if (r >= lim) break;
            const size_t off = (size_t)r * C + cg * 8;
            uint4 o;
            unsigned bits = 0;
            if (MATH == 0) {
                o = make_uint4(rx[u].x + rr[u].x, rx[u].y + rr[u].y, rx[u].z + rr[u].z, rx[u].w + rr[u].w);
                bits = o.x & 255u;
            } else {
                const unsigned wx[4] = {rx[u].x, rx[u].y, rx[u].z, rx[u].w}, wr[4] = {rr[u].x, rr[u].y, rr[u].z, rr[u].w};
                unsigned w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = fmaf(__uint_as_float(wx[i] << 16), scale[2 * i], shift[2 * i]) + __uint_as_float(wr[i] << 16);
                    float b = fmaf(__uint_as_float(wx[i] & 0xFFFF0000u), scale[2 * i + 1], shift[2 * i + 1]) + __uint_as_float(wr[i] & 0xFFFF0000u);
                    a = a > 0.f ? a : 0.f; b = b > 0.f ? b : 0.f;
                    bits |= (a > 0.f ? 1u : 0u) << (2 * i) | (b > 0.f ? 1u : 0u) << (2 * i + 1);
                    w[i] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(a)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(b)) << 16);
                }
                o = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(y + off) = o;
            if (MASK) mask[(size_t)r * G + cg] = (unsigned char)bits;
        }
        r0 += step;
        if (r0 < lim) load_rows(r0);
    }
}

// the float4 kernel of mall_order.hip on the same bytes (reference rate)
__global__ __launch_bounds__(256) void add4(const float4* g, const float4* x, float4* z)
{
    const float4* pg = g + (size_t)blockIdx.x * 4096;
    const float4* px = x + (size_t)blockIdx.x * 4096;
    float4* pz = z + (size_t)blockIdx.x * 4096;
#pragma unroll 4
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const float4 a = pg[i], b = px[i];
        pz[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

static unsigned short *X, *R, *Y;
static float *Gm, *Bt;
static unsigned char* Mk;
static const long long M = 65536;
static const int C = 1024;

template <int MATH, bool MASK, bool PARAMS, int WALK, int UNR>
static void run(const char* name, int nb)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((apply<MATH, MASK, PARAMS, WALK, UNR>), dim3(nb), dim3(256), 0, 0, X, R, Y, Gm, Bt, M, C, Mk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 1 && ms < best) best = ms;
    }
    const double bytes = 3.0 * M * C * 2 + (MASK ? M * C / 8 : 0);
    printf("%-58s blocks %5d | %6.1f us | %5.2f TB/s\n", name, nb, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)M * C * 2;
    (void)hipMalloc(&X, bytes); (void)hipMalloc(&R, bytes); (void)hipMalloc(&Y, bytes); (void)hipMalloc(&Mk, M * C / 8);
    (void)hipMalloc(&Gm, C * 4); (void)hipMalloc(&Bt, C * 4);
    (void)hipMemset(X, 0x3c, bytes); (void)hipMemset(R, 0x3d, bytes); (void)hipMemset(Gm, 0, C * 4); (void)hipMemset(Bt, 0, C * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(add4, dim3(2048), dim3(256), 0, 0, (const float4*)X, (const float4*)R, (float4*)Y);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 1 && ms < best) best = ms;
    }
    printf("%-58s blocks %5d | %6.1f us | %5.2f TB/s\n", "float4 add, 64 KiB contiguous per block (reference)", 2048, best * 1e3, 3.0 * bytes / (best * 1e-3) / 1e12);
    run<1, true, true, 0, 4>("as bn_nhwc.hip: math + mask + params, grid-stride, 4 rows", 2048);
    run<1, false, true, 0, 4>("  without the mask bytes", 2048);
    run<1, false, false, 0, 4>("  without mask and parameter loads", 2048);
    run<0, false, false, 0, 4>("  words passed through (no bf16 math)", 2048);
    run<0, false, false, 1, 4>("  passed through, contiguous range per block", 2048);
    run<0, false, false, 1, 8>("  passed through, contiguous, 8 rows in flight", 2048);
    run<0, false, false, 1, 4>("  passed through, contiguous, 4096 blocks", 4096);
    run<0, false, false, 1, 4>("  passed through, contiguous, 8192 blocks", 8192);
    run<0, false, false, 1, 2>("  passed through, contiguous, 2 rows in flight, 8192 blocks", 8192);
    run<1, true, true, 1, 4>("math + mask + params, contiguous, 8192 blocks", 8192);
    run<1, true, true, 0, 4>("math + mask + params, grid-stride, 8192 blocks", 8192);
    run<1, true, true, 0, 2>("math + mask + params, grid-stride, 2 rows, 16384 blocks", 16384);
    return 0;
}
