// micro-bench: does the ORDER in which a consumer kernel walks a buffer its producer has just written (or read) matter?
// The 256 MiB Infinity Cache keeps the most recently touched lines; a consumer that starts at the producer's tail finds them,
// one that starts at its head finds what the producer's own later traffic has already pushed out.
//   case 1: W writes Y (134 MB)                    -> R reads Y and X (cold), writes Z         (conv -> BN apply + residual)
//   case 2: P reads G and X (2 x 134 MB), reduces  -> A reads G and X again, writes Z          (BN backward: sums -> apply)
// each consumer in the producer's block order ("fwd") and in the opposite order ("rev").
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_order.hip -o tools/micro/mall_order && tools/micro/mall_order
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHUNK_F4 4096            // 16-byte words per block and buffer: 64 KiB contiguous

__global__ __launch_bounds__(256) void writer(float4* y, int nblk)
{
    float4* p = y + (size_t)blockIdx.x * CHUNK_F4;
    const float v = (float)blockIdx.x;
    for (int i = threadIdx.x; i < CHUNK_F4; i += 256) p[i] = make_float4(v, v, v, v);
}

__global__ __launch_bounds__(256) void partial(const float4* g, const float4* x, float* sums, int nblk)
{
    const float4* pg = g + (size_t)blockIdx.x * CHUNK_F4;
    const float4* px = x + (size_t)blockIdx.x * CHUNK_F4;
    float s = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < CHUNK_F4; i += 256) {
        const float4 a = pg[i], b = px[i];
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    if (s == 12345.678f) sums[blockIdx.x] = s;
}

template <bool REV>
__global__ __launch_bounds__(256) void apply(const float4* g, const float4* x, float4* z, int nblk)
{
    const int blk = REV ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const float4* pg = g + (size_t)blk * CHUNK_F4;
    const float4* px = x + (size_t)blk * CHUNK_F4;
    float4* pz = z + (size_t)blk * CHUNK_F4;
#pragma unroll 4
    for (int i = threadIdx.x; i < CHUNK_F4; i += 256) {
        const float4 a = pg[i], b = px[i];
        pz[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

static float timed(hipEvent_t e0, hipEvent_t e1)
{
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const size_t bytes = 134217728;                  // 65536 rows x 1024 channels x 2 B
    const int nblk = (int)(bytes / 16 / CHUNK_F4);   // 2048 blocks
    float4 *y, *x, *z, *junk;
    float* sums;
    hipMalloc(&y, bytes); hipMalloc(&x, bytes); hipMalloc(&z, bytes); hipMalloc(&junk, 4 * bytes); hipMalloc(&sums, 4 * nblk);
    hipMemset(y, 0, bytes); hipMemset(x, 0, bytes); hipMemset(z, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rev = 0; rev < 2; ++rev) {
        float best_r = 1e9f, best_a = 1e9f, best_w = 1e9f, best_p = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            // case 1
            hipMemsetAsync(junk, rep, 4 * bytes, 0);                       // 512 MiB through the caches: cold start
            hipEventRecord(e0);
            hipLaunchKernelGGL(writer, dim3(nblk), dim3(256), 0, 0, y, nblk);
            hipEventRecord(e1);
            float w = timed(e0, e1);
            hipEventRecord(e0);
            if (rev) hipLaunchKernelGGL(apply<true>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            else hipLaunchKernelGGL(apply<false>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            hipEventRecord(e1);
            float r = timed(e0, e1);
            // case 2
            hipMemsetAsync(junk, rep, 4 * bytes, 0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(partial, dim3(nblk), dim3(256), 0, 0, y, x, sums, nblk);
            hipEventRecord(e1);
            float p = timed(e0, e1);
            hipEventRecord(e0);
            if (rev) hipLaunchKernelGGL(apply<true>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            else hipLaunchKernelGGL(apply<false>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            hipEventRecord(e1);
            float a = timed(e0, e1);
            if (rep) {
                if (w < best_w) best_w = w;
                if (r < best_r) best_r = r;
                if (p < best_p) best_p = p;
                if (a < best_a) best_a = a;
            }
        }
        printf("%s | case 1: writer %.1f us (%.2f TB/s), consumer (402 MB) %.1f us = %.2f TB/s | case 2: sums (268 MB) %.1f us = %.2f TB/s, "
               "apply (402 MB) %.1f us = %.2f TB/s\n", rev ? "consumer in REVERSE block order" : "consumer in the producer's order ",
               best_w * 1e3, bytes / (best_w * 1e-3) / 1e12, best_r * 1e3, 3.0 * bytes / (best_r * 1e-3) / 1e12, best_p * 1e3,
               2.0 * bytes / (best_p * 1e-3) / 1e12, best_a * 1e3, 3.0 * bytes / (best_a * 1e-3) / 1e12);
    }
    // the same consumer run twice back to back without anything between (everything it can keep is kept)
    for (int rev = 0; rev < 2; ++rev) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (rev) hipLaunchKernelGGL(apply<true>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            else hipLaunchKernelGGL(apply<false>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            hipEventRecord(e1);
            float a = timed(e0, e1);
            if (rep && a < best) best = a;
        }
        printf("apply replayed back to back, %s: %.1f us = %.2f TB/s\n", rev ? "reverse" : "forward", best * 1e3,
               3.0 * bytes / (best * 1e-3) / 1e12);
    }
    // alternating directions: each launch starts where the previous one ended
    {
        float best[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 10; ++rep) {
            const int rev = rep & 1;
            hipEventRecord(e0);
            if (rev) hipLaunchKernelGGL(apply<true>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            else hipLaunchKernelGGL(apply<false>, dim3(nblk), dim3(256), 0, 0, y, x, z, nblk);
            hipEventRecord(e1);
            float a = timed(e0, e1);
            if (rep > 1 && a < best[rev]) best[rev] = a;
        }
        printf("apply replayed in ALTERNATING directions: forward launches %.1f us = %.2f TB/s, reverse launches %.1f us = %.2f TB/s\n",
               best[0] * 1e3, 3.0 * bytes / (best[0] * 1e-3) / 1e12, best[1] * 1e3, 3.0 * bytes / (best[1] * 1e-3) / 1e12);
    }
    return 0;
}
