// micro-bench: fill rate of LDS tiles by LDS-DMA (buffer_load_dwordx4 ... lds) with the access shapes of igemm.hip.
// One 512-thread block per CU; per step a wave issues NA "A" pieces (8 rows x 128 B of THIS block's 256 rows, row pitch
// pa bytes, slab j = step % slabs) and NB "B" pieces (8 rows x 128 B of a 256-row table shared by every block, pitch pb),
// waits until all but the youngest (depth-1) steps have landed, optionally meets the block at a barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_fill.hip -o tools/micro/dma_fill && tools/micro/dma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, voff, soff, 0, 0);
}

template <int NA, int NB, int DEPTH, bool BARRIER>
__global__ __launch_bounds__(512) void fill_kernel(const unsigned char* a, const unsigned char* b, int pa, int pb, int slabs_a,
                                                   int slabs_b, int rep_a, int steps, int rows_per_block, long long a_bytes, long long b_bytes,
                                                   unsigned* sink)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[160 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)b_bytes, 0x00020000);
    // XCD-aware block order as in igemm: XCD k takes the k-th contiguous eighth of the row tiles
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
    int avoff[NA > 0 ? NA : 1], bvoff[NB > 0 ? NB : 1];
    for (int g = 0; g < NA; ++g) {
        const int row = (wave * NA + g) * 8 + (lane >> 3);                    // row of the block's tile
        avoff[g] = (lid * rows_per_block + row % rows_per_block) * pa + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    }
    for (int g = 0; g < NB; ++g) {
        const int row = (wave * NB + g) * 8 + (lane >> 3);
        bvoff[g] = row * pb + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    }
    constexpr int STAGE = (NA + NB) * 8 * 1024;
    constexpr int NST = (160 * 1024) / STAGE < DEPTH + 1 ? (160 * 1024) / STAGE : DEPTH + 1;
    static_assert(NST >= DEPTH, "depth does not fit in LDS");
    int ja = 0, jb = 0, st = 0, ra = 0;
    auto issue = [&]() {
        unsigned char* base = smem + st * STAGE + wave * (NA + NB) * 1024;
#pragma unroll
        for (int g = 0; g < NA; ++g) dma16(ars, base + g * 1024, avoff[g], ja * 128);
#pragma unroll
        for (int g = 0; g < NB; ++g) dma16(brs, base + (NA + g) * 1024, bvoff[g], jb * 128);
        if (++ra == rep_a) { ra = 0; ja = ja + 1 == slabs_a ? 0 : ja + 1; }      // a slab is re-read rep_a times (3x3 taps)
        jb = jb + 1 == slabs_b ? 0 : jb + 1;
        st = st + 1 == NST ? 0 : st + 1;
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue();
    for (int s = 0; s < steps; ++s) {
        issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * (NA + NB)) : "memory");
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = reinterpret_cast<unsigned*>(smem)[lane];
}

template <int NA, int NB, int DEPTH, bool BARRIER>
static void run(const char* name, const unsigned char* a, const unsigned char* b, int pa, int pb, int slabs_a, int slabs_b,
                int rep_a, long long a_bytes, long long b_bytes, unsigned* sink, int blocks)
{
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<NA, NB, DEPTH, BARRIER>), dim3(blocks), dim3(512), 0, 0, a, b, pa, pb, slabs_a, slabs_b, rep_a, steps,
                           256, a_bytes, b_bytes, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)blocks * steps * (NA + NB) * 8 * 1024;
    printf("%-44s A%d B%d depth %d %s | %7.3f us/step | %6.1f GB/s per CU | %6.2f TB/s chip\n", name, NA, NB, DEPTH,
           BARRIER ? "barrier" : "free   ", best * 1e3 / steps, bytes / blocks / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int blocks = 256;
    const long long a_bytes = 256ll << 20, b_bytes = 16ll << 20;
    unsigned char *a, *b;
    unsigned* sink;
    hipMalloc(&a, a_bytes); hipMalloc(&b, b_bytes); hipMalloc(&sink, 4096);
    hipMemset(a, 1, a_bytes); hipMemset(b, 2, b_bytes);
    // layer3 3x3 (K = 256 bf16): A pitch 512 B, 4 slabs; B pitch 9 taps x 512 B, 36 slabs
    printf("-- layer3 3x3 bf16 shapes: A rows 512 B apart (4 slabs, re-read), B rows 4608 B apart (36 slabs)\n");
    run<4, 4, 1, true>("A+B one step in flight", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<4, 4, 2, true>("A+B two steps in flight", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<4, 4, 2, false>("A+B two steps in flight", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 1, true>("A only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 2, true>("A only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 4, true>("A only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<8, 0, 2, true>("A only (64 KiB steps)", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<0, 4, 1, true>("B only (every block the same rows)", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<0, 4, 2, true>("B only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<0, 4, 4, true>("B only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    run<0, 4, 4, false>("B only", a, b, 512, 4608, 4, 36, 9, a_bytes, b_bytes, sink, blocks);
    printf("-- layer3 1x1 1024->256: A rows 2048 B apart (16 slabs, streamed once per block), B rows 2048 B apart\n");
    run<4, 4, 1, true>("A+B", a, b, 2048, 2048, 16, 16, 1, a_bytes, b_bytes, sink, blocks);
    run<4, 4, 2, true>("A+B", a, b, 2048, 2048, 16, 16, 1, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 2, true>("A only", a, b, 2048, 2048, 16, 16, 1, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 4, true>("A only", a, b, 2048, 2048, 16, 16, 1, a_bytes, b_bytes, sink, blocks);
    printf("-- contiguous rows (pitch 128 B: the tile is one 32 KiB run)\n");
    run<4, 0, 2, true>("A only, private contiguous", a, b, 128, 128, 1, 1, 1, a_bytes, b_bytes, sink, blocks);
    run<4, 0, 4, true>("A only, private contiguous", a, b, 128, 128, 1, 1, 1, a_bytes, b_bytes, sink, blocks);
    run<0, 4, 4, true>("B only, shared contiguous", a, b, 128, 128, 1, 1, 1, a_bytes, b_bytes, sink, blocks);
    return 0;
}
