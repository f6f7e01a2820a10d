// micro-bench: LDS-DMA fill rate with the access shape of wgrad.hip (K9d) — no LDS reads, no MFMA.
// One 512-thread block per CU.  A k-step of a block is ROWS pixel rows of two operands: 256 B-wide column windows of dY
// (row pitch 2N bytes) and of X (row pitch 2K bytes); a wave-instruction moves 4 rows x 256 B (PIECE4) or 8 rows x 128 B.
// `share` blocks (consecutive logical ids, dealt to ONE XCD as wgrad does) walk the SAME pixel range with different
// column windows / taps, i.e. they re-read each other's lines; stream = 1: the range advances by ROWS per step (every
// line is new to the L2 once per group), stream = 0: the same ROWS rows every step (L2-resident after the first step).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_fill_wgrad.hip -o tools/micro/dma_fill_wgrad && tools/micro/dma_fill_wgrad
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds, int voff, int soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 rsrc(const void* base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)a; r[1] = (int)((a >> 32) & 0xFFFFu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}

// ROWS pixels per step; NST stages; DIST = steps between issue and need (<= NST - 1)
template <int ROWS, int NST, int DIST, bool PIECE4>
__global__ __launch_bounds__(512) void fill_kernel(const unsigned char* dY, const unsigned char* X, int N2, int K2, int ncolY, int ncolX,
                                                   int share, int steps, int stream, long long bytesY, long long bytesX, unsigned* sink, int taps, int dilW)
{
    constexpr int STAGE = ROWS * 1024;                       // 4 sub-tiles of [ROWS][256 B]
    constexpr int PCS = ROWS / 8;                            // pieces per wave and step
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lid = blockIdx.x;
    {
        const int total = gridDim.x, base = total >> 3, rem = total & 7;
        const int xcd = lid & 7, slot = lid >> 3;
        lid = xcd * base + (xcd < rem ? xcd : rem) + slot;
    }
    const int member = lid % share, group = lid / share;
    const int sub = wave >> 1;
    const bool is_x = sub >= 2;
    const int pitch = is_x ? K2 : N2;
    // column window of this member (different members: different windows, wrapped)
    // as in wgrad: member = (n tile, k tile, tap), tap fastest; a 3x3 tap shifts the X rows by (ty - 1) * dilW + (tx - 1) * dil
    const int tap = member % taps, tile = member / taps;
    const int col = is_x ? ((tile % ncolX) * 512 + (sub - 2) * 256) : (((tile / ncolX) % ncolY) * 512 + sub * 256);
    const int shift = (is_x && taps == 9) ? (tap / 3) * dilW + (tap % 3) * (dilW >> 7) : 0;     // (>= 0: the window starts one tap row down)
    const i32x4 rs = rsrc(is_x ? X : dY, (unsigned)(is_x ? bytesX : bytesY));
    const int rows_per_group = stream ? steps * ROWS : ROWS;
    const long long m0 = (long long)group * (rows_per_group + (taps == 9 ? 2 * dilW + 64 : 0)) + shift;
    int voff[PCS];
#pragma unroll
    for (int pc = 0; pc < PCS; ++pc) {
        int row, chunk;
        if (PIECE4) { row = (ROWS / 2) * (wave & 1) + 4 * pc + (lane >> 4); chunk = lane & 15; }
        else { row = (ROWS / 2) * (wave & 1) + 8 * (pc >> 1) + (lane >> 3); chunk = (lane & 7) + 8 * (pc & 1); }
        voff[pc] = (int)((m0 + row) * pitch + col + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))));
    }
    const unsigned lds0 = (unsigned)(size_t)smem + (unsigned)(sub * ROWS * 256 + (wave & 1) * PCS * 1024);
    auto issue = [&](int s) {
        const unsigned dst = lds0 + (unsigned)((s % NST) * STAGE);
        const int soff = stream ? s * ROWS * pitch : 0;
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) dma16(rs, dst + pc * 1024, voff[pc], soff);
    };
#pragma unroll
    for (int d = 0; d < DIST; ++d) issue(d);
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PCS * (DIST - 1)) : "memory");
        issue(s + DIST < steps ? s + DIST : steps - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = reinterpret_cast<unsigned*>(smem)[lane];
}

template <int ROWS, int NST, int DIST, bool PIECE4>
static void run(const char* name, const unsigned char* dY, const unsigned char* X, int N, int K, int share, int stream, int steps,
                long long bytesY, long long bytesX, unsigned* sink, int taps = 1, int dilW = 0, int blocks = 256)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    const int groups = (blocks + share - 1) / share;
    if ((long long)groups * (steps * ROWS + 2 * dilW + 64) * N * 2 > bytesY || (long long)groups * (steps * ROWS + 2 * dilW + 64) * K * 2 > bytesX) { printf("%s: buffer too small\n", name); return; }
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<ROWS, NST, DIST, PIECE4>), dim3(blocks), dim3(512), 0, 0, dY, X, 2 * N, 2 * K, N / 256, K / 256, share, steps,
                           stream, bytesY, bytesX, sink, taps, dilW);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)blocks * steps * ROWS * 1024;
    printf("%-34s N=%4d K=%4d share %2d %s rows %2d x %d stages dist %d %s | %7.3f us/step | %6.1f GB/s per CU | %6.2f TB/s chip | unique %.2f TB/s\n",
           name, N, K, share, stream ? "stream  " : "resident", ROWS, NST, DIST, PIECE4 ? "4x256B" : "8x128B", best * 1e3 / steps,
           bytes / blocks / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12,
           stream ? (double)groups * steps * ROWS * (N + K) * 2 / (best * 1e-3) / 1e12 : 0.0);
    fflush(stdout);
}

int main()
{
    const long long bytesY = 1ll << 30, bytesX = 1ll << 30;      // (descriptor sizes are 32-bit: 1 GiB each)
    unsigned char *dY, *X;
    unsigned* sink;
    hipMalloc(&dY, bytesY); hipMalloc(&X, bytesX); hipMalloc(&sink, 4096);
    hipMemset(dY, 1, bytesY); hipMemset(X, 2, bytesX);
    printf("-- layer4 3x3 (N = K = 512, 36 blocks share a pixel range)\n");
    run<64, 2, 1, true>("two stages of 64 (round 2)", dY, X, 512, 512, 36, 1, 256, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 512, 512, 36, 1, 512, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 512, 512, 36, 0, 512, bytesY, bytesX, sink);
    run<32, 4, 3, false>("ring of four", dY, X, 512, 512, 36, 1, 512, bytesY, bytesX, sink);
    run<32, 4, 3, false>("ring of four", dY, X, 512, 512, 36, 0, 512, bytesY, bytesX, sink);
    printf("-- with the row shifts of the nine taps (dilation 4, W = 128: 512 rows per tap row)\n");
    run<32, 4, 3, true>("ring of four, 3x3 taps", dY, X, 512, 512, 36, 1, 293, bytesY, bytesX, sink, 9, 512);
    run<32, 4, 3, true>("ring of four, 3x3 taps, 252 blocks", dY, X, 512, 512, 36, 1, 293, bytesY, bytesX, sink, 9, 512, 252);
    run<32, 4, 3, true>("ring of four, 3x3 taps dil 2", dY, X, 256, 256, 9, 1, 74, bytesY, bytesX, sink, 9, 256);
    run<32, 4, 3, true>("ring of four, no shift, same steps", dY, X, 512, 512, 36, 1, 293, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four, no shift, same steps", dY, X, 256, 256, 9, 1, 74, bytesY, bytesX, sink);
    printf("-- the same with fewer sharers\n");
    run<32, 4, 3, true>("ring of four", dY, X, 512, 512, 4, 1, 512, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 512, 512, 1, 1, 128, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 512, 512, 1, 0, 512, bytesY, bytesX, sink);
    printf("-- layer3 3x3 (N = K = 256, 9 blocks share)\n");
    run<32, 4, 3, true>("ring of four", dY, X, 256, 256, 9, 1, 512, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 256, 256, 9, 0, 512, bytesY, bytesX, sink);
    printf("-- layer3 1x1 (N = 1024, K = 256, 4 blocks share)\n");
    run<32, 4, 3, true>("ring of four", dY, X, 1024, 256, 4, 1, 64, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 1024, 256, 4, 0, 64, bytesY, bytesX, sink);
    printf("-- layer4 1x1 (N = 2048, K = 512, 16 blocks share)\n");
    run<32, 4, 3, true>("ring of four", dY, X, 2048, 512, 16, 1, 128, bytesY, bytesX, sink);
    run<32, 4, 3, true>("ring of four", dY, X, 2048, 512, 16, 0, 128, bytesY, bytesX, sink);
    return 0;
}
