// Micro-check: HBM read+write rate of y[m][c] = r[m][c] + 1 on a [65536][1024] 16-bit map (the residual epilogue of
// the 256->1024 1x1 convolutions moves exactly these bytes) as a function of the CONTIGUOUS RUN one wave touches:
//   run = 128 B : lane l -> row 8*i + (l >> 3), 16 B chunk (l & 7) of a 64-channel column strip (igemm epilogue:
//                 8 rows x 128 B per wave instruction, row stride 2 KiB)
//   run = 512 B : 2 rows x 512 B per wave instruction (a 256-channel block tile written row-wise)
//   run = 1 KiB : 1 row x 1 KiB (streaming kernels: bn_nhwc.hip)
// Work per wave is the same in all variants (a 32-row x (run) tile walked 32 / rows-per-instruction times), blocks of
// 512 threads own 256 rows x 256 channels like an igemm block tile.
//     hipcc --offload-arch=gfx950 -O2 tools/micro/rw_granularity.hip -o tools/micro/rw_granularity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int M = 65536, C = 1024;

// RUN16 = 16-byte chunks per contiguous run (8 -> 128 B, 32 -> 512 B, 64 -> 1 KiB)
// ONE = hold 128 KiB of LDS so that only one 8-wave block is resident per CU, as for an igemm block
template <int RUN16, bool ONE>
__global__ __launch_bounds__(512) void rw_kernel(const uint4* __restrict__ r, uint4* __restrict__ y, int never)
{
    __shared__ unsigned char pad[ONE ? 131072 : 16];
    if (never) pad[threadIdx.x] = (unsigned char)never;          // keeps the allocation

    // block tile: 256 rows x 256 channels (= 32 chunks of 16 B per row); 1024 channels = 4 column tiles
    const int bm = blockIdx.x >> 2, bn = blockIdx.x & 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int ROWS_PER_INSTR = 64 / RUN16;          // rows one wave instruction touches
    constexpr int RUNS_PER_TILEROW = (RUN16 >= 32) ? 1 : 32 / RUN16;   // column strips of the 256-channel tile
    // the block's 256 x 32 chunks are split over 8 waves: wave -> (row group, column strip)
    const int strip = wave % RUNS_PER_TILEROW, rgrp = wave / RUNS_PER_TILEROW;
    constexpr int RGRPS = 8 / RUNS_PER_TILEROW;
    constexpr int ROWS_PER_WAVE = 256 / RGRPS;
    const int lrow = lane / RUN16, lch = lane % RUN16;
    const size_t row_chunks = C / 8;                    // 128 chunks of 16 B per row
    constexpr int ITER = RUN16 == 64 ? 16 : ROWS_PER_WAVE / ROWS_PER_INSTR;      // 16 instructions per wave in every variant
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        size_t idx;
        if (RUN16 == 64) {
            // 1 KiB run: a wave covers half a row of the WHOLE 1024-channel map; the four column-tile blocks of a row
            // block split its rows instead: block (bm, bn) takes rows bm*256 + bn*64 .. +64, wave 8 of them
            const int row = bm * 256 + bn * 64 + wave * 8 + (it & 7);
            idx = (size_t)row * row_chunks + (it >> 3) * 64 + lane;
        } else {
            const int row = bm * 256 + rgrp * ROWS_PER_WAVE + it * ROWS_PER_INSTR + lrow;
            idx = (size_t)row * row_chunks + bn * 32 + strip * RUN16 + lch;
        }
        uint4 v = r[idx];
        v.x += 1u; v.y += 1u; v.z += 1u; v.w += 1u;
        y[idx] = v;
    }
}

template <int RUN16, bool ONE>
static void run(const uint4* r, uint4* y, const char* name)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = (M / 256) * 4;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rw_kernel<RUN16, ONE>), dim3(blocks), dim3(512), 0, 0, r, y, 0);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int k = 0; k < reps; ++k) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((rw_kernel<RUN16, ONE>), dim3(blocks), dim3(512), 0, 0, r, y, 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double bytes = 2.0 * M * C * 2;
    printf("%-28s avg %.3f ms  best %.3f ms  -> %.2f TB/s (best %.2f)\n", name, sum / reps, best, bytes / (sum / reps) / 1e9,
           bytes / best / 1e9);
}

int main()
{
    uint4 *r, *y;
    const size_t bytes = (size_t)M * C * 2;
    CHECK(hipMalloc(&r, bytes));
    CHECK(hipMalloc(&y, bytes));
    CHECK(hipMemset(r, 1, bytes));
    CHECK(hipMemset(y, 0, bytes));
    // 1 KiB variant covers: rows_per_wave = 256/8 = 32 instr... make sure every element is touched exactly once
    run<4, false>(r, y, "run 64 B (16 rows/instr: the xconv epilogue)");
    run<4, true>(r, y, "run 64 B, 1 block/CU");
    run<8, false>(r, y, "run 128 B (8 rows/instr)");
    run<32, false>(r, y, "run 512 B (2 rows/instr)");
    run<64, false>(r, y, "run 1 KiB (1 row/instr)");
    run<8, true>(r, y, "run 128 B, 1 block/CU");
    run<32, true>(r, y, "run 512 B, 1 block/CU");
    run<64, true>(r, y, "run 1 KiB, 1 block/CU");
    // verification of coverage for each variant: y == r + 1 everywhere
    unsigned* h = (unsigned*)malloc(bytes);
    int bad_total = 0;
    for (int v = 0; v < 3; ++v) {
        CHECK(hipMemset(y, 0, bytes));
        const int blocks = (M / 256) * 4;
        if (v == 0) hipLaunchKernelGGL((rw_kernel<8, true>), dim3(blocks), dim3(512), 0, 0, r, y, 0);
        if (v == 1) hipLaunchKernelGGL((rw_kernel<32, true>), dim3(blocks), dim3(512), 0, 0, r, y, 0);
        if (v == 2) hipLaunchKernelGGL((rw_kernel<64, true>), dim3(blocks), dim3(512), 0, 0, r, y, 0);
        CHECK(hipMemcpy(h, y, bytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < bytes / 4; ++i) bad += h[i] != 0x01010102u;
        printf("variant %d: %zu words not written exactly once\n", v, bad);
        bad_total += bad != 0;
    }
    return bad_total;
}
