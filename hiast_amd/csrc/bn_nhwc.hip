// K10b — BatchNorm2d (+ residual add) (+ ReLU), training mode, on CHANNELS-LAST bf16 activations [M = B*H*W][C]
// (reference: Bottleneck.forward, sseg/models/modules/resnet.py:78-98; "frozen" BN still normalises with BATCH
// statistics in train() mode, utils/utils.py:60-65).  Same arithmetic and the same four passes as the NCHW kernels
// of bn_act.hip; this layout is the one the hand-written convolution kernels (igemm.hip) produce and consume, so
// the student forward / backward needs no layout transposes.  gfx950, HBM-bound: every pass moves whole 16-byte
// channel groups, a thread keeps ONE group of 8 channels for all its rows (its scale / shift / statistics live in
// registers), rows are dealt block-cyclically.
//
//   stats      : Σx, Σx² per channel                   -> sums[C][2] (double)           (1 read)
//   apply      : y = relu(x*scale_c + shift_c (+ res))                                   (1 read (+1), 1 write)
//   bwd_stats  : g = dy * (y > 0);  Σg, Σ g*xhat       -> sums[C][2]   (the gate y > 0 is recomputed from x when the
//                forward had no residual input: relu = 2, y is not read)
//   bwd_apply  : dx = gamma*invstd*(g - Σg/n - xhat*Σ(g*xhat)/n);  dres = g
// Partial sums: fp32 per thread (<= a few hundred rows), then double, reduced in a fixed order (no atomics: bitwise
// reproducible).  Between stats and apply the caller may all-reduce sums[C][2] across ranks (SyncBN).
#include <hip/hip_bf16.h>

#include "common.h"

namespace hiast {

constexpr int BNH_MAXBLK = 512;
// Launch geometry of the elementwise passes (round 2, measured per shape with rocprofv3 on tools/ab_bn.py against torch's
// copy / add kernels on the same tensors): rows per thread whose loads are all issued before the first is used, rows a
// block covers per pass, block cap.  What mattered was the block prologue, not the loads in flight: the per-channel
// parameters were 32 scalar loads behind branches on the nullable gamma / beta, waited for BEFORE the first row was
// requested — every block began with two serial memory round trips.  Rows first, parameters as 16-byte loads behind
// them: 256-channel maps 23-32 -> 12-18 us, 1024-channel maps 60-111 -> 42-78 us (torch add: 69 us).
#ifndef BNH_ROWS_PER_BLOCK_PASS
#define BNH_ROWS_PER_BLOCK_PASS 16
#endif
#ifndef BNH_APPLY_MAXBLK
#define BNH_APPLY_MAXBLK 65535
#endif
// How a block walks the rows.  BNH_CONTIG = 1: ONE contiguous range of rows per block (the blocks of a launch, in dispatch
// order, move one front through each tensor); 0: grid-stride (a block returns to the tensor gridDim.x chunks further on:
// several fronts per tensor at any moment).  lim = end of this block's rows.
#ifndef BNH_CONTIG
#define BNH_CONTIG 0      // measured: the contiguous walk is no faster (96.0 vs 94.0 us on the 1024-channel apply + residual)
#endif
#if BNH_CONTIG
#define BNH_ROW_WALK(CHUNK)                                                                                              \
    const long long step = (CHUNK);                                                                                      \
    const long long per_blk = ((M + gridDim.x - 1) / gridDim.x + (CHUNK) - 1) / (CHUNK) * (CHUNK);                       \
    const long long lim = (long long)(blockIdx.x + 1) * per_blk < M ? (long long)(blockIdx.x + 1) * per_blk : M;         \
    long long r0 = (long long)blockIdx.x * per_blk + rsub;
#else
#define BNH_ROW_WALK(CHUNK)                                                                                              \
    const long long step = (long long)gridDim.x * (CHUNK);                                                               \
    const long long lim = M;                                                                                             \
    long long r0 = (long long)blockIdx.x * (CHUNK) + rsub;
#endif
#ifndef BNH_UNR_PART
#define BNH_UNR_PART 8
#endif
#ifndef BNH_UNR_APPLY
#define BNH_UNR_APPLY 4
#endif
#ifndef BNH_UNR_BWD
#define BNH_UNR_BWD 4
#endif

// F16 (every kernel below): the activations are IEEE fp16 rows (HIAST_FMT_FP16) instead of bf16; statistics, scale / shift
// and the arithmetic are fp32 either way
template <bool F16>
__device__ __forceinline__ void bnh_unpack8(const uint4 r, float (&v)[8])
{
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = H16<F16>::lo(w[i]);
        v[2 * i + 1] = H16<F16>::hi(w[i]);
    }
}

// 16-byte row segment of a streamed activation as a NON-TEMPORAL load: the tensors are 33-134 MB, read once per pass and next
// touched by another kernel much later — read with the default policy they push the lines the NEXT kernel wants (this pass's
// output, the weights) out of the L2 / Infinity Cache.  BNH_NT_LOADS: bit 0 = the statistics passes (bnh_partial), bit 1 = the
// forward apply pass, bit 2 = the backward apply pass (A/B builds: tools/build_variant.sh ... -DBNH_NT_LOADS=<mask>).  Round 6,
// in the bench step on one box, two runs each (profiles/r06_ab_nt_loads.txt): mask 0 54.67 / 54.67 ms, 1: 54.67 / 54.63, 5: 54.50 /
// 54.59, 7: 54.32 / 54.30 — stand-alone the apply passes measure 3-11 us SLOWER with it (tools/ab_bn.py): the gain is the next
// kernel's.  The same policy on the residual rows of xconv / xconv2 / xconv_bs (+0.26 ms) and on the once-read A operand of the
// 1x1 tile-kernel launches (+0.07 ms) lost: those stay on the default policy.
#ifndef BNH_NT_LOADS
#define BNH_NT_LOADS 7
#endif
template <int WHICH>
__device__ __forceinline__ uint4 bnh_ld16(const unsigned short* p)
{
    if ((BNH_NT_LOADS >> WHICH) & 1) {
        const h_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const h_u32x4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}

template <bool F16>
__device__ __forceinline__ void bnh_load8(const unsigned short* p, float (&v)[8])
{
    bnh_unpack8<F16>(*reinterpret_cast<const uint4*>(p), v);
}

// 8 consecutive per-channel floats as two 16-byte loads; p == nullptr -> the constant dflt (selected, not branched on)
__device__ __forceinline__ void bnh_param8(const float* p, const float* some_valid, int c0, float dflt, float (&v)[8])
{
    const float* q = (p ? p : some_valid) + c0;
    const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
    const bool has = p != nullptr;
    v[0] = has ? a.x : dflt; v[1] = has ? a.y : dflt; v[2] = has ? a.z : dflt; v[3] = has ? a.w : dflt;
    v[4] = has ? b.x : dflt; v[5] = has ? b.y : dflt; v[6] = has ? b.z : dflt; v[7] = has ? b.w : dflt;
}

// nt: streaming store (tensors from H_NT_MIN_BYTES on: common.h)
template <bool F16>
__device__ __forceinline__ void bnh_store8(unsigned short* p, const float (&v)[8], bool nt = false)
{
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = H16<F16>::pack(v[2 * i], v[2 * i + 1]);
    h_store16(p, w[0], w[1], w[2], w[3], nt);
}

// Thread t of a 256-thread block owns channel group cg = t % G (G = C/8, a power of two <= 256) and rows
// r0 + (t / G) + k * RPP, RPP = 256 / G rows per pass.
// BWD = false: (Σx, Σx²) of x;  BWD = true: (Σg, Σ g*xhat) with g = dy * (y > 0 | all).
// GATE (ReLU gate of the backward passes): 0 = none, 1 = y > 0 (y is read), 2 = recomputed from x as
// x*scale + shift > 0 — valid when the forward had no residual input, and saves reading y in both backward passes,
// 3 = the bit mask the forward wrote ([M][C/8] bytes, passed in place of y): 1/16 of y's bytes.
template <bool BWD, int GATE, bool F16 = false>
__global__ __launch_bounds__(256) void bnh_partial_kernel(const unsigned short* __restrict__ a,    // x | dy
                                                          const unsigned short* __restrict__ y,
                                                          const unsigned short* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, long long M, int C,
                                                          float* __restrict__ partial)          // [nblk][C][2]
{
    __shared__ float s_red[256 * 16];
    const int G = C >> 3, RPP = 256 / G;
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    float s1[8], s2[8], mean[8], invstd[8], gsc[8], gsh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; mean[k] = 0.f; invstd[k] = 0.f; gsc[k] = 0.f; gsh[k] = 0.f; }
    // UNR rows per thread in flight (all loads of a chunk are issued before the first is used); a block walks chunks of
    // UNR * RPP consecutive rows.  The rows of the FIRST chunk are requested before the per-channel parameters: both
    // round trips then overlap (with the parameters first, every block started with one exposed latency).  The rows are
    // accumulated in ascending order within a thread.
    constexpr int UNR = BNH_UNR_PART;
    const long long stride = RPP;
    BNH_ROW_WALK(RPP * UNR)

    uint4 ra[UNR], ry[UNR], rx[UNR];
    unsigned rbits[UNR];
    auto load_rows = [&](long long rb) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + u * stride;
            const long long rc = r < lim ? r : (rb < lim ? rb : 0);     // tail: re-read a valid row, not accumulated
            const size_t off = (size_t)rc * C + cg * 8;
            ra[u] = bnh_ld16<0>(a + off);
            if (BWD) {
                if (GATE == 1) ry[u] = bnh_ld16<0>(y + off);
                rbits[u] = GATE == 3 ? reinterpret_cast<const unsigned char*>(y)[(size_t)rc * G + cg] : 0u;
                rx[u] = bnh_ld16<0>(x + off);
            }
        }
    };
    load_rows(r0);
    if (BWD) {
        float gm[8], bt[8];
        bnh_param8(save_mean, save_mean, cg * 8, 0.f, mean);
        bnh_param8(save_invstd, save_mean, cg * 8, 0.f, invstd);
        if (GATE == 2) {
            bnh_param8(gamma, save_mean, cg * 8, 1.0f, gm);
            bnh_param8(beta, save_mean, cg * 8, 0.0f, bt);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                gsc[k] = gm[k] * invstd[k];
                gsh[k] = fmaf(-mean[k], gsc[k], bt[k]);
            }
        }
    }
    while (r0 < lim) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (r0 + u * stride >= lim) break;
            float v[8];
            bnh_unpack8<F16>(ra[u], v);
            if (!BWD) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1[k] += v[k]; s2[k] = fmaf(v[k], v[k], s2[k]); }
            } else {
                float yy[8], xx[8];
                if (GATE == 1) bnh_unpack8<F16>(ry[u], yy);
                const unsigned bits = rbits[u];
                bnh_unpack8<F16>(rx[u], xx);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool open = GATE == 0 || (GATE == 1 ? yy[k] > 0.f
                                                    : (GATE == 3 ? ((bits >> k) & 1u) != 0u : fmaf(xx[k], gsc[k], gsh[k]) > 0.f));
                    const float g = open ? v[k] : 0.f;
                    s1[k] += g;
                    s2[k] = fmaf(g, (xx[k] - mean[k]) * invstd[k], s2[k]);
                }
            }
        }
        r0 += step;
        if (r0 < lim) load_rows(r0);
    }
    // fold the RPP row-subsets of each channel group (fixed order)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_red[threadIdx.x * 16 + k] = s1[k];
        s_red[threadIdx.x * 16 + 8 + k] = s2[k];
    }
    __syncthreads();
    if (rsub == 0) {
        float* dst = partial + ((size_t)blockIdx.x * C + cg * 8) * 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t1 = 0.f, t2 = 0.f;
            for (int j = 0; j < RPP; ++j) {
                t1 += s_red[(j * G + cg) * 16 + k];
                t2 += s_red[(j * G + cg) * 16 + 8 + k];
            }
            dst[2 * k] = t1;
            dst[2 * k + 1] = t2;
        }
    }
}

// sums[i] = Σ_blk partial[blk][i] (i over C*2) in double, fixed order: a block owns 16 consecutive entries, its
// 16 thread rows take blk = j, j+16, ... and are folded in ascending j.
__global__ __launch_bounds__(256) void bnh_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                           double* __restrict__ sums)
{
    __shared__ double s[16][17];
    const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;                   // C*2 is a multiple of 16
    double acc = 0.0;
    {   // sixteen rows in flight (a latency chain: with four the launch took 12 us for 2 MB), added in ascending order
        int b = j;
        for (; b + 16 * 15 < nblk; b += 16 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(b + 16 * u) * C * 2 + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += (double)v[u];
        }
        for (; b < nblk; b += 16) acc += (double)partial[(size_t)b * C * 2 + i];
    }
    s[j][e] = acc;
    __syncthreads();
    if (j == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s[q][e];
        sums[i] = t;
    }
}

// mean / invstd of the batch (double arithmetic, once per channel — not once per thread) + running statistics
__global__ __launch_bounds__(256) void bnh_prep_fwd_kernel(const double* __restrict__ sums, double count, float momentum,
                                                           float eps, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / count;
    double var = sums[2 * c + 1] / count - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)m;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (run_mean) {           // torch: running = (1-m)*running + m*batch; unbiased variance
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// finalize + prep in one launch (single-rank forward: no all-reduce between them): a block owns 2 channels = 4
// consecutive (Σx, Σx²) entries and folds the partial rows with 64 row-lanes per entry (rows j, j + 64, ... per lane, then
// the lanes in ascending order: a fixed order).  The 16-entry / 16-lane form of round 1 left a 256-channel layer with 32
// blocks of 16 dependent loads each: 11 us for a launch that moves 0.5 MB, 103 times per step.
__global__ __launch_bounds__(256) void bnh_finalize_prep_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                double count, float momentum, float eps,
                                                                float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd)
{
    __shared__ double s[64][5];
    const int e = threadIdx.x & 3, j = threadIdx.x >> 2;
    const int i = blockIdx.x * 4 + e;
    // the two threads that finish a channel ask for its running statistics now, not after the reduction
    const int c = blockIdx.x * 2 + (int)threadIdx.x;
    float rm = 0.f, rv = 0.f;
    if (threadIdx.x < 2 && run_mean) { rm = run_mean[c]; rv = run_var[c]; }
    double acc = 0.0;
    {
        int b = j;
        for (; b + 64 * 3 < nblk; b += 64 * 4) {        // four rows in flight per lane
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = partial[(size_t)(b + 64 * u) * C * 2 + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (double)v[u];
        }
        for (; b < nblk; b += 64) acc += (double)partial[(size_t)b * C * 2 + i];
    }
    s[j][e] = acc;
    __syncthreads();
    if (threadIdx.x < 2) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
        for (int q = 0; q < 64; ++q) { s1 += s[q][2 * threadIdx.x]; s2 += s[q][2 * threadIdx.x + 1]; }
        const double m = s1 / count;
        double var = s2 / count - m * m;
        var = var < 0.0 ? 0.0 : var;
        const float mean = (float)m;
        save_mean[c] = mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            run_mean[c] = (1.0f - momentum) * rm + momentum * mean;
            run_var[c] = (1.0f - momentum) * rv + momentum * (float)unb;
        }
    }
}

template <bool RES, bool RELU, bool F16 = false>
__global__ __launch_bounds__(256) void bnh_apply_kernel(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ res,
                                                        unsigned short* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd, long long M, int C,
                                                        unsigned char* __restrict__ mask)   // optional [M][C/8]: y > 0 bits
{
    const int G = C >> 3, RPP = 256 / G;
    const bool nt_out = (long long)M * C * 2 >= H_NT_MIN_BYTES;      // wave-uniform: stream large outputs past the L2
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    // rows of the first chunk are requested before the per-channel parameters (see bnh_partial_kernel)
    constexpr int UNR = BNH_UNR_APPLY;
    const long long stride = RPP;
    BNH_ROW_WALK(RPP * UNR)

    uint4 rx[UNR], rres[UNR];
    auto load_rows = [&](long long rb) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + u * stride;
            const size_t off = (size_t)(r < lim ? r : (rb < lim ? rb : 0)) * C + cg * 8;
            rx[u] = bnh_ld16<1>(x + off);
            if (RES) rres[u] = bnh_ld16<1>(res + off);
        }
    };
    load_rows(r0);
    float scale[8], shift[8];
    {
        float gm[8], bt[8], mn[8], is[8];
        bnh_param8(gamma, save_mean, cg * 8, 1.0f, gm);
        bnh_param8(beta, save_mean, cg * 8, 0.0f, bt);
        bnh_param8(save_mean, save_mean, cg * 8, 0.f, mn);
        bnh_param8(save_invstd, save_mean, cg * 8, 0.f, is);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            scale[k] = gm[k] * is[k];
            shift[k] = fmaf(-mn[k], scale[k], bt[k]);
        }
    }
    while (r0 < lim) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = r0 + u * stride;
            if (r >= lim) break;
            const size_t off = (size_t)r * C + cg * 8;
            float v[8], rr[8];
            bnh_unpack8<F16>(rx[u], v);
            if (RES) bnh_unpack8<F16>(rres[u], rr);
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float o = fmaf(v[k], scale[k], shift[k]);
                if (RES) o += rr[k];
                if (RELU) o = o > 0.f ? o : 0.f;
                bits |= (o > 0.f ? 1u : 0u) << k;
                v[k] = o;
            }
            bnh_store8<F16>(y + off, v, nt_out);
            if (mask) mask[(size_t)r * G + cg] = (unsigned char)bits;
        }
        r0 += step;
        if (r0 < lim) load_rows(r0);
    }
}

template <int GATE, bool DRES, bool F16 = false>
__global__ __launch_bounds__(256) void bnh_bwd_apply_kernel(
    const unsigned short* __restrict__ dy, const unsigned short* __restrict__ y, const unsigned short* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd,
    const double* __restrict__ sums, double inv_count, unsigned short* __restrict__ dx,
    unsigned short* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta, long long M, int C)
{
    const int G = C >> 3, RPP = 256 / G;
    const bool nt_out = (long long)M * C * 2 >= H_NT_MIN_BYTES;      // wave-uniform: stream large outputs past the L2
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    // rows of the first chunk are requested before the per-channel parameters (see bnh_partial_kernel)
    constexpr int UNR = BNH_UNR_BWD;
    const long long stride = RPP;
    BNH_ROW_WALK(RPP * UNR)

    uint4 rg[UNR], ry[UNR], rx[UNR];
    unsigned rbits[UNR];
    auto load_rows = [&](long long rb) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + u * stride;
            const long long rc = r < lim ? r : (rb < lim ? rb : 0);
            const size_t off = (size_t)rc * C + cg * 8;
            rg[u] = bnh_ld16<2>(dy + off);
            if (GATE == 1) ry[u] = bnh_ld16<2>(y + off);
            rbits[u] = GATE == 3 ? reinterpret_cast<const unsigned char*>(y)[(size_t)rc * G + cg] : 0u;
            rx[u] = bnh_ld16<2>(x + off);
        }
    };
    load_rows(r0);
    float mean[8], invstd[8], k0[8], mg[8], mgx[8], gsh[8];
    {
        float gm[8], bt[8];
        bnh_param8(save_mean, save_mean, cg * 8, 0.f, mean);
        bnh_param8(save_invstd, save_mean, cg * 8, 0.f, invstd);
        bnh_param8(gamma, save_mean, cg * 8, 1.0f, gm);
        bnh_param8(beta, save_mean, cg * 8, 0.0f, bt);
        const double2* sp = reinterpret_cast<const double2*>(sums) + cg * 8;       // (Σg, Σ g*xhat) per channel
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 ss = sp[k];
            if (blockIdx.x == 0 && rsub == 0) {
                if (dbeta) dbeta[cg * 8 + k] = (float)ss.x;
                if (dgamma) dgamma[cg * 8 + k] = (float)ss.y;
            }
            k0[k] = gm[k] * invstd[k];
            gsh[k] = GATE == 2 ? fmaf(-mean[k], k0[k], bt[k]) : 0.f;
            mg[k] = (float)(ss.x * inv_count);
            mgx[k] = (float)(ss.y * inv_count);
        }
    }
    while (r0 < lim) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = r0 + u * stride;
            if (r >= lim) break;
            const size_t off = (size_t)r * C + cg * 8;
            float g[8], yy[8], xx[8];
            bnh_unpack8<F16>(rg[u], g);
            if (GATE == 1) bnh_unpack8<F16>(ry[u], yy);
            const unsigned bits = rbits[u];
            bnh_unpack8<F16>(rx[u], xx);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool open = GATE == 0 || (GATE == 1 ? yy[k] > 0.f
                                                : (GATE == 3 ? ((bits >> k) & 1u) != 0u : fmaf(xx[k], k0[k], gsh[k]) > 0.f));
                const float gg = open ? g[k] : 0.f;
                g[k] = gg;
                xx[k] = k0[k] * (gg - mg[k] - (xx[k] - mean[k]) * invstd[k] * mgx[k]);
            }
            bnh_store8<F16>(dx + off, xx, nt_out);
            if (DRES) bnh_store8<F16>(dres + off, g, nt_out);
        }
        r0 += step;
        if (r0 < lim) load_rows(r0);
    }
}

static int bnh_nblk(long long M, int C)
{
    const int rpp = 256 / (C / 8);
    long long nb = (M + (long long)rpp * 16 - 1) / ((long long)rpp * 16);      // >= 16 passes per block
    nb = nb < 1 ? 1 : (nb > BNH_MAXBLK ? BNH_MAXBLK : nb);
    return (int)nb;
}

}  // namespace hiast

static int bnh_check(const void* x, long long M, int C)
{
    if (!x) return HIAST_E_ARG;
    if (M <= 0 || C <= 0) return HIAST_E_ARG;
    if (C < 8 || C > 2048 || (C & (C - 1)) != 0 || (((uintptr_t)x) & 15)) return HIAST_E_RANGE;
    return 0;
}

extern "C" size_t hiast_bn_nhwc_workspace_bytes(int C) { return (size_t)hiast::BNH_MAXBLK * C * 2 * sizeof(float); }

#define BNH_FMT_CHECK()                                                             \
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;          \
    const bool f16 = fmt == HIAST_FMT_FP16

extern "C" int hiast_bn_nhwc_stats(const void* x, int64_t M, int C, double* sums, void* workspace,
                                   size_t workspace_bytes, int fmt, hiast_stream_t stream)
{
    BNH_FMT_CHECK();
    int e = bnh_check(x, M, C);
    if (e) return e;
    if (!sums || !workspace) return HIAST_E_ARG;
    const int nblk = hiast::bnh_nblk(M, C);
    if (workspace_bytes < (size_t)nblk * C * 2 * sizeof(float)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    if (f16)
        hipLaunchKernelGGL((hiast::bnh_partial_kernel<false, 0, true>), dim3(nblk), dim3(256), 0, st, (const unsigned short*)x,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (long long)M, C, (float*)workspace);
    else
        hipLaunchKernelGGL((hiast::bnh_partial_kernel<false, 0>), dim3(nblk), dim3(256), 0, st, (const unsigned short*)x,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (long long)M, C, (float*)workspace);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::bnh_finalize_kernel, dim3(C * 2 / 16), dim3(256), 0, st, (const float*)workspace, nblk, C,
                       sums);
    HIAST_CHECK_LAUNCH();
    return 0;
}

// sums from per-block partials produced elsewhere (the statistics epilogue of hiast_igemm_bn_act): [nblk][C][2] fp32
extern "C" int hiast_bn_nhwc_stats_from_partial(const float* partial, int nblk, int C, double* sums,
                                                hiast_stream_t stream)
{
    if (!partial || !sums) return HIAST_E_ARG;
    if (nblk <= 0 || C <= 0) return HIAST_E_ARG;
    if (C % 8 != 0) return HIAST_E_RANGE;
    hipLaunchKernelGGL(hiast::bnh_finalize_kernel, dim3(C * 2 / 16), dim3(256), 0, (hipStream_t)stream, partial, nblk, C,
                       sums);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_bn_nhwc_apply(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, const double* sums, double count,
                                   float momentum, float eps, int relu, float* save_mean, float* save_invstd,
                                   int64_t M, int C, void* mask, int fmt, hiast_stream_t stream)
{
    BNH_FMT_CHECK();
    int e = bnh_check(x, M, C);
    if (e) return e;
    if (!y || !sums || !save_mean || !save_invstd || count <= 0) return HIAST_E_ARG;
    if ((((uintptr_t)y) | ((uintptr_t)res)) & 15) return HIAST_E_RANGE;
    const int rpp = 256 / (C / 8);
    long long nb = (M + (long long)rpp * BNH_ROWS_PER_BLOCK_PASS - 1) / ((long long)rpp * BNH_ROWS_PER_BLOCK_PASS);
    nb = nb < 1 ? 1 : (nb > BNH_APPLY_MAXBLK ? BNH_APPLY_MAXBLK : nb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hiast::bnh_prep_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, count, momentum, eps,
                       running_mean, running_var, save_mean, save_invstd, C);
    HIAST_CHECK_LAUNCH();
#define L1(RES, RELU, F)                                                                                           \
    hipLaunchKernelGGL((hiast::bnh_apply_kernel<RES, RELU, F>), dim3((unsigned)nb), dim3(256), 0, st,              \
                       (const unsigned short*)x, (const unsigned short*)res, (unsigned short*)y, gamma, beta,      \
                       save_mean, save_invstd, (long long)M, C, (unsigned char*)mask)
#define L(RES, RELU) do { if (f16) L1(RES, RELU, true); else L1(RES, RELU, false); } while (0)
    if (res) { if (relu) L(true, true); else L(true, false); }
    else { if (relu) L(false, true); else L(false, false); }
#undef L
#undef L1
    HIAST_CHECK_LAUNCH();
    return 0;
}

// single-rank forward fed by per-block partial sums (the statistics epilogue of hiast_igemm_bn_act): finalize + prep in
// one launch, then the elementwise pass
extern "C" int hiast_bn_nhwc_apply_partial(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, const float* partial, int nblk,
                                           double count, float momentum, float eps, int relu, float* save_mean,
                                           float* save_invstd, int64_t M, int C, void* mask, int fmt,
                                           hiast_stream_t stream)
{
    BNH_FMT_CHECK();
    int e = bnh_check(x, M, C);
    if (e) return e;
    if (!y || !partial || !save_mean || !save_invstd || count <= 0 || nblk <= 0) return HIAST_E_ARG;
    if ((((uintptr_t)y) | ((uintptr_t)res)) & 15) return HIAST_E_RANGE;
    const int rpp = 256 / (C / 8);
    long long nb = (M + (long long)rpp * BNH_ROWS_PER_BLOCK_PASS - 1) / ((long long)rpp * BNH_ROWS_PER_BLOCK_PASS);
    nb = nb < 1 ? 1 : (nb > BNH_APPLY_MAXBLK ? BNH_APPLY_MAXBLK : nb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hiast::bnh_finalize_prep_kernel, dim3(C / 2), dim3(256), 0, st, partial, nblk, C, count,
                       momentum, eps, running_mean, running_var, save_mean, save_invstd);
    HIAST_CHECK_LAUNCH();
#define L1(RES, RELU, F)                                                                                           \
    hipLaunchKernelGGL((hiast::bnh_apply_kernel<RES, RELU, F>), dim3((unsigned)nb), dim3(256), 0, st,              \
                       (const unsigned short*)x, (const unsigned short*)res, (unsigned short*)y, gamma, beta,      \
                       save_mean, save_invstd, (long long)M, C, (unsigned char*)mask)
#define L(RES, RELU) do { if (f16) L1(RES, RELU, true); else L1(RES, RELU, false); } while (0)
    if (res) { if (relu) L(true, true); else L(true, false); }
    else { if (relu) L(false, true); else L(false, false); }
#undef L
#undef L1
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_bn_nhwc_bwd_stats(const void* dy, const void* y, const void* x, const float* gamma,
                                       const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                       int64_t M, int C, double* sums, void* workspace, size_t workspace_bytes, int fmt,
                                       hiast_stream_t stream)
{
    BNH_FMT_CHECK();
    int e = bnh_check(x, M, C);
    if (e) return e;
    if (!dy || !save_mean || !save_invstd || !sums || !workspace || ((relu == 1 || relu == 3) && !y)) return HIAST_E_ARG;
    if (relu < 0 || relu > 3 || (((uintptr_t)dy) & 15) || (relu == 1 && (((uintptr_t)y) & 15))) return HIAST_E_RANGE;
    const int nblk = hiast::bnh_nblk(M, C);
    if (workspace_bytes < (size_t)nblk * C * 2 * sizeof(float)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
#define L1(G, F)                                                                                                    \
    hipLaunchKernelGGL((hiast::bnh_partial_kernel<true, G, F>), dim3(nblk), dim3(256), 0, st, (const unsigned short*)dy, \
                       (const unsigned short*)y, (const unsigned short*)x, gamma, beta, save_mean, save_invstd,      \
                       (long long)M, C, (float*)workspace)
#define L(G) do { if (f16) L1(G, true); else L1(G, false); } while (0)
    if (relu == 1) L(1); else if (relu == 2) L(2); else if (relu == 3) L(3); else L(0);
#undef L
#undef L1
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::bnh_finalize_kernel, dim3(C * 2 / 16), dim3(256), 0, st, (const float*)workspace, nblk, C,
                       sums);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_bn_nhwc_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma,
                                       const float* beta, const float* save_mean, const float* save_invstd,
                                       const double* sums,
                                       double count, int relu, void* dx, void* dres, float* dgamma, float* dbeta,
                                       int64_t M, int C, int fmt, hiast_stream_t stream)
{
    BNH_FMT_CHECK();
    int e = bnh_check(x, M, C);
    if (e) return e;
    if (!dy || !save_mean || !save_invstd || !sums || !dx || ((relu == 1 || relu == 3) && !y) || count <= 0)
        return HIAST_E_ARG;
    if (relu < 0 || relu > 3 || ((((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)dres)) & 15) ||
        (relu == 1 && (((uintptr_t)y) & 15)))
        return HIAST_E_RANGE;
    const int rpp = 256 / (C / 8);
    long long nb = (M + (long long)rpp * BNH_ROWS_PER_BLOCK_PASS - 1) / ((long long)rpp * BNH_ROWS_PER_BLOCK_PASS);
    nb = nb < 1 ? 1 : (nb > BNH_APPLY_MAXBLK ? BNH_APPLY_MAXBLK : nb);
    hipStream_t st = (hipStream_t)stream;
#define L1(G, DRES, F)                                                                                             \
    hipLaunchKernelGGL((hiast::bnh_bwd_apply_kernel<G, DRES, F>), dim3((unsigned)nb), dim3(256), 0, st,             \
                       (const unsigned short*)dy, (const unsigned short*)y, (const unsigned short*)x, gamma, beta,  \
                       save_mean, save_invstd, sums, 1.0 / count, (unsigned short*)dx, (unsigned short*)dres, dgamma, \
                       dbeta, (long long)M, C)
#define L(G, DRES) do { if (f16) L1(G, DRES, true); else L1(G, DRES, false); } while (0)
    if (relu == 1) { if (dres) L(1, true); else L(1, false); }
    else if (relu == 3) { if (dres) L(3, true); else L(3, false); }
    else if (relu == 2) { if (dres) L(2, true); else L(2, false); }
    else { if (dres) L(0, true); else L(0, false); }
#undef L
#undef L1
    HIAST_CHECK_LAUNCH();
    return 0;
}
