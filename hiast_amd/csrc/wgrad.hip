// K9d — weight gradient of the trunk convolutions (1x1 and 3x3, dilated / strided) on channels-last bf16
// activations (reference: autograd of nn.Conv2d in Bottleneck.forward, sseg/models/modules/resnet.py:78-98, under
// apex O1: half-precision operands, fp32 accumulation):
//
//     dW[n][tap][k] = Σ_m dY[m][n] * X[m + off(tap)][k]          m over all B*Ho*Wo output pixels
//
// A GEMM that reduces over the PIXEL index: both operands are pixel-major ([M][N] and [M][K] rows), the opposite
// of what an MFMA fragment wants (8 consecutive reduction elements per lane).  The tiles are therefore written into
// LDS exactly as they lie in memory — by LDS-DMA, 4 rows x 256 B per wave-instruction, zero rows for out-of-image
// taps via the buffer out-of-range rule — and every fragment is built by two transposing ds_read_b64_tr_b16 reads
// (swizzle (b) of the CDNA4 guide on 256-byte rows: conflict free for the transposed reads).
// Block = one (256 n) x (256 k) x tap tile of dW over one pixel range; 8 waves (wave tile 128 x 64); k-step = 32
// pixels; a ring of four LDS stages (the DMA of step t+3 flies during the MFMAs of step t).  Pixel ranges are reduced in a fixed
// order by wgrad_reduce_kernel (bitwise reproducible, no float atomics), which also writes torch's [N][K][kh][kw]
// layout.
// The k-step (second form of round 2): the LDS-DMA is issued through inline asm and the transposing reads stay builtins —
// the compiler then folds their offsets into the instructions, keeps the two halves of a fragment in one register tuple
// and counts lgkmcnt itself, where the first form (asm reads, builtin DMA) spent ~600 VALU / branch instructions per
// k-step on addresses, joins and bounds tests for 32 MFMAs.  Out-of-range rows come from the descriptor's size, per-piece
// constants ride in the scalar offset, the 3x3 pixel walk is branch-free, fragments are double-buffered across the four
// sub-steps (182 / 209 VGPRs).  layer3 3x3 0.124 -> 0.105 ms, layer4 3x3 0.408 -> 0.319 ms (with the reduce pass).
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace hiast {

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float wg_f32x16;
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* wg_lds_ptr;

struct WGeo {
    int H, W, Ho, Wo, stride, dil;
};

typedef int wg_i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA as INLINE ASM (64 lanes x 16 bytes from buffer offset voff + soff to LDS address lds + 16*lane).  The compiler
// orders every LDS load it can see behind ALL pending LDS-DMAs it knows of (s_waitcnt vmcnt(0)); round 2 first hid the
// loads (asm ds_read_b64_tr_b16), which left it with ~100 register copies per k-step joining the two halves of every
// fragment and 48 address computations.  Hiding the DMA instead keeps the transposing reads as builtins: the compiler
// folds their tile offsets into the instruction, allocates the halves of a fragment as one register tuple and counts
// lgkmcnt itself; the DMA waits are counted by hand (s_waitcnt vmcnt below).
__device__ __forceinline__ void wg_dma16(wg_i32x4 rs, unsigned lds, int voff, int soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ wg_i32x4 wg_rsrc(const void* base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    wg_i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((a >> 32) & 0xFFFFu);
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ wg_s16x4 wg_tr_read(const unsigned char* p)
{
    typedef wg_s16x4 __attribute__((address_space(3))) * lds_p;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p);
}
__device__ __forceinline__ wg_bf16x8 wg_join(const wg_s16x4& v0, const wg_s16x4& v1)
{
    return __builtin_bit_cast(wg_bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// [rows][128 x 16-bit] sub-tile with 256-byte rows: 16-byte chunk ch of row r lives at chunk ch ^ swz(r)
__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int wg_off(int row, int ch) { return 256 * row + 16 * (ch ^ wg_swz(row)); }

constexpr int WG_ROWS = 32;                       // pixels per k-step
constexpr int WG_SUB = WG_ROWS * 256;             // one [32][128] sub-tile: 8 KiB
constexpr int WG_STAGE = 4 * WG_SUB;              // dY (2 sub-tiles) + X (2 sub-tiles): 32 KiB
constexpr int WG_NST = 4;                         // stages in the LDS ring (128 KiB)
constexpr int WG_PCS = WG_ROWS / 8;               // DMA pieces (4 rows x 256 B) per wave and k-step

// F16: dY and X rows are IEEE fp16 (HIAST_FMT_FP16) instead of bf16; the kernel never decodes a value, only the MFMA differs
// S1 (3x3 only): stride 1 and Wo >= 20 — the tap shift is a constant number of pixels and rides in the buffer descriptor
// XCD-aware block order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (private L2
// each).  The logical order is pixel range OUTER / (n tile, k tile, tap) INNER, and XCD k takes the k-th contiguous
// share of it: the blocks that re-read one pixel range's dY and X rows (all taps, all tiles) sit on ONE XCD at the
// same time, so those rows leave HBM once instead of once per XCD they were dealt to.
__device__ __forceinline__ int wg_logical_block(int lid, int total)
{
    const int base = total >> 3, rem = total & 7;
    const int xcd = lid & 7, slot = lid >> 3;
    return xcd * base + (xcd < rem ? xcd : rem) + slot;
}

// One block = one (256 n) x (256 k) x tap tile `t` of dW over the pixel range of `split`: partial tile -> P[split][n][tap][k]
template <int TAPS, bool F16, bool S1>
__device__ __forceinline__ void wgrad_tn_body(unsigned char* smem, const unsigned short* __restrict__ dY,
                                              const unsigned short* __restrict__ X, float* __restrict__ P, int M, int N,
                                              int K, const WGeo& geo, int m_per_split, int t, int split)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // wave tile: n rows [128*wm, +128), k cols [64*wn, +64)
    const int kt_tiles = K / 256;
    const int tap = t % TAPS; t /= TAPS;                  // t: (n tile, k tile, tap)
    const int k0 = (t % kt_tiles) * 256, n0 = (t / kt_tiles) * 256;
    const int m_begin = split * m_per_split;              // a multiple of 32
    const int m_end = (m_begin + m_per_split < M) ? m_begin + m_per_split : M;
    const int nk = (m_end - m_begin + WG_ROWS - 1) / WG_ROWS;
    const int oy = TAPS == 1 ? 0 : (tap / 3 - 1) * geo.dil, ox = TAPS == 1 ? 0 : (tap % 3 - 1) * geo.dil;

    constexpr int OOB = (int)0x80000000;
    const float inv_hw = 1.0f / (float)(geo.Ho * geo.Wo), inv_wo = 1.0f / (float)geo.Wo;
    const size_t in_pix = (TAPS == 1) ? (size_t)M : (size_t)(M / (geo.Ho * geo.Wo)) * geo.H * geo.W;
    // Buffer descriptors that END at this block's last pixel row: the tail rows of the last k-step are out of range and
    // arrive as zeros without a per-lane test (3x3: X is addressed per input pixel and keeps its own bounds select).
    const wg_i32x4 yrs = wg_rsrc(dY, (unsigned)((size_t)m_end * N * 2));
    // S1: input pixel = output pixel + oy * W + ox.  The descriptor starts that many pixels into X (before X for the upper
    // taps: lanes that would read there are out of the image and never sent) and ends with X, so a fetch is addressed by
    // the OUTPUT pixel with the per-lane constants and scalar offsets of the dY rows; per lane only the in-image test is left.
    const long long tap_bytes = S1 ? (long long)(oy * geo.W + ox) * K * 2 : 0;
    const wg_i32x4 xrs = wg_rsrc((const unsigned char*)X + tap_bytes,
                                 (unsigned)((long long)((TAPS == 1 ? (size_t)m_end : in_pix) * K * 2) - tap_bytes));

    // DMA pieces of one k-step: 32 wave-instructions (4 sub-tiles x 8 groups of 4 rows).  Wave w owns sub-tile w >> 1
    // (0, 1: dY columns n0.. / n0+128..; 2, 3: X columns k0.. / k0+128..) and its row groups 4*(w & 1) + pc, pc = 0..3:
    // row = 16*(w & 1) + 4*pc + (lane >> 4), logical chunk (lane & 15) ^ swz(row) with swz(row) = ((lane>>4 & 3) << 2) |
    // (pc & 3).  Everything that depends on pc or on the k-step but not on the lane goes into the instruction's scalar
    // offset: four per-lane offsets (pc & 3) serve all pieces.
    const int sub = wave >> 1, l4 = lane >> 4;
    const bool is_x = sub >= 2;
    const int pitch2 = (is_x ? K : N) * 2;                                    // bytes per pixel row of this wave's operand
    const int col2 = (is_x ? k0 + (sub - 2) * 128 : n0 + sub * 128) * 2;
    const int row0 = (WG_ROWS / 2) * (wave & 1) + l4;
    int pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        pv[j] = (m_begin + row0) * pitch2 + col2 + 16 * ((lane & 15) ^ (((l4 & 3) << 2) | j));
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned dst0 = lds0 + (unsigned)(sub * WG_SUB + (wave & 1) * WG_PCS * 1024);
    // 3x3: the input pixel of this lane's row is WALKED, not decoded: (xx, yy) = input column / row incl. the tap offset and
    // lin = its linear pixel index move 4 output pixels per piece and 20 behind the last piece of a step (the next step
    // starts 32 on), with one row wrap and one image wrap per move (Wo >= 20; narrower maps decode every step).  All
    // increments are scalars; the only multiply left is one 24-bit mad per piece.  (The first form decoded (image, row,
    // column) per step and multiplied out the address per piece: five quarter-rate integer multiplies and ~40 VALU
    // instructions per piece, 0.3 us of VALU time per 32-pixel step on the waves that fetch X — the DMA-only time of
    // the layer4 3x3 was 207 us against 116 us for the same requests issued without the arithmetic.)
    const int st_ = geo.stride;
    const int xx_end = geo.Wo * st_ + ox, yy_end = geo.Ho * st_ + oy;          // first (xx, yy) behind a row / an image
    const int wo_s = geo.Wo * st_, ho_s = geo.Ho * st_;
    const int row_jump = st_ * geo.W - wo_s, img_jump = geo.H * geo.W - ho_s * geo.W;
    const bool walk = geo.Wo >= 20;
    int cx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cx[j] = col2 + 16 * ((lane & 15) ^ (((l4 & 3) << 2) | j));
    auto decode = [&](int m, int& xx, int& yy, int& lin) {
        // float reciprocal + one-step correction: exact for m < 2^24, no integer division
        const int hw = geo.Ho * geo.Wo;
        int img = (int)(((float)m + 0.5f) * inv_hw);
        int r = m - img * hw;
        const bool lo = r < 0, hi = r >= hw;
        img += hi ? 1 : (lo ? -1 : 0);
        r += lo ? hw : (hi ? -hw : 0);
        int yo = (int)(((float)r + 0.5f) * inv_wo);
        int xo = r - yo * geo.Wo;
        const bool lo2 = xo < 0, hi2 = xo >= geo.Wo;
        yo += hi2 ? 1 : (lo2 ? -1 : 0);
        xo += lo2 ? geo.Wo : (hi2 ? -geo.Wo : 0);
        yy = yo * st_ + oy;
        xx = xo * st_ + ox;
        lin = (img * geo.H + yy) * geo.W + xx;
    };
    auto advance = [&](int n, int& xx, int& yy, int& lin) {       // n output pixels on, n <= Wo
        xx += n * st_;
        lin += n * st_;
        const bool w = xx >= xx_end;
        xx -= w ? wo_s : 0;
        yy += w ? st_ : 0;
        lin += w ? row_jump : 0;
        const bool w2 = yy >= yy_end;
        yy -= w2 ? ho_s : 0;
        lin += w2 ? img_jump : 0;
    };
    auto issue = [&](int kt, int buf, int pc, bool on, int& xx, int& yy, int& lin) {
#ifdef WG_ABL_NODMA       // diagnostic build (tools/build_variant.sh): the loop without its LDS-DMA
        return;
#endif
        const unsigned dst = dst0 + (unsigned)(buf * WG_STAGE + pc * 1024);
        if (TAPS == 1 || !is_x) {
            wg_dma16(is_x ? xrs : yrs, dst, on ? pv[pc & 3] : OOB, (kt * WG_ROWS + 4 * pc) * pitch2);
        } else if (S1) {
            const bool ok = on & ((unsigned)yy < (unsigned)geo.H) & ((unsigned)xx < (unsigned)geo.W);
            wg_dma16(xrs, dst, ok ? pv[pc & 3] : OOB, (kt * WG_ROWS + 4 * pc) * pitch2);
            const int n = (pc == WG_PCS - 1) ? WG_ROWS - 4 * (WG_PCS - 1) : 4;
            xx += n;
            const bool w = xx >= xx_end;
            xx -= w ? wo_s : 0;
            yy += w ? 1 : 0;
            yy -= (yy >= yy_end) ? ho_s : 0;
        } else {
            if (pc == 0 && !walk) decode(m_begin + kt * WG_ROWS + row0, xx, yy, lin);
            const int m = m_begin + row0 + (kt * WG_ROWS + 4 * pc);
            const bool ok = on & (m < m_end) & ((unsigned)yy < (unsigned)geo.H) & ((unsigned)xx < (unsigned)geo.W);
            const int voff = __mul24(lin, pitch2) + cx[pc & 3];
            wg_dma16(xrs, dst, ok ? voff : OOB, 0);
            advance((pc == WG_PCS - 1 && walk) ? WG_ROWS - 4 * (WG_PCS - 1) : 4, xx, yy, lin);
        }
    };

    wg_f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed fragment: 32 columns starting at c0 of a sub-tile, reduction rows kb .. kb+7 for this lane half.
    // Per-lane byte offsets of the two halves (v0: rows kb + fq, v1: rows kb + 4 + fq) for kk = 0; sub-step kk adds
    // 16 rows = 4096 bytes (the swizzle of a row does not depend on kk); a stage holds two sub-steps.
    const int grp4 = lane >> 4, t16 = lane & 15;
    const int fq = t16 >> 2, fp = t16 & 3;
    auto frag_off = [&](int c0, int half) {
        const int kb = 8 * (grp4 >> 1) + 4 * half;
        const int ch = ((c0 + 16 * (grp4 & 1)) >> 3) + (fp >> 1);
        return wg_off(kb + fq, ch) + 8 * (fp & 1);
    };
    int oa[4][2], ob[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h) oa[a][h] = wm * WG_SUB + frag_off(a * 32, h);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) ob[b][h] = 2 * WG_SUB + (wn >> 1) * WG_SUB + frag_off((wn & 1) * 64 + b * 32, h);

    // Ring of WG_NST stages of 32 pixels.  The barrier at the top of step t says "everyone's pieces of step t + 1 have
    // landed and everyone has left stage t - 1": the DMA of step t + 3 goes into the stage just left, a piece has two to
    // three steps (>= 2 us) to arrive, and the first fragments of step t + 1 are read during the last MFMAs of step t — a
    // wave comes out of the barrier with its operands in registers.  (First form: two stages of 64 pixels, the next stage
    // requested during the current one, s_waitcnt vmcnt(0) + __syncthreads() per step, fragments read behind the barrier.)
    // Waits are counted: at the top of step t only the 4 pieces of step t + 2 may still be in flight.
    int w_xx = 0, w_yy = 0, w_lin = 0;                         // 3x3: (xx, yy, lin) of the walk
    if (TAPS != 1 && is_x) decode(m_begin + row0, w_xx, w_yy, w_lin);
#pragma unroll
    for (int s0 = 0; s0 < WG_NST - 1; ++s0)
#pragma unroll
        for (int pc = 0; pc < WG_PCS; ++pc) issue(s0, s0, pc, s0 < nk, w_xx, w_yy, w_lin);
    wg_bf16x8 fa[2][4], fb[2][2];                                 // fragments of the sub-steps (even | odd set)
    auto read_set = [&](int set, const unsigned char* st, int kk) {
#ifdef WG_ABL_NOREAD      // diagnostic build: MFMAs on whatever the registers hold
        return;
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a)
            fa[set][a] = wg_join(wg_tr_read(st + oa[a][0] + kk * 4096), wg_tr_read(st + oa[a][1] + kk * 4096));
#pragma unroll
        for (int b = 0; b < 2; ++b)
            fb[set][b] = wg_join(wg_tr_read(st + ob[b][0] + kk * 4096), wg_tr_read(st + ob[b][1] + kk * 4096));
    };
    auto mfma_set = [&](int set) {
#ifdef WG_ABL_NOMFMA      // diagnostic build: the fragments are read (kept alive) but not multiplied
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(fa[set][a]));
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(fb[set][b]));
        return;
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = H16<F16>::mfma32(fa[set][a], fb[set][b], acc[a][b]);
    };
    static_assert(WG_ROWS == 32 && WG_NST == 4 && WG_PCS == 4, "the loop below is written for two sub-steps and four stages");
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WG_PCS * 2) : "memory");          // step 0 is there
    read_set(0, smem, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WG_PCS) : "memory");
        const bool more = kt + 3 < nk;
        const int nbuf = (kt + 3) & 3;
        const unsigned char* st = smem + (kt & 3) * WG_STAGE;
        const unsigned char* stn = smem + ((kt + 1) & 3) * WG_STAGE;
        // two DMA pieces of step kt + 3, the fragments of the next sub-step, this sub-step's MFMAs: reads and DMA issue ride
        // in the shadow of the MFMAs
        issue(kt + 3, nbuf, 0, more, w_xx, w_yy, w_lin);
        issue(kt + 3, nbuf, 1, more, w_xx, w_yy, w_lin);
        read_set(1, st, 1);
        mfma_set(0);
        issue(kt + 3, nbuf, 2, more, w_xx, w_yy, w_lin);
        issue(kt + 3, nbuf, 3, more, w_xx, w_yy, w_lin);
        read_set(0, stn, 0);                                      // (behind the last step: a stage nobody needs)
        mfma_set(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the tail's out-of-range pieces)

    // partial P[split][n][tap][k]
    float* Ps = P + (size_t)split * N * TAPS * K;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = k0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                h_store_f32_out(Ps + ((size_t)n * TAPS + tap) * K + k, acc[a][b][r]);      // (plain stores: non-temporal ones measured 3 % slower here)
            }
        }
}

template <int TAPS, bool F16 = false, bool S1 = false>
__global__ __launch_bounds__(512) void wgrad_tn_kernel(const unsigned short* __restrict__ dY,
                                                       const unsigned short* __restrict__ X, float* __restrict__ P,
                                                       int M, int N, int K, WGeo geo, int m_per_split)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_NST * WG_STAGE];
    const int lid = wg_logical_block(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    wgrad_tn_body<TAPS, F16, S1>(smem, dY, X, P, M, N, K, geo, m_per_split, lid % (int)gridDim.x, lid / (int)gridDim.x);
}

// ---- grouped launch: the weight gradients of SEVERAL convolutions (the three of one bottleneck) in one grid ----------
// A weight-gradient launch fills the chip with one 256 x 256 tile per CU, whatever the layer: a layer3 1x1 has 4 tiles, so
// 64 pixel ranges each write a 256 KiB fp32 partial tile (64 MB written and read again by the reduction, against 168 MB of
// operands), a layer3 3x3 has 9 tiles x 28 ranges (64 MB of partials against 67 MB of operands).  The three convolutions
// of a bottleneck TOGETHER have 17 tiles: 15 pixel ranges x 17 tiles fill the chip once with 64 MB of partials for all
// three (was 192 MB), every block runs 137 k-steps instead of 32 ... 73, and the HBM-bound 1x1 tiles run beside the
// L2-fed 3x3 tiles on the same XCD.  (The operands of the three products exist at different times of a backward pass;
// the host defers the launch to the end of the block's backward: functional._WGroupFn.)
constexpr int WG_MAXJOBS = 4;
struct WJob {
    const unsigned short* dY;
    const unsigned short* X;
    float* P;               // partials of this job: [nsplit][N][taps][K]
    float* dw;              // result, torch layout [N][K][kh][kw]
    int M, N, K, taps, s1, m_per_split, nsplit, f4_begin;
    WGeo geo;
};
// map[logical block] = job << 14 | tile << 8 | pixel range.  The jobs of a launch may split the pixel index differently
// (WG_MAXBLOCKS blocks in all): a block of a 1x1 job streams its rows from HBM once and is bound by that stream, a block of a
// 3x3 job re-reads its rows nine times from L2 and is bound by the matrix pipes — with the SAME number of pixels per block the
// 1x1 blocks finish last.  The host gives the 1x1 jobs more, shorter pixel ranges (wgrad_group_plan) and orders the blocks by
// their position along the pixel index, so that the blocks that share rows (all tiles of one job and range) still sit on one
// XCD at the same time.
constexpr int WG_MAXBLOCKS = 256;
struct WGroup {
    WJob j[WG_MAXJOBS];
    int njobs, nblocks;
    unsigned short map[WG_MAXBLOCKS];
};

template <bool F16>
__global__ __launch_bounds__(512) void wgrad_group_kernel(const WGroup g)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_NST * WG_STAGE];
    const int lid = wg_logical_block(blockIdx.x, gridDim.x);
    const unsigned e = __builtin_amdgcn_readfirstlane((unsigned)g.map[lid]);
    const int ji = (int)(e >> 14), t = (int)((e >> 8) & 63u), split = (int)(e & 255u);
    const WJob& jb = g.j[ji];
    if (jb.taps == 1)
        wgrad_tn_body<1, F16, false>(smem, jb.dY, jb.X, jb.P, jb.M, jb.N, jb.K, jb.geo, jb.m_per_split, t, split);
    else if (jb.s1)
        wgrad_tn_body<9, F16, true>(smem, jb.dY, jb.X, jb.P, jb.M, jb.N, jb.K, jb.geo, jb.m_per_split, t, split);
    else
        wgrad_tn_body<9, F16, false>(smem, jb.dY, jb.X, jb.P, jb.M, jb.N, jb.K, jb.geo, jb.m_per_split, t, split);
}

// dW[n][k][tap] (torch [N][K][kh][kw]) = Σ_s P[s][n][tap][k], ascending s (fixed order: bitwise reproducible).
// thread = (n, tap, 4 consecutive k): the partials are read as they lie — 16 bytes per lane, eight splits in flight
// (the first form, one float per thread and four loads in flight, streamed the 67 MB of a layer3 1x1 at 1.8 TB/s:
// 38 us x 81 launches per training step) — only the small result is written with the tap stride.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ dw, int N,
                                                           int K, int taps, int nsplit)
{
    const long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;       // index of a float4 along [n][tap][k]
    const size_t per = (size_t)N * taps * K;
    if (idx4 * 4 >= (long long)per) return;
    const long long idx = idx4 * 4;
    const int k = (int)(idx % K);
    const long long nt = idx / K;
    const int t = (int)(nt % taps), n = (int)(nt / taps);
    const float4* p = reinterpret_cast<const float4*>(P + idx);
    const size_t per4 = per / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + u) * per4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {           // the additions keep their order
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
    }
    for (; s < nsplit; ++s) {
        const float4 v = p[(size_t)s * per4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = dw + ((size_t)n * K + k) * taps + t;
    if (taps == 1) {
        *reinterpret_cast<float4*>(d) = acc;
    } else {
        d[0] = acc.x; d[taps] = acc.y; d[2 * taps] = acc.z; d[3 * taps] = acc.w;
    }
}

// the same reduction for every job of a grouped launch in ONE grid (f4_begin: first float4 index of a job's [n][tap][k])
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const WGroup g)
{
    long long idx4 = (long long)blockIdx.x * 256 + threadIdx.x;
    int ji = 0;
#pragma unroll
    for (int q = 1; q < WG_MAXJOBS; ++q)
        if (q < g.njobs && idx4 >= g.j[q].f4_begin) ji = q;
    const WJob& jb = g.j[ji];
    idx4 -= jb.f4_begin;
    const int N = jb.N, K = jb.K, taps = jb.taps, nsplit = jb.nsplit;
    const size_t per = (size_t)N * taps * K;
    if (idx4 * 4 >= (long long)per) return;
    const long long idx = idx4 * 4;
    const int k = (int)(idx % K);
    const long long nt = idx / K;
    const int t = (int)(nt % taps), n = (int)(nt / taps);
    const float4* p = reinterpret_cast<const float4*>(jb.P + idx);
    const size_t per4 = per / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + u) * per4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {           // the additions keep their order
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
    }
    for (; s < nsplit; ++s) {
        const float4 v = p[(size_t)s * per4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = jb.dw + ((size_t)n * K + k) * taps + t;
    if (taps == 1) {
        *reinterpret_cast<float4*>(d) = acc;
    } else {
        d[0] = acc.x; d[taps] = acc.y; d[2 * taps] = acc.z; d[3 * taps] = acc.w;
    }
}

static int wgrad_nsplit(long long M, int tiles, int taps)
{
    // one block per CU at a time (128 KiB of LDS each) and every block costs a 256 KiB partial tile written here and read
    // again by the reduction: ONE round of <= 256 blocks (two rounds doubled the partial traffic — 129 MB against 67 MB
    // of operands on the layer3 3x3 — for the same MFMA time per CU); at least 8 k-steps (512 pixels) per block
    long long s = hiast_grid_cus() / tiles;
    const long long smax = M / 512 > 0 ? M / 512 : 1;
    s = s < 1 ? 1 : (s > smax ? smax : s);
    s = s > 64 ? 64 : s;
    return (int)s;
}

}  // namespace hiast

extern "C" size_t hiast_conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int taps)
{
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin % 256 != 0 || Cout % 256 != 0 || (taps != 1 && taps != 9)) return 0;
    const long long M = (long long)B * Ho * Wo;
    const int tiles = (Cout / 256) * (Cin / 256) * taps;
    return (size_t)hiast::wgrad_nsplit(M, tiles, taps) * Cout * taps * Cin * sizeof(float);
}

extern "C" int hiast_conv_wgrad_nhwc(const void* dy, const void* x, float* dw, int B, int H, int W, int Cin, int Cout,
                                     int taps, int stride, int dil, int fmt, void* workspace, size_t workspace_bytes,
                                     hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    const bool f16 = fmt == HIAST_FMT_FP16;
    if (!dy || !x || !dw || !workspace) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || stride <= 0 || dil <= 0) return HIAST_E_ARG;
    if (Cin % 256 != 0 || Cout % 256 != 0 || (taps != 1 && taps != 9) || (taps == 1 && stride != 1)) return HIAST_E_RANGE;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace)) & 15) return HIAST_E_RANGE;
    const int Ho = taps == 1 ? H : (H - 1) / stride + 1, Wo = taps == 1 ? W : (W - 1) / stride + 1;
    const long long M = (long long)B * Ho * Wo;
    if ((size_t)M * Cout * 2 >= (1ull << 31) || (size_t)B * H * W * Cin * 2 >= (1ull << 31) || M >= (1ll << 24))
        return HIAST_E_RANGE;
    // 3x3: the pixel walk moves 4 columns with one row wrap; its linear pixel index feeds a 24-bit multiply
    if (taps == 9 && (Wo < 4 || (long long)B * H * W + 64 >= (1ll << 23))) return HIAST_E_RANGE;
    const int tiles = (Cout / 256) * (Cin / 256) * taps;
    const int nsplit = hiast::wgrad_nsplit(M, tiles, taps);
    if (workspace_bytes < (size_t)nsplit * Cout * taps * Cin * sizeof(float)) return HIAST_E_WS;
    int mps = (int)((M + nsplit - 1) / nsplit);
    mps = ((mps + hiast::WG_ROWS - 1) / hiast::WG_ROWS) * hiast::WG_ROWS;
    hiast::WGeo geo = {H, W, Ho, Wo, stride, dil};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(tiles, nsplit);
#define WL(T, F, S)                                                                                             \
    hipLaunchKernelGGL((hiast::wgrad_tn_kernel<T, F, S>), grid, dim3(512), 0, st, (const unsigned short*)dy,     \
                       (const unsigned short*)x, (float*)workspace, (int)M, Cout, Cin, geo, mps)
    const bool s1 = stride == 1 && Wo >= 20;
    if (taps == 1) { if (f16) WL(1, true, false); else WL(1, false, false); }
    else if (s1) { if (f16) WL(9, true, true); else WL(9, false, true); }
    else { if (f16) WL(9, true, false); else WL(9, false, false); }
#undef WL
    HIAST_CHECK_LAUNCH();
    const long long total = (long long)Cout * Cin * taps / 4;           // float4 per thread (Cin % 256 == 0)
    hipLaunchKernelGGL(hiast::wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)workspace, dw, Cout, Cin, taps, nsplit);
    HIAST_CHECK_LAUNCH();
    return 0;
}

// ---- grouped launch (see wgrad_group_kernel) ---------------------------------------------------------------------------
static int wgrad_group_plan(const hiast_wgrad_job* jobs, int njobs, hiast::WGroup& g, size_t& ws_bytes)
{
    if (!jobs || njobs < 1 || njobs > hiast::WG_MAXJOBS) return HIAST_E_ARG;
    int tiles_of[hiast::WG_MAXJOBS];
    int tiles = 0, tiles1 = 0, tiles9 = 0;
    long long f4 = 0, mmin = -1;
    for (int i = 0; i < njobs; ++i) {
        const hiast_wgrad_job& q = jobs[i];
        if (q.B <= 0 || q.H <= 0 || q.W <= 0 || q.stride <= 0 || q.dil <= 0) return HIAST_E_ARG;
        if (q.Cin % 256 != 0 || q.Cout % 256 != 0 || (q.taps != 1 && q.taps != 9) || (q.taps == 1 && q.stride != 1))
            return HIAST_E_RANGE;
        const int Ho = q.taps == 1 ? q.H : (q.H - 1) / q.stride + 1, Wo = q.taps == 1 ? q.W : (q.W - 1) / q.stride + 1;
        const long long M = (long long)q.B * Ho * Wo;
        if ((size_t)M * q.Cout * 2 >= (1ull << 31) || (size_t)q.B * q.H * q.W * q.Cin * 2 >= (1ull << 31) || M >= (1ll << 24))
            return HIAST_E_RANGE;
        if (q.taps == 9 && (Wo < 4 || (long long)q.B * q.H * q.W + 64 >= (1ll << 23))) return HIAST_E_RANGE;
        hiast::WJob& j = g.j[i];
        j.dY = (const unsigned short*)q.dy;
        j.X = (const unsigned short*)q.x;
        j.dw = q.dw;
        j.M = (int)M; j.N = q.Cout; j.K = q.Cin; j.taps = q.taps;
        j.s1 = (q.taps == 9 && q.stride == 1 && Wo >= 20) ? 1 : 0;
        j.geo = {q.H, q.W, Ho, Wo, q.stride, q.dil};
        j.f4_begin = (int)f4;
        tiles_of[i] = (q.Cout / 256) * (q.Cin / 256) * q.taps;
        if (tiles_of[i] > 64) return HIAST_E_RANGE;          // (6 bits of the block map)
        tiles += tiles_of[i];
        (q.taps == 1 ? tiles1 : tiles9) += tiles_of[i];
        f4 += (long long)q.Cout * q.Cin * q.taps / 4;
        if (f4 >= (1ll << 31)) return HIAST_E_RANGE;
        mmin = (mmin < 0 || M < mmin) ? M : mmin;
    }
    if (tiles > hiast::WG_MAXBLOCKS || tiles > hiast_grid_cus()) return HIAST_E_RANGE;
    // ONE round of <= 256 blocks (see wgrad_nsplit), at least 16 k-steps per block.  Uniform split first; when the launch
    // mixes 1x1 and 3x3 jobs, the 1x1 jobs get `ratio` times shorter pixel ranges (their blocks take about that much longer
    // per pixel: HBM-streamed rows against L2-fed ones; HIAST_WGROUP_RATIO, percent, tuning): the pair (s1, s9) with
    // tiles1 * s1 + tiles9 * s9 <= 256 that minimises max(ratio / s1, 1 / s9)
    const long long smax = mmin / 512 > 0 ? (mmin / 512 > 64 ? 64 : mmin / 512) : 1;
    const int cus = hiast_grid_cus() < hiast::WG_MAXBLOCKS ? hiast_grid_cus() : hiast::WG_MAXBLOCKS;   // one round of blocks
    long long s1 = cus / tiles, s9;
    s1 = s1 < 1 ? 1 : (s1 > smax ? smax : s1);
    s9 = s1;
    static const int ratio_pct = [] { const char* e = getenv("HIAST_WGROUP_RATIO"); const int v = e ? atoi(e) : 100; return v >= 25 && v <= 400 ? v : 100; }();
    if (tiles1 > 0 && tiles9 > 0 && ratio_pct != 100) {
        double best = (double)ratio_pct / 100.0 / (double)s1;
        if (1.0 / (double)s9 > best) best = 1.0 / (double)s9;
        for (long long a = 1; a <= smax; ++a) {
            const long long left = cus - (long long)tiles1 * a;
            if (left < tiles9) break;
            long long b = left / tiles9;
            b = b > smax ? smax : b;
            const double c1 = (double)ratio_pct / 100.0 / (double)a, c9 = 1.0 / (double)b;
            const double c = c1 > c9 ? c1 : c9;
            if (c < best - 1e-12) { best = c; s1 = a; s9 = b; }
        }
    }
    g.njobs = njobs;
    size_t off = 0;
    int nblocks = 0;
    struct Ent { double key; int job, tile, split; };
    Ent ents[hiast::WG_MAXBLOCKS];
    for (int i = 0; i < njobs; ++i) {
        hiast::WJob& j = g.j[i];
        const long long ns = j.taps == 1 ? s1 : s9;
        j.nsplit = (int)ns;
        int mps = (int)((j.M + ns - 1) / ns);
        j.m_per_split = ((mps + hiast::WG_ROWS - 1) / hiast::WG_ROWS) * hiast::WG_ROWS;
        j.P = (float*)off;                               // offset for now; the launch adds the workspace base
        off += (size_t)ns * j.N * j.taps * j.K * sizeof(float);
        for (int sp = 0; sp < (int)ns; ++sp)
            for (int t = 0; t < tiles_of[i]; ++t) {
                if (nblocks >= hiast::WG_MAXBLOCKS) return HIAST_E_RANGE;
                ents[nblocks++] = {((double)sp + 0.5) / (double)ns, i, t, sp};
            }
    }
    // order by position along the pixel index (then job, pixel range, tile): a stable insertion sort of <= 256 entries
    for (int a = 1; a < nblocks; ++a) {
        const Ent e = ents[a];
        int b = a - 1;
        while (b >= 0 && (ents[b].key > e.key || (ents[b].key == e.key && ents[b].job > e.job))) { ents[b + 1] = ents[b]; --b; }
        ents[b + 1] = e;
    }
    for (int a = 0; a < nblocks; ++a) g.map[a] = (unsigned short)((ents[a].job << 14) | (ents[a].tile << 8) | ents[a].split);
    g.nblocks = nblocks;
    ws_bytes = off;
    return 0;
}

extern "C" size_t hiast_conv_wgrad_group_workspace_bytes(const hiast_wgrad_job* jobs, int njobs)
{
    hiast::WGroup g;
    size_t n = 0;
    return wgrad_group_plan(jobs, njobs, g, n) == 0 ? n : 0;
}

extern "C" int hiast_conv_wgrad_group_nhwc(const hiast_wgrad_job* jobs, int njobs, int fmt, void* workspace,
                                           size_t workspace_bytes, hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if (!workspace) return HIAST_E_ARG;
    hiast::WGroup g;
    size_t need = 0;
    const int e = wgrad_group_plan(jobs, njobs, g, need);
    if (e) return e;
    if (workspace_bytes < need) return HIAST_E_WS;
    uintptr_t al = (uintptr_t)workspace;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].dy || !jobs[i].x || !jobs[i].dw) return HIAST_E_ARG;
        al |= (uintptr_t)jobs[i].dy | (uintptr_t)jobs[i].x | (uintptr_t)jobs[i].dw;
        g.j[i].P = (float*)((unsigned char*)workspace + (size_t)g.j[i].P);
    }
    if (al & 15) return HIAST_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)g.nblocks;
    if (fmt == HIAST_FMT_FP16) hipLaunchKernelGGL(hiast::wgrad_group_kernel<true>, dim3(blocks), dim3(512), 0, st, g);
    else hipLaunchKernelGGL(hiast::wgrad_group_kernel<false>, dim3(blocks), dim3(512), 0, st, g);
    HIAST_CHECK_LAUNCH();
    const hiast::WJob& last = g.j[njobs - 1];
    const long long total4 = (long long)last.f4_begin + (long long)last.N * last.K * last.taps / 4;
    hipLaunchKernelGGL(hiast::wgrad_group_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, g);
    HIAST_CHECK_LAUNCH();
    return 0;
}
