// K3/K4 — pseudo-label kernels (workflows/pseudo_label_generator.py:192-201 and :67-105 of the
// reference).  gfx950 only.
//
// pass 1: fused bilinear upsample (align_corners) + softmax + max/argmax + per-class fp16-bin
//         confidence histogram.  Full-resolution logits (39.8 MB/img) never exist: each thread
//         owns one output column X of one "band" (the 8-9 output rows that share a pair of
//         low-res source rows), keeps the horizontally-lerped top/bottom row values of all C
//         classes in registers (2*C VGPRs) and walks down the band.
//         Histogram: wave-aggregated — lanes holding the same (class, bin) key elect a leader
//         that issues ONE integer atomic of the popcount.
// pass 2: threshold select + per-image class counts + exact integer Σ prob.
#include "common.h"

namespace hiast {

// Scattered histogram layout of the pass-1 workspace.  On real maps a few classes and the confidences 0.5 .. 1 take nearly
// all pixels: in the natural [class][bin] layout their ~2000 counters are ~60 lines of 128 B, and integer atomics on
// one line execute one after the other at the memory side (tools/micro/int_atomics.hip: 3.1 M adds inside 60 lines 0.42 ms,
// over >= 240 lines 0.12 ms = the chip's rate of 26 G atomics/s).  The workspace stores counter (class c, bin) at word
// ((bin & 255) * C + c) * 61 + (bin >> 8): neighbouring bins are 256 planes apart, the hot counters lie on several hundred
// lines.  hist_merge_kernel adds the workspace into the caller's [class][bin] histogram.
constexpr int HIST_PLANE = (HIAST_NBINS + 255) / 256;                 // 61 words per (low byte, class)
template <int C>
__device__ __forceinline__ unsigned hist_slot(unsigned key)
{
    const unsigned c = key / HIAST_NBINS, bin = key - c * HIAST_NBINS;
    return ((bin & 255u) * C + c) * HIST_PLANE + (bin >> 8);
}
template <int C>
__global__ __launch_bounds__(256) void hist_merge_kernel(const uint32_t* __restrict__ ws, uint32_t* __restrict__ hist)
{
    const unsigned key = blockIdx.x * 256u + threadIdx.x;
    if (key >= (unsigned)C * HIAST_NBINS) return;
    const unsigned v = ws[hist_slot<C>(key)];
    if (v) hist[key] += v;                       // (one thread per counter; calls on one stream are ordered)
}

// One work item of pass 1: 256 output columns (tile xt) of one band j of image b; thread t owns one column.
// MODE 0: counts straight into hist [C][NBINS]; 1: into the scattered workspace; both count the top TOPB bins in LDS
// (s_cnt = unsigned [C * TOPB], one block = one item).  MODE 2: all bins >= 0.5 (fp16 0x3800 .. 0x3C00) are counted in LDS
// as packed 16-bit pairs (s_cnt = unsigned [C * UPW]) by a persistent block that works through many items before it
// flushes; the rest goes to the scattered workspace.
constexpr int TOPB = 128;
constexpr int UPLO = 0x3800;                                          // fp16(0.5)
constexpr int UPW = (HIAST_NBINS - UPLO + 1) / 2;                     // 513 words per class
template <int C, int MODE>
__device__ __forceinline__ void pass1_item(const float* __restrict__ logits, int h, int w, int H, int W, float sh, float sw,
                                           float* __restrict__ maxprob, uint8_t* __restrict__ argmax,
                                           uint32_t* __restrict__ hist, unsigned* s_cnt, int b, int j, int xt, int t)
{
    const int X = xt * 256 + t;
    const int Y0 = band_start(sh, j, h, H);
    const int Y1 = band_start(sh, j + 1, h, H);
    if (Y0 >= Y1) return;                           // uniform over the item (nothing is counted)
    const bool live = X < W;
    const int Xc = live ? X : W - 1;

    const Src sx = src_of(sw, Xc, w);
    const int y0 = j, y1 = j + (j < h - 1 ? 1 : 0);
    const float* base = logits + (size_t)b * C * h * w;

    float top[C], bot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* p = base + (size_t)c * h * w;
        float a = p[y0 * w + sx.i0], bb = p[y0 * w + sx.i1];
        float cc = p[y1 * w + sx.i0], d = p[y1 * w + sx.i1];
        top[c] = lerp_h(a, bb, sx.l0, sx.l1);
        bot[c] = lerp_h(cc, d, sx.l0, sx.l1);
    }

    for (int Y = Y0; Y < Y1; ++Y) {
        const Src sy = src_of(sh, Y, h);
        float z[C];
        float m = 0.0f;
        int am = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            z[c] = lerp_v(top[c], bot[c], sy.l0, sy.l1);
            if (c == 0 || z[c] > m) { m = z[c]; am = c; }   // first max wins ties
        }
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) s = s + a_expf(z[c] - m);
        const float prob = 1.0f / s;                         // IEEE division

        unsigned key = 0xFFFFFFFFu;
        if (live) {
            const size_t o = ((size_t)b * H + Y) * W + X;
            maxprob[o] = prob;
            argmax[o] = (uint8_t)am;
            const unsigned bin = __half_as_ushort(__float2half_rn(prob));
            if (MODE == 2) {
                if (bin >= (unsigned)UPLO && bin < HIAST_NBINS) {
                    const unsigned i = bin - UPLO;
                    atomicAdd(&s_cnt[am * UPW + (int)(i >> 1)], (i & 1u) ? 65536u : 1u);
                } else if (bin < HIAST_NBINS) key = (unsigned)am * HIAST_NBINS + bin;
            } else {
                if (bin >= HIAST_NBINS - TOPB && bin < HIAST_NBINS)
                    atomicAdd(&s_cnt[am * TOPB + (int)(bin - (HIAST_NBINS - TOPB))], 1u);
                else if (bin < HIAST_NBINS) key = (unsigned)am * HIAST_NBINS + bin;
            }
        }
        // Histogram update of the keys not counted in LDS.  Two wave-aggregated rounds take out the keys many lanes share
        // (flat regions: 64 same-address atomics would serialise), the lanes that are left add their own count: on
        // realistic confidences nearly every lane of a wave holds a different (class, fp16 bin) key, and the fully
        // aggregated loop then ran 64 ballot / shuffle rounds per wave row (1.85 ms per 8-image batch; the integer sums
        // are the same in any order).
        unsigned long long todo = __ballot(key != 0xFFFFFFFFu);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (!todo) break;
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned k = __shfl(key, leader, 64);
            const unsigned long long same = __ballot(key == k);
            if (lane_id() == leader) atomicAdd(&hist[MODE ? hist_slot<C>(k) : k], (unsigned)__popcll(same));
            todo &= ~same;
        }
        if ((todo >> lane_id()) & 1ull) atomicAdd(&hist[MODE ? hist_slot<C>(key) : key], 1u);
    }
}

template <int C, bool SCATTER>
__global__ __launch_bounds__(256) void plabel_pass1_kernel(
    const float* __restrict__ logits, int h, int w, int H, int W, float sh, float sw,
    float* __restrict__ maxprob, uint8_t* __restrict__ argmax, uint32_t* __restrict__ hist)
{
    // The TOP fp16 bins (confidence >= 1 - 128 * 2^-11 ~ 0.94) of every class are counted in LDS first and flushed once
    // per block: on confident predictions most pixels of the whole batch fall into a handful of (class, bin) counters
    // (0.99 ms per 8-image batch with every lane adding to global memory).
    __shared__ unsigned s_top[C * TOPB];
    for (int i = threadIdx.x; i < C * TOPB; i += 256) s_top[i] = 0u;
    __syncthreads();
    pass1_item<C, SCATTER ? 1 : 0>(logits, h, w, H, W, sh, sw, maxprob, argmax, hist, s_top, blockIdx.z, blockIdx.y, blockIdx.x,
                                   threadIdx.x);
    __syncthreads();
    for (int i = threadIdx.x; i < C * TOPB; i += 256) {
        const unsigned v = s_top[i];
        const unsigned k = (unsigned)(i / TOPB) * HIAST_NBINS + (HIAST_NBINS - TOPB) + (unsigned)(i % TOPB);
        if (v) atomicAdd(&hist[SCATTER ? hist_slot<C>(k) : k], v);
    }
}

// Persistent form (round 3): 1024 threads = four 256-column groups that share ONE LDS histogram of the bins >= 0.5 and work
// through `chunk` items each between flushes (host: 4 * chunk * pixels per item < 65536, the packed counters cannot
// overflow).  On real maps two or three classes and the confidences 0.5 .. 1 hold nearly all pixels — about 2000 hot
// counters; a 2300-pixel item has ~1 pixel per hot counter (aggregating per item saves nothing: 2.9 M global atomics per
// batch either way, in chains of ~1000 per address), a block that counts 18 000 pixels before it flushes sends each hot
// counter once: ~0.5 M atomics in chains of <= 256.
template <int C>
__global__ __launch_bounds__(1024) void plabel_pass1_persistent_kernel(
    const float* __restrict__ logits, int h, int w, int H, int W, float sh, float sw, float* __restrict__ maxprob,
    uint8_t* __restrict__ argmax, uint32_t* __restrict__ ws, int nx, int items, int chunk)
{
    __shared__ unsigned s_up[C * UPW];
    const int tid = threadIdx.x, sub = tid >> 8, t = tid & 255;
    for (int i = tid; i < C * UPW; i += 1024) s_up[i] = 0u;
    __syncthreads();
    const int groups = gridDim.x * 4, g = blockIdx.x * 4 + sub;
    const int per_group = (items + groups - 1) / groups;              // every group runs the same number of rounds
    for (int k0 = 0; k0 < per_group; k0 += chunk) {
        const int k1 = k0 + chunk < per_group ? k0 + chunk : per_group;
        for (int k = k0; k < k1; ++k) {
            const int it = k * groups + g;                            // x tile fastest, then band, then image
            if (it < items) {
                const int xt = it % nx, r = it / nx;
                pass1_item<C, 2>(logits, h, w, H, W, sh, sw, maxprob, argmax, ws, s_up, r / h, r % h, xt, t);
            }
        }
        __syncthreads();
        for (int i = tid; i < C * UPW; i += 1024) {
            const unsigned v = s_up[i];
            if (v) {
                const unsigned c = (unsigned)i / UPW, p = (unsigned)i % UPW;
                const unsigned k = c * HIAST_NBINS + UPLO + 2u * p;
                if (v & 0xFFFFu) atomicAdd(&ws[hist_slot<C>(k)], v & 0xFFFFu);
                if (v >> 16) atomicAdd(&ws[hist_slot<C>(k + 1u)], v >> 16);
                s_up[i] = 0u;
            }
        }
        __syncthreads();
    }
}

template <int C>
static int launch_pass1(const float* logits, int B, int h, int w, int H, int W, float* maxprob,
                        uint8_t* argmax, uint32_t* hist, uint32_t* ws, hipStream_t st)
{
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    dim3 grid((W + 255) / 256, h, B);
    // a band holds the output rows Y with floor(Y * (h-1)/(H-1)) == j: at most (H-1)/(h-1) + 2 of them; the persistent form
    // needs 4 * chunk * 256 * rows < 65536 (packed LDS counters)
    const int rows = h > 1 ? (H - 1) / (h - 1) + 2 : H;
    const int nx = (W + 255) / 256;
    const long long items = (long long)nx * h * B;
    const int chunk = 65535 / (4 * 256 * rows);
    if (ws && chunk >= 1 && items < (1ll << 30)) {
        const hipError_t e0 = hipMemsetAsync(ws, 0, (size_t)256 * C * HIST_PLANE * sizeof(uint32_t), st);
        if (e0 != hipSuccess) return (int)e0;
        const int cus = hiast_grid_cus();                 // persistent: one 1024-thread block per CU
        const int blocks = (int)(items / 4 < cus ? (items + 3) / 4 : cus);
        hipLaunchKernelGGL(plabel_pass1_persistent_kernel<C>, dim3(blocks), dim3(1024), 0, st, logits, h, w, H, W, sh, sw,
                           maxprob, argmax, ws, nx, (int)items, chunk);
        HIAST_CHECK_LAUNCH();
        hipLaunchKernelGGL(hist_merge_kernel<C>, dim3((C * HIAST_NBINS + 255) / 256), dim3(256), 0, st, ws, hist);
    } else if (ws) {
        const hipError_t e0 = hipMemsetAsync(ws, 0, (size_t)256 * C * HIST_PLANE * sizeof(uint32_t), st);
        if (e0 != hipSuccess) return (int)e0;
        hipLaunchKernelGGL((plabel_pass1_kernel<C, true>), grid, dim3(256), 0, st, logits, h, w, H, W, sh, sw, maxprob,
                           argmax, ws);
        HIAST_CHECK_LAUNCH();
        hipLaunchKernelGGL(hist_merge_kernel<C>, dim3((C * HIAST_NBINS + 255) / 256), dim3(256), 0, st, ws, hist);
    } else {
        hipLaunchKernelGGL((plabel_pass1_kernel<C, false>), grid, dim3(256), 0, st, logits, h, w, H, W, sh, sw, maxprob,
                           argmax, hist);
    }
    HIAST_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// pass 2: VEC pixels per thread per iteration (VEC=4: float4 + uchar4 when every image row base
// stays 16-byte aligned; VEC=1 for ragged sizes), class statistics reduced per wave over the
// distinct labels present, then per block in LDS, then one integer atomic per class.
template <bool HAS_THR, int VEC>
__global__ __launch_bounds__(256) void plabel_pass2_kernel(
    const float* __restrict__ maxprob, const uint8_t* __restrict__ argmax,
    const float* __restrict__ thr_up, int C, long long HW, uint8_t* __restrict__ plbl,
    long long* __restrict__ count, unsigned long long* __restrict__ sumprob_fx)
{
    __shared__ unsigned s_cnt[HIAST_MAX_CLASSES];
    __shared__ unsigned long long s_sum[HIAST_MAX_CLASSES];
    __shared__ float s_thr[HIAST_MAX_CLASSES];
    const int b = blockIdx.y;
    if (threadIdx.x < HIAST_MAX_CLASSES) {
        s_cnt[threadIdx.x] = 0;
        s_sum[threadIdx.x] = 0;
        s_thr[threadIdx.x] = (HAS_THR && (int)threadIdx.x < C) ? thr_up[threadIdx.x] : 0.0f;
    }
    __syncthreads();

    const float* mp = maxprob + (size_t)b * HW;
    const uint8_t* am = argmax + (size_t)b * HW;
    uint8_t* out = plbl + (size_t)b * HW;
    const long long nvec = HW / VEC;
    // block-uniform trip count: every lane stays in the loop for the ballots / shuffles
    for (long long base = (long long)blockIdx.x * 256; base < nvec; base += (long long)gridDim.x * 256) {
        const long long v = base + threadIdx.x;
        const bool in = v < nvec;
        float pv[VEC];
        uint8_t lv[VEC];
        if (VEC == 4) {
            float4 p = make_float4(0, 0, 0, 0);
            uchar4 l = make_uchar4(255, 255, 255, 255);
            if (in) {
                p = reinterpret_cast<const float4*>(mp)[v];
                l = reinterpret_cast<const uchar4*>(am)[v];
            }
            pv[0] = p.x; pv[VEC > 1 ? 1 : 0] = p.y; pv[VEC > 2 ? 2 : 0] = p.z; pv[VEC > 3 ? 3 : 0] = p.w;
            lv[0] = l.x; lv[VEC > 1 ? 1 : 0] = l.y; lv[VEC > 2 ? 2 : 0] = l.z; lv[VEC > 3 ? 3 : 0] = l.w;
        } else {
            pv[0] = in ? mp[v] : 0.0f;
            lv[0] = in ? am[v] : (uint8_t)255;
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            uint8_t lab = lv[k];
            if (in && lab < C) {
                if (HAS_THR && pv[k] < s_thr[lab]) lab = HIAST_IGNORE;
            } else {
                lab = HIAST_IGNORE;
            }
            lv[k] = lab;
            // exact: prob is a multiple of 2^-30 for prob >= 2^-7
            const unsigned long long fx = (unsigned long long)((double)pv[k] * 1073741824.0);
            unsigned long long todo = __ballot(lab != HIAST_IGNORE);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int cl = __shfl((int)lab, leader, 64);
                const unsigned long long same = __ballot((int)lab == cl);
                const unsigned long long s = wave_sum_u64((int)lab == cl ? fx : 0ull);
                if (lane_id() == leader) {
                    atomicAdd(&s_cnt[cl], (unsigned)__popcll(same));
                    atomicAdd(&s_sum[cl], s);
                }
                todo &= ~same;
            }
        }
        if (in) {
            if (VEC == 4)
                reinterpret_cast<uchar4*>(out)[v] = make_uchar4(lv[0], lv[VEC > 1 ? 1 : 0],
                                                                lv[VEC > 2 ? 2 : 0], lv[VEC > 3 ? 3 : 0]);
            else
                out[v] = lv[0];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < C && s_cnt[threadIdx.x]) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&count[(size_t)b * C + threadIdx.x]),
                  (unsigned long long)s_cnt[threadIdx.x]);
        atomicAdd(&sumprob_fx[threadIdx.x], s_sum[threadIdx.x]);
    }
}

}  // namespace hiast

extern "C" size_t hiast_plabel_pass1_workspace_bytes(int C)
{
    return C > 0 && C <= HIAST_MAX_CLASSES ? (size_t)256 * C * hiast::HIST_PLANE * sizeof(uint32_t) : 0;
}

extern "C" int hiast_plabel_pass1(const float* logits_lr, int B, int C, int h, int w, int H, int W,
                                  float* maxprob, uint8_t* argmax, uint32_t* hist, void* workspace,
                                  size_t workspace_bytes, hiast_stream_t stream)
{
    if (!logits_lr || !maxprob || !argmax || !hist) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (H < h || W < w || B > 65535 || h > 65535) return HIAST_E_RANGE;
    if (workspace && (workspace_bytes < hiast_plabel_pass1_workspace_bytes(C) || (((uintptr_t)workspace) & 3))) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* ws = (uint32_t*)workspace;
    switch (C) {
        case 19: return hiast::launch_pass1<19>(logits_lr, B, h, w, H, W, maxprob, argmax, hist, ws, st);
        case 16: return hiast::launch_pass1<16>(logits_lr, B, h, w, H, W, maxprob, argmax, hist, ws, st);
        case 9: return hiast::launch_pass1<9>(logits_lr, B, h, w, H, W, maxprob, argmax, hist, ws, st);
        case 2: return hiast::launch_pass1<2>(logits_lr, B, h, w, H, W, maxprob, argmax, hist, ws, st);
        default: return HIAST_E_RANGE;   // class counts of the reference's datasets (19 / 9) + tests
    }
}

extern "C" int hiast_plabel_pass2(const float* maxprob, const uint8_t* argmax, const float* thr_up,
                                  int B, int C, int64_t HW, uint8_t* plbl, int64_t* count,
                                  uint64_t* sumprob_fx, hiast_stream_t stream)
{
    if (!maxprob || !argmax || !plbl || !count || !sumprob_fx) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || HW <= 0) return HIAST_E_ARG;
    if (C > HIAST_MAX_CLASSES || B > 65535) return HIAST_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = !(((uintptr_t)maxprob) & 15) && !(((uintptr_t)argmax) & 3) &&
                      !(((uintptr_t)plbl) & 3) && (HW % 4 == 0);
    const long long nvec = vec4 ? HW / 4 : HW;
    int gx = (int)((nvec + 256 * 4 - 1) / (256 * 4));
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    dim3 grid(gx, B);
    long long* cnt = (long long*)count;
    unsigned long long* sfx = (unsigned long long*)sumprob_fx;
#define HIAST_P2(T, V)                                                                            \
    hipLaunchKernelGGL((hiast::plabel_pass2_kernel<T, V>), grid, dim3(256), 0, st, maxprob, argmax, \
                       thr_up, C, (long long)HW, plbl, cnt, sfx)
    if (thr_up) { if (vec4) HIAST_P2(true, 4); else HIAST_P2(true, 1); }
    else        { if (vec4) HIAST_P2(false, 4); else HIAST_P2(false, 1); }
#undef HIAST_P2
    HIAST_CHECK_LAUNCH();
    return 0;
}
