// K3b / K16 — the generator policies and the validator either side of the IAS path.  gfx950 only.
//
// K3b  hiast_plabel_strided_hist: the CBST policy's confidence sample
//      (workflows/pseudo_label_generator.py:142-158): per class, every `interval`-th element of
//      `probs_pred[lbls_pred == c]` — i.e. of the class's pixels in raster order over the batch — enters the
//      per-class fp16 list.  Here: the rank of a pixel within its class comes from a block-count + scan + in-block
//      ballot ranking, the kept pixels go into the same [C, NBINS] integer histogram pass 1 uses.
// K16  hiast_tta_fused: Validator.get_multi_scale_and_flip_logits + argmax (workflows/validator.py:34-55,92):
//      Σ_scales resample_to_native( softmax(up(z_s)) + flip(softmax(up(z_s,flipped))) ) from the LOW-RES head outputs
//      of every (scale, flip) forward in one kernel: neither the full-resolution logits nor the per-scale probability
//      maps (39.8 MB/img each at 512x1024) are ever stored; the output is the uint8 label map (+ optionally the
//      summed probabilities for the API that returns them).
#include "common.h"

namespace hiast {

constexpr int SH_TILE = 4096;     // pixels per block (16 rounds of 256)

// rank of every lane's pixel among the pixels of ITS class within the 256-pixel round, and the per-class totals of
// the round; s_w[4][MAX_CLASSES] is scratch.  Returns -1 for lanes without a class.
__device__ __forceinline__ int round_rank(int cls, unsigned* s_w, unsigned* s_tot)
{
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 4 * HIAST_MAX_CLASSES) s_w[threadIdx.x] = 0;
    __syncthreads();
    int rk = -1;
    unsigned long long todo = __ballot(cls >= 0);
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(cls, leader, 64);
        const unsigned long long same = __ballot(cls == k);
        if (cls == k) rk = __popcll(same & lt);
        if (lane_id() == leader) s_w[wave * HIAST_MAX_CLASSES + k] = (unsigned)__popcll(same);
        todo &= ~same;
    }
    __syncthreads();
    if (cls >= 0)
        for (int w = 0; w < wave; ++w) rk += (int)s_w[w * HIAST_MAX_CLASSES + cls];
    if (threadIdx.x < HIAST_MAX_CLASSES)
        s_tot[threadIdx.x] = s_w[threadIdx.x] + s_w[HIAST_MAX_CLASSES + threadIdx.x] +
                             s_w[2 * HIAST_MAX_CLASSES + threadIdx.x] + s_w[3 * HIAST_MAX_CLASSES + threadIdx.x];
    __syncthreads();
    return rk;
}

// MODE 0: per-block class counts -> cnt[nblk][C];  MODE 1: histogram of the kept pixels, base[nblk][C] = rank of the
// block's first pixel of each class (mod interval)
template <int MODE>
__global__ __launch_bounds__(256) void strided_hist_kernel(const float* __restrict__ maxprob,
                                                           const uint8_t* __restrict__ argmax, long long N, int C,
                                                           int interval, unsigned* __restrict__ blk,
                                                           uint32_t* __restrict__ hist)
{
    __shared__ unsigned s_w[4 * HIAST_MAX_CLASSES];
    __shared__ unsigned s_tot[HIAST_MAX_CLASSES];
    __shared__ unsigned s_run[HIAST_MAX_CLASSES];
    const long long t0 = (long long)blockIdx.x * SH_TILE;
    if (threadIdx.x < HIAST_MAX_CLASSES)
        s_run[threadIdx.x] = (MODE == 1 && (int)threadIdx.x < C) ? blk[(size_t)blockIdx.x * C + threadIdx.x] : 0u;
    __syncthreads();
    for (int r = 0; r < SH_TILE / 256; ++r) {
        const long long i = t0 + r * 256 + threadIdx.x;
        int cls = -1;
        if (i < N) {
            const int a = argmax[i];
            cls = a < C ? a : -1;
        }
        if (MODE == 0) {
            // counts only: wave-aggregated LDS adds
            unsigned long long todo = __ballot(cls >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int k = __shfl(cls, leader, 64);
                const unsigned long long same = __ballot(cls == k);
                if (lane_id() == leader) atomicAdd(&s_run[k], (unsigned)__popcll(same));
                todo &= ~same;
            }
        } else {
            const int rk = round_rank(cls, s_w, s_tot);
            unsigned key = 0xFFFFFFFFu;
            if (cls >= 0 && ((s_run[cls] + (unsigned)rk) % (unsigned)interval) == 0u) {
                const unsigned bin = __half_as_ushort(__float2half_rn(maxprob[i]));
                if (bin < HIAST_NBINS) key = (unsigned)cls * HIAST_NBINS + bin;
            }
            unsigned long long todo = __ballot(key != 0xFFFFFFFFu);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const unsigned k = __shfl(key, leader, 64);
                const unsigned long long same = __ballot(key == k);
                if (lane_id() == leader) atomicAdd(&hist[k], (unsigned)__popcll(same));
                todo &= ~same;
            }
            __syncthreads();       // every lane has read s_run before it moves on
            if ((int)threadIdx.x < C) s_run[threadIdx.x] = (s_run[threadIdx.x] + s_tot[threadIdx.x]) % (unsigned)interval;
            __syncthreads();
        }
    }
    if (MODE == 0) {
        __syncthreads();
        if ((int)threadIdx.x < C) blk[(size_t)blockIdx.x * C + threadIdx.x] = s_run[threadIdx.x];
    }
}

// exclusive scan over blocks per class (mod interval), seeded with the ranks' offset; also the batch totals
__global__ void strided_scan_kernel(unsigned* __restrict__ blk, int nblk, int C, int interval,
                                    const long long* __restrict__ rank_offset, long long* __restrict__ class_total)
{
    const int c = threadIdx.x;
    if (c >= C) return;
    unsigned run = rank_offset ? (unsigned)(rank_offset[c] % interval) : 0u;
    long long tot = 0;
    for (int b = 0; b < nblk; ++b) {
        const unsigned n = blk[(size_t)b * C + c];
        blk[(size_t)b * C + c] = run;
        run = (run + n) % (unsigned)interval;
        tot += n;
    }
    if (class_total) class_total[c] = tot;
}

// ---------------------------------------------------------------------------------------------------------------
// K16.  One thread per native pixel.  Per scale s (size Hs x Ws, low-res head map hs x ws): the native pixel reads the
// <= 4 bilinear taps of the size-s probability map; each tap's probabilities are softmax(bilinear(z_s)) at that
// size-s pixel (+ the same for the flipped forward, read at the mirrored column) — "HIAST-A arithmetic" for the
// logit interpolation and the softmax, so the label map is bit-reproducible against the oracle's restatement.
struct TtaScale {
    const float* z;       // [B,C,hs,ws] head output of the resized image
    const float* zf;      // [B,C,hs,ws] head output of the horizontally flipped resized image, or null
    int hs, ws, Hs, Ws;
};
struct TtaArgs {
    TtaScale s[HIAST_TTA_MAX_SCALES];
    int n;
};

template <int C>
__device__ __forceinline__ void tap_probs(const float* __restrict__ z, int hs, int ws, float sh, float sw, int Y, int X,
                                          float wgt, float* acc)
{
    const Src sy = src_of(sh, Y, hs), sx = src_of(sw, X, ws);
    float v[C];
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* p = z + (size_t)c * hs * ws;
        const float top = lerp_h(p[sy.i0 * ws + sx.i0], p[sy.i0 * ws + sx.i1], sx.l0, sx.l1);
        const float bot = lerp_h(p[sy.i1 * ws + sx.i0], p[sy.i1 * ws + sx.i1], sx.l0, sx.l1);
        v[c] = lerp_v(top, bot, sy.l0, sy.l1);
        if (c == 0 || v[c] > m) m = v[c];
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        v[c] = a_expf(v[c] - m);
        s = s + v[c];
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = fmaf(wgt, v[c] * inv, acc[c]);
}

template <int C>
__global__ __launch_bounds__(256) void tta_fused_kernel(TtaArgs a, int B, int H, int W, float* __restrict__ probsum,
                                                        uint8_t* __restrict__ label)
{
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int Y = blockIdx.y, b = blockIdx.z;
    if (X >= W) return;
    float tot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) tot[c] = 0.0f;
    for (int si = 0; si < a.n; ++si) {
        const TtaScale sc = a.s[si];
        // native pixel -> size-s grid (the second F.interpolate, validator.py:52)
        const float rh = H > 1 ? (float)(sc.Hs - 1) / (float)(H - 1) : 0.0f;
        const float rw = W > 1 ? (float)(sc.Ws - 1) / (float)(W - 1) : 0.0f;
        const Src ty = src_of(rh, Y, sc.Hs), tx = src_of(rw, X, sc.Ws);
        // low-res head map -> size-s grid (the segmentor's interpolate, self_training_segmentor.py:27)
        const float sh = sc.Hs > 1 ? (float)(sc.hs - 1) / (float)(sc.Hs - 1) : 0.0f;
        const float sw = sc.Ws > 1 ? (float)(sc.ws - 1) / (float)(sc.Ws - 1) : 0.0f;
        const float* z = sc.z + (size_t)b * C * sc.hs * sc.ws;
        const float* zf = sc.zf ? sc.zf + (size_t)b * C * sc.hs * sc.ws : nullptr;
        // r = lerp_v(lerp_h(p00, p01), lerp_h(p10, p11)) per class, expanded into four tap weights in the SAME
        // rounding order as the oracle: acc_top = wl0*p00 (+) wl1*p01 ... is not associative, so the per-scale value is
        // built exactly as lerp_v(lerp_h(.), lerp_h(.)) below
        float p00[C], p01[C], p10[C], p11[C];
#pragma unroll
        for (int c = 0; c < C; ++c) p00[c] = p01[c] = p10[c] = p11[c] = 0.0f;
        tap_probs<C>(z, sc.hs, sc.ws, sh, sw, ty.i0, tx.i0, 1.0f, p00);
        if (zf) tap_probs<C>(zf, sc.hs, sc.ws, sh, sw, ty.i0, sc.Ws - 1 - tx.i0, 1.0f, p00);
        // a tap with weight 0 (scale == native size: every pixel) contributes fmaf(0, p, .) = nothing: skip its softmaxes
        const bool dx = tx.i1 != tx.i0 && tx.l1 != 0.0f, dy = ty.i1 != ty.i0 && ty.l1 != 0.0f;
        if (dx) {
            tap_probs<C>(z, sc.hs, sc.ws, sh, sw, ty.i0, tx.i1, 1.0f, p01);
            if (zf) tap_probs<C>(zf, sc.hs, sc.ws, sh, sw, ty.i0, sc.Ws - 1 - tx.i1, 1.0f, p01);
        }
        if (dy) {
            tap_probs<C>(z, sc.hs, sc.ws, sh, sw, ty.i1, tx.i0, 1.0f, p10);
            if (zf) tap_probs<C>(zf, sc.hs, sc.ws, sh, sw, ty.i1, sc.Ws - 1 - tx.i0, 1.0f, p10);
        }
        if (dx && dy) {
            tap_probs<C>(z, sc.hs, sc.ws, sh, sw, ty.i1, tx.i1, 1.0f, p11);
            if (zf) tap_probs<C>(zf, sc.hs, sc.ws, sh, sw, ty.i1, sc.Ws - 1 - tx.i1, 1.0f, p11);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float q01 = dx ? p01[c] : p00[c];
            const float q10 = dy ? p10[c] : p00[c];
            const float q11 = dx ? (dy ? p11[c] : p01[c]) : (dy ? p10[c] : p00[c]);
            const float top = lerp_h(p00[c], q01, tx.l0, tx.l1);
            const float bot = lerp_h(q10, q11, tx.l0, tx.l1);
            tot[c] = tot[c] + lerp_v(top, bot, ty.l0, ty.l1);
        }
    }
    const size_t pix = ((size_t)b * H + Y) * W + X;
    if (probsum) {
#pragma unroll
        for (int c = 0; c < C; ++c) probsum[((size_t)b * C + c) * H * W + (size_t)Y * W + X] = tot[c];
    }
    if (label) {
        float m = tot[0];
        int am = 0;
#pragma unroll
        for (int c = 1; c < C; ++c)
            if (tot[c] > m) { m = tot[c]; am = c; }
        label[pix] = (uint8_t)am;
    }
}

template <int C>
static int launch_tta(const TtaArgs& a, int B, int H, int W, float* probsum, uint8_t* label, hipStream_t st)
{
    dim3 grid((W + 255) / 256, H, B);
    hipLaunchKernelGGL(tta_fused_kernel<C>, grid, dim3(256), 0, st, a, B, H, W, probsum, label);
    HIAST_CHECK_LAUNCH();
    return 0;
}

}  // namespace hiast

extern "C" size_t hiast_plabel_strided_hist_workspace_bytes(int64_t N, int C)
{
    if (N <= 0 || C <= 0) return 0;
    const long long nblk = (N + hiast::SH_TILE - 1) / hiast::SH_TILE;
    return (size_t)nblk * C * sizeof(unsigned);
}

extern "C" int hiast_plabel_strided_hist(const float* maxprob, const uint8_t* argmax, int64_t N, int C, int interval,
                                         const int64_t* rank_offset, int64_t* class_total, uint32_t* hist,
                                         void* workspace, size_t workspace_bytes, hiast_stream_t stream)
{
    if (!maxprob || !argmax || !hist || !workspace) return HIAST_E_ARG;
    if (N <= 0 || C <= 0 || interval <= 0) return HIAST_E_ARG;
    if (C > HIAST_MAX_CLASSES) return HIAST_E_RANGE;
    const long long nblk = (N + hiast::SH_TILE - 1) / hiast::SH_TILE;
    if (nblk > 0x7fffffffLL) return HIAST_E_RANGE;
    if (workspace_bytes < (size_t)nblk * C * sizeof(unsigned)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    unsigned* blk = (unsigned*)workspace;
    hipLaunchKernelGGL(hiast::strided_hist_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, maxprob, argmax,
                       (long long)N, C, interval, blk, hist);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::strided_scan_kernel, dim3(1), dim3(64), 0, st, blk, (int)nblk, C, interval,
                       (const long long*)rank_offset, (long long*)class_total);
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::strided_hist_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, maxprob, argmax,
                       (long long)N, C, interval, blk, hist);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_tta_fused(const float* const* z, const float* const* zf, const int* hs, const int* ws,
                               const int* Hs, const int* Ws, int n_scales, int B, int C, int H, int W, float* probsum,
                               uint8_t* label, hiast_stream_t stream)
{
    if (!z || !hs || !ws || !Hs || !Ws || (!probsum && !label)) return HIAST_E_ARG;
    if (n_scales <= 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (n_scales > HIAST_TTA_MAX_SCALES || B > 65535 || H > 65535) return HIAST_E_RANGE;
    hiast::TtaArgs a;
    a.n = n_scales;
    for (int i = 0; i < n_scales; ++i) {
        if (!z[i] || hs[i] <= 0 || ws[i] <= 0 || Hs[i] < hs[i] || Ws[i] < ws[i]) return HIAST_E_ARG;
        a.s[i].z = z[i];
        a.s[i].zf = zf ? zf[i] : nullptr;
        a.s[i].hs = hs[i];
        a.s[i].ws = ws[i];
        a.s[i].Hs = Hs[i];
        a.s[i].Ws = Ws[i];
    }
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 19: return hiast::launch_tta<19>(a, B, H, W, probsum, label, st);
        case 16: return hiast::launch_tta<16>(a, B, H, W, probsum, label, st);
        case 9: return hiast::launch_tta<9>(a, B, H, W, probsum, label, st);
        case 2: return hiast::launch_tta<2>(a, B, H, W, probsum, label, st);
        default: return HIAST_E_RANGE;
    }
}
