// K9c host side: weight packing, split planes, stem tail and the launch dispatch of the implicit-GEMM kernel
// (igemm_kernel.h holds the kernel; igemm_f16.hip instantiates its fp16 variants in a translation unit of its own).
#include "igemm_kernel.h"

namespace hiast {

// fp32 conv weight [N][K][taps] (torch layout; taps = kh*kw) -> Wp[N][taps][slabs of 128 B] bf16.
// transpose: pack the weight of the ADJOINT convolution instead (the data-gradient of a stride-1 conv is a conv of
// dY with the channel-transposed, spatially flipped kernel): Wp[k][taps-1-t][.. n ..] = w[n][k][t]; the output then
// has N_out = K rows and K_out = N reduction channels.
// element offset of (row n, tap t, reduction channel k) in a packed weight with KO reduction channels
template <int PL>
__device__ __forceinline__ size_t ig_wp_elem(int n, int t, int k, int taps, int KO)
{
    const size_t row = ((size_t)n * taps + t) * PL * KO;
    return PL == 2 ? row + (size_t)(k >> 5) * 64 + (k & 31) : row + k;      // lo plane: + 32
}

// mode 0: wp = forward weight; 1: wp = adjoint weight; 2: wp = forward, wpt = adjoint (one pass over w)
template <int PL, bool F16 = false>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w,
                                                               unsigned short* __restrict__ wp,
                                                               unsigned short* __restrict__ wpt, int N, int K, int taps,
                                                               int mode)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * taps * K) return;
    const int k = (int)(idx % K);
    const int t = (int)((idx / K) % taps);
    const int n = (int)(idx / ((long long)K * taps));
    const float v = w[((size_t)n * K + k) * taps + t];
    unsigned short h, l = 0;
    if (F16) h = H16<true>::enc(v);
    else ig_split(v, h, l);
    if (mode != 1) {
        unsigned short* d = wp + ig_wp_elem<PL>(n, t, k, taps, K);
        d[0] = h;
        if (PL == 2) d[32] = l;
    }
    if (mode != 0) {                                     // adjoint: rows = k, reduction = n, taps flipped
        unsigned short* d = (mode == 1 ? wp : wpt) + ig_wp_elem<PL>(k, taps - 1 - t, n, taps, N);
        d[0] = h;
        if (PL == 2) d[32] = l;
    }
}

// Tiled form for channel counts that are multiples of 64 (every trunk convolution, the ASPP tap matrices): one block
// takes the 64 (n) x 64 (k) x taps tile of one tensor through LDS, so that the source is read in contiguous runs of
// 64*taps floats per n and BOTH packed forms leave as whole 128-byte lines (forward: 64 k of one (n, tap); adjoint:
// 64 n of one (k, flipped tap), written as two 64-byte halves by consecutive passes of the same block).  The elementwise kernel above issued 2-byte stores 2*N (adjoint) bytes apart and
// reached 0.47 TB/s — 0.73 ms per trunk, three trunks per training step once the packs follow every optimiser / EMA
// step.  PL = 2 walks the n range in two halves (a 32-channel slab holds hi|lo: the same 128-byte line).
constexpr int PK_PITCH = 72;        // ushorts per (n, tap) row of the LDS tile: 64 + 8 (keeps 16-byte alignment)

template <int PL, bool F16 = false>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, unsigned short* __restrict__ wp,
                                          unsigned short* __restrict__ wpt, int N, int K, int taps, int mode, int n0,
                                          int k0, unsigned short* lds)
{
    constexpr int TN = PL == 1 ? 32 : 16;               // rows of n per pass: LDS = TN x taps x 72 x 2 B x PL (41 KiB at 9 taps,
    unsigned short* hi = lds;                           // 4.6 KiB at 1: several blocks per CU instead of one)
    unsigned short* lo = lds + (PL == 2 ? TN * taps * PK_PITCH : 0);
    unsigned short* fwd = (mode == 1) ? nullptr : wp;
    unsigned short* adj = (mode == 0) ? nullptr : (mode == 1 ? wp : wpt);
    const int run = 64 * taps;                                   // contiguous floats per n in the source
    for (int nh = 0; nh < 64; nh += TN) {
        if (nh) __syncthreads();
        for (int e = threadIdx.x; e < TN * run; e += 256) {
            const int n = e / run, r = e - n * run;
            const int k = r / taps, t = r - k * taps;
            const float v = w[((size_t)(n0 + nh + n) * K + k0) * taps + r];
            unsigned short h, l = 0;
            if (F16) h = H16<true>::enc(v);
            else ig_split(v, h, l);
            hi[(n * taps + t) * PK_PITCH + k] = h;
            if (PL == 2) lo[(n * taps + t) * PK_PITCH + k] = l;
        }
        __syncthreads();
        if (fwd) {
            for (int e = threadIdx.x; e < TN * taps * 8; e += 256) {       // (n, t, 8-channel chunk)
                const int c = e & 7, nt = e >> 3;
                const int n = nt / taps, t = nt - n * taps;
                const uint4 vh = *reinterpret_cast<const uint4*>(&hi[nt * PK_PITCH + c * 8]);
                unsigned short* d = fwd + ig_wp_elem<PL>(n0 + nh + n, t, k0 + c * 8, taps, K);
                *reinterpret_cast<uint4*>(d) = vh;
                if (PL == 2) *reinterpret_cast<uint4*>(d + 32) = *reinterpret_cast<const uint4*>(&lo[nt * PK_PITCH + c * 8]);
            }
        }
        if (adj) {
            for (int e = threadIdx.x; e < 64 * taps * (TN / 8); e += 256) {     // (k, t, 8-row chunk of n)
                const int c = e % (TN / 8), kt = e / (TN / 8);
                const int k = kt / taps, t = kt - k * taps;
                unsigned short vh[8], vl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vh[j] = hi[((c * 8 + j) * taps + t) * PK_PITCH + k];
                    if (PL == 2) vl[j] = lo[((c * 8 + j) * taps + t) * PK_PITCH + k];
                }
                unsigned short* d = adj + ig_wp_elem<PL>(k0 + k, taps - 1 - t, n0 + nh + c * 8, taps, N);
                *reinterpret_cast<uint4*>(d) = make_uint4(vh[0] | (unsigned)vh[1] << 16, vh[2] | (unsigned)vh[3] << 16,
                                                          vh[4] | (unsigned)vh[5] << 16, vh[6] | (unsigned)vh[7] << 16);
                if (PL == 2)
                    *reinterpret_cast<uint4*>(d + 32) = make_uint4(vl[0] | (unsigned)vl[1] << 16, vl[2] | (unsigned)vl[3] << 16,
                                                                   vl[4] | (unsigned)vl[5] << 16, vl[6] | (unsigned)vl[7] << 16);
            }
        }
    }
}

template <int PL, bool F16 = false>
__global__ __launch_bounds__(256) void pack_conv_weight_tiled_kernel(const float* __restrict__ w,
                                                                     unsigned short* __restrict__ wp,
                                                                     unsigned short* __restrict__ wpt, int N, int K,
                                                                     int taps, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];       // 32 * taps * PK_PITCH
    const int kt = K / 64;
    pack_tile<PL, F16>(w, wp, wpt, N, K, taps, mode, (int)(blockIdx.x / kt) * 64, (int)(blockIdx.x % kt) * 64, lds);
}

// the same for a LIST of weights in one launch (one ResNet trunk = 104 convolutions whose packed forms are rebuilt
// after every optimiser / EMA step): block = one 64 x 64 (n, k) tile of one tensor; chunk_start = the tile's index
// within its tensor, row-major over (N/64, K/64)
__global__ __launch_bounds__(256) void pack_conv_weight_multi_kernel(const hiast_pack_rec* __restrict__ table,
                                                                     const int32_t* __restrict__ chunk_tensor,
                                                                     const int64_t* __restrict__ chunk_start)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];       // 32 * max taps of the launch * PK_PITCH
    const hiast_pack_rec r = table[chunk_tensor[blockIdx.x]];
    const int kt = r.K / 64;
    const int tile = (int)chunk_start[blockIdx.x];
    const int n0 = (tile / kt) * 64, k0 = (tile % kt) * 64;
    if (r.planes == HIAST_FMT_SPLIT_BF16)                // (the record's `planes` field holds the operand format)
        pack_tile<2>(r.w, (unsigned short*)r.wp, (unsigned short*)r.wpt, r.N, r.K, r.taps, r.mode, n0, k0, lds);
    else if (r.planes == HIAST_FMT_FP16)
        pack_tile<1, true>(r.w, (unsigned short*)r.wp, (unsigned short*)r.wpt, r.N, r.K, r.taps, r.mode, n0, k0, lds);
    else
        pack_tile<1>(r.w, (unsigned short*)r.wp, (unsigned short*)r.wpt, r.N, r.K, r.taps, r.mode, n0, k0, lds);
}

// fp32 [M][C] -> split planes (and back: v = hi + lo, exact in fp32)
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ p,
                                                        long long M, int C)
{
    const long long total8 = M * C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const long long m = (i * 8) / C;
        const int c = (int)((i * 8) - m * C);
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned ph[4], pl_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h0, l0, h1, l1;
            ig_split(v[2 * q], h0, l0);
            ig_split(v[2 * q + 1], h1, l1);
            ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
            pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* dst = p + ig_elem<2>((size_t)m, c, C);
        *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        *reinterpret_cast<uint4*>(dst + 32) = make_uint4(pl_[0], pl_[1], pl_[2], pl_[3]);
    }
}

__global__ __launch_bounds__(256) void from_planes_kernel(const unsigned short* __restrict__ p, float* __restrict__ x,
                                                          long long M, int C)
{
    const long long total8 = M * C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const long long m = (i * 8) / C;
        const int c = (int)((i * 8) - m * C);
        const unsigned short* src = p + ig_elem<2>((size_t)m, c, C);
        const uint4 h = *reinterpret_cast<const uint4*>(src), l = *reinterpret_cast<const uint4*>(src + 32);
        const unsigned wh[4] = {h.x, h.y, h.z, h.w}, wl[4] = {l.x, l.y, l.z, l.w};
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[2 * q] = __uint_as_float(wh[q] << 16) + __uint_as_float(wl[q] << 16);
            v[2 * q + 1] = __uint_as_float(wh[q] & 0xFFFF0000u) + __uint_as_float(wl[q] & 0xFFFF0000u);
        }
        reinterpret_cast<float4*>(x)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(x)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace hiast

// tuning variables of the tile kernel, read once per process; the co-scheduling hint of the calling thread (igemm_kernel.h)
const IgEnv& hiast_ig_env()
{
    static const IgEnv env = [] {
        IgEnv e;
        const char* v = getenv("HIAST_IGEMM_HALF");
        e.half = v ? (atoi(v) != 0) : -1;
        v = getenv("HIAST_IGEMM_BN");
        const int bn = v ? atoi(v) : 0;
        e.bn = (bn == 64 || bn == 128 || bn == 256) ? bn : 0;
        e.ragged_off = getenv("HIAST_IGEMM_RAGGED") != nullptr;
        v = getenv("HIAST_IGEMM_COSCHED");
        e.cosched0 = v && atoi(v) != 0;
        return e;
    }();
    return env;
}
static thread_local int ig_cosched_tls = -1, ig_half_tls = -1;
int hiast_ig_cosched_tls_get() { return ig_cosched_tls; }
int hiast_ig_half_tls_get() { return ig_half_tls; }
extern "C" int hiast_igemm_set_half(int v)
{
    const int prev = ig_half_tls;
    ig_half_tls = v < 0 ? -1 : (v != 0);
    return prev;
}
extern "C" int hiast_igemm_set_cosched(int on)
{
    const int prev = ig_cosched_tls;
    ig_cosched_tls = on < 0 ? -1 : (on != 0);
    return prev;
}

// K9e (xconv.hip): register-resident-weight kernel for the HBM-bound expanding 1x1 shapes
int hiast_xconv_ok(int64_t M, int K, int N, int planes, int taps, int out_f32, int has_bn, int has_res, int relu,
                   int has_gate, int gate_mask, int has_stats);
int hiast_xconv_stats_rows(int64_t M, int N);
int hiast_xconv_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       float* stats, const void* res_gate, int f16, hipStream_t st);
// K9g (xconv2.hip): the same idea for the split-plane 256 -> N launches with BatchNorm(eval) (+ residual + ReLU)
int hiast_xconv2_ok(int64_t M, int K, int N, int planes, int taps, int out_f32, int has_bn, int has_res, int relu,
                    int has_gate, int has_stats);
int hiast_xconv2_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                        const float* var, float eps, const void* res, int relu, void* y, int64_t M, int N, hipStream_t st);
// igemm_f16.hip: the fp16 instantiations of the tile kernel (a translation unit of its own: build time)
int hiast_igemm_launch_f16(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N, int taps,
                           hiast::IGeo geo, float* stats, const void* res_gate, int gate_mask, hipStream_t st, int stats_mode,
                           int out_f32, int stats_rows);

// stats_mode: 0 none, 1 forward BatchNorm sums of the output (stats), 2 backward BatchNorm sums of the activation whose
// gradient the output is (data-gradient launches: res = that BN's input x, mean / var = its batch mean / invstd)
static int igemm_launch_mode(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                             const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                             int taps, int H, int W, int stride, int dil, int fmt, int out_f32, hipStream_t st,
                             float* stats, const void* res_gate, int gate_mask, int stats_mode, int stats_rows)
{
    if (!hiast_fmt_ok(fmt)) return HIAST_E_RANGE;
    const int planes = hiast_fmt_planes(fmt), f16 = fmt == HIAST_FMT_FP16;
    if (stats_mode == 2) {
        if (!stats || !res || !mean || !var || planes != 1 || out_f32 || relu || res_gate || N % 8) return HIAST_E_RANGE;
    } else {
        if (stats && (planes != 1 || out_f32 || res || relu)) return HIAST_E_RANGE;
        if ((stats || res_gate) && mean) return HIAST_E_RANGE;   // the statistics / gradient variants carry no BatchNorm
    }
    if (res_gate && (planes != 1 || !res || relu || out_f32 || (!gate_mask && (((uintptr_t)res_gate) & 15)))) return HIAST_E_RANGE;
    if (!x || !wp || !y || (mean && !var)) return HIAST_E_ARG;
    if (M <= 0 || K <= 0 || N <= 0) return HIAST_E_ARG;
    if ((planes != 1 && planes != 2) || (taps != 1 && taps != 9)) return HIAST_E_RANGE;
    if ((K * planes) % 64 != 0 || N % 64 != 0 || M > (1ll << 31) - 512) return HIAST_E_RANGE;
    if (out_f32 && res) return HIAST_E_RANGE;
    if ((((uintptr_t)x) | ((uintptr_t)wp) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) return HIAST_E_RANGE;
    hiast::IGeo geo = {0, 0, 0, 0, 1, 1};
    size_t in_pix = (size_t)M;
    if (taps != 1 && stride == -2) {
        // transposed stride-2 3x3 (hiast_igemm_dgrad_s2): H x W is the OUTPUT map (the strided convolution's input), the
        // operand rows are its output gradient on ((H-1)/2+1) x ((W-1)/2+1)
        if (H <= 0 || W <= 0 || dil != 1 || planes != 1 || out_f32) return HIAST_E_ARG;
        const int Hs = (H - 1) / 2 + 1, Ws = (W - 1) / 2 + 1;
        if (M % ((int64_t)H * W) != 0) return HIAST_E_ARG;
        geo = {Hs, Ws, H, W, -2, 1};
        in_pix = (size_t)(M / ((int64_t)H * W)) * Hs * Ws;
    } else if (taps != 1) {
        if (H <= 0 || W <= 0 || stride <= 0 || dil <= 0) return HIAST_E_ARG;
        const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
        if (M % ((int64_t)Ho * Wo) != 0) return HIAST_E_ARG;
        geo = {H, W, Ho, Wo, stride, dil};
        in_pix = (size_t)(M / ((int64_t)Ho * Wo)) * H * W;
    }
    // buffer-descriptor addressing: byte offsets and the out-of-range marker need 31 bits
    if (in_pix * planes * K * 2 >= (1ull << 31) || (size_t)N * taps * planes * K * 2 >= (1ull << 31)) return HIAST_E_RANGE;
    if (stats_mode != 2 &&
        hiast_xconv_ok(M, K, N, planes, taps, out_f32, mean != nullptr, res != nullptr, relu, res_gate != nullptr, gate_mask,
                       stats != nullptr)) {
        if (stats && stats_rows != hiast_xconv_stats_rows(M, N)) return HIAST_E_ARG;
        return hiast_xconv_launch(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, stats, res_gate, f16, st);
    }
    if (stats_mode == 0 && !f16 &&
        hiast_xconv2_ok(M, K, N, planes, taps, out_f32, mean != nullptr, res != nullptr, relu, res_gate != nullptr, stats != nullptr))
        return hiast_xconv2_launch(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, N, st);
    if (f16)
        return hiast_igemm_launch_f16(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate,
                                      gate_mask, st, stats_mode, out_f32, stats_rows);
    if (planes == 2) {
        if (out_f32) return launch_igemm_t<2, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode, stats_rows);
        return launch_igemm_t<2, false>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode, stats_rows);
    }
    if (out_f32) return launch_igemm_t<1, true>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode, stats_rows);
    return launch_igemm_t<1, false>(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, geo, stats, res_gate, gate_mask, st, stats_mode, stats_rows);
}

// shared launcher (also used by aspp2.hip for the ASPP tap GEMM)
int hiast_igemm_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       int taps, int H, int W, int stride, int dil, int fmt, int out_f32, hipStream_t st,
                       float* stats, const void* res_gate, int gate_mask, int stats_rows)
{
    return igemm_launch_mode(x, wp, gamma, beta, mean, var, eps, res, relu, y, M, K, N, taps, H, W, stride, dil, fmt,
                             out_f32, st, stats, res_gate, gate_mask, stats ? 1 : 0, stats_rows);
}

// Data gradient of a stride-1 trunk convolution (dy x the adjoint weight, hiast_pack_conv_weight transpose) whose output
// dA is the gradient of A = relu(bn(x)): also emits the per-block sums (Σg, Σ g*xhat) of that BatchNorm's backward,
// partial[hiast_igemm_stats_rows][Cout][2] — hiast_bn_nhwc_stats_from_partial turns them into the sums that
// hiast_bn_nhwc_bwd_apply takes (reference: autograd of conv -> bn -> relu in Bottleneck.forward, resnet.py:78-98).
extern "C" int hiast_igemm_dgrad_bn_stats(const void* dy, const void* wpt, void* da, int B, int H, int W, int Cin, int Cout,
                                          int taps, int dil, const void* bn_x, const float* gamma, const float* beta,
                                          const float* save_mean, const float* save_invstd, float* partial, int partial_rows,
                                          int fmt, hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (!bn_x || !save_mean || !save_invstd || !partial) return HIAST_E_ARG;
    if ((((uintptr_t)save_mean) | ((uintptr_t)save_invstd) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) return HIAST_E_RANGE;
    return igemm_launch_mode(dy, wpt, gamma, beta, save_mean, save_invstd, 0.0f, bn_x, 0, da, (int64_t)B * H * W, Cin, Cout,
                             taps, H, W, 1, dil, fmt, 0, (hipStream_t)stream, partial, nullptr, 0, 2, partial_rows);
}

// Data gradient of a 3x3 / stride-2 / padding-1 trunk convolution (layer2.0.conv2; autograd of that nn.Conv2d in
// Bottleneck.forward, resnet.py:78-98): dx [B,H,W,Cin] = transposed convolution of dy [B,(H-1)/2+1,(W-1)/2+1,Cout] with the
// adjoint-packed weight wpt (hiast_pack_conv_weight transpose) — the tile kernel with the parity test of the UPS variant.
extern "C" int hiast_igemm_dgrad_s2(const void* dy, const void* wpt, void* dx, int B, int H, int W, int Cin, int Cout, int fmt,
                                    hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    return igemm_launch_mode(dy, wpt, nullptr, nullptr, nullptr, nullptr, 0.0f, nullptr, 0, dx, (int64_t)B * H * W, Cout, Cin, 9,
                             H, W, -2, 1, fmt, 0, (hipStream_t)stream, nullptr, nullptr, 0, 0, 0);
}

// rows of the partial-sum buffer of hiast_igemm_dgrad_bn_stats: one per block row of the tile form the launch takes
extern "C" int hiast_igemm_dgrad_bn_stats_rows(int64_t M, int Cin, int Cout, int taps)
{
    if (M <= 0) return 0;
    const int bm = ig_block_rows(M, Cin, Cout, taps, 0);
    return (int)((M + bm - 1) / bm);
}

extern "C" int hiast_igemm_bn_act(const void* x, const void* wp, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, const void* res, int relu, void* y,
                                  int B, int H, int W, int Cin, int Cout, int taps, int stride, int dil, int fmt,
                                  int out_f32, float* stats, int stats_rows, const void* res_gate, int gate_mask,
                                  hiast_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (taps == 1 && stride != 1) return HIAST_E_RANGE;       // strided 1x1: subsample the input first
    const int Ho = taps == 1 ? H : (H - 1) / stride + 1, Wo = taps == 1 ? W : (W - 1) / stride + 1;
    return hiast_igemm_launch(x, wp, gamma, beta, mean, var, eps, res, relu, y, (int64_t)B * Ho * Wo, Cin, Cout, taps, H,
                              W, stride, dil, fmt, out_f32, (hipStream_t)stream, stats, res_gate, gate_mask, stats_rows);
}

#ifdef IG_STAMP
namespace hiast {
__global__ void ig_stamp_copy_kernel(unsigned* dst)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 1024 * 8 * 8 + 32 * 64; i += gridDim.x * blockDim.x) dst[i] = ig_stamp_buf[i];
}
}  // namespace hiast
extern "C" int hiast_igemm_debug_stamps(unsigned* dst_device)
{
    hipLaunchKernelGGL(hiast::ig_stamp_copy_kernel, dim3(64), dim3(256), 0, 0, dst_device);
    return (int)hipDeviceSynchronize();
}
#endif

// rows of the partial-sum buffer of a want_stats launch.  Such a launch is always the PLAIN convolution: igemm_launch_mode
// rejects stats together with a BatchNorm, a residual or a ReLU (HIAST_E_RANGE), so the kernel choice below (xconv for the
// expanding 1x1 shapes, the tile kernel otherwise) is the one the launch makes.
extern "C" int hiast_igemm_stats_rows(int64_t M, int Cin, int Cout, int taps, int fmt)
{
    const int planes = hiast_fmt_planes(fmt);
    if (M <= 0 || Cin <= 0 || Cout <= 0) return 0;
    if (hiast_xconv_ok(M, Cin, Cout, planes, taps, 0, 0, 0, 0, 0, 0, 1)) return hiast_xconv_stats_rows(M, Cout);
    const int bm = ig_block_rows(M, Cin, Cout, taps, 0);
    return (int)((M + bm - 1) / bm);
}

extern "C" int hiast_pack_conv_weight(const float* w, int N, int K, int taps, int fmt, int transpose, void* wp,
                                      void* wpt, hiast_stream_t stream)
{
    if (!w || !wp) return HIAST_E_ARG;
    if (N <= 0 || K <= 0 || taps <= 0) return HIAST_E_ARG;
    const int planes = fmt;                              // 1 bf16 rows | 2 split-bf16 planes | 3 fp16 rows
    if (!hiast_fmt_ok(fmt) || transpose < 0 || transpose > 2) return HIAST_E_RANGE;
    if (transpose == 2 && !wpt) return HIAST_E_ARG;
    const long long total = (long long)N * K * taps;
    if (N % 64 == 0 && K % 64 == 0 && taps <= 9 && !((((uintptr_t)wp) | ((uintptr_t)wpt)) & 15)) {
        const dim3 tg((unsigned)((N / 64) * (K / 64)));
        const size_t lds_bytes = (size_t)32 * taps * hiast::PK_PITCH * sizeof(unsigned short);
        if (planes == HIAST_FMT_SPLIT_BF16)
            hipLaunchKernelGGL(hiast::pack_conv_weight_tiled_kernel<2>, tg, dim3(256), lds_bytes, (hipStream_t)stream, w,
                               (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
        else if (planes == HIAST_FMT_FP16)
            hipLaunchKernelGGL((hiast::pack_conv_weight_tiled_kernel<1, true>), tg, dim3(256), lds_bytes, (hipStream_t)stream, w,
                               (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
        else
            hipLaunchKernelGGL(hiast::pack_conv_weight_tiled_kernel<1>, tg, dim3(256), lds_bytes, (hipStream_t)stream, w,
                               (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
        HIAST_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (planes == HIAST_FMT_SPLIT_BF16)
        hipLaunchKernelGGL(hiast::pack_conv_weight_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
    else if (planes == HIAST_FMT_FP16)
        hipLaunchKernelGGL((hiast::pack_conv_weight_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
    else
        hipLaunchKernelGGL(hiast::pack_conv_weight_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned short*)wp, (unsigned short*)wpt, N, K, taps, transpose);
    HIAST_CHECK_LAUNCH();
    return 0;
}

namespace hiast {

// K9f: BN(eval) + ReLU + MaxPool2d(3, stride 2, padding 1) of the library stem's output, written in the operand format of
// the trunk kernels — one pass instead of three (BN + ReLU, torch's pooling kernel, split / cast).  Thread = 8 channels of an
// output pixel (its nine inputs: the 8 lanes of a pixel read one contiguous run); scale / shift are derived once per thread
// (the channel group of a thread is fixed along its grid-stride walk: the stride is a multiple of C / 8).
// INK: 0 fp32 input, 1 bf16, 2 fp16; F16 (PL = 1): fp16 output rows
template <int INK, int PL, bool F16 = false>
__global__ __launch_bounds__(256) void stem_tail_kernel(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean,
                                                        const float* __restrict__ var, float eps,
                                                        unsigned short* __restrict__ out, int B, int H, int W, int C, int Ho,
                                                        int Wo)
{
    const int G = C >> 3;
    const long long total = (long long)B * Ho * Wo * G;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cg = (int)(i % G);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cg * 8 + k;
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        sc[k] = (gamma ? gamma[c] : 1.0f) * invstd;
        sh[k] = fmaf(-mean[c], sc[k], beta ? beta[c] : 0.0f);
    }
    for (; i < total; i += stride) {
        const long long pix = i / G;
        const int xo = (int)(pix % Wo);
        const int yo = (int)((pix / Wo) % Ho);
        const int b = (int)(pix / ((long long)Wo * Ho));
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * yo - 1 + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo - 1 + dx;
                if (xx < 0 || xx >= W) continue;
                const size_t off = (((size_t)b * H + yy) * W + xx) * C + cg * 8;
                float v[8];
                if (INK != 0) {
                    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(xv) + off);
                    const unsigned w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = H16<INK == 2>::lo(w4[q]);
                        v[2 * q + 1] = H16<INK == 2>::hi(w4[q]);
                    }
                } else {
                    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + off);
                    const float4 c4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + off + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float t = fmaf(v[k], sc[k], sh[k]);
                    t = t > 0.f ? t : 0.f;
                    if (INK != 0) t = H16<INK == 2>::dec(H16<INK == 2>::enc(t));   // the 16-bit activation the pooling kernel saw
                    m[k] = t > m[k] ? t : m[k];
                }
            }
        }
        unsigned ph[4], pl_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h0, l0 = 0, h1, l1 = 0;
            if (PL == 2) {
                ig_split(m[2 * q], h0, l0);
                ig_split(m[2 * q + 1], h1, l1);
            } else {
                h0 = H16<F16>::enc(m[2 * q]);
                h1 = H16<F16>::enc(m[2 * q + 1]);
            }
            ph[q] = (unsigned)h0 | ((unsigned)h1 << 16);
            pl_[q] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* dst = out + ig_elem<PL>((size_t)pix, cg * 8, C);
        *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        if (PL == 2) *reinterpret_cast<uint4*>(dst + 32) = make_uint4(pl_[0], pl_[1], pl_[2], pl_[3]);
    }
}

}  // namespace hiast

extern "C" int hiast_stem_tail(const void* x, int dtype, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, void* out, int fmt, int B, int H, int W, int C,
                               hiast_stream_t stream)
{
    if (!hiast_fmt_ok(fmt)) return HIAST_E_RANGE;
    const int planes = hiast_fmt_planes(fmt), f16 = fmt == HIAST_FMT_FP16;
    if (!x || !mean || !var || !out) return HIAST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return HIAST_E_ARG;
    if (dtype < 0 || dtype > 2 || C % 8 != 0 || (planes == 2 && C % 32 != 0) ||
        ((((uintptr_t)x) | ((uintptr_t)out)) & 15))
        return HIAST_E_RANGE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, G = C / 8;
    const long long total = (long long)B * Ho * Wo * G;
    if ((long long)B * H * W * C >= (1ll << 40)) return HIAST_E_RANGE;
    long long nb = (total + 255) / 256;
    int grid = (int)(nb > 16384 ? 16384 : nb);
    {   // grid-stride = multiple of G, so that a thread keeps its channel group
        long long a = 256, g = G;
        while (a) { const long long t = g % a; g = a; a = t; }
        const int need = (int)(G / g);
        if (grid >= need) grid -= grid % need;
        else return HIAST_E_RANGE;
    }
    hipStream_t st = (hipStream_t)stream;
#define L(BF, PL)                                                                                                  \
    hipLaunchKernelGGL((hiast::stem_tail_kernel<BF, PL>), dim3(grid), dim3(256), 0, st, x, gamma, beta, mean, var, eps, \
                       (unsigned short*)out, B, H, W, C, Ho, Wo)
#define LF(BF)                                                                                                     \
    hipLaunchKernelGGL((hiast::stem_tail_kernel<BF, 1, true>), dim3(grid), dim3(256), 0, st, x, gamma, beta, mean, var, eps, \
                       (unsigned short*)out, B, H, W, C, Ho, Wo)
    if (f16) { if (dtype == 2) LF(2); else if (dtype == 1) LF(1); else LF(0); }
    else if (dtype == 2) { if (planes == 2) L(2, 2); else L(2, 1); }
    else if (dtype == 1) { if (planes == 2) L(1, 2); else L(1, 1); }
    else { if (planes == 2) L(0, 2); else L(0, 1); }
#undef LF
#undef L
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_split_planes(float* x, void* planes, int64_t M, int C, int inverse, hiast_stream_t stream)
{
    if (!x || !planes) return HIAST_E_ARG;
    if (M <= 0 || C <= 0) return HIAST_E_ARG;
    if (C % 32 != 0 || ((((uintptr_t)x) | ((uintptr_t)planes)) & 15)) return HIAST_E_RANGE;
    const long long total8 = (long long)M * C / 8;
    long long nb = (total8 + 255) / 256;
    const int grid = (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
    if (inverse)
        hipLaunchKernelGGL(hiast::from_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)planes, x, (long long)M, C);
    else
        hipLaunchKernelGGL(hiast::to_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (unsigned short*)planes, (long long)M, C);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_pack_conv_weight_multi(const hiast_pack_rec* table, const int32_t* chunk_tensor,
                                            const int64_t* chunk_start, int n_chunks, int max_taps, hiast_stream_t stream)
{
    if (!table || !chunk_tensor || !chunk_start) return HIAST_E_ARG;
    if (n_chunks <= 0 || max_taps <= 0) return HIAST_E_ARG;
    if (max_taps > 9) return HIAST_E_RANGE;
    const size_t lds_bytes = (size_t)32 * max_taps * hiast::PK_PITCH * sizeof(unsigned short);
    hipLaunchKernelGGL(hiast::pack_conv_weight_multi_kernel, dim3(n_chunks), dim3(256), lds_bytes, (hipStream_t)stream, table,
                       chunk_tensor, chunk_start);
    HIAST_CHECK_LAUNCH();
    return 0;
}
