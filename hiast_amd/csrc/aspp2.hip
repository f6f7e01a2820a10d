// K1b — ASPP head on channels-last activations as ONE GEMM + a 33-tap shift-add
// (reference: ASPP_V2.forward, sseg/models/modules/seg_models/deeplab_v2.py:20-24: four dilated 3x3
// cuDNN convolutions over the 2048-channel trunk feature + three adds).
//
// With only Cout = 19 output channels the four convolutions are 33 distinct taps (the four centre taps
// read the same pixel and are pre-summed) x 19 channels = 627 dot products of length Cin per INPUT pixel:
//     T[p][tap*Cout + co] = Σ_ci X[p][ci] * W[tap][co][ci]            (a plain [M x Cin] x [Cin x NP] GEMM)
//     y[co][q]            = bias[co] + Σ_tap T[q + off(tap)][tap*Cout+co]   (zero outside the image)
// so the 67 MB/image feature map is read ONCE (the direct form of aspp.hip re-reads it 33 times through
// L2 and is bound by exact-fp32 MFMA).  The GEMM runs on the bf16 matrix cores: plain bf16 for the
// training step (the reference trains under apex O1, i.e. half-precision convolutions with fp32
// accumulation), split-bf16 (hi*hi + hi*lo + lo*hi, fp32-class) for the fp32 pseudo-label forward.
//
// Backward (bf16 activations, fp32 accumulation, fp32 weight gradients):
//     G[q][tap*Cout+co] = dY[co][q - off(tap)]                               (gather, bf16)
//     dX[q][ci]  = Σ_n G[q][n] * W[n][ci]                                     (GEMM, K = NP)
//     dWt[n][ci] = Σ_q G[q][n] * X[q][ci]                                     (GEMM reducing over pixels:
//                  both operands are pixel-major, so their MFMA fragments are read from LDS with the
//                  transposing ds_read_b64_tr_b16; split over pixel ranges, fixed-order reduce)
#include <hip/hip_bf16.h>

#include "common.h"

int hiast_gemm_nt_launch(const void* x, const float* w, void* y, int64_t M, int K, int N, int dtype, hipStream_t st);
int hiast_igemm_launch(const void* x, const void* wp, const float* gamma, const float* beta, const float* mean,
                       const float* var, float eps, const void* res, int relu, void* y, int64_t M, int K, int N,
                       int taps, int H, int W, int stride, int dil, int planes, int out_f32, hipStream_t st,
                       float* stats, const void* res_gate, int gate_mask, int stats_rows);

namespace hiast {

constexpr int A2_NTAP = 33;

struct Taps2 {
    int dy[A2_NTAP];
    int dx[A2_NTAP];
};

static Taps2 make_taps2(const int* dil)
{
    Taps2 t;
    t.dy[0] = 0; t.dx[0] = 0;
    int k = 1;
    for (int d = 0; d < 4; ++d)
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                if (ky == 1 && kx == 1) continue;
                t.dy[k] = (ky - 1) * dil[d];
                t.dx[k] = (kx - 1) * dil[d];
                ++k;
            }
    return t;
}

__device__ __forceinline__ float aspp2_weight(const float* w0, const float* w1, const float* w2, const float* w3,
                                              int Cin, int tap, int co, int ci)
{
    const size_t base = ((size_t)co * Cin + ci) * 9;
    if (tap == 0) return ((w0[base + 4] + w1[base + 4]) + w2[base + 4]) + w3[base + 4];
    const int d = (tap - 1) >> 3;
    int k = (tap - 1) & 7;
    k += (k >= 4) ? 1 : 0;                       // skip the centre position
    const float* wd = d == 0 ? w0 : (d == 1 ? w1 : (d == 2 ? w2 : w3));
    return wd[base + k];
}

// wt[NP][Cin] (row n = tap*Cout + co; rows >= 33*Cout are zero), wd[Cin][NP] = its transpose, bias[Cout]
__global__ __launch_bounds__(256) void aspp2_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ w3,
                                                         const float* __restrict__ b0, const float* __restrict__ b1,
                                                         const float* __restrict__ b2, const float* __restrict__ b3,
                                                         int Cin, int Cout, int NP, float* __restrict__ wt,
                                                         float* __restrict__ wd, float* __restrict__ bias)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < Cout) bias[idx] = ((b0[idx] + b1[idx]) + b2[idx]) + b3[idx];
    if (idx >= (long long)NP * Cin) return;
    const int n = (int)(idx / Cin), ci = (int)(idx - (long long)n * Cin);
    float v = 0.f;
    if (n < A2_NTAP * Cout) v = aspp2_weight(w0, w1, w2, w3, Cin, n / Cout, n % Cout, ci);
    wt[idx] = v;
    if (wd) wd[(size_t)ci * NP + n] = v;
}

// y[b][co][q] = bias[co] + Σ_tap T[b, q + off(tap)][tap*Cout + co], taps in ascending order.
// Block = 64 consecutive pixels of one image; a 32-lane half-wave owns one pixel at a time (lane = co), so a
// tap read is one contiguous Cout*4-byte segment; results go through LDS so that the NCHW store is
// 64 consecutive pixels per channel.
// TT: 0 = T is fp32 (fp32 / bf16 rows, split planes), 2 = fp16: the fp16 forward stores the tap products in the type of its
// activations (half the bytes of the head's largest tensor, written and read once; the reference's apex-O1 head rounds each
// dilated convolution's output to fp16 and adds the four in fp16).
template <int TT>
__global__ __launch_bounds__(256) void aspp2_shift_add_kernel(const void* __restrict__ Tv,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int h, int w, int Cout, int NP, Taps2 taps)
{
    const float* T = reinterpret_cast<const float*>(Tv);
    const unsigned short* T16 = reinterpret_cast<const unsigned short*>(Tv);
    __shared__ float s[64][33];
    const int hw = h * w;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const int co = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t tb = (size_t)b * hw * NP;
    const float bv = co < Cout ? bias[co] : 0.f;
    for (int i = grp; i < 64; i += 8) {
        const int q = p0 + i;
        float acc = bv;
        if (q < hw && co < Cout) {
            const int qy = q / w, qx = q - qy * w;
            // branch-free, 11 taps per group: the loads of a group are all issued before the first is used (with a branch per
            // tap and three taps in flight the launch moved its 168 MB at 1.2 TB/s); same order of additions as before
#pragma unroll 11
            for (int t = 0; t < A2_NTAP; ++t) {
                const int yy = qy + taps.dy[t], xx = qx + taps.dx[t];
                const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
                const size_t at = tb + (size_t)(in ? yy * w + xx : q) * NP + t * Cout + co;
                const float v = TT == 0 ? T[at] : (TT == 1 ? H16<false>::dec(T16[at]) : H16<true>::dec(T16[at]));
                acc = in ? acc + v : acc;
            }
        }
        s[i][co] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * Cout; e += 256) {
        const int c = e >> 6, i = e & 63;
        if (p0 + i < hw) y[((size_t)b * Cout + c) * hw + p0 + i] = s[i][c];
    }
}

// G[q][n = tap*Cout+co] = bf16(dY[b][co][q - off(tap)]) (0 outside the image and for n >= 33*Cout);
// one thread = 8 consecutive n of one pixel (one 16-byte store).
template <bool F16>
__global__ __launch_bounds__(256) void aspp2_gather_kernel(const float* __restrict__ dy,
                                                           unsigned short* __restrict__ G, int B, int h, int w,
                                                           int Cout, int NP, Taps2 taps)
{
    const int hw = h * w;
    const int per_row = NP / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * hw * per_row) return;
    const long long pq = idx / per_row;
    const int n0 = (int)(idx - pq * per_row) * 8;
    const int b = (int)(pq / hw), q = (int)(pq - (long long)b * hw);
    const int qy = q / w, qx = q - qy * w;
    const float* dyb = dy + (size_t)b * Cout * hw;
    unsigned pk[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        unsigned short v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + j + u;
            float f = 0.f;
            if (n < A2_NTAP * Cout) {
                const int t = n / Cout, co = n - t * Cout;
                const int yy = qy - taps.dy[t], xx = qx - taps.dx[t];
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) f = dyb[(size_t)co * hw + yy * w + xx];
            }
            v[u] = H16<F16>::enc(f);
        }
        pk[j >> 1] = (unsigned)v[0] | ((unsigned)v[1] << 16);
    }
    *reinterpret_cast<uint4*>(G + (size_t)pq * NP + n0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

// ------------------------------------------------------------------------------------------------
// P[s][i][j] = Σ_{m in split s} A[m][i] * Bm[m][j]      A: [M][I] bf16, Bm: [M][J] bf16, P fp32
// Block tile 128(i) x 128(j), 4 waves as 2 x 2 (64 x 64 each, v_mfma_f32_32x32x16_bf16), k-step = 32 rows
// of m.  Both tiles are stored in LDS as they arrive ([m][128] rows of 256 B, 16-byte chunks XOR-swizzled
// so that both the staging stores and the transposing fragment reads are conflict free) and every MFMA
// fragment (8 consecutive m for one i / j) is two ds_read_b64_tr_b16.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int tn_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <bool F16>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const unsigned short* __restrict__ A,
                                                           const unsigned short* __restrict__ Bm,
                                                           float* __restrict__ P, int M, int I, int J,
                                                           int m_per_split)
{
    constexpr int TILE_BYTES = 32 * 256;                 // [32 m][128] bf16
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // [buf][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.x * 128, j0 = blockIdx.y * 128, split = blockIdx.z;
    const int m_begin = split * m_per_split;
    const int m_end = (m_begin + m_per_split < M) ? m_begin + m_per_split : M;
    const int nk = (m_end - m_begin + 31) / 32;

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging: chunk f = tid + 256*u (u = 0,1): row = f >> 4 (0..31), ch = f & 15
    uint4 ra[2], rb[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = tid + 256 * u, row = f >> 4, ch = f & 15;
            const int m = m_begin + kt * 32 + row;
            const int mc = m < m_end ? m : m_end - 1;      // valid address; zeroed below
            uint4 va = *reinterpret_cast<const uint4*>(A + (size_t)mc * I + i0 + ch * 8);
            uint4 vb = *reinterpret_cast<const uint4*>(Bm + (size_t)mc * J + j0 + ch * 8);
            if (m >= m_end) { va = make_uint4(0, 0, 0, 0); vb = va; }
            ra[u] = va;
            rb[u] = vb;
        }
    };
    auto lds_store = [&](int buf) {
        unsigned char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = tid + 256 * u, row = f >> 4, ch = f & 15;
            *reinterpret_cast<uint4*>(base + tn_off(row, ch)) = ra[u];
            *reinterpret_cast<uint4*>(base + TILE_BYTES + tn_off(row, ch)) = rb[u];
        }
    };
    // transposed fragment read: 32 columns starting at c0 (a multiple of 32), k rows kb .. kb+7 for lane half hh
    const int grp = lane >> 4, t16 = lane & 15;
    const int q = t16 >> 2, p = t16 & 3;
    auto frag = [&](const unsigned char* tile, int c0, int kk) -> bf16x8_t {
        const int kb = kk * 16 + 8 * (grp >> 1);
        const int ch = ((c0 + 16 * (grp & 1)) >> 3) + (p >> 1);
        const int o0 = tn_off(kb + q, ch) + 8 * (p & 1);
        const int o1 = tn_off(kb + 4 + q, ch) + 8 * (p & 1);
        typedef s16x4_t __attribute__((address_space(3))) * lds_p;
        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + o0));
        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + o1));
        s16x8_t v;
        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
        v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        return __builtin_bit_cast(bf16x8_t, v);
    };

    if (nk > 0) {
        gload(0);
        lds_store(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* ta = smem + buf * 2 * TILE_BYTES;
        const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = frag(ta, wm * 64 + a * 32, kk);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = frag(tb, wn * 64 + b * 32, kk);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = H16<F16>::mfma32(fa[a], fb[b], acc[a][b]);
        }
        if (kt + 1 < nk) lds_store(buf ^ 1);
        __syncthreads();
    }

    float* Ps = P + (size_t)split * I * J;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = j0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Ps[(size_t)i * J + j] = acc[a][b][r];
            }
        }
}

// dW_d[co][ci][ky][kx] = Σ_s P[s][tap(d,ky,kx)*Cout + co][ci] (ascending s); thread = (n, ci), ci fastest
__global__ __launch_bounds__(256) void aspp2_wgrad_unpack_kernel(const float* __restrict__ P,
                                                                 float* __restrict__ dw0, float* __restrict__ dw1,
                                                                 float* __restrict__ dw2, float* __restrict__ dw3,
                                                                 int Cin, int Cout, int NP, int nsplit)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)A2_NTAP * Cout * Cin) return;
    const int n = (int)(idx / Cin), ci = (int)(idx - (long long)n * Cin);
    const int tap = n / Cout, co = n - tap * Cout;
    float acc = 0.f;
    for (int s = 0; s < nsplit; ++s) acc += P[((size_t)s * NP + n) * Cin + ci];
    const size_t base = ((size_t)co * Cin + ci) * 9;
    if (tap == 0) {
        dw0[base + 4] = acc; dw1[base + 4] = acc; dw2[base + 4] = acc; dw3[base + 4] = acc;
    } else {
        const int d = (tap - 1) >> 3;
        int k = (tap - 1) & 7;
        k += (k >= 4) ? 1 : 0;
        float* dw = d == 0 ? dw0 : (d == 1 ? dw1 : (d == 2 ? dw2 : dw3));
        dw[base + k] = acc;
    }
}

// db[co] = Σ_b Σ_p dY[b][co][p]  (one block of 1024 threads per output channel, fixed order, double accumulation;
// 16-byte loads, four independent partial sums per thread: the first form — 256 threads, one float per dependent double
// add — took 102 us for the 5 MB of an 8-image batch)
__global__ __launch_bounds__(1024) void aspp2_db_kernel(const float* __restrict__ dy, float* __restrict__ db, int B,
                                                        int Cout, int hw)
{
    __shared__ double s[1024];
    const int co = blockIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const bool vec = (hw & 3) == 0 && (((uintptr_t)dy) & 15) == 0;
    for (int n = 0; n < B; ++n) {
        const float* p = dy + ((size_t)n * Cout + co) * hw;
        if (vec) {
            for (int i = threadIdx.x * 4; i < hw; i += 4096) {
                const float4 v = *reinterpret_cast<const float4*>(p + i);
                a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
            }
        } else {
            for (int i = threadIdx.x; i < hw; i += 1024) a0 += (double)p[i];
        }
    }
    s[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = (float)s[0];
}

static int wgrad2_nsplit(long long M)
{
    int s = 16;
    while (s > 1 && M / s < 1024) s >>= 1;
    return s;
}

}  // namespace hiast

static int aspp2_np(int Cout) { return ((hiast::A2_NTAP * Cout + 127) / 128) * 128; }

static int aspp2_check(int B, int Cin, int h, int w, int Cout)
{
    if (B <= 0 || Cin <= 0 || h <= 0 || w <= 0 || Cout <= 0) return HIAST_E_ARG;
    if (Cin % 128 != 0 || Cout > 32 || B > 65535) return HIAST_E_RANGE;
    if ((long long)B * h * w * (long long)(Cin > aspp2_np(Cout) ? Cin : aspp2_np(Cout)) >= (1ll << 40)) return HIAST_E_RANGE;
    return 0;
}

extern "C" int hiast_aspp2_np(int Cout) { return Cout > 0 && Cout <= 32 ? aspp2_np(Cout) : 0; }

extern "C" size_t hiast_aspp2_workspace_bytes(int B, int Cin, int h, int w, int Cout, int backward)
{
    if (aspp2_check(B, Cin, h, w, Cout)) return 0;
    const size_t M = (size_t)B * h * w, NP = aspp2_np(Cout);
    const size_t fwd = M * NP * sizeof(float);                                    // T
    if (!backward) return fwd + 256;
    const size_t g = M * NP * 2;                                                  // G (bf16)
    const size_t part = (size_t)hiast::wgrad2_nsplit((long long)M) * NP * Cin * sizeof(float);
    const size_t bwd = ((g + 255) / 256) * 256 + part;
    return (fwd > bwd ? fwd : bwd) + 256;
}

extern "C" int hiast_aspp2_pack_weights(const float* w0, const float* w1, const float* w2, const float* w3,
                                        const float* b0, const float* b1, const float* b2, const float* b3,
                                        int Cin, int Cout, float* wt, float* wd, float* bias,
                                        hiast_stream_t stream)
{
    if (!w0 || !w1 || !w2 || !w3 || !b0 || !b1 || !b2 || !b3 || !wt || !bias) return HIAST_E_ARG;
    int e = aspp2_check(1, Cin, 1, 1, Cout);
    if (e) return e;
    const int NP = aspp2_np(Cout);
    const long long total = (long long)NP * Cin;
    hipLaunchKernelGGL(hiast::aspp2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w0, w1, w2, w3, b0, b1, b2, b3, Cin, Cout, NP, wt, wd, bias);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_aspp2_fwd(const void* x_nhwc, int dtype, const void* wt, const float* bias, float* y, int B,
                               int Cin, int h, int w, int Cout, const int* dil, void* workspace,
                               size_t workspace_bytes, hiast_stream_t stream)
{
    if (!x_nhwc || !wt || !bias || !y || !dil || !workspace) return HIAST_E_ARG;
    int e = aspp2_check(B, Cin, h, w, Cout);
    if (e) return e;
    if (dtype < 0 || dtype > 3) return HIAST_E_RANGE;      // 0 fp32 rows | HIAST_FMT_BF16 | _SPLIT_BF16 | _FP16
    const int NP = aspp2_np(Cout);
    const long long M = (long long)B * h * w;
    if (workspace_bytes < (size_t)M * NP * sizeof(float)) return HIAST_E_WS;
    if ((((uintptr_t)x_nhwc) | ((uintptr_t)workspace) | ((uintptr_t)wt)) & 15) return HIAST_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    float* T = (float*)workspace;
    // fp16 rows (the reference's apex-O1 type): the tap products are stored as fp16 (HIAST_ASPP_T32=1: fp32, the round-2
    // form).  bf16 rows keep the fp32 T: 8 significant bits per tap product would double the error of that forward.
    static const bool t32 = [] { const char* v = getenv("HIAST_ASPP_T32"); return v && atoi(v) != 0; }();
    const bool t16 = dtype == HIAST_FMT_FP16 && !t32;
    if (dtype == 0)      // fp32 rows, fp32 weights: the register-staged kernel splits on the fly
        e = hiast_gemm_nt_launch(x_nhwc, (const float*)wt, T, M, Cin, NP, 0, st);
    else                 // bf16 rows (1) / split planes (2) / fp16 rows (3), weights packed by hiast_pack_conv_weight: LDS-DMA kernel
        e = hiast_igemm_launch(x_nhwc, wt, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, T, M, Cin, NP, 1, 0, 0,
                               1, 1, dtype, t16 ? 0 : 1, st, nullptr, nullptr, 0, 0);
    if (e) return e;
    const hiast::Taps2 taps = hiast::make_taps2(dil);
    const dim3 grid((h * w + 63) / 64, B);
    if (!t16)
        hipLaunchKernelGGL(hiast::aspp2_shift_add_kernel<0>, grid, dim3(256), 0, st, T, bias, y, h, w, Cout, NP, taps);
    else
        hipLaunchKernelGGL(hiast::aspp2_shift_add_kernel<2>, grid, dim3(256), 0, st, T, bias, y, h, w, Cout, NP, taps);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_aspp2_bwd(const void* x_nhwc, const float* dy, const void* wd, void* dx_nhwc, float* dw0,
                               float* dw1, float* dw2, float* dw3, float* db, int B, int Cin, int h, int w, int Cout,
                               const int* dil, int fmt, void* workspace, size_t workspace_bytes, hiast_stream_t stream)
{
    if (fmt != HIAST_FMT_BF16 && fmt != HIAST_FMT_FP16) return HIAST_E_RANGE;
    const bool f16 = fmt == HIAST_FMT_FP16;
    if (!dy || !dil || !workspace) return HIAST_E_ARG;
    const bool want_w = dw0 || dw1 || dw2 || dw3 || db;
    if (want_w && (!x_nhwc || !dw0 || !dw1 || !dw2 || !dw3 || !db)) return HIAST_E_ARG;
    if (dx_nhwc && !wd) return HIAST_E_ARG;
    int e = aspp2_check(B, Cin, h, w, Cout);
    if (e) return e;
    const int NP = aspp2_np(Cout);
    const long long M = (long long)B * h * w;
    if (workspace_bytes < hiast_aspp2_workspace_bytes(B, Cin, h, w, Cout, 1) - 256) return HIAST_E_WS;
    if ((((uintptr_t)x_nhwc) | ((uintptr_t)workspace) | ((uintptr_t)wd) | ((uintptr_t)dx_nhwc)) & 15) return HIAST_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    const hiast::Taps2 taps = hiast::make_taps2(dil);
    unsigned short* G = (unsigned short*)workspace;
    const size_t g_bytes = (((size_t)M * NP * 2 + 255) / 256) * 256;
    float* P = (float*)((unsigned char*)workspace + g_bytes);
    const long long gthreads = M * (NP / 8);
    if (f16)
        hipLaunchKernelGGL(hiast::aspp2_gather_kernel<true>, dim3((unsigned)((gthreads + 255) / 256)), dim3(256), 0, st, dy, G,
                           B, h, w, Cout, NP, taps);
    else
        hipLaunchKernelGGL(hiast::aspp2_gather_kernel<false>, dim3((unsigned)((gthreads + 255) / 256)), dim3(256), 0, st, dy, G,
                           B, h, w, Cout, NP, taps);
    HIAST_CHECK_LAUNCH();
    if (dx_nhwc) {
        // dX[M][Cin] = G[M][NP] * wd[Cin][NP]^T, wd packed bf16 (hiast_pack_conv_weight, planes = 1)
        e = hiast_igemm_launch(G, wd, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, dx_nhwc, M, NP, Cin, 1, 0, 0, 1,
                               1, fmt, 0, st, nullptr, nullptr, 0, 0);
        if (e) return e;
    }
    if (want_w) {
        const int nsplit = hiast::wgrad2_nsplit(M);
        int mps = (int)((M + nsplit - 1) / nsplit);
        mps = ((mps + 31) / 32) * 32;
        dim3 grid(NP / 128, Cin / 128, nsplit);
        if (f16)
            hipLaunchKernelGGL(hiast::gemm_tn_bf16_kernel<true>, grid, dim3(256), 0, st, G, (const unsigned short*)x_nhwc, P,
                               (int)M, NP, Cin, mps);
        else
            hipLaunchKernelGGL(hiast::gemm_tn_bf16_kernel<false>, grid, dim3(256), 0, st, G, (const unsigned short*)x_nhwc, P,
                               (int)M, NP, Cin, mps);
        HIAST_CHECK_LAUNCH();
        const long long total = (long long)hiast::A2_NTAP * Cout * Cin;
        hipLaunchKernelGGL(hiast::aspp2_wgrad_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P,
                           dw0, dw1, dw2, dw3, Cin, Cout, NP, nsplit);
        HIAST_CHECK_LAUNCH();
        hipLaunchKernelGGL(hiast::aspp2_db_kernel, dim3(Cout), dim3(1024), 0, st, dy, db, B, Cout, h * w);
        HIAST_CHECK_LAUNCH();
    }
    return 0;
}
