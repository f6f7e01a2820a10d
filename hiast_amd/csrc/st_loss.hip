// K5-K8 — fused self-training loss on LOW-RES logits (reference: SelfTrainingSegmentor.compute_loss,
// sseg/models/segmentors/self_training_segmentor.py:30-53,128-163; losses.py:32-65,75-89; the
// bilinear upsample of forward() :27 and the teacher softmax of
// workflows/trainer/consistency_self_training_trainer.py:113-119 are recomputed in registers).
//
// Thread = one output column X of one band (the output rows sharing source rows j, j+1): the
// horizontally-lerped top/bottom values of all C classes (student and teacher) stay in VGPRs
// while the thread walks down the band.  No full-resolution tensor is read or written: per image
// the kernels touch the two low-res logit maps, the label map and (bwd) the low-res gradient.
//
// fwd: per-block partial sums (double) -> fixed-order finalize  => bitwise reproducible.
// bwd: per-thread column gradients are folded into low-res cells through LDS in a fixed order,
//      written as per-block partial tiles, and summed by a combine kernel => no float atomics,
//      bitwise reproducible.
#include "common.h"

namespace hiast {

constexpr int LOSS_THREADS = 256;

template <typename LT>
__device__ __forceinline__ int load_label(const LT* p, size_t i) { return (int)p[i]; }

// ------------------------------------------------------------------------------------------ fwd
template <int C, bool TEACHER, typename LT>
__global__ __launch_bounds__(LOSS_THREADS) void st_loss_fwd_kernel(
    const float* __restrict__ zs_lr, const float* __restrict__ zt_lr, const LT* __restrict__ plbl,
    int h, int w, int H, int W, float sh, float sw, int region, double* __restrict__ partial)
{
    const int b = blockIdx.z, j = blockIdx.y;
    const int X = blockIdx.x * LOSS_THREADS + threadIdx.x;
    const int Y0 = band_start(sh, j, h, H), Y1 = band_start(sh, j + 1, h, H);
    const bool live = X < W && Y0 < Y1;
    const int Xc = X < W ? X : W - 1;
    const Src sx = src_of(sw, Xc, w);
    const int y0 = j, y1 = j + (j < h - 1 ? 1 : 0);

    float st[C], sb[C], tt[TEACHER ? C : 1], tb[TEACHER ? C : 1];
    {
        const float* base = zs_lr + (size_t)b * C * h * w;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* p = base + (size_t)c * h * w;
            st[c] = lerp_h(p[y0 * w + sx.i0], p[y0 * w + sx.i1], sx.l0, sx.l1);
            sb[c] = lerp_h(p[y1 * w + sx.i0], p[y1 * w + sx.i1], sx.l0, sx.l1);
        }
        if (TEACHER) {
            const float* tbase = zt_lr + (size_t)b * C * h * w;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* p = tbase + (size_t)c * h * w;
                tt[c] = lerp_h(p[y0 * w + sx.i0], p[y0 * w + sx.i1], sx.l0, sx.l1);
                tb[c] = lerp_h(p[y1 * w + sx.i0], p[y1 * w + sx.i1], sx.l0, sx.l1);
            }
        }
    }

    float a_ce = 0.f, a_kld = 0.f, a_ent = 0.f, a_cst = 0.f;
    int n_conf = 0, n_ign = 0, n_cst = 0;
    const float invC = 1.0f / (float)C;
    if (live) {
        for (int Y = Y0; Y < Y1; ++Y) {
            const Src sy = src_of(sh, Y, h);
            const int y = load_label(plbl, ((size_t)b * H + Y) * W + X);
            const bool ign = (y == HIAST_IGNORE);
            float ms = 0.f, zy = 0.f, zsum = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float z = lerp_v(st[c], sb[c], sy.l0, sy.l1);
                ms = (c == 0 || z > ms) ? z : ms;
                zy = (c == y) ? z : zy;
                zsum += z;
            }
            float Ss = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) Ss += __expf(lerp_v(st[c], sb[c], sy.l0, sy.l1) - ms);
            const float lse = ms + __logf(Ss);
            if (!ign) {
                a_ce += lse - zy;
                a_kld += ((float)C * lse - zsum) * invC;
                ++n_conf;
            } else {
                ++n_ign;
            }
            const bool in_region = TEACHER && (region == 2 || (region == 0 ? ign : !ign));
            float mt = 0.f, invSt = 0.f;
            if (TEACHER && in_region) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float z = lerp_v(tt[c], tb[c], sy.l0, sy.l1);
                    mt = (c == 0 || z > mt) ? z : mt;
                }
                float St = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) St += __expf(lerp_v(tt[c], tb[c], sy.l0, sy.l1) - mt);
                invSt = 1.0f / St;
            }
            if (ign || in_region) {
                float e = 0.f, cs = 0.f;
                int cn = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float logp = lerp_v(st[c], sb[c], sy.l0, sy.l1) - lse;
                    if (ign) e -= __expf(logp) * logp;
                    if (TEACHER && in_region) {
                        const float q = __expf(lerp_v(tt[c], tb[c], sy.l0, sy.l1) - mt) * invSt;
                        const float prod = (-logp) * q;        // losses.py:61
                        cs += prod;
                        cn += (prod != 0.0f) ? 1 : 0;          // losses.py:89
                    }
                }
                a_ent += e;
                a_cst += cs;
                n_cst += cn;
            }
        }
    }

    // block reduction in double, fixed order
    double v[7] = {(double)a_ce, (double)a_kld, (double)a_ent, (double)a_cst,
                   (double)n_conf, (double)n_ign, (double)n_cst};
    __shared__ double s_red[LOSS_THREADS / 64][8];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double r = wave_sum_f64(v[k]);
        if (lane_id() == 0) s_red[threadIdx.x >> 6][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double r = 0.0;
        for (int wv = 0; wv < LOSS_THREADS / 64; ++wv) r += s_red[wv][threadIdx.x];
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk * 8 + threadIdx.x] = r;
    }
}

// sums[k] = Σ_blocks partial[blk][k], fixed association
__global__ __launch_bounds__(256) void st_loss_finalize_kernel(const double* __restrict__ partial,
                                                               int nblk, double* __restrict__ sums)
{
    __shared__ double s[256][8];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int k = 0; k < 7; ++k) acc[k] += partial[(size_t)i * 8 + k];
    for (int k = 0; k < 7; ++k) s[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 7; ++k) s[threadIdx.x][k] += s[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x < 8) sums[threadIdx.x] = threadIdx.x < 7 ? s[0][threadIdx.x] : 0.0;
}

// ------------------------------------------------------------------------------------------ bwd
// Block = the columns whose left source column x0 lies in cells [i0, i0+TI); partial tile
// out[b][c][j][r][xb][TI+1], r = 0: contribution to source row j, r = 1: to source row y1(j).
template <int C, bool TEACHER, typename LT>
__global__ __launch_bounds__(LOSS_THREADS) void st_loss_bwd_kernel(
    const float* __restrict__ zs_lr, const float* __restrict__ zt_lr, const LT* __restrict__ plbl,
    int h, int w, int H, int W, float sh, float sw, int region, int TI,
    const double* __restrict__ sums, const float* __restrict__ coef, float* __restrict__ tiles)
{
    __shared__ float s_g[2][C][LOSS_THREADS];
    __shared__ int s_x0[LOSS_THREADS];
    __shared__ float s_w1[LOSS_THREADS];

    const int b = blockIdx.z, j = blockIdx.y, xb = blockIdx.x, nxb = gridDim.x;
    const int i0 = xb * TI;
    const int i1 = (i0 + TI < w) ? i0 + TI : w;
    const int Xs = band_start(sw, i0, w, W), Xe = band_start(sw, i1, w, W);
    const int X = Xs + (int)threadIdx.x;
    const int Y0 = band_start(sh, j, h, H), Y1 = band_start(sh, j + 1, h, H);
    const bool live = X < Xe && Y0 < Y1;
    const int Xc = X < W ? X : W - 1;
    const Src sx = src_of(sw, Xc, w);
    const int y0 = j, y1 = j + (j < h - 1 ? 1 : 0);

    // loss normalisation (see hiast_st_loss_fwd): coef_i / denominator_i, 0 when coef_i == 0
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
    const float A1 = c0 == 0.f ? 0.f : (float)((double)c0 / sums[4]);
    const float A2 = c1 == 0.f ? 0.f : (float)((double)c1 / ((double)C * sums[4]));
    const float A3 = c2 == 0.f ? 0.f : (float)((double)c2 / ((double)C * sums[5]));
    const float A4 = (!TEACHER || c3 == 0.f) ? 0.f : (float)((double)c3 / sums[6]);

    float st[C], sb[C], tt[TEACHER ? C : 1], tb[TEACHER ? C : 1];
    float gt[C], gb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gt[c] = gb[c] = 0.f;
    {
        const float* base = zs_lr + (size_t)b * C * h * w;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* p = base + (size_t)c * h * w;
            st[c] = lerp_h(p[y0 * w + sx.i0], p[y0 * w + sx.i1], sx.l0, sx.l1);
            sb[c] = lerp_h(p[y1 * w + sx.i0], p[y1 * w + sx.i1], sx.l0, sx.l1);
        }
        if (TEACHER) {
            const float* tbase = zt_lr + (size_t)b * C * h * w;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* p = tbase + (size_t)c * h * w;
                tt[c] = lerp_h(p[y0 * w + sx.i0], p[y0 * w + sx.i1], sx.l0, sx.l1);
                tb[c] = lerp_h(p[y1 * w + sx.i0], p[y1 * w + sx.i1], sx.l0, sx.l1);
            }
        }
    }
    const float invC = 1.0f / (float)C;
    if (live) {
        for (int Y = Y0; Y < Y1; ++Y) {
            const Src sy = src_of(sh, Y, h);
            const int y = load_label(plbl, ((size_t)b * H + Y) * W + X);
            const bool ign = (y == HIAST_IGNORE);
            const float wconf = ign ? 0.f : 1.f, wign = ign ? 1.f : 0.f;
            const bool in_region = TEACHER && (region == 2 || (region == 0 ? ign : !ign));
            const float wreg = in_region ? 1.f : 0.f;
            float ms = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float z = lerp_v(st[c], sb[c], sy.l0, sy.l1);
                ms = (c == 0 || z > ms) ? z : ms;
            }
            float Ss = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) Ss += __expf(lerp_v(st[c], sb[c], sy.l0, sy.l1) - ms);
            const float lse = ms + __logf(Ss);
            float Hent = 0.f;
            if (ign) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float logp = lerp_v(st[c], sb[c], sy.l0, sy.l1) - lse;
                    Hent -= __expf(logp) * logp;
                }
            }
            float mt = 0.f, invSt = 0.f, Q = 0.f;
            if (TEACHER && in_region) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float z = lerp_v(tt[c], tb[c], sy.l0, sy.l1);
                    mt = (c == 0 || z > mt) ? z : mt;
                }
                float St = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) St += __expf(lerp_v(tt[c], tb[c], sy.l0, sy.l1) - mt);
                invSt = 1.0f / St;
#pragma unroll
                for (int c = 0; c < C; ++c) Q += __expf(lerp_v(tt[c], tb[c], sy.l0, sy.l1) - mt) * invSt;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float logp = lerp_v(st[c], sb[c], sy.l0, sy.l1) - lse;
                const float p = __expf(logp);
                float g = ign ? 0.f : A1 * (p - (c == y ? 1.f : 0.f));          // CE
                g += A2 * (wconf * (p - invC));                                  // KLD to uniform
                g += A3 * (wign * (-p * (logp + Hent)));                         // entropy
                if (TEACHER) {
                    const float q = in_region
                                        ? __expf(lerp_v(tt[c], tb[c], sy.l0, sy.l1) - mt) * invSt
                                        : 0.f;
                    g += A4 * (wreg * (p * Q - q));                              // soft CE
                }
                gt[c] = fmaf(sy.l0, g, gt[c]);      // adjoint of lerp_v
                gb[c] = fmaf(sy.l1, g, gb[c]);
            }
        }
    }

    // fold columns into low-res cells (adjoint of lerp_h), fixed order
    s_x0[threadIdx.x] = live ? sx.i0 - i0 : -1000;
    s_w1[threadIdx.x] = sx.l1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        s_g[0][c][threadIdx.x] = live ? gt[c] : 0.f;
        s_g[1][c][threadIdx.x] = live ? gb[c] : 0.f;
    }
    __syncthreads();
    const int ncell = TI + 1;
    const int ncol = Xe - Xs;
    float* out = tiles;
    for (int task = threadIdx.x; task < 2 * C * ncell; task += LOSS_THREADS) {
        const int ii = task % ncell;
        const int c = (task / ncell) % C;
        const int r = task / (ncell * C);
        // columns with x0 in {i0+ii-1, i0+ii}
        int ca = band_start(sw, i0 + ii - 1, w, W) - Xs;
        int cb = band_start(sw, i0 + ii + 1, w, W) - Xs;
        ca = ca < 0 ? 0 : ca;
        cb = cb > ncol ? ncol : cb;
        float acc = 0.f;
        for (int t = ca; t < cb; ++t) {
            const int x0r = s_x0[t];
            const int x1r = x0r + ((x0r + i0) < w - 1 ? 1 : 0);
            const float w1 = s_w1[t];
            const float wt = (x0r == ii ? 1.0f - w1 : 0.f) + (x1r == ii ? w1 : 0.f);
            acc = fmaf(s_g[r][c][t], wt, acc);
        }
        out[(((((size_t)b * C + c) * h + j) * 2 + r) * nxb + xb) * ncell + ii] = acc;
    }
}

// dlogits[b][c][jj][i] = Σ of the (<=4) partial tiles that cover the cell, fixed order
__global__ __launch_bounds__(256) void st_loss_combine_kernel(const float* __restrict__ tiles,
                                                              float* __restrict__ dlogits, int C,
                                                              int h, int w, int TI, int nxb,
                                                              long long total)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % w);
    const int jj = (int)((idx / w) % h);
    const long long bc = idx / ((long long)w * h);
    const int ncell = TI + 1;
    float acc = 0.f;
    // bands whose rows map onto source row jj: (j = jj, r = 0), (j = jj-1, r = 1) and, for the
    // clamped last band (y1 == y0 == h-1), (j = h-1, r = 1)
    for (int k = 0; k < 3; ++k) {
        int j, r;
        if (k == 0) { j = jj; r = 0; }
        else if (k == 1) { j = jj - 1; r = 1; if (j < 0) continue; }
        else { if (jj != h - 1) continue; j = h - 1; r = 1; }
        const int xb = i / TI, ii = i - xb * TI;
        const float* base = tiles + (((size_t)bc * h + j) * 2 + r) * nxb * ncell;
        if (xb < nxb) acc += base[(size_t)xb * ncell + ii];
        if (ii == 0 && xb > 0) acc += base[(size_t)(xb - 1) * ncell + TI];
    }
    dlogits[idx] = acc;
}

struct LossGeom {
    float sh, sw;
    int TI, nxb;
};

static int loss_geom(int h, int w, int H, int W, LossGeom* g)
{
    g->sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    g->sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    int TI = g->sw > 0.f ? (int)(253.0f * g->sw) : w;   // TI cells span <= TI/sw + 1 <= 254 columns
    TI = TI < 1 ? 1 : (TI > 253 ? 253 : TI);
    if (TI > w) TI = w;
    g->TI = TI;
    g->nxb = (w + TI - 1) / TI;
    // every block's column count must fit one thread per column
    float ratio = g->sw > 0.f ? 1.0f / g->sw : (float)W;
    if ((float)TI * ratio + 2.0f > 256.0f && !(g->sw == 0.f && W <= 256)) return HIAST_E_RANGE;
    return 0;
}

}  // namespace hiast

extern "C" size_t hiast_st_loss_workspace_bytes(int B, int C, int h, int w, int H, int W)
{
    hiast::LossGeom g;
    if (hiast::loss_geom(h, w, H, W, &g)) return 0;
    const size_t fwd_blocks = (size_t)((W + 255) / 256) * h * B;
    const size_t fwd = fwd_blocks * 8 * sizeof(double);
    const size_t bwd = (size_t)B * C * h * 2 * g.nxb * (g.TI + 1) * sizeof(float);
    return (fwd > bwd ? fwd : bwd) + 256;
}

static int loss_check(const void* a, const void* l, const void* s, const void* ws, int B, int C, int h,
                      int w, int H, int W, int region)
{
    if (!a || !l || !s || !ws) return HIAST_E_ARG;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return HIAST_E_ARG;
    if (H < h || W < w || B > 65535 || h > 65535 || region < 0 || region > 2) return HIAST_E_RANGE;
    return 0;
}

#define HIAST_LOSS_DISPATCH(KERNEL_CALL)                                          \
    switch (C) {                                                                  \
        case 19: { constexpr int CC = 19; KERNEL_CALL; } break;                   \
        case 16: { constexpr int CC = 16; KERNEL_CALL; } break;                   \
        case 9:  { constexpr int CC = 9;  KERNEL_CALL; } break;                   \
        case 2:  { constexpr int CC = 2;  KERNEL_CALL; } break;                   \
        default: return HIAST_E_RANGE;                                            \
    }

extern "C" int hiast_st_loss_fwd(const float* logits_lr, const float* teacher_lr, const void* plbl,
                                 int plbl_is_i64, int B, int C, int h, int w, int H, int W, int region,
                                 double* sums, void* workspace, size_t workspace_bytes,
                                 hiast_stream_t stream)
{
    int e = loss_check(logits_lr, plbl, sums, workspace, B, C, h, w, H, W, region);
    if (e) return e;
    hiast::LossGeom g;
    if ((e = hiast::loss_geom(h, w, H, W, &g))) return e;
    if (workspace_bytes < hiast_st_loss_workspace_bytes(B, C, h, w, H, W)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((W + 255) / 256, h, B);
    const int nblk = (int)(grid.x * grid.y * grid.z);
    double* partial = (double*)workspace;
#define FWD(T, LT)                                                                                  \
    hipLaunchKernelGGL((hiast::st_loss_fwd_kernel<CC, T, LT>), grid, dim3(hiast::LOSS_THREADS), 0, st, \
                       logits_lr, teacher_lr, (const LT*)plbl, h, w, H, W, g.sh, g.sw, region, partial)
    if (teacher_lr) {
        if (plbl_is_i64) { HIAST_LOSS_DISPATCH(FWD(true, int64_t)) } else { HIAST_LOSS_DISPATCH(FWD(true, uint8_t)) }
    } else {
        if (plbl_is_i64) { HIAST_LOSS_DISPATCH(FWD(false, int64_t)) } else { HIAST_LOSS_DISPATCH(FWD(false, uint8_t)) }
    }
#undef FWD
    HIAST_CHECK_LAUNCH();
    hipLaunchKernelGGL(hiast::st_loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk, sums);
    HIAST_CHECK_LAUNCH();
    return 0;
}

extern "C" int hiast_st_loss_bwd(const float* logits_lr, const float* teacher_lr, const void* plbl,
                                 int plbl_is_i64, int B, int C, int h, int w, int H, int W, int region,
                                 const double* sums, const float* coef, float* dlogits_lr,
                                 void* workspace, size_t workspace_bytes, hiast_stream_t stream)
{
    int e = loss_check(logits_lr, plbl, sums, workspace, B, C, h, w, H, W, region);
    if (e) return e;
    if (!coef || !dlogits_lr) return HIAST_E_ARG;
    hiast::LossGeom g;
    if ((e = hiast::loss_geom(h, w, H, W, &g))) return e;
    if (workspace_bytes < hiast_st_loss_workspace_bytes(B, C, h, w, H, W)) return HIAST_E_WS;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(g.nxb, h, B);
    float* tiles = (float*)workspace;
#define BWD(T, LT)                                                                                  \
    hipLaunchKernelGGL((hiast::st_loss_bwd_kernel<CC, T, LT>), grid, dim3(hiast::LOSS_THREADS), 0, st, \
                       logits_lr, teacher_lr, (const LT*)plbl, h, w, H, W, g.sh, g.sw, region, g.TI,  \
                       sums, coef, tiles)
    if (teacher_lr) {
        if (plbl_is_i64) { HIAST_LOSS_DISPATCH(BWD(true, int64_t)) } else { HIAST_LOSS_DISPATCH(BWD(true, uint8_t)) }
    } else {
        if (plbl_is_i64) { HIAST_LOSS_DISPATCH(BWD(false, int64_t)) } else { HIAST_LOSS_DISPATCH(BWD(false, uint8_t)) }
    }
#undef BWD
    HIAST_CHECK_LAUNCH();
    const long long total = (long long)B * C * h * w;
    hipLaunchKernelGGL(hiast::st_loss_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       st, tiles, dlogits_lr, C, h, w, g.TI, g.nxb, total);
    HIAST_CHECK_LAUNCH();
    return 0;
}
