/*
 * hiast_oracle.c — TEST INFRASTRUCTURE ONLY (the parity oracle), never product code.
 *
 * Plain-C restatement of the integer/byte part of HIAST's pseudo-label path and of the
 * small dense ops around it, used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker.  Nothing under hiast_amd/ may import or link this.
 *
 * Parity status: PINNED against outputs of the reference itself (run in the build
 * container by tests/golden/make_golden.py; fixtures under tests/golden/), see
 * tests/test_oracle_golden.py.  The reference has no tests or golden vectors of its own.
 *
 * "HIAST-A arithmetic": every fp32 step below is written with explicit fmaf/mul/add in a
 * fixed order and this file is compiled with -ffp-contract=off, so the HIP kernels (which
 * spell out the same sequence) can be compared BIT-exactly.  Against torch's own
 * interpolate/softmax the float stages agree to a few ulp (tolerance in the tests), the
 * integer stages exactly.
 *
 * Reference citations are paths under bupt-ai-cz/HIAST `code/`.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_IGNORE 255

/* ---- deterministic exp for x <= 0 (softmax numerators) ---------------------------- */
/* Cody-Waite reduction + degree-7 Taylor in Horner/fmaf form; |rel err| < 1 ulp.
 * x < -87 flushes to 0 (never feeds a max-prob: the max term is exp(0) = 1). */
float orc_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    const float LOG2E = 1.44269502162933349609375f;      /* 0x3FB8AA3B */
    const float LN2_HI = 0.693145751953125f;             /* 0x3F317200 */
    const float LN2_LO = 1.428606765330187045037746429443359375e-06f; /* 0x35BFBE8E */
    float n = rintf(x * LOG2E);
    float r = fmaf(-n, LN2_HI, x);
    r = fmaf(-n, LN2_LO, r);
    float p = 1.984127011382952332496643066406250e-04f;  /* 1/5040 */
    p = fmaf(p, r, 1.388888922519981861114501953125e-03f); /* 1/720 */
    p = fmaf(p, r, 8.33333376795053482055664062500e-03f);  /* 1/120 */
    p = fmaf(p, r, 4.16666679084300994873046875000e-02f);  /* 1/24 */
    p = fmaf(p, r, 1.66666671633720397949218750000e-01f);  /* 1/6 */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;                /* 2^n, n in [-126, 0] */
    return p * s.f;
}

/* ---- K2: bilinear upsample, align_corners=True -------------------------------------- */
/* F.interpolate(..., mode='bilinear', align_corners=True)
 * (sseg/models/segmentors/self_training_segmentor.py:27).  Source index = dst * (in-1)/(out-1),
 * 4-tap lerp, horizontal first (the order of ATen's upsample_bilinear2d). */
static inline float orc_scale(int in, int out)
{
    return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
}

static inline void orc_src(float scale, int dst, int in, int* i0, int* i1, float* l0, float* l1)
{
    float s = scale * (float)dst;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    *i0 = a;
    *i1 = a + (a < in - 1 ? 1 : 0);
    *l1 = s - (float)a;
    *l0 = 1.0f - *l1;
}

static inline float orc_tap4(const float* p, int w, int y0, int y1, int x0, int x1, float hl0,
                             float hl1, float wl0, float wl1)
{
    float a = p[(size_t)y0 * w + x0], b = p[(size_t)y0 * w + x1];
    float c = p[(size_t)y1 * w + x0], d = p[(size_t)y1 * w + x1];
    float top = fmaf(wl1, b, wl0 * a);
    float bot = fmaf(wl1, d, wl0 * c);
    return fmaf(hl1, bot, hl0 * top);
}

void orc_upsample_bilinear_ac(const float* in, float* out, int BC, int h, int w, int H, int W)
{
    float sh = orc_scale(h, H), sw = orc_scale(w, W);
    for (int bc = 0; bc < BC; ++bc) {
        const float* p = in + (size_t)bc * h * w;
        float* o = out + (size_t)bc * H * W;
        for (int Y = 0; Y < H; ++Y) {
            int y0, y1; float hl0, hl1;
            orc_src(sh, Y, h, &y0, &y1, &hl0, &hl1);
            for (int X = 0; X < W; ++X) {
                int x0, x1; float wl0, wl1;
                orc_src(sw, X, w, &x0, &x1, &wl0, &wl1);
                o[(size_t)Y * W + X] = orc_tap4(p, w, y0, y1, x0, x1, hl0, hl1, wl0, wl1);
            }
        }
    }
}

/* adjoint of the above (autograd of F.interpolate); double accumulation, tolerance-class */
void orc_upsample_bilinear_ac_bwd(const float* gout, float* gin, int BC, int h, int w, int H, int W)
{
    float sh = orc_scale(h, H), sw = orc_scale(w, W);
    double* acc = (double*)calloc((size_t)h * w, sizeof(double));
    for (int bc = 0; bc < BC; ++bc) {
        memset(acc, 0, (size_t)h * w * sizeof(double));
        const float* g = gout + (size_t)bc * H * W;
        for (int Y = 0; Y < H; ++Y) {
            int y0, y1; float hl0, hl1;
            orc_src(sh, Y, h, &y0, &y1, &hl0, &hl1);
            for (int X = 0; X < W; ++X) {
                int x0, x1; float wl0, wl1;
                orc_src(sw, X, w, &x0, &x1, &wl0, &wl1);
                double v = g[(size_t)Y * W + X];
                acc[(size_t)y0 * w + x0] += v * hl0 * wl0;
                acc[(size_t)y0 * w + x1] += v * hl0 * wl1;
                acc[(size_t)y1 * w + x0] += v * hl1 * wl0;
                acc[(size_t)y1 * w + x1] += v * hl1 * wl1;
            }
        }
        for (size_t i = 0; i < (size_t)h * w; ++i) gin[(size_t)bc * h * w + i] = (float)acc[i];
    }
    free(acc);
}

/* ---- K3 stage A: upsample + softmax + max/argmax ------------------------------------ */
/* probs = F.softmax(logits, 1); probs_pred, lbls_pred = probs.max(1)
 * (workflows/pseudo_label_generator.py:192-193).  Ties -> first index (torch CPU max).
 * max-prob = exp(0)/Σ = 1/Σ_c exp(z_c - m), Σ accumulated in ascending c. */
static inline uint16_t orc_f32_to_f16_bits(float f);

void orc_plabel_stage_a(const float* logits_lr, int B, int C, int h, int w, int H, int W,
                        float* maxprob, uint8_t* argmax)
{
    float sh = orc_scale(h, H), sw = orc_scale(w, W);
    float z[64];
    for (int b = 0; b < B; ++b) {
        const float* base = logits_lr + (size_t)b * C * h * w;
        for (int Y = 0; Y < H; ++Y) {
            int y0, y1; float hl0, hl1;
            orc_src(sh, Y, h, &y0, &y1, &hl0, &hl1);
            for (int X = 0; X < W; ++X) {
                int x0, x1; float wl0, wl1;
                orc_src(sw, X, w, &x0, &x1, &wl0, &wl1);
                float m = 0.0f; int am = 0;
                for (int c = 0; c < C; ++c) {
                    z[c] = orc_tap4(base + (size_t)c * h * w, w, y0, y1, x0, x1, hl0, hl1, wl0, wl1);
                    if (c == 0 || z[c] > m) { m = z[c]; am = c; }
                }
                float s = 0.0f;
                for (int c = 0; c < C; ++c) s = s + orc_expf(z[c] - m);
                size_t o = ((size_t)b * H + Y) * W + X;
                maxprob[o] = 1.0f / s;
                argmax[o] = (uint8_t)am;
            }
        }
    }
}

/* np.float16(x) for a float32 x: IEEE round-to-nearest-even (pseudo_label_generator.py:201) */
static inline uint16_t orc_f32_to_f16_bits(float f)
{
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t sign = (v.u >> 16) & 0x8000u;
    uint32_t x = v.u & 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);           /* rounds to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                         /* rounds to 0 */
    int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) half++;
    if (e < -14) return (uint16_t)(sign | half);                        /* subnormal (may carry) */
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));
}

uint16_t orc_f16_bits(float f) { return orc_f32_to_f16_bits(f); }

/* per-class histogram over fp16 bit patterns of the max-prob (the multiset the reference
 * keeps as Python lists, pseudo_label_generator.py:198-201) */
void orc_plabel_hist(const float* maxprob, const uint8_t* argmax, int64_t N, int C, int nbins,
                     uint32_t* hist)
{
    for (int64_t i = 0; i < N; ++i) {
        uint16_t b = orc_f32_to_f16_bits(maxprob[i]);
        if (argmax[i] < C && b < nbins) hist[(size_t)argmax[i] * nbins + b]++;
    }
}

/* ---- K4 stage B: select confident pixels --------------------------------------------- */
/* select_and_save_confident_label, pseudo_label_generator.py:67-105:
 *   ignored = prob(f32) < thr[lbl](f64); plbl = lbl; plbl[ignored] = 255
 *   per image, per class pixel counts of plbl; Σ prob over plbl == c (exact, as prob*2^30). */
void orc_plabel_select(const float* maxprob, const uint8_t* argmax, const double* thr /* NULL = NT */,
                       int B, int C, int64_t HW, uint8_t* plbl, int64_t* count, uint64_t* sumprob_fx)
{
    for (int b = 0; b < B; ++b)
        for (int64_t i = 0; i < HW; ++i) {
            size_t o = (size_t)b * HW + i;
            uint8_t l = argmax[o];
            if (thr && (double)maxprob[o] < thr[l]) l = ORC_IGNORE;
            plbl[o] = l;
            if (l < C) {
                count[(size_t)b * C + l]++;
                sumprob_fx[l] += (uint64_t)((double)maxprob[o] * 1073741824.0);
            }
        }
}

/* ---- K1: ASPP (4 dilated 3x3 convs summed), direct form, double accumulation --------- */
/* ASPP_V2.forward, sseg/models/modules/seg_models/deeplab_v2.py:20-24. */
void orc_aspp_fwd(const float* x, const float* const* W, const float* const* bias, float* y, int B,
                  int Cin, int h, int w, int Cout, const int* dil)
{
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < h; ++oy)
                for (int ox = 0; ox < w; ++ox) {
                    double acc = 0.0;
                    for (int i = 0; i < 4; ++i) {
                        acc += bias[i][co];
                        int d = dil[i];
                        for (int ky = 0; ky < 3; ++ky) {
                            int iy = oy + (ky - 1) * d;
                            if (iy < 0 || iy >= h) continue;
                            for (int kx = 0; kx < 3; ++kx) {
                                int ix = ox + (kx - 1) * d;
                                if (ix < 0 || ix >= w) continue;
                                const float* wp = W[i] + ((size_t)co * Cin * 9) + ky * 3 + kx;
                                const float* xp = x + ((size_t)b * Cin * h + iy) * w + ix;
                                for (int ci = 0; ci < Cin; ++ci)
                                    acc += (double)wp[(size_t)ci * 9] * (double)xp[(size_t)ci * h * w];
                            }
                        }
                    }
                    y[(((size_t)b * Cout + co) * h + oy) * w + ox] = (float)acc;
                }
}

/* ---- K12: IoU areas, utils/metrics.py:6-19 ------------------------------------------- */
void orc_confusion_hist(const int64_t* pred, const int64_t* target, int64_t N, int K, int64_t* inter,
                        int64_t* area_pred, int64_t* area_tgt)
{
    for (int64_t i = 0; i < N; ++i) {
        int64_t t = target[i];
        int64_t p = (t == ORC_IGNORE) ? ORC_IGNORE : pred[i];   /* output[target==255] = 255 */
        if (p >= 0 && p < K) area_pred[p]++;
        if (t >= 0 && t < K) area_tgt[t]++;
        if (p == t && p >= 0 && p < K) inter[p]++;
    }
}

/* ---- K11: EMA, utils/utils.py:115-123 ------------------------------------------------- */
/* gamma and (1-gamma) arrive as the float32 roundings of the Python doubles, which is what
 * `tensor * python_float` does in torch. */
void orc_ema_update(float* ema, const float* p, int64_t n, float gamma, float omg)
{
    for (int64_t i = 0; i < n; ++i) ema[i] = ema[i] * gamma + p[i] * omg;
}

/* ---- K16: multi-scale + flip test-time augmentation ----------------------------------- */
/* Validator.get_multi_scale_and_flip_logits (workflows/validator.py:34-55) downstream of the
 * segmentation net's LOW-RES head outputs, in HIAST-A arithmetic:
 *   per scale s: P_s = softmax(interp(z_s -> Hs x Ws)) [+ flip_w(softmax(interp(zf_s -> Hs x Ws)))]
 *                (self_training_segmentor.py:27 + validator.py:37,46-50);
 *   out = sum_s interp(P_s -> H x W) (:52,55); label = argmax (:92, first max wins).
 * softmax(v)_c = expA(v_c - m) * (1/sum), sum ascending in c. */
static void orc_softmax_at(const float* z, int C, int hs, int ws, float sh, float sw, int Y, int X, float* out)
{
    int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
    orc_src(sh, Y, hs, &y0, &y1, &hl0, &hl1);
    orc_src(sw, X, ws, &x0, &x1, &wl0, &wl1);
    float m = 0.0f;
    for (int c = 0; c < C; ++c) {
        out[c] = orc_tap4(z + (size_t)c * hs * ws, ws, y0, y1, x0, x1, hl0, hl1, wl0, wl1);
        if (c == 0 || out[c] > m) m = out[c];
    }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) { out[c] = orc_expf(out[c] - m); s = s + out[c]; }
    float inv = 1.0f / s;
    for (int c = 0; c < C; ++c) out[c] = out[c] * inv;
}

void orc_tta(const float* const* z, const float* const* zf, const int* hs, const int* ws, const int* Hs,
             const int* Ws, int n_scales, int B, int C, int H, int W, float* probsum, uint8_t* label)
{
    float tmp[64], tmp2[64];
    float* acc = (float*)calloc((size_t)B * C * H * W, sizeof(float));
    for (int s = 0; s < n_scales; ++s) {
        int h_ = hs[s], w_ = ws[s], HS = Hs[s], WS = Ws[s];
        float* P = (float*)malloc((size_t)C * HS * WS * sizeof(float));
        float sh = orc_scale(h_, HS), sw = orc_scale(w_, WS);
        float rh = orc_scale(HS, H), rw = orc_scale(WS, W);
        for (int b = 0; b < B; ++b) {
            const float* zb = z[s] + (size_t)b * C * h_ * w_;
            const float* zfb = (zf && zf[s]) ? zf[s] + (size_t)b * C * h_ * w_ : NULL;
            for (int Y = 0; Y < HS; ++Y)
                for (int X = 0; X < WS; ++X) {
                    orc_softmax_at(zb, C, h_, w_, sh, sw, Y, X, tmp);
                    if (zfb) {
                        orc_softmax_at(zfb, C, h_, w_, sh, sw, Y, WS - 1 - X, tmp2);
                        for (int c = 0; c < C; ++c) tmp[c] = tmp[c] + tmp2[c];
                    }
                    for (int c = 0; c < C; ++c) P[((size_t)c * HS + Y) * WS + X] = tmp[c];
                }
            for (int c = 0; c < C; ++c)
                for (int Y = 0; Y < H; ++Y) {
                    int y0, y1; float hl0, hl1;
                    orc_src(rh, Y, HS, &y0, &y1, &hl0, &hl1);
                    for (int X = 0; X < W; ++X) {
                        int x0, x1; float wl0, wl1;
                        orc_src(rw, X, WS, &x0, &x1, &wl0, &wl1);
                        size_t o = (((size_t)b * C + c) * H + Y) * W + X;
                        acc[o] = acc[o] + orc_tap4(P + (size_t)c * HS * WS, WS, y0, y1, x0, x1, hl0, hl1, wl0, wl1);
                    }
                }
        }
        free(P);
    }
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < (size_t)H * W; ++i) {
            float m = 0.0f; int am = 0;
            for (int c = 0; c < C; ++c) {
                float v = acc[((size_t)b * C + c) * H * W + i];
                if (probsum) probsum[((size_t)b * C + c) * H * W + i] = v;
                if (c == 0 || v > m) { m = v; am = c; }
            }
            if (label) label[(size_t)b * H * W + i] = (uint8_t)am;
        }
    free(acc);
}
