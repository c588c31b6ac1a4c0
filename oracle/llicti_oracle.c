/*
 * llicti_oracle.c -- CPU ORACLE (test infrastructure only; see llicti_oracle.h for scope and citations).
 * Build: oracle/Makefile  (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 *
 * All floating point below is written operation by operation: no contraction (-ffp-contract=off),
 * explicit fmaf() where the spec says "fused", IEEE division, round-to-nearest-even throughout.
 */
#include "llicti_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
void orc_set_threads(int n) { g_threads = n; }
static int nthreads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ numerics spec v1: scalar functions */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* coefficients derived by tools/gen_numerics_coeffs.py */
static const float ERFC_P[11] = {
    0x1.058672p+0f, -0x1.5fd388p-1f, 0x1.1e2deep-2f, -0x1.2a869p-5f, -0x1.606bd2p-6f, 0x1.a7e3c6p-8f,
    0x1.93b74ap-9f, -0x1.a18b0cp-11f, -0x1.3ddc9cp-11f, -0x1.d399c2p-18f, 0x1.73901ap-15f };
static const float EXP_Q[5] = { 0x1.0p-1f, 0x1.5554dcp-3f, 0x1.5554eap-5f, 0x1.120b74p-7f, 0x1.6d4328p-10f };
#define LOG2E_F   0x1.715476p+0f
#define LN2_HI_F  0x1.62e4p-1f
#define LN2_LO_F  0x1.7f7d1cp-20f

/* exp(y) for y in [-49, 0]:  2^j * (1 + f + f^2 R(f)),  j = rint(y log2 e),  f = y - j ln2 (two fma steps) */
static inline float exp_spec(float y)
{
    float j = rintf(y * LOG2E_F);
    float f = fmaf(j, -LN2_HI_F, y);
    f = fmaf(j, -LN2_LO_F, f);
    float q = EXP_Q[4];
    q = fmaf(q, f, EXP_Q[3]);
    q = fmaf(q, f, EXP_Q[2]);
    q = fmaf(q, f, EXP_Q[1]);
    q = fmaf(q, f, EXP_Q[0]);
    float f2 = f * f;
    q = fmaf(q, f2, f);
    q = q + 1.0f;
    int32_t ji = (int32_t)j;
    return bits2f(f2bits(q) + ((uint32_t)ji << 23));
}

/* erfc(x), x >= 0:  r = 1/(x+2), t = (x-2) r,  P(t) exp(-x^2) r;  := 0 for x >= 7 (and for NaN) */
static inline float erfc_pos(float x)
{
    if (!(x < 7.0f)) return 0.0f;
    float r = 1.0f / (x + 2.0f);
    float t = (x - 2.0f) * r;
    float p = ERFC_P[10];
    for (int i = 9; i >= 0; --i) p = fmaf(p, t, ERFC_P[i]);
    float s = x * x;
    float e = fmaf(x, x, -s);          /* exact rounding error of s */
    float ex = exp_spec(-s);
    ex = fmaf(-e, ex, ex);             /* exp(-(s+e)) ~= exp(-s) (1 - e) */
    return (p * ex) * r;
}
static inline float erfc_spec(float x)
{
    float v = erfc_pos(fabsf(x));
    return (x < 0.0f) ? 2.0f - v : v;
}
float orc_erfc(float x) { return erfc_spec(x); }

/* compressai constants: scale bound 0.11/255 and weight bound 1e-6 are float32 buffers
 * (entropy_layer_nets.py:149-158); const = -(2**-0.5) (compressai _standardized_cumulative). */
#define SCALE_BOUND ((float)(0.11 / 255.0))
#define WEIGHT_BOUND (1e-6f)
#define NEG_RSQRT2 ((float)(-0.70710678118654752440))

typedef struct { float rsig[5], mu[5], wn[5]; } mix_t;   /* rsig = 1 / max(sigma, bound), one IEEE division */

/* entropy_layer_nets.py:197-200 + LLICTI_nets.py:385-392: slice, cross-channel mean update, bounds, normalise */
static inline void mix_prepare(const float *par, int clr, float yv, float cov, mix_t *m)
{
    float w[5];
    for (int k = 0; k < 5; ++k) {
        float sg = par[5 * clr + k];
        float mu = par[15 + 5 * clr + k];
        float wk = par[30 + 5 * clr + k];
        if (clr == 1) {
            float t = par[45 + k] * yv;
            mu = mu + t;
        } else if (clr == 2) {
            float t1 = par[50 + k] * yv;
            float t2 = par[55 + k] * cov;
            float t = t1 + t2;
            mu = mu + t;
        }
        m->rsig[k] = 1.0f / ((sg > SCALE_BOUND) ? sg : SCALE_BOUND);
        m->mu[k] = mu;
        w[k] = (wk > WEIGHT_BOUND) ? wk : WEIGHT_BOUND;
    }
    float s = (((w[0] + w[1]) + w[2]) + w[3]) + w[4];
    float den = 1e-9f + s;
    for (int k = 0; k < 5; ++k) m->wn[k] = w[k] / den;
}

static inline float mix_cdf(const mix_t *m, float pt)
{
    float acc = 0.0f;
    for (int k = 0; k < 5; ++k) {
        float z = (pt - m->mu[k]) * m->rsig[k];     /* spec: reciprocal once per mixture, then multiply */
        float c = 0.5f * erfc_spec(NEG_RSQRT2 * z);
        float t = m->wn[k] * c;
        acc = (k == 0) ? t : acc + t;
    }
    return acc;
}

/* sample point of table entry i (LLICTI_nets.py:941-942): half-integers / 255 in fp32, the two end
 * points pushed out by 20 and computed in double like the reference's Python scalars */
static inline float sample_pt(int i, int Lp, int minv, int maxv)
{
    if (i == 0) return (float)(((double)minv - 0.5 - 20.0) / 255.0);
    if (i == Lp - 1) return (float)(((double)maxv + 0.5 + 20.0) / 255.0);
    return ((float)minv - 0.5f + (float)i) / 255.0f;
}

/* LLICTI_nets.py:955-983: round(cdf * (65536 - (Lp-1))) -> 16-bit wrap -> + index */
static inline uint16_t cdf_entry(const mix_t *m, int i, int Lp, int minv, int maxv)
{
    float scale = (float)(65536 - (Lp - 1));
    float q = rintf(mix_cdf(m, sample_pt(i, Lp, minv, maxv)) * scale);
    return (uint16_t)((int32_t)q + i);
}

void orc_cdf_row(const float *par, int clr, float yv, float cov, int minv, int maxv, uint16_t *row)
{
    mix_t m;
    mix_prepare(par, clr, yv, cov, &m);
    int Lp = maxv - minv + 2;
    for (int i = 0; i < Lp; ++i) row[i] = cdf_entry(&m, i, Lp, minv, maxv);
}

float orc_cdf_float(const float *par, int clr, float yv, float cov, float pt)
{
    mix_t m;
    mix_prepare(par, clr, yv, cov, &m);
    return mix_cdf(&m, pt);
}

/* ------------------------------------------------------------------ integer colour lift */
static inline int fdiv2(int a) { return a >> 1; }   /* floor division by 2 (torch >= 1.13 '//' on int16) */

void orc_lift(const uint8_t *rgb, int H, int W, int16_t *planes, int16_t minmax[6])
{
    long n = (long)H * W;
    int mnCo = 32767, mxCo = -32768, mnCg = 32767, mxCg = -32768;
    for (long p = 0; p < n; ++p) {
        int R = rgb[p], G = rgb[n + p], B = rgb[2 * n + p];
        int Co = R - B;
        int t = B + fdiv2(Co);
        int Cg = G - t;
        int Y = t + fdiv2(Cg);
        planes[p] = (int16_t)(Y - 127);
        planes[n + p] = (int16_t)Co;
        planes[2 * n + p] = (int16_t)Cg;
        if (Co < mnCo) mnCo = Co;
        if (Co > mxCo) mxCo = Co;
        if (Cg < mnCg) mnCg = Cg;
        if (Cg > mxCg) mxCg = Cg;
    }
    minmax[0] = 0; minmax[1] = (int16_t)mnCo; minmax[2] = (int16_t)mnCg;
    minmax[3] = 255; minmax[4] = (int16_t)mxCo; minmax[5] = (int16_t)mxCg;
}

void orc_unlift(const int16_t *planes, int H, int W, uint8_t *rgb)
{
    long n = (long)H * W;
    for (long p = 0; p < n; ++p) {
        int Y = planes[p] + 127, Co = planes[n + p], Cg = planes[2 * n + p];
        int t = Y - fdiv2(Cg);
        int G = Cg + t;
        int B = t - fdiv2(Co);
        int R = B + Co;
        rgb[p] = (uint8_t)R; rgb[n + p] = (uint8_t)G; rgb[2 * n + p] = (uint8_t)B;
    }
}

/* ------------------------------------------------------------------ level geometry / band access */
void orc_level_geom(int H, int W, int lvl, int *Hl, int *Wl, int *h, int *w, int *padH, int *padW)
{
    int st = 1 << lvl;
    *Hl = (H + st - 1) / st;
    *Wl = (W + st - 1) / st;
    *h = (*Hl + 1) / 2;
    *w = (*Wl + 1) / 2;
    *padH = *Hl & 1;
    *padW = *Wl & 1;
}

static const int BAND_OI[4] = { 0, 1, 0, 1 };   /* source order x00, x11, x01, x10 (lazyDWT cat order) */
static const int BAND_OJ[4] = { 0, 1, 1, 0 };
/* band to predict b = 0,1,2 is x11, x01, x10 = source index b+1 */

/* value of sub-band `src` at band coordinate (i, j) of level lvl, with the conv's replicate clamp in
 * band space and lazyDWT's replicate pad of the odd edge (LLICTI_nets.py:226-240, :511-530) */
static inline int band_px(const int16_t *plane, int H, int W, int lvl, int Hl, int Wl, int h, int w,
                          int src, int i, int j)
{
    if (i < 0) i = 0;
    if (i > h - 1) i = h - 1;
    if (j < 0) j = 0;
    if (j > w - 1) j = w - 1;
    int r = 2 * i + BAND_OI[src];
    int c = 2 * j + BAND_OJ[src];
    if (r >= Hl) r -= 2;
    if (c >= Wl) c -= 2;
    (void)H;
    return plane[((long)r << lvl) * W + ((long)c << lvl)];
}

/* layer-0 tap tables: per band a list of (src, kh, kw, top pad, left pad) (LLICTI_nets.py:651-675) */
typedef struct { int src, kh, kw, pt, pl; } conv_t;
static const conv_t CONVS[3][3] = {
    { { 0, 4, 4, 1, 1 }, { -1, 0, 0, 0, 0 }, { -1, 0, 0, 0, 0 } },
    { { 0, 3, 4, 1, 1 }, { 1, 4, 3, 2, 1 }, { -1, 0, 0, 0, 0 } },
    { { 0, 4, 3, 1, 1 }, { 1, 3, 4, 1, 2 }, { 2, 4, 4, 1, 2 } },
};

typedef struct { int src, ci, dy, dx; } tap_t;
static int build_taps(int band, tap_t *taps)
{
    int k = 0;
    for (int c = 0; c < 3; ++c) {
        const conv_t *cv = &CONVS[band][c];
        if (cv->src < 0) break;
        /* K order of the spec: the kernel's length-4 axis runs fastest (kw == 4: kx; 4x3 kernels: ky) */
        for (int ci = 0; ci < 3; ++ci) {
            if (cv->kw == 4) {
                for (int ky = 0; ky < cv->kh; ++ky)
                    for (int kx = 0; kx < 4; ++kx) {
                        taps[k].src = cv->src; taps[k].ci = ci;
                        taps[k].dy = ky - cv->pt; taps[k].dx = kx - cv->pl;
                        ++k;
                    }
            } else {
                for (int kx = 0; kx < cv->kw; ++kx)
                    for (int ky = 0; ky < 4; ++ky) {
                        taps[k].src = cv->src; taps[k].ci = ci;
                        taps[k].dy = ky - cv->pt; taps[k].dx = kx - cv->pl;
                        ++k;
                    }
            }
        }
    }
    return k;
}

/* ------------------------------------------------------------------ interpolator CNN */
static inline float band_pxf(const float *plane, int W, int lvl, int Hl, int Wl, int h, int w, int src, int i, int j)
{
    if (i < 0) i = 0;
    if (i > h - 1) i = h - 1;
    if (j < 0) j = 0;
    if (j > w - 1) j = w - 1;
    int r = 2 * i + BAND_OI[src];
    int c = 2 * j + BAND_OJ[src];
    if (r >= Hl) r -= 2;
    if (c >= Wl) c -= 2;
    return plane[((long)r << lvl) * W + ((long)c << lvl)];
}

/* fplanes: float [3][H][W] (codec path: int16 / 255; training path: the float lift) */
void orc_band_params_f(const float *fplanes, int H, int W, int lvl, int band,
                       const orc_band_weights *bw, float *out)
{
    int Hl, Wl, h, w, padH, padW;
    orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
    const int K0 = bw->K0;
    tap_t taps[120];
    int nk = build_taps(band, taps);
    if (nk != K0) abort();
    /* transposed copies so the inner loops run over output channels (each channel keeps its own
     * k-ordered fmaf chain; vectorising across channels does not change any chain) */
    float *w0t = (float *)malloc(sizeof(float) * K0 * ORC_NCH);
    float *w1t = (float *)malloc(sizeof(float) * ORC_HEAD * ORC_NCH);
    float *w2t = (float *)malloc(sizeof(float) * ORC_HEAD * 64);
    for (int c = 0; c < ORC_NCH; ++c)
        for (int k = 0; k < K0; ++k) w0t[k * ORC_NCH + c] = bw->w0[c * K0 + k];
    for (int c = 0; c < ORC_NCH; ++c)
        for (int i = 0; i < ORC_HEAD; ++i) w1t[i * ORC_NCH + c] = bw->w1[c * ORC_HEAD + i];
    memset(w2t, 0, sizeof(float) * ORC_HEAD * 64);
    for (int o = 0; o < ORC_NPAR; ++o)
        for (int i = 0; i < ORC_HEAD; ++i) w2t[i * 64 + (o / 15) * 16 + (o % 15)] = bw->w2[o * ORC_HEAD + i];
    const long plane_sz = (long)H * W;

#pragma omp parallel for schedule(static) num_threads(nthreads())
    for (long pos = 0; pos < (long)h * w; ++pos) {
        int i = (int)(pos / w), j = (int)(pos % w);
        float x[120];
        float h0[ORC_NCH], h1[ORC_NCH], o2[64];
        for (int k = 0; k < K0; ++k) {
            x[k] = band_pxf(fplanes + taps[k].ci * plane_sz, W, lvl, Hl, Wl, h, w,
                            taps[k].src, i + taps[k].dy, j + taps[k].dx);
        }
        for (int c = 0; c < ORC_NCH; ++c) h0[c] = bw->b0[c];
        for (int k = 0; k < K0; ++k) {
            const float xv = x[k];
            const float *wr = w0t + (long)k * ORC_NCH;
            for (int c = 0; c < ORC_NCH; ++c) h0[c] = fmaf(wr[c], xv, h0[c]);
        }
        for (int c = 0; c < ORC_NCH; ++c) h0[c] = (h0[c] > 0.0f) ? h0[c] : 0.0f;
        for (int c = 0; c < ORC_NCH; ++c) h1[c] = bw->b1[c];
        for (int g = 0; g < 4; ++g)
            for (int ii = 0; ii < ORC_HEAD; ++ii) {
                const float xv = h0[g * ORC_HEAD + ii];
                const float *wr = w1t + (long)ii * ORC_NCH + g * ORC_HEAD;
                float *hp = h1 + g * ORC_HEAD;
                for (int c = 0; c < ORC_HEAD; ++c) hp[c] = fmaf(wr[c], xv, hp[c]);
            }
        for (int c = 0; c < ORC_NCH; ++c) h1[c] = (h1[c] > 0.0f) ? h1[c] : 0.0f;
        for (int g = 0; g < 4; ++g)
            for (int o = 0; o < 16; ++o) o2[g * 16 + o] = (o < 15) ? bw->b2[g * 15 + o] : 0.0f;
        for (int g = 0; g < 4; ++g)
            for (int ii = 0; ii < ORC_HEAD; ++ii) {
                const float xv = h1[g * ORC_HEAD + ii];
                const float *wr = w2t + (long)ii * 64 + g * 16;
                float *op = o2 + g * 16;
                for (int o = 0; o < 16; ++o) op[o] = fmaf(wr[o], xv, op[o]);
            }
        float *dst = out + pos * ORC_NPAR;
        for (int g = 0; g < 4; ++g)
            for (int o = 0; o < 15; ++o) dst[g * 15 + o] = o2[g * 16 + o];
    }
    free(w0t); free(w1t); free(w2t);
}

void orc_band_params(const int16_t *planes, int H, int W, int lvl, int band,
                     const orc_band_weights *bw, float *out)
{
    const long n = 3L * H * W;
    float *fp = (float *)malloc(sizeof(float) * n);
    for (long k = 0; k < n; ++k) fp[k] = (float)planes[k] / 255.0f;     /* LLICTI_nets.py:143-144, :563-565 */
    orc_band_params_f(fp, H, W, lvl, band, bw, out);
    free(fp);
}

/* ------------------------------------------------------------------ training / validation likelihood path
 * LLICTI.forward (LLICTI_nets.py:101-123): float lift with torch.round (half to even) on fp32 values
 * (:40-49), Y - 127/255 (:110), lazyDWT(pad=False) (:182-245), per band get_params + get_self_infos
 * (:802-811, :862-880, :933-935) -> GaussianConditionalLosslessGMM.forward (entropy_layer_nets.py:160-183)
 * with _likelihood_fk (:117-139).  Every step is an elementwise IEEE fp32 operation in the reference, restated
 * in the same order; the mixture sum runs m = 0..4. */
void orc_lift_train(const uint8_t *rgb, int H, int W, float *fplanes)
{
    const long n = (long)H * W;
    const float meanY = (float)(127.0 / 255.0);
    for (long p = 0; p < n; ++p) {
        const float R = (float)rgb[p] / 255.0f, G = (float)rgb[n + p] / 255.0f, B = (float)rgb[2 * n + p] / 255.0f;
        const float Co = R - B;
        const float t = B + rintf(Co * 255.0f / 2.0f) / 255.0f;
        const float Cg = G - t;
        const float Y = t + rintf(Cg * 255.0f / 2.0f) / 255.0f;
        fplanes[p] = Y - meanY;
        fplanes[n + p] = Co;
        fplanes[2 * n + p] = Cg;
    }
}

/* out: [3 (Y, Co, Cg)][h][w] self-information in bits of band `band` of level `lvl`; params: [h*w][60] */
void orc_selfinfo(const float *fplanes, int H, int W, int lvl, int band, const float *params, float *out)
{
    int Hl, Wl, h, w, padH, padW;
    orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
    const long plane_sz = (long)H * W;

    const float half = (float)(0.5 / 255.0);
    const float kneg = (float)(-0.70710678118654752440);
    const float sbound = (float)(0.11 / 255.0);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const float *par = params + ((long)i * w + j) * ORC_NPAR;
            float v[3];
            for (int c = 0; c < 3; ++c) v[c] = band_pxf(fplanes + c * plane_sz, W, lvl, Hl, Wl, h, w, band + 1, i, j);
            for (int clr = 0; clr < 3; ++clr) {
                float wv[5], lik[5], wsum = 0.0f;
                for (int m = 0; m < 5; ++m) {
                    float sg = par[5 * clr + m], mu = par[15 + 5 * clr + m];
                    if (clr == 1) { const float t = par[45 + m] * v[0]; mu = mu + t; }
                    else if (clr == 2) { const float t1 = par[50 + m] * v[0]; const float t2 = par[55 + m] * v[1]; const float t = t1 + t2; mu = mu + t; }
                    sg = (sg > sbound) ? sg : sbound;
                    const float d = fabsf(v[clr] - mu);
                    const float up = 0.5f * erfc_spec(kneg * ((half - d) / sg));
                    const float lo = 0.5f * erfc_spec(kneg * ((-half - d) / sg));
                    lik[m] = up - lo;
                    const float wk = par[30 + 5 * clr + m];
                    wv[m] = (wk > 1e-6f) ? wk : 1e-6f;
                    wsum = (m == 0) ? wv[m] : wsum + wv[m];
                }
                float L = 0.0f;
                for (int m = 0; m < 5; ++m) { const float t = (wv[m] / wsum) * lik[m]; L = (m == 0) ? t : L + t; }
                if (!(L > 1e-9f)) L = 1e-9f;                   /* likelihood_lower_bound */
                out[((long)clr * h + i) * w + j] = -log2f(L);
            }
        }
}

/* ------------------------------------------------------------------ arithmetic coder (torchac 0.9.3 algorithm) */
typedef struct { uint8_t *out; long cap, n; uint8_t cache; int count; int overflow; } bitw_t;
static inline void bw_put(bitw_t *b, int bit)
{
    b->cache = (uint8_t)((b->cache << 1) | bit);
    if (++b->count == 8) {
        if (b->n < b->cap) b->out[b->n] = b->cache; else b->overflow = 1;
        b->n++;
        b->count = 0;
    }
}
static inline void bw_put_pending(bitw_t *b, int bit, uint64_t *pending)
{
    bw_put(b, bit);
    while (*pending > 0) { bw_put(b, !bit); (*pending)--; }
}

typedef struct { uint32_t low, high; uint64_t pending; bitw_t bw; } acenc_t;
static void acenc_init(acenc_t *e, uint8_t *out, long cap)
{
    e->low = 0; e->high = 0xFFFFFFFFu; e->pending = 0;
    e->bw.out = out; e->bw.cap = cap; e->bw.n = 0; e->bw.cache = 0; e->bw.count = 0; e->bw.overflow = 0;
}
static inline void acenc_put(acenc_t *e, uint32_t c_low, uint32_t c_high)
{
    const uint64_t span = (uint64_t)e->high - (uint64_t)e->low + 1;
    e->high = (e->low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
    e->low = e->low + (uint32_t)((span * (uint64_t)c_low) >> 16);
    for (;;) {
        if (e->high < 0x80000000u) {
            bw_put_pending(&e->bw, 0, &e->pending);
            e->low <<= 1; e->high <<= 1; e->high |= 1;
        } else if (e->low >= 0x80000000u) {
            bw_put_pending(&e->bw, 1, &e->pending);
            e->low <<= 1; e->high <<= 1; e->high |= 1;
        } else if (e->low >= 0x40000000u && e->high < 0xC0000000u) {
            e->pending++;
            e->low <<= 1; e->low &= 0x7FFFFFFFu;
            e->high <<= 1; e->high |= 0x80000001u;
        } else break;
    }
}
static long acenc_finish(acenc_t *e)
{
    e->pending += 1;
    if (e->low < 0x40000000u) bw_put_pending(&e->bw, 0, &e->pending);
    else bw_put_pending(&e->bw, 1, &e->pending);
    if (e->bw.count > 0) { while (e->bw.count != 0) bw_put(&e->bw, 0); }
    return e->bw.overflow ? -1 : e->bw.n;
}

typedef struct { const uint8_t *in; long nbytes, ptr; uint8_t cache; int cached; uint32_t low, high, value; } acdec_t;
static inline void acdec_get(acdec_t *d)
{
    if (d->cached == 0) {
        if (d->ptr == d->nbytes) { d->value <<= 1; return; }
        d->cache = d->in[d->ptr++];
        d->cached = 8;
    }
    d->value <<= 1;
    d->value |= (uint32_t)((d->cache >> (d->cached - 1)) & 1);
    d->cached--;
}
static void acdec_init(acdec_t *d, const uint8_t *in, long nbytes)
{
    d->in = in; d->nbytes = nbytes; d->ptr = 0; d->cache = 0; d->cached = 0;
    d->low = 0; d->high = 0xFFFFFFFFu; d->value = 0;
    for (int i = 0; i < 32; ++i) acdec_get(d);
}
static inline uint16_t acdec_count(const acdec_t *d)
{
    const uint64_t span = (uint64_t)d->high - (uint64_t)d->low + 1;
    return (uint16_t)((((uint64_t)d->value - (uint64_t)d->low + 1) * 0x10000u - 1) / span);
}
static inline void acdec_update(acdec_t *d, uint32_t c_low, uint32_t c_high)
{
    const uint64_t span = (uint64_t)d->high - (uint64_t)d->low + 1;
    d->high = (d->low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
    d->low = d->low + (uint32_t)((span * (uint64_t)c_low) >> 16);
    for (;;) {
        if (d->low >= 0x80000000u || d->high < 0x80000000u) {
            d->low <<= 1; d->high <<= 1; d->high |= 1;
            acdec_get(d);
        } else if (d->low >= 0x40000000u && d->high < 0xC0000000u) {
            d->low <<= 1; d->low &= 0x7FFFFFFFu;
            d->high <<= 1; d->high |= 0x80000001u;
            d->value -= 0x40000000u;
            acdec_get(d);
        } else break;
    }
}

long orc_ac_encode_tables(const uint16_t *cdf, int Lp, const int16_t *sym, long N, uint8_t *out, long cap)
{
    acenc_t e;
    acenc_init(&e, out, cap);
    const int max_symbol = Lp - 2;
    for (long i = 0; i < N; ++i) {
        int s = sym[i];
        uint32_t c_low = cdf[i * Lp + s];
        uint32_t c_high = (s == max_symbol) ? 0x10000u : cdf[i * Lp + s + 1];
        acenc_put(&e, c_low, c_high);
    }
    return acenc_finish(&e);
}

long orc_ac_encode_pairs(const uint32_t *clow, const uint32_t *chigh, long N, uint8_t *out, long cap)
{
    acenc_t e;
    acenc_init(&e, out, cap);
    for (long i = 0; i < N; ++i) acenc_put(&e, clow[i], chigh[i]);
    return acenc_finish(&e);
}

void orc_ac_decode_tables(const uint16_t *cdf, int Lp, const uint8_t *in, long nbytes, long N, int16_t *sym)
{
    acdec_t d;
    acdec_init(&d, in, nbytes);
    const uint16_t max_symbol = (uint16_t)(Lp - 2);
    for (long i = 0; i < N; ++i) {
        const uint16_t count = acdec_count(&d);
        const uint16_t *row = cdf + i * Lp;
        uint16_t left = 0, right = (uint16_t)(max_symbol + 1);
        int found = -1;
        while (left + 1 < right) {
            uint16_t m = (uint16_t)((left + right) / 2);
            uint16_t v = row[m];
            if (v < count) left = m; else if (v > count) right = m; else { found = m; break; }
        }
        uint16_t s = (found >= 0) ? (uint16_t)found : left;
        sym[i] = (int16_t)s;
        if (i == N - 1) break;
        uint32_t c_low = row[s];
        uint32_t c_high = (s == max_symbol) ? 0x10000u : row[s + 1];
        acdec_update(&d, c_low, c_high);
    }
}

/* ------------------------------------------------------------------ stream geometry */
static void stream_dims(int h, int w, int padH, int padW, int band, int *hc, int *wc)
{
    /* LLICTI_nets.py:396-397: rows cropped for x11 (b0) and x10 (b2), columns for x11 (b0) and x01 (b1) */
    *hc = (band == 0 || band == 2) ? h - padH : h;
    *wc = (band == 0 || band == 1) ? w - padW : w;
}

long orc_stream_pairs(const int16_t *planes, int H, int W, const int16_t minmax[6], int lvl, int band, int clr,
                      const float *params, uint32_t *clow, uint32_t *chigh, int16_t *sym)
{
    int Hl, Wl, h, w, padH, padW, hc, wc;
    orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
    stream_dims(h, w, padH, padW, band, &hc, &wc);
    const long plane_sz = (long)H * W;
    const int minv = (clr == 0) ? -127 : minmax[clr];
    const int maxv = (clr == 0) ? 128 : minmax[3 + clr];
    const int shift = (clr == 0) ? 127 : -minmax[clr];   /* _adjust_mean_shifts, LLICTI_nets.py:544-547 */
    const int Lp = maxv - minv + 2;
    const int src = band + 1;
#pragma omp parallel for schedule(static) num_threads(nthreads())
    for (long n = 0; n < (long)hc * wc; ++n) {
        int i = (int)(n / wc), j = (int)(n % wc);
        long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
        int yv = planes[off], cov = planes[plane_sz + off];
        int v = planes[clr * plane_sz + off];
        int s = v + shift;
        mix_t m;
        mix_prepare(params + ((long)i * w + j) * ORC_NPAR, clr, (float)yv / 255.0f, (float)cov / 255.0f, &m);
        clow[n] = cdf_entry(&m, s, Lp, minv, maxv);
        chigh[n] = (s == Lp - 2) ? 0x10000u : cdf_entry(&m, s + 1, Lp, minv, maxv);
        sym[n] = (int16_t)s;
    }
    return (long)hc * wc;
}

/* ------------------------------------------------------------------ whole-image encode */
long orc_encode_image(const uint8_t *rgb, int H, int W, const orc_weights *wts, int full_tables,
                      uint8_t *out, long cap, int32_t seg_len[49])
{
    if (H < 32 || W < 32 || H > 8160 || W > 8160) return -2;
    const long plane_sz = (long)H * W;
    int16_t *planes = (int16_t *)malloc(sizeof(int16_t) * 3 * plane_sz);
    int16_t minmax[6];
    orc_lift(rgb, H, W, planes, minmax);
    int Hl, Wl, h, w, padH, padW;
    long pos = 0;
    /* header (LLICTI_nets.py:346-354) */
    orc_level_geom(H, W, 4, &Hl, &Wl, &h, &w, &padH, &padW);
    const int h4 = h, w4 = w;
    if (cap < 17 + 3L * h4 * w4) { free(planes); return -1; }
    out[pos++] = ORC_NLEV; out[pos++] = (uint8_t)h4; out[pos++] = (uint8_t)w4;
    seg_len[0] = 3;
    memcpy(out + pos, minmax, 12); pos += 12; seg_len[1] = 12;
    int padint = 0;
    for (int l = 0; l < ORC_NLEV; ++l) {
        orc_level_geom(H, W, l, &Hl, &Wl, &h, &w, &padH, &padW);
        padint = 4 * padint + 2 * padH + padW;          /* LLICTI_nets.py:230, level 0 most significant */
    }
    int16_t padi16 = (int16_t)padint;
    memcpy(out + pos, &padi16, 2); pos += 2; seg_len[2] = 2;
    for (int c = 0; c < 3; ++c)                          /* raw DC band = x[::32, ::32], uint8 CHW (:248-252) */
        for (int i = 0; i < h4; ++i)
            for (int j = 0; j < w4; ++j) out[pos++] = rgb[c * plane_sz + (long)(32 * i) * W + 32 * j];
    seg_len[3] = 3 * h4 * w4;

    int si = 4;
    for (int lvl = ORC_NLEV - 1; lvl >= 0; --lvl) {
        orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
        float *params = (float *)malloc(sizeof(float) * (long)h * w * ORC_NPAR);
        uint32_t *clow = (uint32_t *)malloc(sizeof(uint32_t) * (long)h * w);
        uint32_t *chigh = (uint32_t *)malloc(sizeof(uint32_t) * (long)h * w);
        int16_t *sym = (int16_t *)malloc(sizeof(int16_t) * (long)h * w);
        for (int band = 0; band < 3; ++band) {
            orc_band_params(planes, H, W, lvl, band, &wts->band[band], params);
            for (int clr = 0; clr < 3; ++clr) {
                long n;
                long wrote;
                if (!full_tables) {
                    n = orc_stream_pairs(planes, H, W, minmax, lvl, band, clr, params, clow, chigh, sym);
                    wrote = orc_ac_encode_pairs(clow, chigh, n, out + pos, cap - pos);
                } else {
                    /* reference structure: materialise the whole [N][Lp] table, then code from it */
                    int hc, wc;
                    stream_dims(h, w, padH, padW, band, &hc, &wc);
                    n = (long)hc * wc;
                    const int minv = (clr == 0) ? -127 : minmax[clr];
                    const int maxv = (clr == 0) ? 128 : minmax[3 + clr];
                    const int shift = (clr == 0) ? 127 : -minmax[clr];
                    const int Lp = maxv - minv + 2;
                    const int src = band + 1;
                    uint16_t *tab = (uint16_t *)malloc(sizeof(uint16_t) * n * Lp);
#pragma omp parallel for schedule(static) num_threads(nthreads())
                    for (long q = 0; q < n; ++q) {
                        int i = (int)(q / wc), j = (int)(q % wc);
                        long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                        orc_cdf_row(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                    (float)planes[plane_sz + off] / 255.0f, minv, maxv, tab + q * Lp);
                        sym[q] = (int16_t)(planes[clr * plane_sz + off] + shift);
                    }
                    wrote = orc_ac_encode_tables(tab, Lp, sym, n, out + pos, cap - pos);
                    free(tab);
                }
                if (wrote < 0) { free(params); free(clow); free(chigh); free(sym); free(planes); return -1; }
                seg_len[si++] = (int32_t)wrote;
                pos += wrote;
            }
        }
        free(params); free(clow); free(chigh); free(sym);
    }
    free(planes);
    return pos;
}

/* ------------------------------------------------------------------ whole-image decode */
void orc_header_dims(const uint8_t *in, const int32_t seg_len[49], int *H_out, int *W_out)
{
    (void)seg_len;
    int h4 = in[1], w4 = in[2];   /* byte 0: number of scales (AC container) or 0x80|lgM<<4|5 (rANS container) */
    int16_t padi16;
    memcpy(&padi16, in + 15, 2);
    int padint = padi16;
    /* _get_padHW_lev_list (LLICTI_nets.py:533-542): two bits per level, level 4 in the lowest bits */
    int Hc = h4, Wc = w4;
    for (int l = ORC_NLEV - 1; l >= 0; --l) {
        int padW = padint & 1; padint >>= 1;
        int padH = padint & 1; padint >>= 1;
        Hc = 2 * Hc - padH;     /* H_l = 2 h_l - padH_l, and h_l = H_{l+1} */
        Wc = 2 * Wc - padW;
    }
    *H_out = Hc; *W_out = Wc;
}

int orc_decode_image(const uint8_t *in, const int32_t seg_len[49], const orc_weights *wts, int full_tables,
                     uint8_t *rgb, long rgb_cap, int *H_out, int *W_out)
{
    if (seg_len[0] != 3 || seg_len[1] != 12 || seg_len[2] != 2) return -3;
    if (in[0] != ORC_NLEV) return -4;                     /* assert num_scales (LLICTI_nets.py:424) */
    int H, W;
    orc_header_dims(in, seg_len, &H, &W);
    *H_out = H; *W_out = W;
    const long plane_sz = (long)H * W;
    if (rgb_cap < 3 * plane_sz) return -1;
    int16_t minmax[6];
    memcpy(minmax, in + 3, 12);
    const int h4 = in[1], w4 = in[2];
    if (seg_len[3] != 3 * h4 * w4) return -3;
    int16_t *planes = (int16_t *)calloc(3 * plane_sz, sizeof(int16_t));
    /* DC band: uint8 RGB -> YCoCg-R (LLICTI_nets.py:430, :443-444) */
    const uint8_t *dc = in + 17;
    for (int i = 0; i < h4; ++i)
        for (int j = 0; j < w4; ++j) {
            int R = dc[i * w4 + j], G = dc[h4 * w4 + i * w4 + j], B = dc[2 * h4 * w4 + i * w4 + j];
            int Co = R - B, t = B + fdiv2(Co), Cg = G - t, Y = t + fdiv2(Cg);
            long off = (long)(32 * i) * W + 32 * j;
            planes[off] = (int16_t)(Y - 127);
            planes[plane_sz + off] = (int16_t)Co;
            planes[2 * plane_sz + off] = (int16_t)Cg;
        }
    long pos = 17 + seg_len[3];
    int si = 4;
    int Hl, Wl, h, w, padH, padW;
    for (int lvl = ORC_NLEV - 1; lvl >= 0; --lvl) {
        orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
        float *params = (float *)malloc(sizeof(float) * (long)h * w * ORC_NPAR);
        for (int band = 0; band < 3; ++band) {
            orc_band_params(planes, H, W, lvl, band, &wts->band[band], params);
            const int src = band + 1;
            int hc, wc;
            stream_dims(h, w, padH, padW, band, &hc, &wc);
            const long n = (long)hc * wc;
            for (int clr = 0; clr < 3; ++clr) {
                const int minv = (clr == 0) ? -127 : minmax[clr];
                const int maxv = (clr == 0) ? 128 : minmax[3 + clr];
                const int shift = (clr == 0) ? 127 : -minmax[clr];
                const int Lp = maxv - minv + 2;
                const uint16_t max_symbol = (uint16_t)(Lp - 2);
                acdec_t d;
                acdec_init(&d, in + pos, seg_len[si]);
                uint16_t *tab = NULL;
                if (full_tables) {
                    /* reference structure: the whole [N][Lp] table of the stage is materialised first
                     * (get_cdfs, LLICTI_nets.py:489), then the stream is decoded from it (:492) */
                    tab = (uint16_t *)malloc(sizeof(uint16_t) * n * Lp);
#pragma omp parallel for schedule(static) num_threads(nthreads())
                    for (long q = 0; q < n; ++q) {
                        int i = (int)(q / wc), j = (int)(q % wc);
                        long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                        orc_cdf_row(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                    (float)planes[plane_sz + off] / 255.0f, minv, maxv, tab + q * Lp);
                    }
                }
                for (long q = 0; q < n; ++q) {
                    int i = (int)(q / wc), j = (int)(q % wc);
                    long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                    mix_t m;
                    const uint16_t *row = full_tables ? tab + q * Lp : NULL;
                    if (!full_tables)
                        mix_prepare(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                    (float)planes[plane_sz + off] / 255.0f, &m);
                    const uint16_t count = acdec_count(&d);
                    /* torchac binsearch, entries evaluated on demand (same values as the table's) */
                    uint16_t left = 0, right = (uint16_t)(max_symbol + 1);
                    int found = -1;
                    while (left + 1 < right) {
                        uint16_t mid = (uint16_t)((left + right) / 2);
                        uint16_t v = full_tables ? row[mid] : cdf_entry(&m, mid, Lp, minv, maxv);
                        if (v < count) left = mid; else if (v > count) right = mid; else { found = mid; break; }
                    }
                    uint16_t s = (found >= 0) ? (uint16_t)found : left;
                    planes[clr * plane_sz + off] = (int16_t)((int)s - shift);   /* _convert_int16cpu_to_float32gpu */
                    if (q == n - 1) break;
                    uint32_t c_low = full_tables ? row[s] : cdf_entry(&m, s, Lp, minv, maxv);
                    uint32_t c_high = (s == max_symbol) ? 0x10000u
                                    : (full_tables ? row[s + 1] : cdf_entry(&m, s + 1, Lp, minv, maxv));
                    acdec_update(&d, c_low, c_high);
                }
                free(tab);
                pos += seg_len[si++];
            }
        }
        free(params);
    }
    orc_unlift(planes, H, W, rgb);
    free(planes);
    return 0;
}

/* ------------------------------------------------------------------ rANS container "LLICTI-rANS v3" (see header) */
typedef struct { long n; uint32_t *clow, *chigh; int16_t *sym; } stage_syms_t;

#define RANS_MAX_LANES  256                             /* lanes of a stream: 64, 128 ("wide": two 64-symbol chunks per step) or 256 ("xwide": four) */
#define RANS_STATE_BITS 31                              /* a lane state is 2^31 | 31 bits */
#define RANS_MAX_PAY_BITS (RANS_MAX_LANES * RANS_STATE_BITS)   /* what the initial states carry (the tail stream): 1984 / 3968 / 7936 bits */
#define RANS_TAIL_MAX   2047                            /* tail symbols of a 64- / 128-lane stream (the 11-bit T field) */
/* xwide streams (256 lanes) are "v4" (round 6; header: padHW bits 10..15 say so and carry the stream count) -- the 64- and 128-lane kinds keep
 * their v3 bytes.  What v4 changes (spec: llicti_oracle.h, "xwide v4"):
 *   arena    the tail coder's output is no longer cut to the 7936 payload bits the initial states carry: the stream's last T symbols are taken in
 *            blocks of 32 until the output reaches the payload size, and what exceeds it ("spill", < 512 bits) lies at the BOTTOM of the main bit
 *            region, where the main decoder -- reading DOWN -- leaves it.  No unused payload bits, and T is a multiple of 32: an 8-bit field;
 *   one chain  starts from the stream's LAST symbol itself, raw (x = its index: 9 bits of state for one symbol, nothing else), and pushes without
 *            emitting while the state is below the interval: a chain costs ~1 bit (its end marker) instead of the ~31 a start state of 2^31 did;
 *   two chains  (expensive symbols: v3's integer rule) as in v3 -- seeded with n raw symbols each -- at the two ends of the arena;
 *   header   9 bits (T / 32 rounded up, the one-chain flag) on TOP of the bit region under an end marker, instead of a u16 in front and an escape. */
#define RANS_TAIL_MAX_X 8160                            /* = 32 * 255: the cap of an xwide stream's tail (the field is T / 32 rounded up, 8 bits) */
#define RANS_TAIL_BLOCK 32
#define RANS_SPILL_MAX  512                             /* bits by which the tail coder's output may exceed the payload: < 32 symbols x 16 bits */
#define RANS_SEED_LANES 256
#define RANS_SEED_MAX   31
static inline int rans_seed_count(int A, uint32_t *pw)          /* n and A^n: the largest count with A^n <= 2^31 (at most 31) */
{
    int n = 0;
    uint64_t p = 1;
    while (n < RANS_SEED_MAX && p * (uint64_t)A <= (1ull << 31)) { p *= (uint64_t)A; ++n; }
    *pw = (uint32_t)p;
    return n;
}

static inline void put_bits(uint8_t *buf, long pos, int n, uint32_t v)      /* LSB first */
{
    for (int b = 0; b < n; ++b, ++pos)
        if ((v >> b) & 1u) buf[pos >> 3] |= (uint8_t)(1u << (pos & 7));
}
static inline uint32_t get_bits(const uint8_t *buf, long pos, int n)        /* pos >= 0 */
{
    uint32_t v = 0;
    for (int b = 0; b < n; ++b, ++pos) v |= (uint32_t)((buf[pos >> 3] >> (pos & 7)) & 1u) << b;
    return v;
}
/* encoder renormalisation: the smallest n with (x >> n) < freq << 16, for a state x in [2^31, 2^32) */
static inline int rans_emit_bits(uint32_t x, uint32_t freq)
{
    int n = 0;
    while (((uint64_t)x >> n) >= ((uint64_t)freq << 16)) ++n;
    return n;
}
static inline uint32_t rans_push(uint32_t x, uint32_t lo, uint32_t freq) { return ((x / freq) << 16) + (x % freq) + lo; }
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

/* symbols of stream m in a stage of nc symbols: chunks m, m + M, ...; only the stage's last chunk can be partial */
static long rans_stream_count(long nc, int m, int M, int L)
{
    const long nchunks = (nc + L - 1) / L;
    if (nchunks <= m) return 0;
    const long K = (nchunks - m + M - 1) / M;
    const long last = m + (K - 1) * M;
    return L * K - ((last == nchunks - 1 && (nc % L)) ? L - (nc % L) : 0);
}

/* the xwide v4 tail: fills arena (zeroed, RANS_MAX_PAY_BITS + RANS_SPILL_MAX bits), returns T (or -1 on an impossible pair); *alen = the arena's
 * length in bits (>= the payload's 7936: what exceeds it is the spill), *single = one chain */
static long rans_tail_encode_x(const stage_syms_t *sl, int m, int M, long cnt, int A, uint8_t *arena, long *alen, int *single)
{
    const int L = RANS_SEED_LANES;
    const long P = (long)L * RANS_STATE_BITS;
    uint32_t pw;
    const int ns = rans_seed_count(A, &pw);
    /* one chain or two -- v3's integer rule, on the stream's last up to 64 symbols (a mean, so that the streams of a statistically uniform batch
     * decide alike): two chains iff the stream has 2 ns symbols and ns * mean(16 - floor(log2 freq)) >= 32 + ns / 2 */
    int nch = 1;
    if (cnt >= 2 * ns) {
        const long k64 = cnt < 64 ? cnt : 64;
        long wsum = 0;
        for (long jj = 0; jj < k64; ++jj) {
            const long q = cnt - 1 - jj;
            const long n = (long)L * (m + (q / L) * M) + (q % L);
            const uint32_t freq = sl->chigh[n] - sl->clow[n];
            if (freq == 0 || freq > 0x10000u) return -1;
            wsum += clz32(freq) - 15;
        }
        if (2 * wsum * ns >= k64 * (64 + ns)) nch = 2;
    }
    *single = (nch == 1);
    uint32_t x[2] = { 0, 0 };
    long j = 0;
    if (nch == 2) {
        for (int c = 0; c < 2; ++c) {                           /* seeds: n raw symbol indices per chain, radix A, on top of 2^31 */
            uint32_t mul = 1;
            x[c] = 1u << 31;
            for (int i = 0; i < ns && j < cnt; ++i, ++j, mul *= (uint32_t)A) {
                const long q = cnt - 1 - j;
                const int sy = sl->sym[(long)L * (m + (q / L) * M) + (q % L)];
                if (sy < 0 || sy >= A) return -1;
                x[c] += (uint32_t)sy * mul;
            }
        }
    } else if (cnt > 0) {                                       /* one chain: its start state IS the last symbol's index */
        const long q = cnt - 1;
        const int sy = sl->sym[(long)L * (m + (q / L) * M) + (q % L)];
        if (sy < 0 || sy >= A) return -1;
        x[0] = (uint32_t)sy;
        j = 1;
    }
    const long j0 = j;                                          /* first coded symbol */
    const long fixed = nch == 2 ? 64 : 33;                      /* the final states; one chain: + its end marker */
    static __thread uint32_t fld[RANS_TAIL_MAX_X];
    static __thread uint8_t fnb[RANS_TAIL_MAX_X];
    long used[2] = { 0, 0 };
    for (; j < cnt && j < RANS_TAIL_MAX_X; ++j) {
        if (j % RANS_TAIL_BLOCK == 0 && used[0] + used[1] + fixed >= P) break;      /* the payload is full: T is this multiple of 32 */
        const long q = cnt - 1 - j;
        const long n = (long)L * (m + (q / L) * M) + (q % L);
        const uint32_t lo = sl->clow[n], freq = sl->chigh[n] - lo;
        if (freq == 0 || freq > 0x10000u) return -1;
        const int c = (int)((j - j0) % nch);
        const int nb = rans_emit_bits(x[c], freq);              /* (a state still below 2^31 -- one chain, its first symbols -- emits nothing unless the push would overflow) */
        fld[j] = x[c] & ((1u << nb) - 1u);
        fnb[j] = (uint8_t)nb;
        used[c] += nb;
        x[c] = rans_push(x[c] >> nb, lo, freq);
    }
    const long T = j;
    long len = used[0] + used[1] + fixed;
    if (len < P) len = P;
    *alen = len;
    put_bits(arena, 0, 32, x[0]);
    if (nch == 2) put_bits(arena, len - 32, 32, x[1]);
    long pa = 32, pb = len - 32;
    for (long t = T - 1; t >= j0; --t) {                        /* the decoder's order: chain A's fields UP from bit 32, chain B's DOWN from its state */
        if ((t - j0) % nch) { pb -= fnb[t]; put_bits(arena, pb, fnb[t], fld[t]); }
        else { put_bits(arena, pa, fnb[t], fld[t]); pa += fnb[t]; }
    }
    if (nch == 1) put_bits(arena, pa, 1, 1u);                   /* end marker: the arena's highest set bit */
    return T;
}

/* "auto" xwide encodes: the stream count of an image from the image itself -- Mlo, what its size gives, and S = the sum over the n symbols of its
 * LAST stage of (16 - floor(log2 freq)): expensive symbols (S >= 11 n) -> Mlo + ceil(Mlo / 3) (at most 32); cheap symbols (S < 4 n) -> ceil(2 Mlo / 3);
 * otherwise Mlo; and a count whose payloads of 7,936 bits the last stage cannot fill with a tenth to spare (2 S - n < 2 * 8704 M) falls to Mlo (from
 * above) or to ceil(Mlo / 2).  (llicti_amd/csrc/host_types.hpp: rans_auto_pick) */
int orc_auto_streams(int Mlo, const uint32_t *clow, const uint32_t *chigh, long n)
{
    long long S = 0;
    for (long i = 0; i < n; ++i) {
        uint32_t f = chigh[i] - clow[i];
        if (f == 0 || f > 0x10000u) f = 1;
        S += clz32(f) - 15;
    }
    if (n <= 0) return Mlo;
    const int hi = (Mlo + (Mlo + 2) / 3) > 32 ? 32 : Mlo + (Mlo + 2) / 3;
    int M = Mlo;
    if (S >= 11 * (long long)n) M = hi;                          /* expensive symbols: ~2.5 bytes per stream */
    else if (S < 4 * (long long)n) M = (2 * Mlo + 2) / 3;        /* cheap symbols: ~5.1 bytes per stream, long serial tails */
    if (2 * S - n < 2LL * 8704 * M) M = (M > Mlo && 2 * S - n >= 2LL * 8704 * Mlo) ? Mlo : (Mlo + 1) / 2;      /* the last stage must fill the payloads */
    return M;
}

long orc_encode_image_rans(const uint8_t *rgb, int H, int W, const orc_weights *wts, int M,
                           uint8_t *out, long cap, int32_t seg_len[49])
{
    if (H < 32 || W < 32 || H > 8160 || W > 8160) return -2;
    const int wide = (M >> 8) & 3;                      /* M | 0x100: wide streams of 128 lanes (M <= 14); M | 0x200: xwide streams of 256 lanes (v4: M <= 32, 64, 128) */
    const int autoM = (M >> 12) & 1;                    /* M | 0x200 | 0x1000: the count is the size rule's; the encoder picks the image's own from its last stage (orc_auto_streams) */
    M &= 0xFF;
    if (autoM && (wide != 2 || M < 1 || M > 32)) return -2;
    if (wide == 0 && (M < 1 || (M > 32 && M != 64 && M != 128))) return -2;
    if (wide == 1 && (M < 1 || M > 14)) return -2;
    if (wide == 2 && (M < 1 || (M > 32 && M != 64 && M != 128))) return -2;
    if (wide == 3) return -2;
    const int L = 64 << wide;
    const int PAY_BITS = L * RANS_STATE_BITS;
    const int G = M > 32 ? M / 32 : 1;                  /* streams per segment (the reference's list has 45 stream slots: 32 are used) */
    const long plane_sz = (long)H * W;
    int16_t *planes = (int16_t *)malloc(sizeof(int16_t) * 3 * plane_sz);
    int16_t minmax[6];
    orc_lift(rgb, H, W, planes, minmax);
    int Hl, Wl, h, w, padH, padW;
    long pos = 0;
    orc_level_geom(H, W, 4, &Hl, &Wl, &h, &w, &padH, &padW);
    const int h4 = h, w4 = w;
    if (cap < 17 + 3L * h4 * w4) { free(planes); return -1; }
    for (int i = 0; i < 49; ++i) seg_len[i] = 0;
    {   /* byte 0: bit 7 rANS, bit 3 format v3+, bit 6 extended, bits 5,4,2,1,0 = v: M = v + 1 (<= 32) streams of 64 lanes; extended: v = 0, 1:
         * 64 / 128 streams of 64 lanes; 2 .. 15: v - 1 wide streams; v = 16 (byte 0 = 0xE8): xwide streams, v4 -- their count is in the pad field (below).
         * (v = 17 .. 31 were the xwide v3 tags of rounds 4-5: retired, rejected.) */
        const int lat = M > 32 || wide;
        const int v = wide == 2 ? 16 : wide == 1 ? M + 1 : M > 32 ? (M == 64 ? 0 : 1) : M - 1;
        out[pos++] = (uint8_t)(0x88 | (lat << 6) | (((v >> 3) & 3) << 4) | (v & 7));
    }
    out[pos++] = (uint8_t)h4; out[pos++] = (uint8_t)w4;
    seg_len[0] = 3;
    memcpy(out + pos, minmax, 12); pos += 12; seg_len[1] = 12;
    int padint = 0;
    for (int l = 0; l < ORC_NLEV; ++l) {
        orc_level_geom(H, W, l, &Hl, &Wl, &h, &w, &padH, &padW);
        padint = 4 * padint + 2 * padH + padW;
    }
    /* xwide v4: bits 10 .. 15 of the pad field (zero in every other container: the five levels' flags are bits 0 .. 9) = u: M = u for 1 .. 32,
     * u = 33 / 34: 64 / 128 streams (two / four per segment).  A reader of the older formats finds a pad field that contradicts the size and refuses. */
    if (wide == 2) padint |= (M <= 32 ? M : M == 64 ? 33 : 34) << 10;
    int16_t padi16 = (int16_t)(uint16_t)padint;
    memcpy(out + pos, &padi16, 2); pos += 2; seg_len[2] = 2;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < h4; ++i)
            for (int j = 0; j < w4; ++j) out[pos++] = rgb[c * plane_sz + (long)(32 * i) * W + 32 * j];
    seg_len[3] = 3 * h4 * w4;

    /* all 45 stages' (c_low, c_high), in decode order */
    stage_syms_t st[ORC_NSTREAM];
    int si = 0;
    long total = 0;
    for (int lvl = ORC_NLEV - 1; lvl >= 0; --lvl) {
        orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
        float *params = (float *)malloc(sizeof(float) * (long)h * w * ORC_NPAR);
        int16_t *sym = (int16_t *)malloc(sizeof(int16_t) * (long)h * w);
        for (int band = 0; band < 3; ++band) {
            orc_band_params(planes, H, W, lvl, band, &wts->band[band], params);
            for (int clr = 0; clr < 3; ++clr) {
                st[si].clow = (uint32_t *)malloc(sizeof(uint32_t) * (long)h * w);
                st[si].chigh = (uint32_t *)malloc(sizeof(uint32_t) * (long)h * w);
                st[si].n = orc_stream_pairs(planes, H, W, minmax, lvl, band, clr, params, st[si].clow, st[si].chigh, sym);
                st[si].sym = NULL;
                if (si == ORC_NSTREAM - 1) {                 /* the raw seed of an xwide stream's tail coder needs symbol indices */
                    st[si].sym = (int16_t *)malloc(sizeof(int16_t) * (long)h * w);
                    memcpy(st[si].sym, sym, sizeof(int16_t) * st[si].n);
                }
                total += st[si].n;
                ++si;
            }
        }
        free(params); free(sym);
    }
    free(planes);
    if (autoM) {
        /* the image's own count (host_types.hpp: rans_auto_pick), from what the symbols of its last stage cost; the pad field says which */
        M = orc_auto_streams(M, st[ORC_NSTREAM - 1].clow, st[ORC_NSTREAM - 1].chigh, st[ORC_NSTREAM - 1].n);
        const int pf = (padint & 0x3FF) | (M << 10);
        out[15] = (uint8_t)(pf & 0xFF); out[16] = (uint8_t)((pf >> 8) & 0xFF);
    }
    long rc = 0;
    const long bcap = 2 * total + 256;                   /* <= 16 bits per symbol + (xwide) spill and header field */
    uint8_t *bits = (uint8_t *)malloc(bcap);
    const int S = ORC_NSTREAM - 1;                       /* the last stage: the only one the tail may take symbols from */
    for (int m = 0; m < M && rc >= 0; ++m) {
        /* 1. tail: the stream's last T symbols (decode order), single-state coder, pushed last symbol first; its bits
         *    go UP from bit 0 of the payload, its final state (32 bits, leading one = the payload's highest set bit) on top */
        uint8_t pay[(RANS_MAX_PAY_BITS + RANS_SPILL_MAX) / 8 + 8];   /* xwide: the arena (payload ++ spill) */
        memset(pay, 0, sizeof pay);
        const long cnt = rans_stream_count(st[S].n, m, M, L);
        uint32_t xt = 1u << 31;
        long tb = 0, T = 0, alen = PAY_BITS;
        int single = 0;                                      /* xwide: one tail chain instead of two (bit 8 of the stream's header field) */
        if (L == RANS_SEED_LANES) {                          /* xwide v4 */
            T = rans_tail_encode_x(&st[S], m, M, cnt, minmax[5] - minmax[2] + 1, pay, &alen, &single);
            if (T < 0) { rc = -5; break; }
        }
        while (L != RANS_SEED_LANES && T < cnt && T < RANS_TAIL_MAX) {
            const long q = cnt - 1 - T;
            const long n = (long)L * (m + (q / L) * M) + (q % L);
            const uint32_t lo = st[S].clow[n], freq = st[S].chigh[n] - lo;
            if (freq == 0 || freq > 0x10000u) { rc = -5; break; }
            if (T == 0) xt = freq << 15;          /* absorbing start: the first pushed symbol codes to 2^31 + c_low, no bits */
            const int nb = rans_emit_bits(xt, freq);
            if (tb + nb + 32 > PAY_BITS) break;
            put_bits(pay, tb, nb, xt & ((1u << nb) - 1u));
            tb += nb;
            xt = rans_push(xt >> nb, lo, freq);
            ++T;
        }
        if (rc < 0) break;
        if (L != RANS_SEED_LANES) put_bits(pay, tb, 32, xt);
        /* 2. the L lanes start from the payload: lane l = 2^31 | payload bits [31 l, 31 l + 31) */
        uint32_t x[RANS_MAX_LANES];
        for (int l = 0; l < L; ++l) x[l] = (1u << 31) | get_bits(pay, (long)RANS_STATE_BITS * l, RANS_STATE_BITS);
        /* 3. main coder, last decoded symbol first; bits go UP from bit 0 of the stream's bit region (xwide v4: from the end of the spill, which is
         *    what the tail coder's output has beyond the payload -- the main decoder, reading DOWN, stops there and leaves it to the tail decoder) */
        memset(bits, 0, bcap);
        long bp = alen - PAY_BITS;
        for (long e = 0; e < bp; ++e)
            if ((pay[(PAY_BITS + e) >> 3] >> ((PAY_BITS + e) & 7)) & 1u) bits[e >> 3] |= (uint8_t)(1u << (e & 7));
        for (int s = ORC_NSTREAM - 1; s >= 0 && rc >= 0; --s) {
            const long nchunks = (st[s].n + L - 1) / L;
            if (nchunks <= m) continue;
            const long K = (nchunks - m + M - 1) / M;
            for (long k = K - 1; k >= 0 && rc >= 0; --k) {
                const long c = m + k * M;
                for (int l = L - 1; l >= 0; --l) {     /* highest lane first: the decoder renormalises lane-ascending, reading DOWN */
                    const long n = (long)L * c + l;
                    if (n >= st[s].n) continue;
                    if (s == S && L * k + l >= cnt - T) continue;          /* coded by the tail */
                    const uint32_t lo = st[s].clow[n], freq = st[s].chigh[n] - lo;
                    if (freq == 0 || freq > 0x10000u) { rc = -5; break; }
                    const int nb = rans_emit_bits(x[l], freq);
                    put_bits(bits, bp, nb, x[l] & ((1u << nb) - 1u));
                    bp += nb;
                    x[l] = rans_push(x[l] >> nb, lo, freq);
                }
            }
        }
        if (rc < 0) break;
        long hdr = 0;                                        /* bytes in front of the bit region: the u16 of the 64- / 128-lane kinds */
        if (L == RANS_SEED_LANES) {
            /* xwide v4: the header field on TOP of the bit region -- 8 bits T / 32 rounded up (the decoder takes min(32 v, its own count of the
             * stream's last-stage symbols)), 1 bit "one chain" -- and an end marker bit above it; zero bits up to the byte boundary */
            const long v = (T + RANS_TAIL_BLOCK - 1) / RANS_TAIL_BLOCK;
            put_bits(bits, bp, 10, (uint32_t)(v | ((long)single << 8) | (1l << 9)));
            bp += 10;
        } else hdr = 2;
        const long nbytes = (bp + 7) / 8;
        const long padb = 8 * nbytes - bp;               /* unused (zero) bits on top of the region's last byte */
        const long bytes = hdr + nbytes + PAY_BITS / 8;
        if (pos + bytes + 4 * G > cap) { rc = -1; break; }
        if (G > 1) {
            /* M = 64 / 128: segment m / G = G little-endian u32 stream lengths, then its G streams */
            if (m % G == 0) { memset(out + pos, 0, 4 * G); pos += 4 * G; seg_len[4 + m / G] = 4 * G; }
            uint8_t *tab = out + pos - seg_len[4 + m / G] + 4 * (m % G);
            tab[0] = (uint8_t)(bytes & 0xFF); tab[1] = (uint8_t)((bytes >> 8) & 0xFF); tab[2] = (uint8_t)((bytes >> 16) & 0xFF); tab[3] = (uint8_t)(bytes >> 24);
        }
        if (hdr) {
            const long t16 = (T & 0x7FF) | (padb << 11);
            out[pos] = (uint8_t)(t16 & 0xFF); out[pos + 1] = (uint8_t)(t16 >> 8);
        }
        memcpy(out + pos + hdr, bits, nbytes);
        uint8_t *fs = out + pos + hdr + nbytes;
        memset(fs, 0, PAY_BITS / 8);
        for (int l = 0; l < L; ++l) put_bits(fs, (long)RANS_STATE_BITS * l, RANS_STATE_BITS, x[l] & 0x7FFFFFFFu);
        pos += bytes;
        seg_len[4 + m / G] += (int32_t)bytes;
    }
    free(bits);
    for (int s = 0; s < ORC_NSTREAM; ++s) { free(st[s].clow); free(st[s].chigh); free(st[s].sym); }
    return rc < 0 ? rc : pos;
}

/* exact symbol search on the spec's table: largest idx in [0, max_symbol] whose entry is <= slot (idx 0 always qualifies) */
static inline int rans_find(const mix_t *mx, uint32_t slot, int Lp, int minv, int maxv, uint32_t *c_low, uint32_t *c_high)
{
    const int max_symbol = Lp - 2;
    int lo = 0, hi = max_symbol + 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cdf_entry(mx, mid, Lp, minv, maxv) <= slot) lo = mid; else hi = mid;
    }
    *c_low = cdf_entry(mx, lo, Lp, minv, maxv);
    *c_high = (lo == max_symbol) ? 0x10000u : cdf_entry(mx, lo + 1, Lp, minv, maxv);
    return lo;
}

int orc_decode_image_rans(const uint8_t *in, const int32_t seg_len[49], const orc_weights *wts,
                          uint8_t *rgb, long rgb_cap, int *H_out, int *W_out)
{
    if (seg_len[0] != 3 || seg_len[1] != 12 || seg_len[2] != 2) return -3;
    if ((in[0] & 0x88) != 0x88) return -4;                        /* bit 3 clear: the retired v2 format */
    const int tagv = (((in[0] >> 4) & 3) << 3) | (in[0] & 7);
    const int ext = (in[0] >> 6) & 1, wide = !ext || tagv < 2 ? 0 : tagv < 16 ? 1 : 2;
    const int padfield = (int)(uint16_t)(in[15] | (in[16] << 8));
    const int u6 = padfield >> 10;                                /* xwide v4: the stream count; zero in every other container */
    if (wide == 2 && (tagv != 16 || u6 < 1 || u6 > 34)) return -4; /* (tagv 17 .. 31, or no count: the xwide v3 layout of rounds 4-5, retired) */
    if (wide != 2 && u6) return -4;
    const int M = wide == 2 ? (u6 <= 32 ? u6 : u6 == 33 ? 64 : 128) : !ext ? tagv + 1 : tagv == 0 ? 64 : tagv == 1 ? 128 : tagv - 1;
    const int L = 64 << wide;
    const int PAY_BITS = L * RANS_STATE_BITS;
    const int G = M > 32 ? M / 32 : 1;
    int H, W;
    orc_header_dims(in, seg_len, &H, &W);
    *H_out = H; *W_out = W;
    const long plane_sz = (long)H * W;
    if (rgb_cap < 3 * plane_sz) return -1;
    int16_t minmax[6];
    memcpy(minmax, in + 3, 12);
    const int h4 = in[1], w4 = in[2];
    if (seg_len[3] != 3 * h4 * w4) return -3;
    /* streams: u16 (T | pad) | bit region (read DOWN from its top) | L x 31-bit states; xwide v4: bit region under its header field | states */
    long cnt_last[128];                                            /* every stream's share of the LAST stage (level 0, band x10): an xwide tail cannot be longer */
    {
        int Hl0, Wl0, h0, w0, padH0, padW0, hc0, wc0;
        orc_level_geom(H, W, 0, &Hl0, &Wl0, &h0, &w0, &padH0, &padW0);
        stream_dims(h0, w0, padH0, padW0, 2, &hc0, &wc0);
        for (int m = 0; m < M; ++m) cnt_last[m] = rans_stream_count((long)hc0 * wc0, m, M, L);
    }
    uint32_t (*x)[RANS_MAX_LANES] = (uint32_t (*)[RANS_MAX_LANES])malloc(sizeof(uint32_t) * RANS_MAX_LANES * M);
    const uint8_t **bitsp = (const uint8_t **)malloc(sizeof(uint8_t *) * M);
    long *cur = (long *)calloc(M, sizeof(long)), *T = (long *)calloc(M, sizeof(long));
    int bad = 0;
    {
        long pos = 17 + seg_len[3];
        long seg_left = 0;
        for (int m = 0; m < M; ++m) {
            long len = seg_len[4 + m / G];
            if (G > 1) {
                if (m % G == 0) {
                    seg_left = len - 4 * G;
                    if (seg_left < 0) { free(x); free(bitsp); free(cur); free(T); return -3; }
                    pos += 4 * G;
                }
                const uint8_t *seg0 = in + 17 + seg_len[3];           /* the segment's table of G stream lengths */
                for (int sgi = 0; sgi < m / G; ++sgi) seg0 += seg_len[4 + sgi];
                const uint8_t *tab = seg0 + 4 * (m % G);
                len = (long)tab[0] | ((long)tab[1] << 8) | ((long)tab[2] << 16) | ((long)tab[3] << 24);
                seg_left -= len;
                if (seg_left < 0 || (m % G == G - 1 && seg_left != 0)) { free(x); free(bitsp); free(cur); free(T); return -3; }
            }
            if (len < 2 + PAY_BITS / 8) { free(x); free(bitsp); free(cur); free(T); return -3; }
            const uint8_t *sp = in + pos;
            long nbytes, hdr;
            if (L == RANS_SEED_LANES) {
                /* xwide v4: bit region | states.  The region's highest set bit is its end marker; the 9 bits below it are the header field (T / 32
                 * rounded up, bit 8: one tail chain), the main coder's bits lie below that -- read DOWN -- on top of the tail coder's spill. */
                hdr = 0;
                nbytes = len - PAY_BITS / 8;
                const int lastb = sp[nbytes - 1];
                if (lastb == 0) { free(x); free(bitsp); free(cur); free(T); return -3; }
                const long top = 8 * (nbytes - 1) + (31 - clz32((uint32_t)lastb));
                if (top < 9) { free(x); free(bitsp); free(cur); free(T); return -3; }
                const uint32_t f9 = get_bits(sp, top - 9, 9);
                long Tv = 32L * (f9 & 0xFF);
                if (Tv > cnt_last[m]) Tv = cnt_last[m];               /* the stream's whole share of the last stage */
                T[m] = Tv | ((long)((f9 >> 8) & 1) << 16);            /* bit 16: one chain */
                cur[m] = top - 9;
            } else {
                hdr = 2;
                const int t16 = sp[0] | (sp[1] << 8);
                const int padb = (t16 >> 11) & 7;
                nbytes = len - 2 - PAY_BITS / 8;
                if ((t16 >> 14) || nbytes < 0 || (nbytes == 0 && padb)) { free(x); free(bitsp); free(cur); free(T); return -3; }
                T[m] = t16 & 0x7FF;
                cur[m] = 8 * nbytes - padb;                           /* number of data bits */
            }
            bitsp[m] = sp + hdr;
            const uint8_t *fs = sp + hdr + nbytes;
            for (int l = 0; l < L; ++l) x[m][l] = (1u << 31) | get_bits(fs, (long)RANS_STATE_BITS * l, RANS_STATE_BITS);
            pos += len;
        }
    }
    int16_t *planes = (int16_t *)calloc(3 * plane_sz, sizeof(int16_t));
    const uint8_t *dc = in + 17;
    for (int i = 0; i < h4; ++i)
        for (int j = 0; j < w4; ++j) {
            int R = dc[i * w4 + j], G = dc[h4 * w4 + i * w4 + j], B = dc[2 * h4 * w4 + i * w4 + j];
            int Co = R - B, t = B + fdiv2(Co), Cg = G - t, Y = t + fdiv2(Cg);
            long off = (long)(32 * i) * W + 32 * j;
            planes[off] = (int16_t)(Y - 127);
            planes[plane_sz + off] = (int16_t)Co;
            planes[2 * plane_sz + off] = (int16_t)Cg;
        }
    int Hl, Wl, h, w, padH, padW;
    int stage = 0;
    for (int lvl = ORC_NLEV - 1; lvl >= 0; --lvl) {
        orc_level_geom(H, W, lvl, &Hl, &Wl, &h, &w, &padH, &padW);
        float *params = (float *)malloc(sizeof(float) * (long)h * w * ORC_NPAR);
        for (int band = 0; band < 3; ++band) {
            orc_band_params(planes, H, W, lvl, band, &wts->band[band], params);
            const int src = band + 1;
            int hc, wc;
            stream_dims(h, w, padH, padW, band, &hc, &wc);
            const long n_sym = (long)hc * wc;
            const long nchunks = (n_sym + L - 1) / L;
            for (int clr = 0; clr < 3; ++clr, ++stage) {
                const int minv = (clr == 0) ? -127 : minmax[clr];
                const int maxv = (clr == 0) ? 128 : minmax[3 + clr];
                const int shift = (clr == 0) ? 127 : -minmax[clr];
                const int Lp = maxv - minv + 2;
                const int last_stage = (stage == ORC_NSTREAM - 1);
                for (long c = 0; c < nchunks; ++c) {
                    const int m = (int)(c % M);
                    const long k = c / M;
                    const long cnt = last_stage ? rans_stream_count(n_sym, m, M, L) : 0;
                    int nb[RANS_MAX_LANES];
                    for (int l = 0; l < L; ++l) nb[l] = -1;
                    for (int l = 0; l < L; ++l) {
                        const long q = (long)L * c + l;
                        if (q >= n_sym) break;
                        if (last_stage && L * k + l >= cnt - (T[m] & 0xFFFF)) continue;      /* tail symbol: decoded after the stages */
                        int i = (int)(q / wc), j = (int)(q % wc);
                        long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                        mix_t mx;
                        mix_prepare(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                    (float)planes[plane_sz + off] / 255.0f, &mx);
                        const uint32_t slot = x[m][l] & 0xFFFF;
                        uint32_t c_low, c_high;
                        const int s = rans_find(&mx, slot, Lp, minv, maxv, &c_low, &c_high);
                        x[m][l] = (c_high - c_low) * (x[m][l] >> 16) + slot - c_low;
                        planes[clr * plane_sz + off] = (int16_t)(s - shift);
                        nb[l] = clz32(x[m][l]);
                        if (nb[l] > 16) { bad = 1; nb[l] = 16; }
                    }
                    for (int l = 0; l < L; ++l) {       /* renormalise lane-ascending, reading DOWN */
                        if (nb[l] < 0) continue;
                        if (cur[m] < nb[l]) { bad = 1; x[m][l] = (x[m][l] << nb[l]) | (1u << 31); continue; }
                        cur[m] -= nb[l];
                        x[m][l] = (x[m][l] << nb[l]) | get_bits(bitsp[m], cur[m], nb[l]);
                        if (!(x[m][l] >> 31)) { bad = 1; x[m][l] |= 1u << 31; }
                    }
                }
            }
        }
        if (lvl == 0) {
            /* tails: what the L states of a stream are left with is the tail stream; its final state sits on top */
            const int band = 2, clr = 2, src = 3;
            int hc, wc;
            stream_dims(h, w, padH, padW, band, &hc, &wc);
            const long n_sym = (long)hc * wc;
            const int minv = minmax[2], maxv = minmax[5], shift = -minmax[2];
            const int Lp = maxv - minv + 2;
            for (int m = 0; m < M; ++m) {
                const long E = (L == RANS_SEED_LANES) ? cur[m] : 0;   /* xwide v4: what the main decoder leaves at the bottom of the region is the tail coder's spill */
                if (cur[m] != E || E >= RANS_SPILL_MAX) { bad = 1; continue; }      /* (64 / 128 lanes: every bit of the main region must have been read) */
                uint8_t pay[(RANS_MAX_PAY_BITS + RANS_SPILL_MAX) / 8 + 8];
                memset(pay, 0, sizeof pay);
                for (int l = 0; l < L; ++l) put_bits(pay, (long)RANS_STATE_BITS * l, RANS_STATE_BITS, x[m][l] & 0x7FFFFFFFu);
                for (long e = 0; e < E; ++e)
                    if ((bitsp[m][e >> 3] >> (e & 7)) & 1u) pay[(PAY_BITS + e) >> 3] |= (uint8_t)(1u << ((PAY_BITS + e) & 7));
                const long cnt = rans_stream_count(n_sym, m, M, L);
                const int nch = (T[m] >> 16) ? 1 : 2;            /* (xwide) */
                const long Tm = T[m] & 0xFFFF;
                if (Tm > cnt) { bad = 1; continue; }
                if (L == RANS_SEED_LANES) {
                    /* xwide v4: the arena is the payload ++ the spill.  Two chains: seeded, A's state in bits [0, 32) and its fields UP from bit 32,
                     * B's state in the arena's top 32 bits and its fields DOWN from there.  One chain: A alone; it started from the stream's last
                     * symbol (raw) and emitted nothing while its state was below the interval -- so the decoder takes min(clz, bits left) bits, and
                     * "bits left" ends at the chain's end marker, the arena's highest set bit. */
                    const long alen = PAY_BITS + E;
                    uint32_t pw;
                    const int ns = rans_seed_count(Lp - 1, &pw);
                    const long NS = nch == 2 ? (cnt < 2 * ns ? cnt : 2 * ns) : (cnt < 1 ? cnt : 1);
                    if (Tm < NS || (nch == 2 && cnt < 2 * ns)) { bad = 1; continue; }
                    long pa = 32, pb;
                    uint32_t xc[2] = { get_bits(pay, 0, 32), 0 };
                    if (nch == 2) {
                        xc[1] = get_bits(pay, alen - 32, 32);
                        pb = alen - 32;
                        if (!(xc[0] >> 31) || !(xc[1] >> 31)) { bad = 1; continue; }
                    } else {
                        long top = -1;
                        for (long b = alen - 1; b >= 32; --b)
                            if ((pay[b >> 3] >> (b & 7)) & 1u) { top = b; break; }
                        if (top < 32 || (E > 0 && top != alen - 1)) { bad = 1; continue; }      /* a spill ends with the marker */
                        pb = top;
                    }
                    int stop = 0;
                    for (long t = Tm - 1; t >= NS && !stop; --t) {
                        const long q = cnt - 1 - t;
                        const long n = (long)L * (m + (q / L) * M) + (q % L);
                        int i = (int)(n / wc), j = (int)(n % wc);
                        long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                        mix_t mx;
                        mix_prepare(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                    (float)planes[plane_sz + off] / 255.0f, &mx);
                        const int c = (int)((t - NS) % nch);
                        const uint32_t slot = xc[c] & 0xFFFF;
                        uint32_t c_low, c_high;
                        const int s = rans_find(&mx, slot, Lp, minv, maxv, &c_low, &c_high);
                        xc[c] = (c_high - c_low) * (xc[c] >> 16) + slot - c_low;
                        planes[clr * plane_sz + off] = (int16_t)(s - shift);
                        int nb = clz32(xc[c]);
                        if (nch == 1) {
                            if (nb > pb - pa) nb = (int)(pb - pa);       /* the chain's first pushes emitted nothing: what is left is all there is */
                            else if (nb > 16) { bad = 1; stop = 1; break; }
                        } else if (nb > 16 || pa + nb > pb) { bad = 1; stop = 1; break; }
                        if (c) { pb -= nb; xc[c] = (xc[c] << nb) | get_bits(pay, pb, nb); }
                        else { xc[c] = (nb < 32 ? (xc[c] << nb) : 0u) | get_bits(pay, pa, nb); pa += nb; }
                    }
                    if (stop) continue;
                    if (nch == 1) {
                        /* the chain is back at its start: every bit read, the state = the stream's last symbol (zero if it has none) */
                        if (pa != pb) bad = 1;
                        const uint32_t v = xc[0];
                        if (v >= (uint32_t)(Lp - 1) || (cnt < 1 && v)) { bad = 1; continue; }
                        if (cnt >= 1) {
                            const long q = cnt - 1;
                            const long n = (long)L * (m + (q / L) * M) + (q % L);
                            int i2 = (int)(n / wc), j2 = (int)(n % wc);
                            planes[clr * plane_sz + ((long)(2 * i2 + BAND_OI[src]) << lvl) * W + ((long)(2 * j2 + BAND_OJ[src]) << lvl)] = (int16_t)((int)v - shift);
                        }
                        continue;
                    }
                    for (long bq = pa; bq < pb; ++bq)
                        if ((pay[bq >> 3] >> (bq & 7)) & 1u) bad = 1;             /* nothing between the two chains */
                    for (int c = 0; c < nch; ++c) {                               /* the start states: the last 2 n symbols, raw */
                        uint32_t v = xc[c] & 0x7FFFFFFFu;
                        if (!(xc[c] >> 31) || v >= pw) bad = 1;
                        for (int i = 0; i < ns; ++i, v /= (uint32_t)(Lp - 1)) {
                            const int s = (int)(v % (uint32_t)(Lp - 1));
                            const long t = (long)c * ns + i;
                            if (t >= NS) { if (s) bad = 1; continue; }
                            const long q = cnt - 1 - t;
                            const long n = (long)L * (m + (q / L) * M) + (q % L);
                            int i2 = (int)(n / wc), j2 = (int)(n % wc);
                            planes[clr * plane_sz + ((long)(2 * i2 + BAND_OI[src]) << lvl) * W + ((long)(2 * j2 + BAND_OJ[src]) << lvl)] = (int16_t)(s - shift);
                        }
                    }
                    continue;
                }
                long top = -1;
                for (long b = PAY_BITS - 1; b >= 0; --b)
                    if ((pay[b >> 3] >> (b & 7)) & 1u) { top = b; break; }
                if (top < 31) { bad = 1; continue; }
                uint32_t xt = get_bits(pay, top - 31, 32);
                long tc = top - 31;
                uint32_t f_last = 0;
                for (long q = cnt - Tm; q < cnt; ++q) {
                    const long n = (long)L * (m + (q / L) * M) + (q % L);
                    int i = (int)(n / wc), j = (int)(n % wc);
                    long off = ((long)(2 * i + BAND_OI[src]) << lvl) * W + ((long)(2 * j + BAND_OJ[src]) << lvl);
                    mix_t mx;
                    mix_prepare(params + ((long)i * w + j) * ORC_NPAR, clr, (float)planes[off] / 255.0f,
                                (float)planes[plane_sz + off] / 255.0f, &mx);
                    const uint32_t slot = xt & 0xFFFF;
                    uint32_t c_low, c_high;
                    const int s = rans_find(&mx, slot, Lp, minv, maxv, &c_low, &c_high);
                    xt = (c_high - c_low) * (xt >> 16) + slot - c_low;
                    planes[clr * plane_sz + off] = (int16_t)(s - shift);
                    f_last = c_high - c_low;
                    if (q == cnt - 1) break;                     /* the encoder's first symbol: absorbing start, no bits */
                    int nb = clz32(xt);
                    if (nb > 16 || tc < nb) { bad = 1; break; }
                    tc -= nb;
                    xt = (xt << nb) | get_bits(pay, tc, nb);
                }
                if (xt != (Tm ? f_last << 15 : 1u << 31) || tc != 0) bad = 1;   /* the tail coder's start state, and no bit left */
            }
        }
        free(params);
    }
    orc_unlift(planes, H, W, rgb);
    free(planes); free(x); free(bitsp); free(cur); free(T);
    return bad ? -5 : 0;
}
