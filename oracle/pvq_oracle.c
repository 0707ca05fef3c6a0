/*
 * pvq_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see pvq_oracle.h).
 *
 * Plain-C, single-precision restatement of pitchvis_analysis's hot path.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference/pitchvis_analysis/src).
 * Arithmetic is f32 wherever the reference uses f32, in the reference's operation order; build
 * with -ffp-contract=off so no FMA contraction sneaks in (Rust never contracts).
 *
 * Third-party crates the reference calls on this path are not vendored under /root/reference;
 * their published semantics are restated here:
 *   rustfft 6.4.1 / realfft 3.5.0 (Cargo.lock:5841,5679) — unnormalised forward e^{-2*pi*i*jk/N},
 *       R2C returns N/2+1 bins (pinned by vqt.rs:1087-1128).  Restated as an iterative radix-2
 *       FFT with f64-computed twiddles rounded to f32 + the standard half-size-complex split.
 *   sprs 0.11.4 (Cargo.lock:6264) — TriMat::to_csr sorts columns; mul_acc_mat_vec_csr does
 *       y[r] += A[r,c]*x[c] sequentially in ascending column order.
 *   apodize 1.0.0 (Cargo.lock:279) — hanning_iter(n): f64 0.5 - 0.5*cos(2*pi*i/(n-1)).
 *   num-complex 0.4.6 (Cargo.lock:4532) — norm()=hypot, norm_sqr()=re^2+im^2,
 *       exp(a+ib)=from_polar(e^a, b), Complex*f32 and Complex/f32 are component-wise.
 *   find_peaks 0.1.5 (Cargo.lock:2949) — scipy-like PeakFinder (see orc_find_peaks).
 *
 * PARITY UNPINNED against the Rust binary's exact values (no golden vectors exist in the
 * reference; no Rust toolchain).  Pinned by the reference's own property tests, ported 1:1.
 */
#define _GNU_SOURCE
#include "pvq_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } c32;

/* ------------------------------------------------------------------------------------------
 * FFT (restates rustfft/realfft contract; vqt.rs:1087-1128 pins scaling, sign and layout)
 * ---------------------------------------------------------------------------------------- */

static unsigned ilog2u(uint32_t n) { unsigned l = 0; while ((1u << l) < n) l++; return l; }

/* in-place iterative radix-2 DIT, unnormalised; sign=-1 forward, +1 inverse */
static void fft_c32(c32 *a, uint32_t n, int inverse)
{
    if (n <= 1) return;
    unsigned lg = ilog2u(n);
    assert((1u << lg) == n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = 0;
        for (unsigned b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
        if (r > i) { c32 t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    c32 *tw = (c32 *)malloc(sizeof(c32) * (n / 2));
    for (uint32_t k = 0; k < n / 2; k++) {
        double ang = (inverse ? 2.0 : -2.0) * M_PI * (double)k / (double)n;
        tw[k].re = (float)cos(ang);
        tw[k].im = (float)sin(ang);
    }
    for (uint32_t len = 2; len <= n; len <<= 1) {
        uint32_t half = len >> 1, step = n / len;
        for (uint32_t i = 0; i < n; i += len) {
            for (uint32_t j = 0; j < half; j++) {
                c32 w = tw[j * step];
                c32 u = a[i + j], v = a[i + j + half];
                c32 t;
                t.re = v.re * w.re - v.im * w.im;
                t.im = v.re * w.im + v.im * w.re;
                a[i + j].re = u.re + t.re;
                a[i + j].im = u.im + t.im;
                a[i + j + half].re = u.re - t.re;
                a[i + j + half].im = u.im - t.im;
            }
        }
    }
    free(tw);
}

/* R2C of even n via an n/2-point complex FFT (what realfft does); out has n/2+1 bins */
static void rfft_f32(const float *x, uint32_t n, c32 *out)
{
    uint32_t h = n / 2;
    c32 *z = (c32 *)malloc(sizeof(c32) * h);
    for (uint32_t i = 0; i < h; i++) { z[i].re = x[2 * i]; z[i].im = x[2 * i + 1]; }
    fft_c32(z, h, 0);
    for (uint32_t k = 0; k <= h; k++) {
        c32 a = z[k % h];
        c32 b = z[(h - k) % h];
        b.im = -b.im; /* conj(Z[h-k]) */
        double ang = -2.0 * M_PI * (double)k / (double)n;
        float wr = (float)cos(ang), wi = (float)sin(ang);
        float er = 0.5f * (a.re + b.re), ei = 0.5f * (a.im + b.im);   /* even part */
        float dr = 0.5f * (a.re - b.re), di = 0.5f * (a.im - b.im);   /* (Z - conj Z')/2 */
        /* odd part = -i * w * d */
        float tr = wr * dr - wi * di, ti = wr * di + wi * dr;
        out[k].re = er + ti;
        out[k].im = ei - tr;
    }
    free(z);
}


/* planned variants of the same arithmetic (same butterfly order and twiddle values as fft_c32 /
 * rfft_f32 above, tables cached per Vqt instance so the cpu_baseline timing is not dominated by
 * malloc/cos) */
typedef struct {
    uint32_t n;      /* real length */
    uint32_t *rev;   /* bit reversal for n/2 */
    c32 *tw;         /* n/4 twiddles of the n/2-point complex FFT */
    c32 *split;      /* n/2+1 split twiddles e^{-2 pi i k/n} */
    c32 *z;          /* n/2 work */
} rfft_plan;

static void rfft_plan_init(rfft_plan *p, uint32_t n)
{
    uint32_t h = n / 2;
    unsigned lg = ilog2u(h);
    p->n = n;
    p->rev = (uint32_t *)malloc(sizeof(uint32_t) * h);
    for (uint32_t i = 0; i < h; i++) {
        uint32_t r = 0;
        for (unsigned b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
        p->rev[i] = r;
    }
    p->tw = (c32 *)malloc(sizeof(c32) * (h / 2 ? h / 2 : 1));
    for (uint32_t k = 0; k < h / 2; k++) {
        double ang = -2.0 * M_PI * (double)k / (double)h;
        p->tw[k].re = (float)cos(ang); p->tw[k].im = (float)sin(ang);
    }
    p->split = (c32 *)malloc(sizeof(c32) * (h + 1));
    for (uint32_t k = 0; k <= h; k++) {
        double ang = -2.0 * M_PI * (double)k / (double)n;
        p->split[k].re = (float)cos(ang); p->split[k].im = (float)sin(ang);
    }
    p->z = (c32 *)malloc(sizeof(c32) * h);
}

static void rfft_plan_free(rfft_plan *p) { free(p->rev); free(p->tw); free(p->split); free(p->z); }

static void rfft_planned(const rfft_plan *p, const float *x, c32 *out)
{
    uint32_t n = p->n, h = n / 2;
    c32 *a = p->z;
    for (uint32_t i = 0; i < h; i++) { uint32_t r = p->rev[i]; a[r].re = x[2 * i]; a[r].im = x[2 * i + 1]; }
    /* the radix-2 stages of fft_c32, two at a time: the four values of a pair of butterflies stay in registers between the two
     * stages (same operations on the same operands in the same order: bit-identical to one stage per pass, half the passes over
     * the array — the plain one-stage-per-pass loop made this port ~2x slower than the reference's rustfft, VQT_REVIEW.md:363) */
#define ORC_BFLY(u, v, w) do { c32 t_; t_.re = (v).re * (w).re - (v).im * (w).im; t_.im = (v).re * (w).im + (v).im * (w).re; \
        c32 u_ = (u); (u).re = u_.re + t_.re; (u).im = u_.im + t_.im; (v).re = u_.re - t_.re; (v).im = u_.im - t_.im; } while (0)
    uint32_t len = 2;
    for (; len * 2 <= h; len <<= 2) {
        uint32_t half = len >> 1, step1 = h / len, len2 = len * 2, step2 = h / len2;
        for (uint32_t i = 0; i < h; i += len2) {
            for (uint32_t j = 0; j < half; j++) {
                c32 w1 = p->tw[j * step1], w2a = p->tw[j * step2], w2b = p->tw[(j + half) * step2];
                c32 a0 = a[i + j], a1 = a[i + j + half], a2 = a[i + j + len], a3 = a[i + j + len + half];
                ORC_BFLY(a0, a1, w1);
                ORC_BFLY(a2, a3, w1);
                ORC_BFLY(a0, a2, w2a);
                ORC_BFLY(a1, a3, w2b);
                a[i + j] = a0; a[i + j + half] = a1; a[i + j + len] = a2; a[i + j + len + half] = a3;
            }
        }
    }
    for (; len <= h; len <<= 1) {   /* an odd number of stages: the last one alone */
        uint32_t half = len >> 1, step = h / len;
        for (uint32_t i = 0; i < h; i += len) {
            for (uint32_t j = 0; j < half; j++) {
                c32 w = p->tw[j * step];
                ORC_BFLY(a[i + j], a[i + j + half], w);
            }
        }
    }
#undef ORC_BFLY
    for (uint32_t k = 0; k <= h; k++) {
        c32 za = a[k % h];
        c32 zb = a[(h - k) % h];
        zb.im = -zb.im;
        float wr = p->split[k].re, wi = p->split[k].im;
        float er = 0.5f * (za.re + zb.re), ei = 0.5f * (za.im + zb.im);
        float dr = 0.5f * (za.re - zb.re), di = 0.5f * (za.im - zb.im);
        float tr = wr * dr - wi * di, ti = wr * di + wi * dr;
        out[k].re = er + ti;
        out[k].im = ei - tr;
    }
}

void orc_fft_complex(float *a, uint32_t n, int inverse) { fft_c32((c32 *)a, n, inverse); }
void orc_fft_real(const float *x, uint32_t n, float *out) { rfft_f32(x, n, (c32 *)out); }

/* ------------------------------------------------------------------------------------------
 * kernel data structures
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    uint32_t rows, cols, nnz;
    uint32_t *row_ptr; /* rows+1 */
    uint32_t *col_idx;
    c32 *val;
} csr;

typedef struct {
    uint32_t w0, w1;  /* vqt.rs:391 window */
    csr mat;          /* filter_bank */
    csr neg;          /* negative_filter_bank; nnz==0 <=> None */
} wgroup;

typedef struct {
    float freq, window_length;
    uint32_t M, minwin;
} fparams; /* vqt.rs:370-384 */

struct orc_vqt {
    orc_params p;
    uint32_t n_bins, n_groups;
    fparams *fp;
    wgroup *groups;
    float window_center;
    double delay_s;
    /* scratch, vqt.rs:425-431 */
    float *input;
    c32 *spectrum;
    c32 *x_vqt;
    c32 *neg_part;
    rfft_plan *plans; /* one per group */
};

void orc_default_params(orc_params *p)
{ /* vqt.rs:180-214, 333-348 */
    p->sr = 22050.0f;
    p->n_fft = 2 * 16384;
    p->min_freq = 55.0f;
    p->octaves = 7;
    p->buckets_per_octave = 12 * 7;
    p->sparsity_quantile = 0.999f;
    p->quality = 1.6f / 1.0f;
    p->gamma = 4.8f * (1.6f / 1.0f);
}

static uint32_t f32_to_u32_sat(float x)
{ /* Rust `as u32`/`as usize`: truncates toward zero, saturates, NaN -> 0 */
    if (!(x > 0.0f)) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

/* vqt.rs:517-587 */
static int filter_bank_params(const orc_params *p, fparams *out, float err[2])
{
    uint32_t n_bins = p->buckets_per_octave * p->octaves;
    float bpo = (float)p->buckets_per_octave;
    float highest = p->min_freq * powf(2.0f, (float)(n_bins - 1) / bpo);
    float nyq = p->sr / 2.0f;
    if (highest > nyq) { err[0] = highest; err[1] = nyq; return ORC_ABOVE_NYQUIST; }

    float r = powf(2.0f, 1.0f / bpo);
    float alpha = (r * r - 1.0f) / (r * r + 1.0f);
    for (uint32_t k = 0; k < n_bins; k++) {
        float freq = p->min_freq * powf(2.0f, (float)k / bpo);
        float wl = p->quality * p->sr / (alpha * freq + p->gamma);
        float min_scaled_sr = ceilf(freq * 2.0f * 1.15f);
        uint32_t k1 = f32_to_u32_sat(floorf(log2f(p->sr / min_scaled_sr)));
        uint32_t k2 = f32_to_u32_sat(floorf(log2f((float)p->n_fft / wl)));
        out[k].freq = freq;
        out[k].window_length = wl;
        out[k].M = 1u << k1;
        out[k].minwin = p->n_fft >> k2;
    }
    if (out[0].window_length > (float)p->n_fft) {
        err[0] = out[0].window_length; err[1] = (float)p->n_fft;
        return ORC_WINDOW_EXCEEDS_NFFT;
    }
    return ORC_OK;
}

static int cmp_f32(const void *a, const void *b)
{ float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }

/* vqt.rs:769-852.  v must hold S = (w1-w0)/M complex values. */
static void calculate_filter(float sr, float q, uint32_t M, fparams fp, uint32_t w0, uint32_t w1,
                             float window_center, c32 *v, uint32_t S)
{
    float scaled_freq = fp.freq * (float)M;
    float scaled_wl = fp.window_length / (float)M;
    uint32_t L = f32_to_u32_sat(roundf(scaled_wl));
    float scaled_center = (window_center - (float)w0) / (float)M;
    uint32_t c = f32_to_u32_sat(floorf(scaled_center));
    assert(S == (w1 - w0) / M);
    assert(L <= S);
    assert(c >= L / 2);
    uint32_t begin = c - L / 2;
    assert(begin + L <= S);

    for (uint32_t i = 0; i < S; i++) { v[i].re = 0.0f; v[i].im = 0.0f; }
    const float two_pi = 2.0f * 3.14159274101257324f; /* Complex::i()*2.0*PI, f32 */
    for (uint32_t i = 0; i < L; i++) {
        /* apodize::hanning_iter: f64 */
        double x = (M_PI * (double)i) / (double)(L - 1);
        double w = 0.5 - 0.5 * cos(2.0 * x);
        /* (i * 2.0 * PI * (i as f32) * scaled_freq / sr), left to right in f32 */
        float ang = two_pi * (float)i;
        ang = ang * scaled_freq;
        ang = ang / sr;
        float wf = (float)w;
        v[begin + i].re = wf * cosf(ang);
        v[begin + i].im = wf * sinf(ang);
    }
    /* L1 normalise, vqt.rs:802-805 */
    float norm_1 = 0.0f;
    for (uint32_t i = 0; i < S; i++) norm_1 += hypotf(v[i].re, v[i].im);
    for (uint32_t i = 0; i < S; i++) { v[i].re /= norm_1; v[i].im /= norm_1; }
    /* FFT + conj, vqt.rs:808-811 */
    fft_c32(v, S, 0);
    for (uint32_t i = 0; i < S; i++) v[i].im = -v[i].im;
    /* sparsify, vqt.rs:813-842 */
    float *mag = (float *)malloc(sizeof(float) * S);
    for (uint32_t i = 0; i < S; i++) mag[i] = hypotf(v[i].re, v[i].im);
    qsort(mag, S, sizeof(float), cmp_f32);
    float sum = 0.0f;
    for (uint32_t i = 0; i < S; i++) sum += mag[i];
    float accum = 0.0f;
    uint32_t idx = 0;
    float target = (1.0f - q) * sum;
    while (accum < target && idx < S) { accum += mag[idx]; idx++; }
    float cutoff = idx == 0 ? 0.0f : mag[idx - 1];
    for (uint32_t i = 0; i < S; i++)
        if (hypotf(v[i].re, v[i].im) < cutoff) { v[i].re = 0.0f; v[i].im = 0.0f; }
    free(mag);
}

typedef struct { uint32_t row, col; c32 v; } triplet;

static int cmp_triplet(const void *a, const void *b)
{
    const triplet *x = (const triplet *)a, *y = (const triplet *)b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return 0;
}

static void triplets_to_csr(triplet *t, uint32_t n, uint32_t rows, uint32_t cols, csr *m)
{ /* sprs TriMat::to_csr: sorted by (row, col); duplicates do not occur here */
    qsort(t, n, sizeof(triplet), cmp_triplet);
    m->rows = rows; m->cols = cols; m->nnz = n;
    m->row_ptr = (uint32_t *)calloc(rows + 1, sizeof(uint32_t));
    m->col_idx = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    m->val = (c32 *)malloc(sizeof(c32) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) {
        m->row_ptr[t[i].row + 1]++;
        m->col_idx[i] = t[i].col;
        m->val[i] = t[i].v;
    }
    for (uint32_t r = 0; r < rows; r++) m->row_ptr[r + 1] += m->row_ptr[r];
}

/* vqt.rs:599-759 */
int orc_vqt_new(const orc_params *p, orc_vqt **out, float err[2])
{
    uint32_t n_bins = p->buckets_per_octave * p->octaves;
    fparams *fp = (fparams *)malloc(sizeof(fparams) * n_bins);
    int rc = filter_bank_params(p, fp, err);
    if (rc != ORC_OK) { free(fp); *out = NULL; return rc; }

    orc_vqt *v = (orc_vqt *)calloc(1, sizeof(orc_vqt));
    v->p = *p; v->n_bins = n_bins; v->fp = fp;
    float n_fft_f = (float)p->n_fft;
    float window_center = n_fft_f - fp[0].window_length / 2.0f; /* vqt.rs:604-605 */
    v->window_center = window_center;

    /* rate groups, vqt.rs:616-642 */
    typedef struct { uint32_t M, w0, w1, first, count; } rate_group;
    rate_group *rg = (rate_group *)malloc(sizeof(rate_group) * n_bins);
    uint32_t n_rg = 0;
    for (uint32_t k = 0; k < n_bins;) {
        uint32_t e = k + 1;
        while (e < n_bins && fp[e].M == fp[e - 1].M) e++;
        uint32_t ws = 0;
        for (uint32_t i = k; i < e; i++) if (fp[i].minwin > ws) ws = fp[i].minwin;
        uint32_t w0, w1;
        if ((window_center + (float)ws / 2.0f) < n_fft_f) {
            w0 = f32_to_u32_sat(window_center - (float)ws / 2.0f);
            w1 = f32_to_u32_sat(window_center + (float)ws / 2.0f);
        } else {
            w0 = p->n_fft - ws; w1 = p->n_fft;
        }
        rg[n_rg].M = fp[k].M; rg[n_rg].w0 = w0; rg[n_rg].w1 = w1;
        rg[n_rg].first = k; rg[n_rg].count = e - k;
        n_rg++;
        k = e;
    }

    float kernel_gain = sqrtf(p->sr); /* vqt.rs:646 */

    /* window groups, vqt.rs:653-754 */
    v->groups = (wgroup *)calloc(n_rg, sizeof(wgroup));
    uint32_t n_g = 0;
    uint32_t max_ws = 0, max_rows = 0;
    for (uint32_t a = 0; a < n_rg;) {
        uint32_t b = a + 1;
        while (b < n_rg && rg[b].w0 == rg[b - 1].w0 && rg[b].w1 == rg[b - 1].w1) b++;
        uint32_t w0 = rg[a].w0, w1 = rg[a].w1, ws = w1 - w0;
        uint32_t n_spec = ws / 2 + 1;
        uint32_t n_filters = 0;
        for (uint32_t i = a; i < b; i++) n_filters += rg[i].count;
        size_t cap = (size_t)n_filters * ws; /* generous upper bound */
        triplet *tm = (triplet *)malloc(sizeof(triplet) * (cap < 16 ? 16 : cap / 4 + 16));
        triplet *tn = (triplet *)malloc(sizeof(triplet) * (cap < 16 ? 16 : cap / 4 + 16));
        size_t capm = cap < 16 ? 16 : cap / 4 + 16, capn = capm;
        uint32_t nm = 0, nn = 0, row = 0;
        c32 *buf = (c32 *)malloc(sizeof(c32) * ws);
        for (uint32_t i = a; i < b; i++) {
            uint32_t M = rg[i].M;
            uint32_t S = ws / M;
            for (uint32_t f = 0; f < rg[i].count; f++) {
                calculate_filter(p->sr, p->sparsity_quantile, M, fp[rg[i].first + f], w0, w1,
                                 window_center, buf, S);
                /* remap, vqt.rs:725-735 */
                for (uint32_t j = 0; j < S; j++) {
                    c32 z = buf[j];
                    if (z.re == 0.0f && z.im == 0.0f) continue;
                    c32 val;
                    val.re = z.re * kernel_gain; val.im = z.im * kernel_gain;
                    val.re = val.re / (float)ws; val.im = val.im / (float)ws;
                    if (j <= S / 2) {
                        if (nm == capm) { capm *= 2; tm = (triplet *)realloc(tm, sizeof(triplet) * capm); }
                        tm[nm].row = row; tm[nm].col = j; tm[nm].v = val; nm++;
                    } else {
                        if (nn == capn) { capn *= 2; tn = (triplet *)realloc(tn, sizeof(triplet) * capn); }
                        tn[nn].row = row; tn[nn].col = S - j;
                        tn[nn].v.re = val.re; tn[nn].v.im = -val.im; nn++;
                    }
                }
                row++;
            }
        }
        free(buf);
        wgroup *g = &v->groups[n_g++];
        g->w0 = w0; g->w1 = w1;
        triplets_to_csr(tm, nm, n_filters, n_spec, &g->mat);
        triplets_to_csr(tn, nn, n_filters, n_spec, &g->neg);
        free(tm); free(tn);
        if (ws > max_ws) max_ws = ws;
        if (n_filters > max_rows) max_rows = n_filters;
        a = b;
    }
    free(rg);
    v->n_groups = n_g;
    /* Duration::from_secs_f32, vqt.rs:756 */
    v->delay_s = (double)((n_fft_f - window_center) / p->sr);

    v->input = (float *)malloc(sizeof(float) * max_ws);
    v->spectrum = (c32 *)malloc(sizeof(c32) * (max_ws / 2 + 1));
    v->x_vqt = (c32 *)malloc(sizeof(c32) * n_bins);
    v->neg_part = (c32 *)malloc(sizeof(c32) * (max_rows ? max_rows : 1));
    v->plans = (rfft_plan *)malloc(sizeof(rfft_plan) * n_g);
    for (uint32_t g = 0; g < n_g; g++) rfft_plan_init(&v->plans[g], v->groups[g].w1 - v->groups[g].w0);
    *out = v;
    return ORC_OK;
}

static void csr_free(csr *m) { free(m->row_ptr); free(m->col_idx); free(m->val); }

void orc_vqt_free(orc_vqt *v)
{
    if (!v) return;
    for (uint32_t g = 0; g < v->n_groups; g++) { csr_free(&v->groups[g].mat); csr_free(&v->groups[g].neg); }
    for (uint32_t g = 0; g < v->n_groups; g++) rfft_plan_free(&v->plans[g]);
    free(v->plans);
    free(v->groups); free(v->fp); free(v->input); free(v->spectrum); free(v->x_vqt); free(v->neg_part);
    free(v);
}

uint32_t orc_n_bins(const orc_vqt *v) { return v->n_bins; }
uint32_t orc_n_groups(const orc_vqt *v) { return v->n_groups; }
double orc_delay_seconds(const orc_vqt *v) { return v->delay_s; }
float orc_window_center(const orc_vqt *v) { return v->window_center; }

void orc_filter_params(const orc_vqt *v, float *freq, float *wl, uint32_t *M, uint32_t *minwin)
{
    for (uint32_t k = 0; k < v->n_bins; k++) {
        freq[k] = v->fp[k].freq; wl[k] = v->fp[k].window_length;
        M[k] = v->fp[k].M; minwin[k] = v->fp[k].minwin;
    }
}

void orc_group_info(const orc_vqt *v, uint32_t g, uint32_t info[5])
{
    const wgroup *w = &v->groups[g];
    info[0] = w->w0; info[1] = w->w1; info[2] = w->mat.rows; info[3] = w->mat.nnz; info[4] = w->neg.nnz;
}

void orc_group_csr(const orc_vqt *v, uint32_t g, int neg, uint32_t *row_ptr, uint32_t *col_idx, float *values)
{
    const csr *m = neg ? &v->groups[g].neg : &v->groups[g].mat;
    memcpy(row_ptr, m->row_ptr, sizeof(uint32_t) * (m->rows + 1));
    memcpy(col_idx, m->col_idx, sizeof(uint32_t) * m->nnz);
    memcpy(values, m->val, sizeof(c32) * m->nnz);
}

/* sprs::prod::mul_acc_mat_vec_csr: y[r] += sum_c A[r,c]*x[c], ascending column order */
static void mul_acc_mat_vec_csr(const csr *m, const c32 *x, c32 *y)
{
    for (uint32_t r = 0; r < m->rows; r++) {
        c32 acc = y[r];
        for (uint32_t i = m->row_ptr[r]; i < m->row_ptr[r + 1]; i++) {
            c32 a = m->val[i], b = x[m->col_idx[i]];
            float pr = a.re * b.re - a.im * b.im;
            float pi = a.re * b.im + a.im * b.re;
            acc.re += pr; acc.im += pi;
        }
        y[r] = acc;
    }
}

/* vqt.rs:873-913 */
static void frame_complex(orc_vqt *v, const float *x)
{
    for (uint32_t k = 0; k < v->n_bins; k++) { v->x_vqt[k].re = 0.0f; v->x_vqt[k].im = 0.0f; }
    uint32_t offset = 0;
    for (uint32_t g = 0; g < v->n_groups; g++) {
        wgroup *w = &v->groups[g];
        uint32_t ws = w->w1 - w->w0;
        memcpy(v->input, x + w->w0, sizeof(float) * ws); /* vqt.rs:881-882 */
        rfft_planned(&v->plans[g], v->input, v->spectrum);
        uint32_t n_filters = w->mat.rows;
        mul_acc_mat_vec_csr(&w->mat, v->spectrum, v->x_vqt + offset);
        if (w->neg.nnz > 0) {
            for (uint32_t r = 0; r < n_filters; r++) { v->neg_part[r].re = 0.0f; v->neg_part[r].im = 0.0f; }
            mul_acc_mat_vec_csr(&w->neg, v->spectrum, v->neg_part);
            for (uint32_t r = 0; r < n_filters; r++) {
                v->x_vqt[offset + r].re += v->neg_part[r].re;
                v->x_vqt[offset + r].im += -v->neg_part[r].im;
            }
        }
        offset += n_filters;
    }
}

/* vqt.rs:922-954 */
void orc_power_to_db(const float *xc, uint32_t n, float *out)
{
    const float REF_POWER = 0.3f * 0.3f;
    const float A_MIN = 1e-6f * 1e-6f;
    const float TOP_DB = 60.0f;
    float ref_db = 10.0f * log10f(REF_POWER);
    float mx = -3.40282347e+38f, mn = 3.40282347e+38f; /* f32::MIN, f32::MAX */
    for (uint32_t i = 0; i < n; i++) {
        float re = xc[2 * i], im = xc[2 * i + 1];
        float ns = re * re + im * im;
        float d = 10.0f * log10f(fmaxf(ns, A_MIN)) - ref_db;
        out[i] = d;
        mx = fmaxf(mx, d);
        mn = fminf(mn, d);
    }
    float floor_db = mx - TOP_DB;
    mn = fmaxf(mn, floor_db);
    for (uint32_t i = 0; i < n; i++) {
        float clamped = fmaxf(out[i], floor_db);
        out[i] = (mn > 0.0f) ? (clamped - mn) : fmaxf(clamped, 0.0f);
    }
}

void orc_calculate_vqt_instant_complex(orc_vqt *v, const float *x, float *out)
{
    frame_complex(v, x);
    memcpy(out, v->x_vqt, sizeof(c32) * v->n_bins);
}

void orc_calculate_vqt_instant_in_db(orc_vqt *v, const float *x, float *out_db)
{
    frame_complex(v, x);
    orc_power_to_db((const float *)v->x_vqt, v->n_bins, out_db);
}

void orc_group_spectrum(orc_vqt *v, uint32_t g, const float *x, float *out)
{
    wgroup *w = &v->groups[g];
    rfft_f32(x + w->w0, w->w1 - w->w0, (c32 *)out);
}

void orc_calculate_batch(orc_vqt *v, const float *pcm, size_t n_lead, size_t hop, size_t n_frames,
                         float *out_db, float *out_cplx)
{
    uint32_t n_fft = v->p.n_fft;
    float *ring = (float *)malloc(sizeof(float) * n_fft);
    for (size_t f = 0; f < n_frames; f++) {
        /* last n_fft samples ending at n_lead + (f+1)*hop; zeros before the buffer start */
        long long end = (long long)(n_lead + (f + 1) * hop);
        long long beg = end - (long long)n_fft;
        for (uint32_t j = 0; j < n_fft; j++) {
            long long s = beg + j;
            ring[j] = s >= 0 ? pcm[s] : 0.0f;
        }
        frame_complex(v, ring);
        if (out_cplx) memcpy(out_cplx + 2 * f * v->n_bins, v->x_vqt, sizeof(c32) * v->n_bins);
        orc_power_to_db((const float *)v->x_vqt, v->n_bins, out_db + f * v->n_bins);
    }
    free(ring);
}

/* util.rs:62-79 */
void orc_test_create_sines(const orc_params *p, const float *freqs, uint32_t n_freqs, float t_diff,
                           float *wave)
{
    const float PI_F = 3.14159274101257324f;
    for (uint32_t i = 0; i < p->n_fft; i++) wave[i] = 0.0f;
    for (uint32_t k = 0; k < n_freqs; k++) {
        float f = freqs[k];
        for (uint32_t i = 0; i < p->n_fft; i++) {
            float t = ((float)i + t_diff * p->sr) * 2.0f * PI_F / p->sr;
            float amp = sinf(t * f) / 12.0f;
            wave[i] += amp;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * peaks
 * ---------------------------------------------------------------------------------------- */

/*
 * find_peaks 0.1.5 PeakFinder, restated from its documented (scipy.signal.find_peaks-like)
 * behaviour: plateau-aware strict local maxima (a rise before, a fall after; the first and last
 * sample are never peaks); Peak.position is the half-open plateau range and
 * middle_position() = (start+end)/2; height and prominence lower bounds are inclusive;
 * prominence = height - max(min over the left walk, min over the right walk), each walk stopping
 * at the first strictly higher sample or the signal edge; min_distance keeps the higher of two
 * peaks closer than `distance` (applied after height, before prominence as in scipy; with
 * buckets_per_octave = 36 the distance is 1 and the rule is a no-op).
 * Wrapper: peak_detection.rs:26-51.
 */
uint32_t orc_find_peaks(const float *x, uint32_t n, uint32_t bpo, float min_prom, float min_height,
                        uint32_t *out_idx)
{
    if (n < 3) return 0;
    uint32_t *mid = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint32_t np = 0;
    uint32_t i = 1, i_max = n - 1;
    while (i < i_max) {
        if (x[i - 1] < x[i]) {
            uint32_t ia = i + 1;
            while (ia < i_max && x[ia] == x[i]) ia++;
            if (x[ia] < x[i]) {
                mid[np++] = (i + ia) / 2; /* range i..ia, middle_position */
                i = ia;
            }
        }
        i++;
    }
    /* height */
    uint32_t k = 0;
    for (uint32_t j = 0; j < np; j++) if (x[mid[j]] >= min_height) mid[k++] = mid[j];
    np = k;
    /* distance: peak_detection.rs:37-40 */
    uint32_t dist = f32_to_u32_sat(roundf((float)bpo * 0.4f / 12.0f));
    if (dist > 0 && np > 1) {
        /* scipy _select_by_peak_distance: visit by descending height, drop neighbours closer than dist */
        uint8_t *keep = (uint8_t *)malloc(np);
        uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * np);
        memset(keep, 1, np);
        for (uint32_t j = 0; j < np; j++) order[j] = j;
        for (uint32_t a = 1; a < np; a++) { /* stable insertion sort ascending by height */
            uint32_t t = order[a]; uint32_t b = a;
            while (b > 0 && x[mid[order[b - 1]]] > x[mid[t]]) { order[b] = order[b - 1]; b--; }
            order[b] = t;
        }
        for (uint32_t a = np; a-- > 0;) {
            uint32_t j = order[a];
            if (!keep[j]) continue;
            for (uint32_t b = j; b-- > 0 && mid[j] - mid[b] < dist;) keep[b] = 0;
            for (uint32_t b = j + 1; b < np && mid[b] - mid[j] < dist; b++) keep[b] = 0;
        }
        k = 0;
        for (uint32_t j = 0; j < np; j++) if (keep[j]) mid[k++] = mid[j];
        np = k;
        free(keep); free(order);
    }
    /* prominence */
    k = 0;
    for (uint32_t j = 0; j < np; j++) {
        uint32_t pk = mid[j];
        float h = x[pk];
        float lmin = h, rmin = h;
        for (uint32_t a = pk; a-- > 0;) { if (x[a] > h) break; if (x[a] < lmin) lmin = x[a]; }
        for (uint32_t a = pk + 1; a < n; a++) { if (x[a] > h) break; if (x[a] < rmin) rmin = x[a]; }
        float prom = h - fmaxf(lmin, rmin);
        if (prom >= min_prom) mid[k++] = pk;
    }
    np = k;
    /* min_bin filter, peak_detection.rs:45-50 */
    uint32_t min_bin = ((bpo / 12) + 1) / 2;
    k = 0;
    for (uint32_t j = 0; j < np; j++) if (mid[j] >= min_bin) out_idx[k++] = mid[j];
    free(mid);
    return k;
}

void orc_default_analysis_params(orc_analysis_params *a)
{ /* analysis.rs:72-98 */
    a->peak_min_prominence = 10.0f; a->peak_min_height = 4.0f;
    a->bass_min_prominence = 5.0f; a->bass_min_height = 3.5f;
    a->highest_bassnote = 12 * 2 + 4;
    a->harmonic_threshold = 0.3f;
}

/* analysis.rs:332-349 */
uint32_t orc_find_peaks_split(const float *vqt, uint32_t n, uint32_t bpo, const orc_analysis_params *a,
                              uint32_t *out_idx)
{
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t k = 0;
    uint32_t nb = orc_find_peaks(vqt, n, bpo, a->bass_min_prominence, a->bass_min_height, tmp);
    for (uint32_t j = 0; j < nb; j++) if (tmp[j] <= a->highest_bassnote) out_idx[k++] = tmp[j];
    uint32_t ng = orc_find_peaks(vqt, n, bpo, a->peak_min_prominence, a->peak_min_height, tmp);
    for (uint32_t j = 0; j < ng; j++) if (tmp[j] > a->highest_bassnote) out_idx[k++] = tmp[j];
    free(tmp);
    return k; /* ascending: bass part <= 28 < general part */
}

static float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* peak_detection.rs:61-148 */
uint32_t orc_enhance_peaks_continuous(const uint32_t *peaks, uint32_t n_peaks, const float *vqt,
                                      float min_freq, uint32_t octaves, uint32_t bpo_u,
                                      float *out_center, float *out_size)
{
    uint32_t n_buckets = octaves * bpo_u;
    float bpo = (float)bpo_u;
    for (uint32_t j = 0; j < n_peaks; j++) {
        uint32_t p = peaks[j];
        if (p < 1 || p > n_buckets - 2) { out_center[j] = (float)p; out_size[j] = vqt[p]; continue; }
        float f_prev = min_freq * powf(2.0f, (float)(p - 1) / bpo);
        float f_curr = min_freq * powf(2.0f, (float)p / bpo);
        float f_next = min_freq * powf(2.0f, (float)(p + 1) / bpo);
        float lf0 = logf(f_prev), lf1 = logf(f_curr), lf2 = logf(f_next);
        float a0 = vqt[p - 1], a1 = vqt[p], a2 = vqt[p + 1];
        float denom = (lf0 - lf1) * (lf0 - lf2) * (lf1 - lf2);
        if (fabsf(denom) < 1.1920929e-07f) { out_center[j] = (float)p; out_size[j] = vqt[p]; continue; }
        float a = (lf2 * (a1 - a0) + lf0 * (a2 - a1) + lf1 * (a0 - a2)) / denom;
        float b = ((lf2 * lf2) * (a0 - a1) + (lf0 * lf0) * (a1 - a2) + (lf1 * lf1) * (a2 - a0)) / denom;
        float lfp = (fabsf(a) < 1.1920929e-07f) ? lf1 : clampf(-b / (2.0f * a), lf0, lf2);
        float f_peak = expf(lfp);
        float center = bpo * log2f(f_peak / min_freq);
        float cc = clampf(center, 0.0f, (float)n_buckets - 1.0f);
        uint32_t lower = f32_to_u32_sat(floorf(cc));
        uint32_t upper = lower + 1 < n_buckets - 1 ? lower + 1 : n_buckets - 1;
        float fract = cc - truncf(cc);
        float size = vqt[lower] * (1.0f - fract) + vqt[upper] * fract;
        out_center[j] = cc;
        out_size[j] = fmaxf(size, 0.0f);
    }
    /* sort_by center (stable insertion sort), peak_detection.rs:145 */
    for (uint32_t a = 1; a < n_peaks; a++) {
        float c = out_center[a], s = out_size[a]; uint32_t b = a;
        while (b > 0 && out_center[b - 1] > c) { out_center[b] = out_center[b - 1]; out_size[b] = out_size[b - 1]; b--; }
        out_center[b] = c; out_size[b] = s;
    }
    return n_peaks;
}

/* peak_detection.rs:172-241 */
void orc_promote_bass_peaks_with_harmonics(const float *center, float *size, uint32_t n_peaks,
                                           const float *vqt, float min_freq, uint32_t octaves,
                                           uint32_t bpo_u, uint32_t highest_bassnote, float thr)
{
    uint32_t n_buckets = octaves * bpo_u;
    float bpo = (float)bpo_u;
    static const float weights[4] = {0.5f, 0.3f, 0.15f, 0.05f};
    for (uint32_t j = 0; j < n_peaks; j++) {
        if (center[j] > (float)highest_bassnote) continue;
        float f0 = min_freq * powf(2.0f, center[j] / bpo);
        float p0 = powf(10.0f, size[j] / 10.0f);
        float score = 0.0f;
        for (uint32_t h = 2; h <= 5; h++) {
            float hf = f0 * (float)h;
            if (!(hf >= min_freq)) continue;
            float hb = (log2f(hf) - log2f(min_freq)) * bpo;
            if (hb >= 0.0f && hb < (float)n_buckets) {
                uint32_t lo = f32_to_u32_sat(floorf(hb));
                uint32_t hi = f32_to_u32_sat(ceilf(hb));
                if (hi > n_buckets - 1) hi = n_buckets - 1;
                float frac = hb - truncf(hb);
                float adb = (lo == hi) ? vqt[lo] : (vqt[lo] * (1.0f - frac) + vqt[hi] * frac);
                float hp = powf(10.0f, adb / 10.0f);
                float tp = p0 * thr;
                if (hp > tp) score += hp * weights[h - 2];
            }
        }
        if (score > 0.0f) {
            float boost = 1.0f + 0.5f * (score / fmaxf(p0, 1e-6f));
            float capped = fminf(boost, 1.5f);
            size[j] += 10.0f * log10f(capped);
        }
    }
}

uint32_t orc_analyze_frame(const float *vqt, uint32_t n, float min_freq, uint32_t octaves, uint32_t bpo,
                           const orc_analysis_params *a, uint32_t *out_idx, float *out_center,
                           float *out_size)
{
    uint32_t np = orc_find_peaks_split(vqt, n, bpo, a, out_idx);
    orc_enhance_peaks_continuous(out_idx, np, vqt, min_freq, octaves, bpo, out_center, out_size);
    orc_promote_bass_peaks_with_harmonics(out_center, out_size, np, vqt, min_freq, octaves, bpo,
                                          a->highest_bassnote, a->harmonic_threshold);
    return np;
}

/* libm pass-throughs so that the NumPy-side restatement of AnalysisState (oracle/analysis_state.py)
 * evaluates exp / pow with the same glibc routines the reference's f32::exp / f32::powf resolve to */
float orc_expf(float x) { return expf(x); }
float orc_powf(float x, float y) { return powf(x, y); }
/* the same libm calls over arrays (oracle/analysis_state.py's vectorised form: one call per frame instead of one per bin) */
void orc_expf_v(const float *x, uint32_t n, float *out) { for (uint32_t i = 0; i < n; ++i) out[i] = expf(x[i]); }
void orc_powf_v(float base, const float *y, uint32_t n, float *out) { for (uint32_t i = 0; i < n; ++i) out[i] = powf(base, y[i]); }
