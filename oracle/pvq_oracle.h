/*
 * pvq_oracle.h — CPU ORACLE for the pitchvis VQT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, single-precision restatement of the reference algorithm
 * (heinzelotto/pitchvis, crate pitchvis_analysis).  It is the checker the HIP path is
 * compared against; it is NOT part of the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libpvq.so) never
 * links, loads or calls anything in this directory.
 *
 * Pinning status: the reference ships NO golden vectors and cannot be compiled here (Rust,
 * no toolchain).  The oracle is pinned by the reference's own property / known-answer tests
 * ported 1:1 (tests/test_oracle_reference_properties.py: vqt.rs:996-1128, lib.rs:16-72,
 * analysis.rs:415-428) plus analytic known answers and an independent float64 model
 * (oracle/model_f64.py).  Exact-value parity with the Rust binary (rustfft 6.4.1 butterfly
 * order, find_peaks 0.1.5 internals) is UNPINNED.
 */
#ifndef PVQ_ORACLE_H
#define PVQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vqt.rs:238-262, 278-331 */
typedef struct {
    float sr;
    uint32_t n_fft;
    float min_freq;
    uint32_t octaves;
    uint32_t buckets_per_octave;
    float sparsity_quantile;
    float quality;
    float gamma;
} orc_params;

/* vqt.rs:350-366 */
enum { ORC_OK = 0, ORC_ABOVE_NYQUIST = 1, ORC_WINDOW_EXCEEDS_NFFT = 2 };

typedef struct orc_vqt orc_vqt;

/* vqt.rs:333-348 */
void orc_default_params(orc_params *p);

/* vqt.rs:465-505.  err_detail[0..1] = (highest_frequency, nyquist) or (window_length, n_fft). */
int orc_vqt_new(const orc_params *p, orc_vqt **out, float err_detail[2]);
void orc_vqt_free(orc_vqt *v);

uint32_t orc_n_bins(const orc_vqt *v);
uint32_t orc_n_groups(const orc_vqt *v);
double orc_delay_seconds(const orc_vqt *v); /* vqt.rs:756 */
float orc_window_center(const orc_vqt *v);  /* vqt.rs:605 */

/* per-bin filter params, vqt.rs:517-587: out arrays of n_bins */
void orc_filter_params(const orc_vqt *v, float *freq, float *window_length,
                       uint32_t *sr_downscaling_factor, uint32_t *min_window_size);

/* group g: info[0]=window begin, [1]=window end, [2]=rows, [3]=nnz, [4]=neg nnz (0 = None) */
void orc_group_info(const orc_vqt *v, uint32_t g, uint32_t info[5]);
/* CSR copy-out; neg=0 -> filter_bank, neg=1 -> negative_filter_bank.  values = interleaved re,im */
void orc_group_csr(const orc_vqt *v, uint32_t g, int neg, uint32_t *row_ptr, uint32_t *col_idx,
                   float *values);

/* vqt.rs:866-916: x has n_fft samples; out has n_bins dB values. */
void orc_calculate_vqt_instant_in_db(orc_vqt *v, const float *x, float *out_db);
/* same, but returns the complex coefficients before power_to_db (interleaved re,im) */
void orc_calculate_vqt_instant_complex(orc_vqt *v, const float *x, float *out_cplx);
/* per-group half spectrum (unnormalised R2C, vqt.rs:884-887): out has window/2+1 complex */
void orc_group_spectrum(orc_vqt *v, uint32_t g, const float *x, float *out_cplx);

/* vqt.rs:922-954 */
void orc_power_to_db(const float *x_cplx, uint32_t n, float *out_db);

/*
 * Batch framing (defined by the build, mirrors train.rs:276-277,306-310,341 and
 * audio_desktop.rs:113-115): a ring buffer of n_fft zeros; after hop f (0-based) has been
 * shifted in, frame f analyses the last n_fft samples, i.e. stream samples
 * [(f+1)*hop - n_fft, (f+1)*hop), zeros before the stream start.  `n_lead` samples of real
 * history precede the first hop inside `pcm` (0 at stream start; >0 for a shard with halo).
 * pcm holds n_lead + n_frames*hop samples.  out_db: [n_frames][n_bins].
 * out_cplx (optional, may be NULL): [n_frames][n_bins][2].
 */
void orc_calculate_batch(orc_vqt *v, const float *pcm, size_t n_lead, size_t hop, size_t n_frames,
                         float *out_db, float *out_cplx);

/* util.rs:62-79: test stimulus, n_fft samples */
void orc_test_create_sines(const orc_params *p, const float *freqs, uint32_t n_freqs, float t_diff,
                           float *wave);

/* ---- peaks: analysis_modules/peak_detection.rs ---- */

/* peak_detection.rs:26-51 (find_peaks 0.1.5 semantics restated, see .c).  Writes ascending bin
 * indices to out_idx (capacity n), returns the count. */
uint32_t orc_find_peaks(const float *vqt, uint32_t n, uint32_t buckets_per_octave,
                        float min_prominence, float min_height, uint32_t *out_idx);

/* analysis.rs:72-98 defaults */
typedef struct {
    float peak_min_prominence, peak_min_height;         /* 10.0, 4.0 */
    float bass_min_prominence, bass_min_height;         /* 5.0, 3.5 */
    uint32_t highest_bassnote;                          /* 28 */
    float harmonic_threshold;                           /* 0.3 */
} orc_analysis_params;
void orc_default_analysis_params(orc_analysis_params *a);

/* analysis.rs:332-349: bass/general split; ascending indices, returns count */
uint32_t orc_find_peaks_split(const float *vqt, uint32_t n, uint32_t buckets_per_octave,
                              const orc_analysis_params *a, uint32_t *out_idx);

/* peak_detection.rs:61-148; out_center/out_size sorted by center; returns count (= n_peaks) */
uint32_t orc_enhance_peaks_continuous(const uint32_t *peaks, uint32_t n_peaks, const float *vqt,
                                      float min_freq, uint32_t octaves, uint32_t buckets_per_octave,
                                      float *out_center, float *out_size);

/* peak_detection.rs:172-241 (in place on size[]) */
void orc_promote_bass_peaks_with_harmonics(const float *center, float *size, uint32_t n_peaks,
                                           const float *vqt, float min_freq, uint32_t octaves,
                                           uint32_t buckets_per_octave, uint32_t highest_bassnote,
                                           float harmonic_threshold);

/* stateless frame analysis = analysis.rs:332-361 with smoothing disabled (pass-through EMA,
 * analysis.rs:251-269 / util.rs:117-120).  Returns count; peaks ascending, continuous sorted. */
uint32_t orc_analyze_frame(const float *vqt, uint32_t n, float min_freq, uint32_t octaves,
                           uint32_t buckets_per_octave, const orc_analysis_params *a,
                           uint32_t *out_idx, float *out_center, float *out_size);

/* glibc expf / powf (used by oracle/analysis_state.py) */
float orc_expf(float x);
float orc_powf(float x, float y);
void orc_expf_v(const float *x, uint32_t n, float *out);
void orc_powf_v(float base, const float *y, uint32_t n, float *out);

/* FFT contracts (vqt.rs:1087-1128): unnormalised complex forward/inverse, R2C */
void orc_fft_complex(float *re_im_interleaved, uint32_t n, int inverse);
void orc_fft_real(const float *x, uint32_t n, float *out_cplx /* n/2+1 */);

#ifdef __cplusplus
}
#endif
#endif
