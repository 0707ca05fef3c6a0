/*
 * pvq.h — C ABI of libpvq: MI355X-native batched VQT pitch-analysis engine.
 *
 * Drop-in boundary for the hot path of heinzelotto/pitchvis' `pitchvis_analysis` crate.  The
 * reference has no FFI of its own (consumers link the rlib); every entry point
 * below names the Rust item it replaces (paths relative to pitchvis_analysis/src/).  A Rust shim
 * that re-exposes `Vqt` / `VqtParameters` / `VqtError` over these symbols is shown in
 * INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, caller-allocated outputs, no exceptions cross the ABI
 * (every entry point catches: where the reference panics in kernel construction, vqt.rs:785-792,
 * the call returns PVQ_ERR_INVALID_ARG with the reference's panic text; std::bad_alloc and the like
 * become PVQ_ERR_INTERNAL), every fallible call returns a pvq_status.  A handle is NOT thread-safe (the reference takes
 * `&mut self`, vqt.rs:866); distinct handles are independent (one per worker thread / stream,
 * as pitchvis_train/src/train.rs:146-155 does).  All compute runs on the GPU: there is no CPU
 * fallback, and every compute entry point fails with PVQ_ERR_NO_DEVICE on a handle created
 * without a device.
 */
#ifndef PVQ_H
#define PVQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVQ_ABI_VERSION 4   /* 4: + pvq_vqt_calculate_batch_db_streams, pvq_vqt_analyze_batch_streams (many streams per call).  3: + pvq_vqt_set_workspace_limit, pvq_plan_shard, pvq_vqt_analyze_batch_multi, pvq_analysis_batch_*, profiling mode 2; per-call NaN flag semantics of the synchronous entry points */

/* replaces VqtParameters + VqtRange (vqt.rs:238-262, 278-331), flattened POD */
typedef struct pvq_vqt_params {
    float sr;                    /* VqtParameters::sr */
    uint32_t n_fft;              /* VqtParameters::n_fft */
    float min_freq;              /* VqtRange::min_freq */
    uint32_t octaves;            /* VqtRange::octaves (u8 in the reference) */
    uint32_t buckets_per_octave; /* VqtRange::buckets_per_octave (u16 in the reference) */
    float sparsity_quantile;     /* VqtParameters::sparsity_quantile */
    float quality;               /* VqtParameters::quality */
    float gamma;                 /* VqtParameters::gamma */
} pvq_vqt_params;

/* replaces VqtError (vqt.rs:350-366) plus the panics of vqt.rs:867-871 / analysis.rs:289 */
typedef enum pvq_status {
    PVQ_OK = 0,
    PVQ_ERR_ABOVE_NYQUIST = 1,        /* VqtError::AboveNyquist{highest_frequency, nyquist_frequency} */
    PVQ_ERR_WINDOW_EXCEEDS_NFFT = 2,  /* VqtError::WindowExceedsNFft{window_length, n_fft} */
    PVQ_ERR_BAD_LENGTH = 3,           /* assert_eq!(x.len(), n_fft) vqt.rs:867-871 */
    PVQ_ERR_INVALID_ARG = 4,
    PVQ_ERR_NO_DEVICE = 5,            /* handle has no GPU context (created with device_id < 0) */
    PVQ_ERR_DEVICE = 6,               /* a HIP call failed; see pvq_last_error() */
    PVQ_ERR_UNSUPPORTED = 7,          /* geometry outside what the kernels support */
    PVQ_ERR_INTERNAL = 8,             /* out of host memory or an unexpected C++ exception, caught at the ABI; pvq_last_error() has the text */
    PVQ_ERR_NONFINITE_INPUT = 9       /* a NaN / Inf sample reached a frame (see pvq_vqt_input_status) */
} pvq_status;

const char *pvq_status_string(pvq_status s);
/* thread-local text of the last failing call on this thread */
const char *pvq_last_error(void);
uint32_t pvq_abi_version(void);

/* replaces `impl Default for VqtParameters` (vqt.rs:333-348) and DEFAULT_* (vqt.rs:180-214) */
void pvq_vqt_default_params(pvq_vqt_params *p);

typedef struct pvq_vqt pvq_vqt;

/*
 * replaces Vqt::new (vqt.rs:465-505).  Builds the multi-rate sparse kernel on the host
 * (vqt.rs:517-852) and, when device_id >= 0, uploads it to that GPU.  device_id < 0 creates a
 * host-only "plan" handle: getters work, compute entry points return PVQ_ERR_NO_DEVICE.
 * On PVQ_ERR_ABOVE_NYQUIST / PVQ_ERR_WINDOW_EXCEEDS_NFFT err_detail receives the two fields of
 * the corresponding VqtError variant.
 */
pvq_status pvq_vqt_create(const pvq_vqt_params *params, int device_id, pvq_vqt **out,
                          float err_detail[2]);
/* replaces Drop for Vqt (the viewer swaps the instance at runtime, pitchvis_viewer/src/app/common.rs:1133) */
void pvq_vqt_destroy(pvq_vqt *v);

/* replaces Vqt::params() (vqt.rs:507-509) */
void pvq_vqt_get_params(const pvq_vqt *v, pvq_vqt_params *out);
/* replaces VqtRange::n_buckets() (vqt.rs:259-261) */
uint32_t pvq_vqt_n_bins(const pvq_vqt *v);
/* replaces `pub delay: Duration` (vqt.rs:449, :756), in seconds */
double pvq_vqt_delay_seconds(const pvq_vqt *v);
/* number of trailing samples of the n_fft buffer that any window group reads (the window union) */
uint32_t pvq_vqt_window_union(const pvq_vqt *v);
/* replaces Filter::bandwidth_3db_in_hz (vqt.rs:421, :817-818; find_3db_points / calculate_bandwidth, vqt.rs:956-989):
 * the -3 dB band of every bin's filter in Hz, read off its decimated frequency response; n_bins floats each */
pvq_status pvq_vqt_bandwidths_3db(const pvq_vqt *v, float *lo_hz, float *hi_hz);
/* the warn!() lines of the kernel construction (vqt.rs:695-709: a coverage gap between the -3 dB bands of neighbouring
 * filters — "decrease quality to close the gap"), in bin order: their number, and line i copied NUL-terminated into buf
 * (truncated to cap bytes) */
uint32_t pvq_vqt_warning_count(const pvq_vqt *v);
pvq_status pvq_vqt_warning(const pvq_vqt *v, uint32_t i, char *buf, size_t cap);

/* replaces Vqt::kernel() (vqt.rs:511-513) -> VqtKernel{window_groups} (vqt.rs:388-415) */
uint32_t pvq_vqt_n_groups(const pvq_vqt *v);
/* info[0]=window.0, [1]=window.1, [2]=filter_bank.rows(), [3]=filter_bank.nnz(),
 * [4]=negative_filter_bank nnz (0 <=> None) */
pvq_status pvq_vqt_group_info(const pvq_vqt *v, uint32_t group, uint32_t info[5]);
/* CSR copy-out of WindowGroup::filter_bank (negative=0) or ::negative_filter_bank (negative=1);
 * values are interleaved (re, im) Complex32 */
pvq_status pvq_vqt_group_csr(const pvq_vqt *v, uint32_t group, int negative, uint32_t *row_ptr,
                             uint32_t *col_idx, float *values);
/* per-bin FilterParams (vqt.rs:370-384), arrays of n_bins */
pvq_status pvq_vqt_filter_params(const pvq_vqt *v, float *freq, float *window_length,
                                 uint32_t *sr_downscaling_factor, uint32_t *minimum_needed_window_size);

/*
 * replaces Vqt::calculate_vqt_instant_in_db (vqt.rs:866-916): x = exactly n_fft host samples,
 * the last of which is "now"; out_db = n_bins host floats.  len != n_fft -> PVQ_ERR_BAD_LENGTH
 * (the reference panics).  Synchronous, and shaped for latency: only the window union of x (its last
 * pvq_vqt_window_union samples: the kernel reads nothing before them) is staged, page-locked, on a stream of the handle's
 * own; a non-finite sample among them returns PVQ_ERR_NONFINITE_INPUT before anything is launched (out_db untouched).
 * Always the FFT path (one frame has nothing to share with a neighbour), whatever pvq_vqt_set_algo says.
 */
pvq_status pvq_vqt_calculate_instant_db(pvq_vqt *v, const float *x, size_t len, float *out_db);

/*
 * Batched form of the same call, the data-parallel path the reference runs with one Vqt per
 * rayon worker (pitchvis_train/src/train.rs:146-155, :276-310, :341).
 * Framing: a ring buffer of n_fft zeros; hop f (0-based) shifts `hop` samples in, then frame f
 * analyses the last n_fft samples (pitchvis_audio/src/audio_desktop.rs:113-115 shift semantics).
 * `pcm` holds n_lead + n_frames*hop samples; the first n_lead are history that precedes hop 0
 * (0 at stream start, > 0 for a shard that carries a halo); samples before pcm[0] are zeros.
 * out_db: [n_frames][n_bins] row-major.  Host pointers; synchronous.
 */
pvq_status pvq_vqt_calculate_batch_db(pvq_vqt *v, const float *pcm, size_t n_lead, size_t hop,
                                      size_t n_frames, float *out_db);

/*
 * Device-pointer form: d_pcm, d_out_db (and optional d_out_cplx: [n_frames][n_bins][2], the
 * complex coefficients before power_to_db, may be NULL) are device memory on the handle's GPU.
 * Enqueued on `stream` (a hipStream_t; NULL = default stream); returns without synchronising.
 */
pvq_status pvq_vqt_calculate_batch_db_device(pvq_vqt *v, const float *d_pcm, size_t n_lead,
                                             size_t hop, size_t n_frames, float *d_out_db,
                                             float *d_out_cplx, void *stream);

/*
 * NaN / Inf policy.  The reference's callers never pass non-finite samples to the transform (the audio
 * callback drops such chunks, pitchvis_audio/src/audio_desktop.rs:102-105), and the NaNs they would cause make its peak
 * stage panic (peak_detection.rs:145 `partial_cmp().unwrap()`).  Here a non-finite sample inside one of a frame's windows
 * raises a sticky flag on the device while the frame's dB values are computed; the frames it touches are unspecified.
 * The synchronous host-buffer batch entry points (pvq_vqt_calculate_batch_db, pvq_train_frames_db) check the flag themselves
 * and return PVQ_ERR_NONFINITE_INPUT; such a call reports its own input only — it clears, unreported, whatever an earlier
 * asynchronous call of the same handle raised and nobody polled.  pvq_vqt_calculate_instant_db does not use the flag at all: it
 * checks its window union on the host while staging it, launches nothing on a non-finite sample, and neither reads nor clears what
 * an earlier asynchronous call left behind.  For the asynchronous device-pointer entry points call this: it waits for `stream`,
 * returns PVQ_ERR_NONFINITE_INPUT if the flag is up (PVQ_OK otherwise) and clears it (read and clear are one stream-ordered step on
 * `stream`: poll on the stream the work was queued on).  The flag is per handle.
 *
 * Ordering of one handle's calls.  The handle owns its workspaces, so its device calls are ordered: a call on another stream than
 * the handle's previous call — pvq_vqt_calculate_instant_db and pvq_stream_frame_db run on streams of their own — first makes its
 * stream wait, on the device, for everything the previous call queued (the host does not block; calls that stay on one stream pay
 * nothing).  The handle itself is still not thread-safe: one handle per thread, as one `&mut Vqt` in the reference.
 */
pvq_status pvq_vqt_input_status(pvq_vqt *v, void *stream);

/* algorithm selection for the batch path (default PVQ_ALGO_AUTO) */
typedef enum pvq_algo {
    PVQ_ALGO_AUTO = 0,
    PVQ_ALGO_FFT = 1,      /* per-frame LDS FFT per window group (any hop) */
    PVQ_ALGO_BLOCKDFT = 2  /* hop-block DFT on fp32 MFMA + phase combine: every hop block is transformed once for all the frames that
                            * share it.  Takes a power-of-two hop (>= 64) that divides every window (doubling tree); a multiple of 64 that
                            * the longest window holds at most 16 times, e.g. 1 600 (whole blocks + the window's remainder, Horner combine);
                            * or a hop whose 2-, 4-, 8- or 16-fold is one of those, e.g. 800 or 320 (that many interleaved block grids).
                            * PVQ_ALGO_AUTO picks it where it applies and pays (pvq_vqt_resolve_algo); forcing it on another hop
                            * (735, say) returns PVQ_ERR_UNSUPPORTED */
} pvq_algo;
pvq_status pvq_vqt_set_algo(pvq_vqt *v, pvq_algo algo);
/* which algorithm the last batch call actually used */
pvq_algo pvq_vqt_last_algo(const pvq_vqt *v);
/* which algorithm a batch of n_frames frames (all streams of a *_streams call together) at this hop takes under the current setting.
 * PVQ_ALGO_AUTO takes the block-DFT path where it applies and is the faster one: from 384 frames on for a power-of-two hop; for a
 * general hop (1 600, 800 ...) from where its launch floor — a K loop hop / 2 deep per tile — is paid back, ~1 700 frames at
 * 48 kHz / 252 bins / hop 1 600 (smaller batches: the FFT path, a workgroup per frame).  The two paths agree to the parity bars, not
 * bit for bit: a caller that needs the same bits for every batch size fixes the path with pvq_vqt_set_algo. */
pvq_algo pvq_vqt_resolve_algo(pvq_vqt *v, size_t hop, size_t n_frames);

/* arithmetic of the block-DFT GEMM and kernel product (default PVQ_GEMM_F32).  Both accumulate in fp32 and meet the
 * same parity bars; PVQ_GEMM_BF16X3 writes each fp32 operand exactly as three bf16 terms and uses the bf16 matrix
 * cores (six exact partial products per fp32 product, dropped terms < 2^-24): ~10 % faster end to end. */
typedef enum pvq_gemm_precision {
    PVQ_GEMM_F32 = 0,     /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32)  */
    PVQ_GEMM_BF16X3 = 1   /* split-bf16: 6 bf16 MFMAs per fp32 product block, error at fp32 rounding level */
} pvq_gemm_precision;
pvq_status pvq_vqt_set_gemm_precision(pvq_vqt *v, pvq_gemm_precision p);
/* twiddle tables (FFT twiddles, real-split factors, block-DFT matrix and combine phases) rounded to IEEE half
 * before use, accumulation in fp32: the "fp16 FFT twiddles" variant of BASELINE.json configs[3].  Off by default;
 * switching rebuilds the device tables.  Expect ~1e-3 relative error (tests/test_configs_gpu.py reports it). */
pvq_status pvq_vqt_set_twiddle_fp16(pvq_vqt *v, int enable);
/* Upper bound, in bytes, of the block-DFT path's intermediate spectrum workspace of this handle (grow-only device memory,
 * ~5.6 KB per frame of a sub-batch at 48 kHz / 252 bins, more with more spectrum columns).  A batch longer than the workspace
 * holds is processed in sub-batches of a multiple of 64 frames, at most 147 456.  Default (and bytes == 0): 1 GiB.  The
 * reference's trainer keeps one Vqt per worker thread (pitchvis_train/src/train.rs:146-155): size this per handle. */
pvq_status pvq_vqt_set_workspace_limit(pvq_vqt *v, uint64_t bytes);
/* complex spectrum columns per hop block the block-DFT GEMM computes (padded), 0 before its first use */
uint32_t pvq_vqt_blockdft_columns(const pvq_vqt *v);

/* ---- peak / note detection: analysis_modules/peak_detection.rs + analysis.rs:332-361 ---- */

/* replaces the peak-related fields of AnalysisParameters (analysis.rs:36-65, defaults :72-98) */
typedef struct pvq_analysis_params {
    float peak_min_prominence;     /* peak_config.min_prominence = 10.0 */
    float peak_min_height;         /* peak_config.min_height = 4.0 */
    float bass_min_prominence;     /* bassline_peak_config.min_prominence = 5.0 */
    float bass_min_height;         /* bassline_peak_config.min_height = 3.5 */
    uint32_t highest_bassnote;     /* 28 */
    float harmonic_threshold;      /* 0.3 */
} pvq_analysis_params;
void pvq_analysis_default_params(pvq_analysis_params *a);

/*
 * Per-frame stateless analysis of dB frames, i.e. AnalysisState::preprocess's peak pipeline
 * (analysis.rs:332-361: find_peaks bass/general split -> enhance_peaks_continuous ->
 * promote_bass_peaks_with_harmonics) with the EMA in pass-through mode
 * (update_vqt_smoothing_duration(None), analysis.rs:251-269).
 * d_db: [n_frames][n_bins] device.  Outputs (device, any may be NULL):
 *   d_peak_mask  [n_frames][ceil(n_bins/32)] u32 bitmask of AnalysisState::peaks
 *   d_peak_count [n_frames] number of peaks
 *   d_center/d_size [n_frames][max_peaks] AnalysisState::peaks_continuous, ascending center;
 *                   entries beyond the frame's count are untouched.
 */
pvq_status pvq_analyze_batch_device(pvq_vqt *v, const float *d_db, size_t n_frames,
                                    const pvq_analysis_params *a, uint32_t *d_peak_mask,
                                    uint32_t *d_peak_count, float *d_center, float *d_size,
                                    uint32_t max_peaks, void *stream);

/* Host-pointer convenience wrapper of the above (uploads, runs on the GPU, downloads). */
pvq_status pvq_analyze_batch(pvq_vqt *v, const float *db, size_t n_frames,
                             const pvq_analysis_params *a, uint32_t *peak_mask, uint32_t *peak_count,
                             float *center, float *size, uint32_t max_peaks);

/* The whole hot path in one call: PCM -> VQT dB frames -> peaks (device pointers, async). */
pvq_status pvq_vqt_analyze_batch_device(pvq_vqt *v, const float *d_pcm, size_t n_lead, size_t hop,
                                        size_t n_frames, const pvq_analysis_params *a,
                                        float *d_out_db, uint32_t *d_peak_mask,
                                        uint32_t *d_peak_count, float *d_center, float *d_size,
                                        uint32_t max_peaks, void *stream);


/* ---- many streams, one call ----------------------------------------------------------------------------
 * The reference's batch driver analyses MANY independent files side by side, one Vqt per rayon worker
 * (pitchvis_train/src/train.rs:146-163), and a stereo recording is two streams.  Here one handle takes all of them in one
 * call: d_pcm is a HOST array of n_streams device pointers; stream s holds n_lead[s] + n_frames[s] * hop samples (n_lead
 * NULL: no history anywhere) and is framed exactly as by pvq_vqt_calculate_batch_db_device; its frame f goes to row
 * s * out_stride_frames + f of d_out_db [n_streams][out_stride_frames][n_bins] (out_stride_frames >= every n_frames[s]) — the
 * layout pvq_analysis_batch_preprocess_device reads, so PCM -> VQT -> AnalysisState::preprocess of many streams never leaves
 * the device.  Rows a stream does not fill are zero frames.  With a power-of-two hop (the block-DFT path) every stage covers
 * ALL streams with one launch — 64 streams of 2 048 frames cost what one 131 072-frame stream costs, not 64 launch ramps and
 * tails per stage; every value equals, bit for bit, what the single-stream call computes for that stream.  Streams of at most
 * 2 048 frames are first copied one behind the other into a staging buffer of the handle (grow-only device memory, at most 512 MiB:
 * a tile of the GEMM then never stops at a stream's end; 256 streams of 512 frames run at 0.86 instead of 0.61 of the single-stream
 * rate); longer ones are read where they are.  Asynchronous on `stream`; NaN / Inf policy as for the device-pointer entry points
 * (pvq_vqt_input_status). */
pvq_status pvq_vqt_calculate_batch_db_streams(pvq_vqt *v, const float *const *d_pcm, const size_t *n_lead,
                                              const size_t *n_frames, uint32_t n_streams, size_t hop, float *d_out_db,
                                              size_t out_stride_frames, void *stream);
/* The same with the per-frame peak pipeline behind it (as pvq_vqt_analyze_batch_device): peak outputs are laid out by the
 * same rows, [n_streams][out_stride_frames][...]; any may be NULL (center and size go together). */
pvq_status pvq_vqt_analyze_batch_streams(pvq_vqt *v, const float *const *d_pcm, const size_t *n_lead,
                                         const size_t *n_frames, uint32_t n_streams, size_t hop,
                                         const pvq_analysis_params *a, float *d_out_db, size_t out_stride_frames,
                                         uint32_t *d_peak_mask, uint32_t *d_peak_count, float *d_center, float *d_size,
                                         uint32_t max_peaks, void *stream);

/* ---- several devices, one stream ---------------------------------------------------------------------
 * The reference's only data-parallel driver hands every rayon worker its own Vqt (pitchvis_train/src/train.rs:146-155:
 * par_iter().map_init(|| Vqt::new(..), ..)).  The same shape here: one handle per worker (each on its own device, or
 * several on one), ONE long stream split into contiguous frame ranges, each with a halo of window_union - hop samples of
 * history, kernel tables replicated, no collective on the data path. */
typedef struct pvq_shard {
    uint64_t first_frame;   /* global index of the shard's first frame */
    uint64_t n_frames;
    uint64_t sample_begin;  /* first sample of the stream the shard must hold, counted from the stream's first hop */
    uint64_t sample_end;    /* one past the last */
    uint64_t n_lead;        /* samples of [sample_begin, sample_end) that are history before the shard's first hop */
} pvq_shard;
/* contiguous split of n_frames_total frames over `world` shards (the first n_frames_total % world take one more) */
pvq_status pvq_plan_shard(uint64_t n_frames_total, uint64_t hop, uint64_t window_union, uint32_t rank, uint32_t world,
                          pvq_shard *out);
/* PCM -> dB frames -> peaks of one HOST stream on n_handles handles at once, one host thread each; the handles must have
 * been created with the same parameters, and none may appear twice.  pcm: [n_lead + n_frames * hop]; outputs are host
 * arrays laid out as for pvq_vqt_analyze_batch_device (peak outputs may be NULL; center and size go together).  Every
 * output value equals, bit for bit, what one handle computes for the whole stream. */
pvq_status pvq_vqt_analyze_batch_multi(pvq_vqt *const *handles, uint32_t n_handles, const float *pcm, size_t n_lead,
                                       size_t hop, size_t n_frames, const pvq_analysis_params *a, float *out_db,
                                       uint32_t *peak_mask, uint32_t *peak_count, float *center, float *size,
                                       uint32_t max_peaks);

/* ---- stateful per-stream analysis: AnalysisState (analysis.rs:119-410), host side ------------------
 * preprocess() is a recurrence over frames (bin EMAs, calmness EMAs and the scene calmness feed the
 * next frame's smoothing horizons: analysis.rs:295-319, calmness.rs:23-95), so it is sequential per
 * stream and runs on the host, fed with dB frames computed on the GPU.  One handle per stream. */

/* replaces AnalysisParameters (analysis.rs:36-65), Default at :72-98; durations in nanoseconds */
typedef struct pvq_analysis_full_params {
    uint32_t spectrogram_length;                       /* 400 (unused by the crate itself) */
    float peak_min_prominence, peak_min_height;        /* peak_config 10.0 / 4.0 */
    float bass_min_prominence, bass_min_height;        /* bassline_peak_config 5.0 / 3.5 */
    uint32_t highest_bassnote;                         /* 28 */
    uint64_t vqt_smoothing_duration_base_ns;           /* 70 ms */
    float vqt_smoothing_calmness_min, vqt_smoothing_calmness_max;  /* 0.6 / 2.0 */
    uint64_t note_calmness_smoothing_duration_ns;      /* 3500 ms */
    uint64_t scene_calmness_smoothing_duration_ns;     /* 800 ms */
    uint64_t tuning_inaccuracy_smoothing_duration_ns;  /* 4000 ms */
    float harmonic_threshold;                          /* 0.3 */
} pvq_analysis_full_params;
void pvq_analysis_full_default_params(pvq_analysis_full_params *p);

typedef struct pvq_analysis_state pvq_analysis_state;
/* replaces AnalysisState::new(range, params) (analysis.rs:192) */
pvq_status pvq_analysis_state_create(float min_freq, uint32_t octaves, uint32_t buckets_per_octave,
                                     const pvq_analysis_full_params *params, pvq_analysis_state **out);
void pvq_analysis_state_destroy(pvq_analysis_state *s);
/* replaces update_vqt_smoothing_duration(Option<Duration>) (analysis.rs:251); has_duration = 0 is None */
pvq_status pvq_analysis_state_update_vqt_smoothing_duration(pvq_analysis_state *s, int has_duration, uint64_t duration_ns);
/* replaces preprocess(&[f32], Duration) (analysis.rs:288); a wrong length returns PVQ_ERR_BAD_LENGTH
 * where the reference panics (analysis.rs:289) */
pvq_status pvq_analysis_state_preprocess(pvq_analysis_state *s, const float *x_vqt, size_t len, uint64_t frame_time_ns);
/* replaces bin_to_frequency (analysis.rs:407) */
float pvq_analysis_state_bin_to_frequency(const pvq_analysis_state *s, uint32_t bin);

/* the `pub` result fields (analysis.rs:119-177); arrays have n_buckets entries */
typedef enum pvq_analysis_field {
    PVQ_FIELD_X_VQT_SMOOTHED = 0,  /* x_vqt_smoothed[i].get() */
    PVQ_FIELD_X_VQT_PEAKFILTERED = 1,
    PVQ_FIELD_X_VQT_AFTERGLOW = 2,
    PVQ_FIELD_CALMNESS = 3,        /* calmness[i].get() */
    PVQ_FIELD_PITCH_ACCURACY = 4,
    PVQ_FIELD_PITCH_DEVIATION = 5
} pvq_analysis_field;
uint32_t pvq_analysis_state_n_buckets(const pvq_analysis_state *s);
pvq_status pvq_analysis_state_get_field(const pvq_analysis_state *s, pvq_analysis_field f, float *out);
/* peaks (ascending bin indices) and peaks_continuous (ascending center); return the total count */
uint32_t pvq_analysis_state_get_peaks(const pvq_analysis_state *s, uint32_t *out, uint32_t capacity);
uint32_t pvq_analysis_state_get_peaks_continuous(const pvq_analysis_state *s, float *center, float *size, uint32_t capacity);
float pvq_analysis_state_scene_calmness(const pvq_analysis_state *s);            /* smoothed_scene_calmness.get() */
float pvq_analysis_state_tuning_grid_inaccuracy(const pvq_analysis_state *s);    /* smoothed_tuning_grid_inaccuracy.get() */

/* ---- AnalysisState for MANY streams, on the GPU ------------------------------------------------------
 * preprocess() cannot be split over time, but streams are independent (the trainer analyses many files side by side,
 * pitchvis_train/src/train.rs:146-155): one wavefront owns one stream and walks its frames in order, thousands of streams in
 * parallel ("replicas only": no exchange between streams or devices).  Same arithmetic, same operation order as the host
 * pvq_analysis_state above; every stream starts as AnalysisState::new leaves it (analysis.rs:192-241) and keeps its state
 * between calls. */
typedef struct pvq_analysis_batch pvq_analysis_batch;
/* per-frame results, DEVICE pointers, any may be NULL (center and size go together, with max_peaks > 0).  Per-bin fields:
 * [n_streams][n_frames][n_bins]; peak_mask [..][..][ceil(n_bins/32)]; center / size [..][..][max_peaks] (ascending center, entries
 * beyond the frame's count untouched); peak_count, scene_calmness, tuning_grid_inaccuracy [n_streams][n_frames]. */
typedef struct pvq_analysis_batch_outputs {
    float *x_vqt_smoothed, *x_vqt_peakfiltered, *x_vqt_afterglow, *calmness, *pitch_accuracy, *pitch_deviation;
    uint32_t *peak_mask, *peak_count;
    float *center, *size;
    uint32_t max_peaks;
    float *scene_calmness, *tuning_grid_inaccuracy;
} pvq_analysis_batch_outputs;
/* n_streams AnalysisState::new(range, params) on device `device_id` (params NULL: AnalysisParameters::default) */
pvq_status pvq_analysis_batch_create(int device_id, float min_freq, uint32_t octaves, uint32_t buckets_per_octave,
                                     const pvq_analysis_full_params *params, uint32_t n_streams, pvq_analysis_batch **out);
void pvq_analysis_batch_destroy(pvq_analysis_batch *b);
/* AnalysisState::update_vqt_smoothing_duration (analysis.rs:251) on every stream */
pvq_status pvq_analysis_batch_update_vqt_smoothing_duration(pvq_analysis_batch *b, int has_duration, uint64_t duration_ns);
/* AnalysisState::preprocess (analysis.rs:288) for n_frames frames of every stream, in order: d_db [n_streams][n_frames][n_bins]
 * (device).  frame_time_ns applies to every frame unless frame_times_ns (HOST array of n_frames) is given.  Asynchronous on
 * `stream` — except that a call whose frame time(s) differ from the previous call's first waits for `stream` (the table of EMA weights
 * 1 - exp(-2 frame_time / horizon), built on the host with its libm so that the device follows the host AnalysisState bit for bit, is
 * replaced). */
pvq_status pvq_analysis_batch_preprocess_device(pvq_analysis_batch *b, const float *d_db, size_t n_frames, uint64_t frame_time_ns,
                                                const uint64_t *frame_times_ns, const pvq_analysis_batch_outputs *outs,
                                                void *stream);
/* The consumers' loop in one call, for many streams: per stream and hop what pitchvis_viewer does per rendered frame
 * (vqt_system.rs:40-68 -> Vqt::calculate_vqt_instant_in_db, analysis_system.rs:10-20 -> AnalysisState::preprocess) and pitchvis_serial
 * per 1 / 30 s (main.rs:205-215).  d_pcm / n_lead as for pvq_vqt_calculate_batch_db_streams, n_frames frames of EVERY stream (the
 * batch's n_streams of them); the dB frames [n_streams][n_frames][n_bins] go through a buffer of the batch object (grow-only device
 * memory) — or through d_db if the caller wants them — and never leave the device.  v and b must sit on the same device and share
 * the VqtRange.  frame_time_ns: the hop's duration (hop / sr, what a live consumer measures between two analyses).  Asynchronous. */
pvq_status pvq_analysis_batch_preprocess_pcm(pvq_analysis_batch *b, pvq_vqt *v, const float *const *d_pcm, const size_t *n_lead,
                                             size_t n_frames, size_t hop, uint64_t frame_time_ns, float *d_db,
                                             const pvq_analysis_batch_outputs *outs, void *stream);
/* the state of one stream after the last call (synchronises): a pub field as pvq_analysis_state_get_field, and the two scalars */
pvq_status pvq_analysis_batch_get_field(pvq_analysis_batch *b, uint32_t stream_index, pvq_analysis_field f, float *out);
pvq_status pvq_analysis_batch_get_scalars(pvq_analysis_batch *b, uint32_t stream_index, float *scene_calmness,
                                          float *tuning_grid_inaccuracy);

/* ------------------------------------------------------------------------------------------------
 * Callers either side of the path: host code around the GPU frames.
 * ---------------------------------------------------------------------------------------------- */

/* dagc::MonoAgc (dagc_fork/src/lib.rs:19-87).  create: PVQ_ERR_INVALID_ARG with the reference's Error text
 * for a bad desired_output_rms / distortion_factor (lib.rs:36-49). */
typedef struct pvq_mono_agc pvq_mono_agc;
pvq_status pvq_mono_agc_create(float desired_output_rms, float distortion_factor, pvq_mono_agc **out);
void pvq_mono_agc_destroy(pvq_mono_agc *a);
void pvq_mono_agc_freeze_gain(pvq_mono_agc *a, int freeze);      /* lib.rs:62 */
int pvq_mono_agc_is_gain_frozen(const pvq_mono_agc *a);           /* lib.rs:67 */
float pvq_mono_agc_gain(const pvq_mono_agc *a);                   /* lib.rs:72 */
void pvq_mono_agc_process(pvq_mono_agc *a, float *samples, size_t n);   /* lib.rs:76-86, in place */

/* pitchvis_train as a batch (pitchvis_train/src/train.rs).
 * chunk length: (delay_ms * sr / 1000) / 64 * 64, train.rs:128-129 */
size_t pvq_train_chunk_samples(const pvq_vqt *v);
/* train.rs:286-310 over n_chunks rendered chunks of `chunk` samples: downmix (l + r) / 2 (right may be NULL),
 * silence gate (sum of squares < 1e-6 freezes the gain for the chunk), AGC in place.  mono_out
 * [n_chunks * chunk]; gain_out [n_chunks] (may be NULL) = agc.gain() after each chunk. */
pvq_status pvq_train_condition_stream(pvq_mono_agc *a, const float *left, const float *right, size_t n_chunks,
                                      size_t chunk, float *mono_out, float *gain_out);
/* train.rs:341 for every step-th chunk (STEP_SIZE_IN_CHUNKS = 3, train.rs:44), on the GPU: frame f = VQT dB of
 * the last n_fft conditioned samples after chunk (f+1)*step (zeros before the stream, the ring buffer of
 * train.rs:268-269).  out_db [n_chunks / step][n_bins]. */
pvq_status pvq_train_frames_db(pvq_vqt *v, const float *mono, size_t n_chunks, size_t chunk, size_t step, float *out_db);
/* train.rs:317-337, 347, 443-460: row f = n_bins dB values, then 128 targets from the active keys of frame
 * f-1 (a key's value: largest (mix_left + mix_right) / 2 * agc_gain over its voices; target = value > 0.5).
 * voice_ptr [n_frames + 1] indexes the voice arrays; agc_gain [n_frames].  out_rows [n_frames][n_bins + 128].
 * PVQ_ERR_INVALID_ARG for a key outside 0..127 (the reference would panic). */
pvq_status pvq_train_rows(const float *db, size_t n_frames, uint32_t n_bins, const uint32_t *voice_ptr,
                          const int32_t *voice_key, const float *voice_gain_left, const float *voice_gain_right,
                          const float *agc_gain, float *out_rows);
/* train.rs:192-208: flat little-endian f32 .npy, shape (n,) */
pvq_status pvq_npy_write_f32(const char *path, const float *data, uint64_t n);

/* Streaming front end: the pitchvis_audio RingBuffer contract (pitchvis_audio/src/lib.rs:17-22,
 * audio_desktop.rs:88-131) with the ring resident on the device.  The ring holds buf_size samples, zeros at
 * start; push() is the audio callback: a chunk containing a non-finite sample is dropped (audio_desktop.rs:
 * 97-100), otherwise the silence gate + MonoAgc(0.07, 0.0001) run over it (with_agc != 0), it is shifted into
 * the ring (drain + extend) and gain / chunk_size_ms are updated.  frame_db() is the consumer of
 * pitchvis_serial/src/main.rs:205-211: the VQT of the newest n_fft samples.  Not thread-safe: serialise push
 * and frame_db like the reference's mutex does. */
typedef struct pvq_stream pvq_stream;
pvq_status pvq_stream_create(pvq_vqt *v, size_t buf_size, int with_agc, pvq_stream **out);
void pvq_stream_destroy(pvq_stream *s);
pvq_status pvq_stream_push(pvq_stream *s, const float *data, size_t n);
float pvq_stream_gain(const pvq_stream *s);             /* RingBuffer.gain */
float pvq_stream_chunk_size_ms(const pvq_stream *s);    /* RingBuffer.chunk_size_ms */
pvq_status pvq_stream_frame_db(pvq_stream *s, float *out_db);
/* host copy of the newest n_last samples of the ring (n_last <= buf_size) */
pvq_status pvq_stream_read(pvq_stream *s, float *out, size_t n_last);

/* pitchvis_colors::calculate_color (pitchvis_colors/src/lib.rs:86-117); colors: 12 RGB triples in [0, 1] */
void pvq_calculate_color(uint16_t buckets_per_octave, float bucket, const float *colors, float gray_level,
                         float easing_pow, float out_rgb[3]);
/* pitchvis_serial::update_serial (pitchvis_serial/src/main.rs:122-175): 0xFF, 16-bit triple count (big endian),
 * then one RGB triple (each byte <= 0xFE) per bucket.  center/size: AnalysisState::peaks_continuous.  out must
 * hold 3 + 3 * n_buckets bytes; returns the number of bytes written. */
size_t pvq_led_frame(uint32_t n_buckets, uint16_t buckets_per_octave, const float *center, const float *size,
                     uint32_t n_peaks, const float *colors, float gray_level, float easing_pow, uint8_t *out);

/* Page-locked host memory for the host-buffer entry points (pvq_vqt_calculate_batch_db, pvq_analyze_batch,
 * pvq_train_frames_db): with pageable buffers those calls are bound by staged PCIe copies (~16 GB/s); buffers from
 * here are DMA-able directly.  NULL on failure (pvq_last_error). */
void *pvq_host_alloc(size_t bytes);
void pvq_host_free(void *p);

/* timing hook for bench.py: elapsed GPU milliseconds of the dominant kernel launches of the
 * last batch call, measured with HIP events on the stream the kernels were launched on.
 * Enable with pvq_vqt_set_profiling(v, 1) (resets the statistics); reading synchronises.  enable == 2 times only the transform's
 * main kernel (the fused GEMM + tree, or the FFT-path kernel): two event records per step instead of eight — the events themselves
 * cost ~3 us each on the stream. */
pvq_status pvq_vqt_set_profiling(pvq_vqt *v, int enable);
/* out_ms[i] for kernel slot i (see pvq_vqt_kernel_name); returns the number of slots filled */
uint32_t pvq_vqt_last_kernel_ms(pvq_vqt *v, float *out_ms, uint32_t capacity);
/* out_n[i] = launches of kernel slot i recorded since profiling was enabled (a batch call may
 * launch a kernel once per sub-batch); out_ms above is the mean per launch */
uint32_t pvq_vqt_last_kernel_launches(pvq_vqt *v, uint32_t *out_n, uint32_t capacity);
/* flop issued by the matrix instructions of the last block-DFT GEMM launch (tiles x rows x columns x depth x 2, padding and
 * recomputed rows included); 0 after a call that took the FFT path */
double pvq_vqt_last_gemm_flop(const pvq_vqt *v);
/* shader clock (MHz) the chip held inside the GEMM kernel's K loop during the last profiled launch: median over sampled
 * workgroups of shader-clock ticks / 100 MHz ticks; 0 if nothing was measured.  Synchronises the device. */
float pvq_vqt_last_sclk_mhz(pvq_vqt *v);
/* frames one launch of the frame kernels processed in the last batch call (the sub-batch size) */
uint32_t pvq_vqt_last_frames_per_launch(const pvq_vqt *v);
const char *pvq_vqt_kernel_name(uint32_t slot);

#ifdef __cplusplus
}
#endif
#endif /* PVQ_H */
