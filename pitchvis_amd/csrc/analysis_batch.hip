// analysis_batch.hip — AnalysisState::preprocess for many streams, one wavefront per stream (analysis_batch.hpp).
//
// Every arithmetic step below is the host AnalysisState's (analysis_host.cpp, itself the reference's operation order, file:line
// cited there), in f32 with FMA contraction off.  Two things make the GPU values follow the host's through the recurrence:
//   * the libm calls of the reference (exp, ln, log2, log10, powf) are evaluated in double and rounded once — the correctly rounded
//     f32 result in all but ~1e-8 of the calls, which is what glibc's expf / logf / log2f / powf deliver too;
//   * the sums the reference accumulates bin by bin (scene calmness: calmness.rs:62-92; tuning inaccuracy: pitch_analysis.rs:55-66)
//     are accumulated in that same order here, not as a tree.
// One 64-lane wave owns a stream: its bins sit at lane + 64 k (k < NK), the per-bin EMA states live in registers for the whole
// call, the three find_peaks passes per frame (bass / general split on the smoothed frame, general on the raw frame) reuse the
// wave routine of the batched peak kernels (peaks_device.hpp: bit-identical peak sets).
#include "analysis_batch.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "peaks_device.hpp"
#include "vqt_engine.hpp"

namespace pvq {

#define PVQ_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_last_error(std::string(#call) + " failed: " + hipGetErrorString(e_));              \
            return PVQ_ERR_DEVICE;                                                                 \
        }                                                                                          \
    } while (0)

struct AbArgs {
    const float* db;   // [n_streams][n_frames][n_bins]
    int n_streams, n_frames, n_bins, bpo, octaves;
    float min_freq;
    float peak_prom, peak_h, bass_prom, bass_h;
    int highest_bassnote;
    float harm_thr;
    unsigned long long base_ms;
    int smooth_has;
    float calm_min, calm_max;
    unsigned long long note_ns, scene_ns, tuning_ns;
    unsigned long long frame_ns;
    const unsigned long long* frame_times;   // optional, [n_frames]
    const float* lnf;
    int dist, min_bin, radius;
    float *smoothed, *calm, *released, *afterglow, *peakfiltered, *pitch_acc, *pitch_dev, *scene, *tuning;   // state
    AnalysisBatchOutputs o;
    unsigned scratch_bytes;   // peaks_scratch_bytes(n_bins, dist)
    unsigned wave_bytes;      // LDS bytes per wave
};

namespace {
// the reference's libm calls: double evaluation, one rounding
__device__ __forceinline__ float ab_exp(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float ab_log2(float x) { return (float)log2((double)x); }
__device__ __forceinline__ float ab_log10(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float ab_pow(float b, float y) { return (float)pow((double)b, (double)y); }
// core::time::Duration::as_secs_f32 (analysis_host.hpp)
__device__ __forceinline__ float ab_secs(unsigned long long ns) {
    return (float)(ns / 1000000000ull) + (float)(ns % 1000000000ull) / 1000000000.0f;
}
__device__ __forceinline__ unsigned long long ab_trunc_u64(float x) {
    if (!(x > 0.0f)) return 0ull;
    if (x >= 18446744073709551616.0f) return ~0ull;
    return (unsigned long long)x;
}
__device__ __forceinline__ void ab_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
}  // namespace

template <int NK>
__global__ __launch_bounds__(256) void analysis_batch_preprocess(AbArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * 4 + wave;
    if (s >= a.n_streams) return;   // (no workgroup barrier below: waves are independent)
    const int n = a.n_bins, npad = (n + 63) / 64 * 64, words = (n + 31) / 32;
    unsigned char* base = ab_lds + (size_t)wave * a.wave_bytes;
    float* rowA = reinterpret_cast<float*>(base);               // the smoothed frame (find_peaks input), later pitch_accuracy
    float* rowB = rowA + npad;                                   // the raw frame, later pitch_deviation
    float* cw = rowB + npad;                                     // per-bin / per-peak terms of the sequential sums
    float* cs = cw + npad;
    float* pc_c = cs + npad;                                     // peaks_continuous of the frame: center, size (npad / 2 each)
    float* pc_s = pc_c + npad / 2;
    unsigned char* flag = reinterpret_cast<unsigned char*>(pc_s + npad / 2);   // is-peak / around-a-raw-peak flags
    unsigned char* scratch = flag + npad;                        // peaks_wave_nk's scratch
    const uint16_t* plist = reinterpret_cast<const uint16_t*>(scratch + npad);

    PeakParamsDev ps{};   // analysis.rs:332-349: bass config at or below highest_bassnote, general config above
    ps.n_bins = n; ps.bpo = a.bpo; ps.min_freq = a.min_freq; ps.lnf = a.lnf;
    ps.peak_min_prominence = a.peak_prom; ps.peak_min_height = a.peak_h;
    ps.bass_min_prominence = a.bass_prom; ps.bass_min_height = a.bass_h;
    ps.highest_bassnote = a.highest_bassnote; ps.harmonic_threshold = a.harm_thr;
    ps.dist = a.dist; ps.min_bin = a.min_bin;
    ps.mask = a.o.peak_mask ? a.o.peak_mask + (size_t)s * a.n_frames * words : nullptr;
    ps.count = a.o.peak_count ? a.o.peak_count + (size_t)s * a.n_frames : nullptr;
    ps.center = nullptr; ps.size = nullptr; ps.max_peaks = 0;
    PeakParamsDev pg = ps;   // calmness.rs:40: the general config on the whole raw frame
    pg.bass_min_prominence = a.peak_prom; pg.bass_min_height = a.peak_h;
    pg.mask = nullptr; pg.count = nullptr;

    float y_sm[NK], y_calm[NK], y_rel[NK], y_glow[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int bin = lane + 64 * k;
        const bool in = bin < n;
        y_sm[k] = in ? a.smoothed[(size_t)s * n + bin] : 0.0f;
        y_calm[k] = in ? a.calm[(size_t)s * n + bin] : 0.0f;
        y_rel[k] = in ? a.released[(size_t)s * n + bin] : 0.0f;
        y_glow[k] = in ? a.afterglow[(size_t)s * n + bin] : 0.0f;
    }
    float scene = a.scene[s], tuning = a.tuning[s];
    const float bpo_f = (float)a.bpo, n_f = (float)n;
    const float note_s = ab_secs(a.note_ns), scene_s = ab_secs(a.scene_ns), tuning_s = ab_secs(a.tuning_ns);
    float pf_last[NK], acc_last[NK], dev_last[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) pf_last[k] = acc_last[k] = dev_last[k] = 0.0f;

    for (int f = 0; f < a.n_frames; ++f) {
        const size_t fr = (size_t)s * a.n_frames + f;
        const unsigned long long dt_ns = a.frame_times ? a.frame_times[f] : a.frame_ns;
        const float dt_s = ab_secs(dt_ns);
        // ---- analysis.rs:295-323: per-bin EMA with a frequency- and calmness-dependent horizon
        const float cm = a.calm_min + (a.calm_max - a.calm_min) * scene;
        float xr[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) {
                const float x = a.db[fr * n + bin];
                xr[k] = x;
                if (!a.smooth_has) {
                    y_sm[k] = x;   // util.rs:117-120
                } else {
                    unsigned long long hms = 0;
                    if (a.base_ms > 0) {
                        const float octave_fraction = (float)bin / bpo_f / (float)a.octaves;
                        const float frequency_multiplier = 1.5f - 0.5f * octave_fraction;
                        hms = ab_trunc_u64((float)a.base_ms * frequency_multiplier * cm);
                    }
                    const float alpha = 1.0f - ab_exp(-2.0f * dt_s / ab_secs(hms * 1000000ull));
                    y_sm[k] = y_sm[k] + alpha * (x - y_sm[k]);
                }
                rowA[bin] = y_sm[k];
                rowB[bin] = x;
            } else {
                xr[k] = 0.0f;
            }
        }
        ab_wave_sync();
        // ---- analysis.rs:332-349: peaks of the smoothed frame (mask / count go straight to the outputs)
        uint32_t total = 0;
        peaks_wave_nk<NK>(rowA, scratch, (size_t)f, ps, lane, &total);
        ab_wave_sync();
        // ---- peak_detection.rs:61-148, :172-241: one lane per peak (ascending bins = ascending centres: a centre stays
        //      between its peak's neighbours and peaks are never adjacent)
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < npad) flag[lane + 64 * k] = 0;
        for (uint32_t idx = lane; idx < total; idx += 64) {
            const int p = plist[idx];
            float ctr, sz;
            if (p < 1 || p > n - 2) {
                ctr = (float)p;
                sz = rowA[p];
            } else {
                const float l0 = a.lnf[p - 1], l1 = a.lnf[p], l2 = a.lnf[p + 1];
                const float a0 = rowA[p - 1], a1 = rowA[p], a2 = rowA[p + 1];
                const float denom = (l0 - l1) * (l0 - l2) * (l1 - l2);
                if (fabsf(denom) < 1.1920929e-07f) {
                    ctr = (float)p;
                    sz = rowA[p];
                } else {
                    const float qa = (l2 * (a1 - a0) + l0 * (a2 - a1) + l1 * (a0 - a2)) / denom;
                    const float qb = ((l2 * l2) * (a0 - a1) + (l0 * l0) * (a1 - a2) + (l1 * l1) * (a2 - a0)) / denom;
                    const float lfp = (fabsf(qa) < 1.1920929e-07f) ? l1 : pk_clampf(-qb / (2.0f * qa), l0, l2);
                    const float f_peak = ab_exp(lfp);
                    const float center = bpo_f * ab_log2(f_peak / a.min_freq);
                    const float cc = pk_clampf(center, 0.0f, n_f - 1.0f);
                    const int lower = (int)floorf(cc);
                    const int upper = min(lower + 1, n - 1);
                    const float fract = cc - truncf(cc);
                    ctr = cc;
                    sz = fmaxf(rowA[lower] * (1.0f - fract) + rowA[upper] * fract, 0.0f);
                }
            }
            if (!(ctr > (float)a.highest_bassnote)) {   // promote_bass_peaks_with_harmonics
                const float f0 = a.min_freq * ab_pow(2.0f, ctr / bpo_f);
                const float p0 = ab_pow(10.0f, sz / 10.0f);
                float score = 0.0f;
                const float wts[4] = {0.5f, 0.3f, 0.15f, 0.05f};
#pragma unroll
                for (int h = 2; h <= 5; ++h) {
                    const float hf = f0 * (float)h;
                    if (hf >= a.min_freq) {
                        const float hb = (ab_log2(hf) - ab_log2(a.min_freq)) * bpo_f;
                        if (hb >= 0.0f && hb < n_f) {
                            const int lo = (int)floorf(hb);
                            const int hi = min((int)ceilf(hb), n - 1);
                            const float frac = hb - truncf(hb);
                            const float adb = (lo == hi) ? rowA[lo] : (rowA[lo] * (1.0f - frac) + rowA[hi] * frac);
                            const float hp = ab_pow(10.0f, adb / 10.0f);
                            if (hp > p0 * a.harm_thr) score += hp * wts[h - 2];
                        }
                    }
                }
                if (score > 0.0f) {
                    const float boost = fminf(1.0f + 0.5f * (score / fmaxf(p0, 1e-6f)), 1.5f);
                    sz += 10.0f * ab_log10(boost);
                }
            }
            pc_c[idx] = ctr;
            pc_s[idx] = sz;
            flag[p] = 1;
            if (a.o.center && idx < a.o.max_peaks) {
                a.o.center[fr * a.o.max_peaks + idx] = ctr;
                a.o.size[fr * a.o.max_peaks + idx] = sz;
            }
        }
        ab_wave_sync();
        // ---- afterglow.rs:27-36, :10-21
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) {
                pf_last[k] = flag[bin] ? y_sm[k] : 0.0f;
                float g = y_glow[k];
                g *= 0.85f - 0.15f * ((float)bin / n_f);
                if (g < y_sm[k]) g = y_sm[k];
                y_glow[k] = g;
                if (a.o.x_vqt_smoothed) a.o.x_vqt_smoothed[fr * n + bin] = y_sm[k];
                if (a.o.x_vqt_peakfiltered) a.o.x_vqt_peakfiltered[fr * n + bin] = pf_last[k];
                if (a.o.x_vqt_afterglow) a.o.x_vqt_afterglow[fr * n + bin] = g;
            }
        }
        // ---- pitch_analysis.rs:55-66: tuning-grid inaccuracy, power-weighted, accumulated in peak order
        for (uint32_t idx = lane; idx < total; idx += 64) {
            const float power = ab_pow(10.0f, pc_s[idx] / 10.0f);
            const float semis = pc_c[idx] * 12.0f / bpo_f;
            cw[idx] = fabsf(semis - roundf(semis)) * power;
            cs[idx] = power;
        }
        ab_wave_sync();
        {
            float inaccuracy_sum = 0.0f, power_sum = 0.0f;
            for (uint32_t idx = 0; idx < total; ++idx) {   // (every lane runs the same sequence on broadcast reads: a uniform result)
                power_sum += cs[idx];
                inaccuracy_sum += cw[idx];
            }
            const float avg = power_sum > 0.0f ? inaccuracy_sum / power_sum : 0.0f;
            const float alpha_t = 1.0f - ab_exp(-2.0f * dt_s / tuning_s);
            tuning = tuning + alpha_t * (100.0f * avg - tuning);
        }
        ab_wave_sync();
        // ---- calmness.rs:23-95: peaks of the RAW frame mark the bins "around a note"
        uint32_t total_raw = 0;
        peaks_wave_nk<NK>(rowB, scratch, (size_t)f, pg, lane, &total_raw);
        ab_wave_sync();
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < npad) flag[lane + 64 * k] = 0;
        ab_wave_sync();
        for (uint32_t idx = lane; idx < total_raw; idx += 64) {
            const int p = plist[idx];
            const int lo = max(0, p - a.radius), hi = min(n, p + a.radius);
            for (int i = lo; i < hi; ++i) flag[i] = 1;
        }
        ab_wave_sync();
        {
            const float alpha_c = 1.0f - ab_exp(-2.0f * dt_s / note_s);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int bin = lane + 64 * k;
                if (bin < n) {
                    float ws = 0.0f, w = 0.0f;
                    if (flag[bin]) {
                        y_calm[k] = y_calm[k] + alpha_c * (1.0f - y_calm[k]);
                        y_rel[k] = y_calm[k];
                        const float power = ab_pow(10.0f, y_sm[k] / 10.0f);
                        ws = y_calm[k] * power;
                        w = power;
                    } else {
                        y_calm[k] = y_calm[k] + alpha_c * (0.0f - y_calm[k]);
                        y_rel[k] = y_rel[k] + alpha_c * (0.0f - y_rel[k]);
                        if (y_rel[k] > 0.01f) {
                            w = y_rel[k] * 0.3f;
                            ws = y_rel[k] * w;
                        }
                    }
                    cw[bin] = ws;
                    cs[bin] = w;
                    if (a.o.calmness) a.o.calmness[fr * n + bin] = y_calm[k];
                }
            }
        }
        ab_wave_sync();
        {
            float weighted_sum = 0.0f, weight_sum = 0.0f;   // in bin order, as the reference's loop (adding a skipped bin's 0 changes nothing)
            for (int bin = 0; bin < n; ++bin) {
                weighted_sum += cw[bin];
                weight_sum += cs[bin];
            }
            if (weight_sum > 0.0f) {
                const float alpha_s = 1.0f - ab_exp(-2.0f * dt_s / scene_s);
                scene = scene + alpha_s * (weighted_sum / weight_sum - scene);
            }
        }
        ab_wave_sync();
        // ---- pitch_analysis.rs:12-42: per-bin accuracy / deviation at the peaks' nearest bins, later peaks overwrite earlier ones
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < npad) {
                rowA[lane + 64 * k] = 0.0f;
                rowB[lane + 64 * k] = 0.0f;
            }
        ab_wave_sync();
        if (lane == 0) {
            for (uint32_t idx = 0; idx < total; ++idx) {
                const float semis = pc_c[idx] * 12.0f / bpo_f;
                const float deviation = semis - roundf(semis);
                const float accuracy = fmaxf(1.0f - 2.0f * fabsf(deviation), 0.0f);
                const float rc = roundf(pc_c[idx]);
                if (rc >= 0.0f && rc < n_f) {
                    rowA[(int)rc] = accuracy;
                    rowB[(int)rc] = deviation;
                }
            }
        }
        ab_wave_sync();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) {
                acc_last[k] = rowA[bin];
                dev_last[k] = rowB[bin];
                if (a.o.pitch_accuracy) a.o.pitch_accuracy[fr * n + bin] = acc_last[k];
                if (a.o.pitch_deviation) a.o.pitch_deviation[fr * n + bin] = dev_last[k];
            }
        }
        if (lane == 0) {
            if (a.o.scene_calmness) a.o.scene_calmness[fr] = scene;
            if (a.o.tuning_grid_inaccuracy) a.o.tuning_grid_inaccuracy[fr] = tuning;
        }
        ab_wave_sync();
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int bin = lane + 64 * k;
        if (bin < n) {
            a.smoothed[(size_t)s * n + bin] = y_sm[k];
            a.calm[(size_t)s * n + bin] = y_calm[k];
            a.released[(size_t)s * n + bin] = y_rel[k];
            a.afterglow[(size_t)s * n + bin] = y_glow[k];
            if (a.n_frames > 0) {
                a.peakfiltered[(size_t)s * n + bin] = pf_last[k];
                a.pitch_acc[(size_t)s * n + bin] = acc_last[k];
                a.pitch_dev[(size_t)s * n + bin] = dev_last[k];
            }
        }
    }
    if (lane == 0) {
        a.scene[s] = scene;
        a.tuning[s] = tuning;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
AnalysisBatch::~AnalysisBatch() {
    if (device_id_ >= 0) (void)hipSetDevice(device_id_);
    for (float* p : {d_smoothed_, d_calm_, d_released_, d_afterglow_, d_peakfiltered_, d_pitch_acc_, d_pitch_dev_, d_scene_, d_tuning_, d_lnf_})
        if (p) (void)hipFree(p);
    if (d_times_) (void)hipFree(d_times_);
}

pvq_status AnalysisBatch::create(int device_id, const VqtRange& range, const FullAnalysisParameters& params, uint32_t n_streams,
                                 std::unique_ptr<AnalysisBatch>& out) {
    out.reset();
    if (device_id < 0) {
        set_last_error("the batched AnalysisState runs on a GPU; there is no CPU fallback (the host AnalysisState is the single-stream face)");
        return PVQ_ERR_NO_DEVICE;
    }
    const uint32_t n = range.n_buckets();
    if (!(range.min_freq > 0.0f) || range.octaves == 0 || range.buckets_per_octave == 0 || n_streams == 0) {
        set_last_error("invalid VqtRange or zero streams");
        return PVQ_ERR_INVALID_ARG;
    }
    if (n < 3 || n > 1024) {
        set_last_error("unsupported: the batched AnalysisState takes 3 .. 1024 bins");
        return PVQ_ERR_UNSUPPORTED;
    }
    PVQ_HIP(hipSetDevice(device_id));
    std::unique_ptr<AnalysisBatch> b(new AnalysisBatch());
    b->device_id_ = device_id;
    b->range_ = range;
    b->params_ = params;
    b->n_streams_ = n_streams;
    const size_t per = (size_t)n_streams * n * sizeof(float);
    for (float** p : {&b->d_smoothed_, &b->d_calm_, &b->d_released_, &b->d_afterglow_, &b->d_peakfiltered_, &b->d_pitch_acc_, &b->d_pitch_dev_}) {
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(p), per));
        PVQ_HIP(hipMemset(*p, 0, per));   // analysis.rs:192-241: every EMA starts at 0
    }
    for (float** p : {&b->d_scene_, &b->d_tuning_}) {
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(p), n_streams * sizeof(float)));
        PVQ_HIP(hipMemset(*p, 0, n_streams * sizeof(float)));
    }
    VqtParameters vp;
    vp.range = range;
    std::vector<float> lnf;
    bin_log_frequencies(vp, lnf);
    PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_lnf_), lnf.size() * sizeof(float)));
    PVQ_HIP(hipMemcpy(b->d_lnf_, lnf.data(), lnf.size() * sizeof(float), hipMemcpyHostToDevice));
    out = std::move(b);
    return PVQ_OK;
}

void AnalysisBatch::update_vqt_smoothing_duration(bool has_duration, Duration d) {
    params_.vqt_smoothing_duration_base = has_duration ? d : Duration::from_millis(0);
    smooth_has_ = has_duration;
}

pvq_status AnalysisBatch::preprocess_device(const float* d_db, size_t n_frames, Duration frame_time, const uint64_t* frame_times_ns,
                                            const AnalysisBatchOutputs& outs, hipStream_t stream) {
    if (n_frames == 0) return PVQ_OK;
    if (!d_db) {
        set_last_error("null frame pointer");
        return PVQ_ERR_INVALID_ARG;
    }
    if ((outs.center != nullptr) != (outs.size != nullptr) || (outs.center && outs.max_peaks == 0)) {
        set_last_error("center and size go together, with max_peaks > 0");
        return PVQ_ERR_INVALID_ARG;
    }
    if (n_frames > 0x7FFFFFFFull) {
        set_last_error("too many frames in one call");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    AbArgs a{};
    a.db = d_db;
    a.n_streams = (int)n_streams_;
    a.n_frames = (int)n_frames;
    a.n_bins = (int)range_.n_buckets();
    a.bpo = (int)range_.buckets_per_octave;
    a.octaves = (int)range_.octaves;
    a.min_freq = range_.min_freq;
    a.peak_prom = params_.peak_config.min_prominence;
    a.peak_h = params_.peak_config.min_height;
    a.bass_prom = params_.bassline_peak_config.min_prominence;
    a.bass_h = params_.bassline_peak_config.min_height;
    a.highest_bassnote = (int)params_.highest_bassnote;
    a.harm_thr = params_.harmonic_threshold;
    a.base_ms = params_.vqt_smoothing_duration_base.as_millis();
    a.smooth_has = smooth_has_ ? 1 : 0;
    a.calm_min = params_.vqt_smoothing_calmness_min;
    a.calm_max = params_.vqt_smoothing_calmness_max;
    a.note_ns = params_.note_calmness_smoothing_duration.ns;
    a.scene_ns = params_.scene_calmness_smoothing_duration.ns;
    a.tuning_ns = params_.tuning_inaccuracy_smoothing_duration.ns;
    a.frame_ns = frame_time.ns;
    a.frame_times = nullptr;
    if (frame_times_ns) {
        if (times_cap_ < n_frames) {
            if (d_times_) PVQ_HIP(hipFree(d_times_));
            d_times_ = nullptr;
            times_cap_ = 0;
            PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&d_times_), n_frames * sizeof(unsigned long long)));
            times_cap_ = n_frames;
        }
        PVQ_HIP(hipMemcpyAsync(d_times_, frame_times_ns, n_frames * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
        a.frame_times = d_times_;
    }
    a.lnf = d_lnf_;
    {   // peak_detection.rs:37, :45 and calmness.rs:37, as the batched peak kernels derive them
        const float dist_f = std::round((float)a.bpo * 0.4f / 12.0f);
        a.dist = dist_f > 0.0f ? (int)dist_f : 0;
        a.min_bin = ((a.bpo / 12) + 1) / 2;
        a.radius = a.bpo / 12 / 3;
    }
    a.smoothed = d_smoothed_; a.calm = d_calm_; a.released = d_released_; a.afterglow = d_afterglow_;
    a.peakfiltered = d_peakfiltered_; a.pitch_acc = d_pitch_acc_; a.pitch_dev = d_pitch_dev_;
    a.scene = d_scene_; a.tuning = d_tuning_;
    a.o = outs;
    const int npad = (a.n_bins + 63) / 64 * 64;
    a.scratch_bytes = (unsigned)peaks_scratch_bytes(a.n_bins, a.dist);
    a.wave_bytes = (unsigned)(sizeof(float) * (4 * npad + npad) + npad + a.scratch_bytes + 15) / 16 * 16;
    const dim3 grid((n_streams_ + 3) / 4);
    const size_t lds = (size_t)a.wave_bytes * 4;
    if (a.n_bins <= 256)
        hipLaunchKernelGGL(analysis_batch_preprocess<4>, grid, dim3(256), lds, stream, a);
    else if (a.n_bins <= 512) {
        PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(analysis_batch_preprocess<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(analysis_batch_preprocess<8>, grid, dim3(256), lds, stream, a);
    } else {
        PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(analysis_batch_preprocess<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(analysis_batch_preprocess<16>, grid, dim3(256), lds, stream, a);
    }
    PVQ_HIP(hipGetLastError());
    return PVQ_OK;
}

pvq_status AnalysisBatch::get_field(uint32_t stream_index, int field, float* out) {
    if (stream_index >= n_streams_ || !out) {
        set_last_error("stream index out of range or null output");
        return PVQ_ERR_INVALID_ARG;
    }
    const float* src = nullptr;
    switch (field) {
        case PVQ_FIELD_X_VQT_SMOOTHED: src = d_smoothed_; break;
        case PVQ_FIELD_X_VQT_PEAKFILTERED: src = d_peakfiltered_; break;
        case PVQ_FIELD_X_VQT_AFTERGLOW: src = d_afterglow_; break;
        case PVQ_FIELD_CALMNESS: src = d_calm_; break;
        case PVQ_FIELD_PITCH_ACCURACY: src = d_pitch_acc_; break;
        case PVQ_FIELD_PITCH_DEVIATION: src = d_pitch_dev_; break;
        default: set_last_error("unknown field"); return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    PVQ_HIP(hipDeviceSynchronize());
    const size_t n = range_.n_buckets();
    PVQ_HIP(hipMemcpy(out, src + (size_t)stream_index * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return PVQ_OK;
}

pvq_status AnalysisBatch::get_scalars(uint32_t stream_index, float* scene_calmness, float* tuning_grid_inaccuracy) {
    if (stream_index >= n_streams_) {
        set_last_error("stream index out of range");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    PVQ_HIP(hipDeviceSynchronize());
    if (scene_calmness) PVQ_HIP(hipMemcpy(scene_calmness, d_scene_ + stream_index, sizeof(float), hipMemcpyDeviceToHost));
    if (tuning_grid_inaccuracy) PVQ_HIP(hipMemcpy(tuning_grid_inaccuracy, d_tuning_ + stream_index, sizeof(float), hipMemcpyDeviceToHost));
    return PVQ_OK;
}

}  // namespace pvq
