// analysis_host.cpp — host-side AnalysisState (see analysis_host.hpp).  f32 arithmetic in the
// reference's operation order; compile with -ffp-contract=off.
#include "analysis_host.hpp"

#include <algorithm>
#include <cmath>

namespace pvq {

namespace {
uint32_t trunc_sat_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(x);
}
uint64_t trunc_sat_u64(float x) {
    if (!(x > 0.0f)) return 0ull;
    if (x >= 18446744073709551616.0f) return ~0ull;
    return static_cast<uint64_t>(x);
}
float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
}  // namespace

// util.rs:108: the EMA weight.  One definition for the host AnalysisState and for the weight tables the batched GPU form is
// handed (analysis_batch.hip): both get this translation unit's expf (g++, no fast-math, no FMA contraction).
float ema_alpha(Duration timestep, Duration horizon) {
    return 1.0f - std::exp(-2.0f * timestep.as_secs_f32() / horizon.as_secs_f32());
}

// util.rs:106-121
void EmaMeasurement::update_with_timestep(float new_value, Duration timestep) {
    if (has_) {
        const float alpha = ema_alpha(timestep, horizon_);
        update_with_alpha(new_value, alpha);
    } else {
        y_ = new_value;
    }
}

// peak_detection.rs:26-51.  find_peaks 0.1.5 PeakFinder (un-vendored, Cargo.lock:2949) restated with its
// documented scipy-like behaviour: plateau-aware strict local maxima, edges never peaks,
// middle_position() = (start+end)/2 of the half-open plateau range, inclusive height / prominence
// bounds, greedy min-distance suppression from the highest peak down.
std::vector<uint32_t> find_peaks(const PeakDetectionParameters& cfg, const float* x, uint32_t n, uint32_t bpo) {
    std::vector<uint32_t> mid;
    if (n < 3) return mid;
    const uint32_t i_max = n - 1;
    for (uint32_t i = 1; i < i_max; ++i) {
        if (x[i - 1] < x[i]) {
            uint32_t ia = i + 1;
            while (ia < i_max && x[ia] == x[i]) ++ia;
            if (x[ia] < x[i]) {
                mid.push_back((i + ia) / 2);
                i = ia;
            }
        }
    }
    std::vector<uint32_t> kept;
    for (uint32_t p : mid)
        if (x[p] >= cfg.min_height) kept.push_back(p);
    const uint32_t dist = trunc_sat_u32(std::round(static_cast<float>(bpo) * 0.4f / 12.0f));
    if (dist > 0 && kept.size() > 1) {
        const size_t np = kept.size();
        std::vector<size_t> order(np);
        for (size_t j = 0; j < np; ++j) order[j] = j;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return x[kept[a]] < x[kept[b]]; });
        std::vector<char> keep(np, 1);
        for (size_t a = np; a-- > 0;) {
            const size_t j = order[a];
            if (!keep[j]) continue;
            for (size_t b = j; b-- > 0 && kept[j] - kept[b] < dist;) keep[b] = 0;
            for (size_t b = j + 1; b < np && kept[b] - kept[j] < dist; ++b) keep[b] = 0;
        }
        std::vector<uint32_t> tmp;
        for (size_t j = 0; j < np; ++j)
            if (keep[j]) tmp.push_back(kept[j]);
        kept.swap(tmp);
    }
    std::vector<uint32_t> out;
    const uint32_t min_bin = ((bpo / 12) + 1) / 2;
    for (uint32_t pk : kept) {
        const float h = x[pk];
        float lmin = h, rmin = h;
        for (uint32_t a = pk; a-- > 0;) {
            if (x[a] > h) break;
            lmin = std::min(lmin, x[a]);
        }
        for (uint32_t a = pk + 1; a < n; ++a) {
            if (x[a] > h) break;
            rmin = std::min(rmin, x[a]);
        }
        if (h - std::max(lmin, rmin) >= cfg.min_prominence && pk >= min_bin) out.push_back(pk);
    }
    return out;
}

// peak_detection.rs:61-148
std::vector<ContinuousPeak> enhance_peaks_continuous(const std::vector<uint32_t>& peaks, const float* vqt, const VqtRange& range) {
    const uint32_t n = range.n_buckets();
    const float bpo = static_cast<float>(range.buckets_per_octave);
    std::vector<ContinuousPeak> out;
    out.reserve(peaks.size());
    for (uint32_t p : peaks) {
        if (p < 1 || p > n - 2) {
            out.push_back(ContinuousPeak{static_cast<float>(p), vqt[p]});
            continue;
        }
        const float f_prev = range.min_freq * std::pow(2.0f, static_cast<float>(p - 1) / bpo);
        const float f_curr = range.min_freq * std::pow(2.0f, static_cast<float>(p) / bpo);
        const float f_next = range.min_freq * std::pow(2.0f, static_cast<float>(p + 1) / bpo);
        const float l0 = std::log(f_prev), l1 = std::log(f_curr), l2 = std::log(f_next);
        const float a0 = vqt[p - 1], a1 = vqt[p], a2 = vqt[p + 1];
        const float denom = (l0 - l1) * (l0 - l2) * (l1 - l2);
        if (std::fabs(denom) < 1.1920929e-07f) {
            out.push_back(ContinuousPeak{static_cast<float>(p), vqt[p]});
            continue;
        }
        const float a = (l2 * (a1 - a0) + l0 * (a2 - a1) + l1 * (a0 - a2)) / denom;
        const float b = ((l2 * l2) * (a0 - a1) + (l0 * l0) * (a1 - a2) + (l1 * l1) * (a2 - a0)) / denom;
        const float lfp = (std::fabs(a) < 1.1920929e-07f) ? l1 : clampf(-b / (2.0f * a), l0, l2);
        const float f_peak = std::exp(lfp);
        const float center = bpo * std::log2(f_peak / range.min_freq);
        const float cc = clampf(center, 0.0f, static_cast<float>(n) - 1.0f);
        const uint32_t lower = trunc_sat_u32(std::floor(cc));
        const uint32_t upper = std::min(lower + 1, n - 1);
        const float fract = cc - std::trunc(cc);
        const float size = vqt[lower] * (1.0f - fract) + vqt[upper] * fract;
        out.push_back(ContinuousPeak{cc, std::max(size, 0.0f)});
    }
    std::stable_sort(out.begin(), out.end(), [](const ContinuousPeak& x, const ContinuousPeak& y) { return x.center < y.center; });
    return out;
}

// peak_detection.rs:172-241
void promote_bass_peaks_with_harmonics(std::vector<ContinuousPeak>& peaks, const float* vqt, const VqtRange& range,
                                       uint32_t highest_bassnote, float thr) {
    const uint32_t n = range.n_buckets();
    const float bpo = static_cast<float>(range.buckets_per_octave);
    static const float weights[4] = {0.5f, 0.3f, 0.15f, 0.05f};
    for (ContinuousPeak& pk : peaks) {
        if (pk.center > static_cast<float>(highest_bassnote)) continue;
        const float f0 = range.min_freq * std::pow(2.0f, pk.center / bpo);
        const float p0 = std::pow(10.0f, pk.size / 10.0f);
        float score = 0.0f;
        for (uint32_t h = 2; h <= 5; ++h) {
            const float hf = f0 * static_cast<float>(h);
            if (!(hf >= range.min_freq)) continue;
            const float hb = (std::log2(hf) - std::log2(range.min_freq)) * bpo;
            if (hb >= 0.0f && hb < static_cast<float>(n)) {
                const uint32_t lo = trunc_sat_u32(std::floor(hb));
                const uint32_t hi = std::min(trunc_sat_u32(std::ceil(hb)), n - 1);
                const float frac = hb - std::trunc(hb);
                const float adb = (lo == hi) ? vqt[lo] : (vqt[lo] * (1.0f - frac) + vqt[hi] * frac);
                const float hp = std::pow(10.0f, adb / 10.0f);
                if (hp > p0 * thr) score += hp * weights[h - 2];
            }
        }
        if (score > 0.0f) {
            const float boost = std::min(1.0f + 0.5f * (score / std::max(p0, 1e-6f)), 1.5f);
            pk.size += 10.0f * std::log10(boost);
        }
    }
}

// analysis.rs:192-241
AnalysisState::AnalysisState(const VqtRange& r, const FullAnalysisParameters& p)
    : params(p), range(r), smoothed_scene_calmness(true, p.scene_calmness_smoothing_duration, 0.0f),
      smoothed_tuning_grid_inaccuracy(true, p.tuning_inaccuracy_smoothing_duration, 0.0f) {
    const uint32_t n = range.n_buckets();
    x_vqt_smoothed.reserve(n);
    for (uint32_t bin = 0; bin < n; ++bin) {
        const float octave_fraction =
            static_cast<float>(bin) / static_cast<float>(range.buckets_per_octave) / static_cast<float>(range.octaves);
        const float frequency_multiplier = 1.5f - 0.5f * octave_fraction;
        const float duration_ms = static_cast<float>(params.vqt_smoothing_duration_base.as_millis()) * frequency_multiplier;
        x_vqt_smoothed.emplace_back(true, Duration::from_millis(trunc_sat_u64(duration_ms)), 0.0f);
    }
    x_vqt_peakfiltered.assign(n, 0.0f);
    x_vqt_afterglow.assign(n, 0.0f);
    ml_midi_base_pitches.assign(128, 0.0f);
    calmness.assign(n, EmaMeasurement(true, params.note_calmness_smoothing_duration, 0.0f));
    released_note_calmness_.assign(n, EmaMeasurement(true, params.note_calmness_smoothing_duration, 0.0f));
    pitch_accuracy.assign(n, 0.0f);
    pitch_deviation.assign(n, 0.0f);
}

// analysis.rs:251-270
void AnalysisState::update_vqt_smoothing_duration(bool has_duration, Duration d) {
    params.vqt_smoothing_duration_base = has_duration ? d : Duration::from_millis(0);
    for (size_t bin = 0; bin < x_vqt_smoothed.size(); ++bin) {
        if (has_duration) {
            const float octave_fraction =
                static_cast<float>(bin) / static_cast<float>(range.buckets_per_octave) / static_cast<float>(range.octaves);
            const float frequency_multiplier = 1.5f - 0.5f * octave_fraction;
            const float duration_ms = static_cast<float>(d.as_millis()) * frequency_multiplier;
            x_vqt_smoothed[bin].set_time_horizon(true, Duration::from_millis(trunc_sat_u64(duration_ms)));
        } else {
            x_vqt_smoothed[bin].set_time_horizon(false, Duration{});
        }
    }
}

float AnalysisState::bin_to_frequency(uint32_t bin) const {
    return range.min_freq * std::pow(2.0f, static_cast<float>(bin) / static_cast<float>(range.buckets_per_octave));
}

// analysis.rs:288-404
bool AnalysisState::preprocess(const float* x_vqt, size_t len, Duration frame_time) {
    const uint32_t n = range.n_buckets();
    if (len != n) return false;  // the reference asserts (analysis.rs:289)

    const float scene = smoothed_scene_calmness.get();
    const float calmness_multiplier =
        params.vqt_smoothing_calmness_min + (params.vqt_smoothing_calmness_max - params.vqt_smoothing_calmness_min) * scene;
    const uint64_t base_ms = params.vqt_smoothing_duration_base.as_millis();
    for (uint32_t bin = 0; bin < n; ++bin) {
        if (base_ms > 0) {
            const float octave_fraction =
                static_cast<float>(bin) / static_cast<float>(range.buckets_per_octave) / static_cast<float>(range.octaves);
            const float frequency_multiplier = 1.5f - 0.5f * octave_fraction;
            const float duration_ms = static_cast<float>(base_ms) * frequency_multiplier * calmness_multiplier;
            x_vqt_smoothed[bin].set_time_horizon(true, Duration::from_millis(trunc_sat_u64(duration_ms)));
        }
        x_vqt_smoothed[bin].update_with_timestep(x_vqt[bin], frame_time);
    }
    std::vector<float> sm(n);
    for (uint32_t bin = 0; bin < n; ++bin) sm[bin] = x_vqt_smoothed[bin].get();

    // peaks: bass config for p <= highest_bassnote, general config above (analysis.rs:332-349)
    std::vector<uint32_t> pk;
    for (uint32_t p : find_peaks(params.bassline_peak_config, sm.data(), n, range.buckets_per_octave))
        if (p <= params.highest_bassnote) pk.push_back(p);
    for (uint32_t p : find_peaks(params.peak_config, sm.data(), n, range.buckets_per_octave))
        if (p > params.highest_bassnote) pk.push_back(p);

    std::vector<ContinuousPeak> pc = enhance_peaks_continuous(pk, sm.data(), range);
    promote_bass_peaks_with_harmonics(pc, sm.data(), range, params.highest_bassnote, params.harmonic_threshold);

    // afterglow.rs:27-36, :10-21
    std::vector<char> is_peak(n, 0);
    for (uint32_t p : pk) is_peak[p] = 1;
    for (uint32_t i = 0; i < n; ++i) x_vqt_peakfiltered[i] = is_peak[i] ? sm[i] : 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        float& g = x_vqt_afterglow[i];
        g *= 0.85f - 0.15f * (static_cast<float>(i) / static_cast<float>(n));
        if (g < sm[i]) g = sm[i];
    }
    peaks = pk;
    peaks_continuous = pc;

    // calmness.rs:23-95 (peaks of the *raw* frame)
    {
        std::vector<char> around(n, 0);
        const int radius = static_cast<int>(range.buckets_per_octave / 12 / 3);
        for (uint32_t p : find_peaks(params.peak_config, x_vqt, n, range.buckets_per_octave)) {
            const int lo = std::max(0, static_cast<int>(p) - radius);
            const int hi = std::min(static_cast<int>(n), static_cast<int>(p) + radius);
            for (int i = lo; i < hi; ++i) around[i] = 1;
        }
        float weighted_sum = 0.0f, weight_sum = 0.0f;
        for (uint32_t bin = 0; bin < n; ++bin) {
            if (around[bin]) {
                calmness[bin].update_with_timestep(1.0f, frame_time);
                released_note_calmness_[bin] = calmness[bin];
                const float power = std::pow(10.0f, sm[bin] / 10.0f);
                weighted_sum += calmness[bin].get() * power;
                weight_sum += power;
            } else {
                calmness[bin].update_with_timestep(0.0f, frame_time);
                released_note_calmness_[bin].update_with_timestep(0.0f, frame_time);
                const float rel = released_note_calmness_[bin].get();
                if (rel > 0.01f) {
                    const float w = rel * 0.3f;
                    weighted_sum += rel * w;
                    weight_sum += w;
                }
            }
        }
        if (weight_sum > 0.0f) smoothed_scene_calmness.update_with_timestep(weighted_sum / weight_sum, frame_time);
    }

    // pitch_analysis.rs:48-75
    {
        float inaccuracy_sum = 0.0f, power_sum = 0.0f;
        for (const ContinuousPeak& p : peaks_continuous) {
            const float power = std::pow(10.0f, p.size / 10.0f);
            power_sum += power;
            const float semis = p.center * 12.0f / static_cast<float>(range.buckets_per_octave);
            inaccuracy_sum += std::fabs(semis - std::round(semis)) * power;
        }
        const float avg = power_sum > 0.0f ? inaccuracy_sum / power_sum : 0.0f;
        smoothed_tuning_grid_inaccuracy.update_with_timestep(100.0f * avg, frame_time);
    }
    // pitch_analysis.rs:12-42
    std::fill(pitch_accuracy.begin(), pitch_accuracy.end(), 0.0f);
    std::fill(pitch_deviation.begin(), pitch_deviation.end(), 0.0f);
    for (const ContinuousPeak& p : peaks_continuous) {
        const float semis = p.center * 12.0f / static_cast<float>(range.buckets_per_octave);
        const float deviation = semis - std::round(semis);
        const float accuracy = std::max(1.0f - 2.0f * std::fabs(deviation), 0.0f);
        const uint32_t bin = trunc_sat_u32(std::round(p.center));
        if (bin < n) {
            pitch_accuracy[bin] = accuracy;
            pitch_deviation[bin] = deviation;
        }
    }
    return true;
}

}  // namespace pvq
