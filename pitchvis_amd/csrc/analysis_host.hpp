// analysis_host.hpp — host-side AnalysisState.
//
// Mirrors pitchvis_analysis::analysis::AnalysisState (reference pitchvis_analysis/src/analysis.rs:119-410)
// and its modules (analysis_modules/{peak_detection,afterglow,calmness,pitch_analysis}.rs, util.rs:91-137).
// preprocess() is a recurrence over frames — bin EMAs, calmness EMAs and the scene calmness feed the
// next frame's smoothing horizons (analysis.rs:295-319, calmness.rs:23-95) — so it is sequential per
// stream and lives on the host, consuming dB frames the GPU produced.  The batch-parallel, stateless
// part of it (peaks of a frame with smoothing off) is the GPU peak kernel (peaks_device.hpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "vqt_host.hpp"

namespace pvq {

// std::time::Duration: integer nanoseconds
struct Duration {
    uint64_t ns = 0;
    static Duration from_millis(uint64_t ms) { return Duration{ms * 1000000ull}; }
    uint64_t as_millis() const { return ns / 1000000ull; }
    // core::time::Duration::as_secs_f32: (secs as f32) + (nanos as f32) / 1e9
    float as_secs_f32() const {
        return static_cast<float>(ns / 1000000000ull) + static_cast<float>(ns % 1000000000ull) / 1000000000.0f;
    }
};

// util.rs:108: alpha = 1 - exp(-2 timestep / time_horizon), in f32 with the host's expf
float ema_alpha(Duration timestep, Duration horizon);

// util.rs:91-137
class EmaMeasurement {
   public:
    EmaMeasurement() = default;
    EmaMeasurement(bool has_horizon, Duration horizon, float value) : has_(has_horizon), horizon_(horizon), y_(value) {}
    void update_with_timestep(float new_value, Duration timestep);
    void update_with_alpha(float new_value, float alpha) { y_ = y_ + alpha * (new_value - y_); }
    void set_time_horizon(bool has_horizon, Duration horizon) {
        has_ = has_horizon;
        horizon_ = horizon;
    }
    float get() const { return y_; }

   private:
    bool has_ = false;
    Duration horizon_{};
    float y_ = 0.0f;
};

// analysis_modules/peak_detection.rs:9-23
struct PeakDetectionParameters {
    float min_prominence;
    float min_height;
};
struct ContinuousPeak {
    float center;
    float size;
};

// analysis.rs:35-98
struct FullAnalysisParameters {
    uint32_t spectrogram_length = 400;
    PeakDetectionParameters peak_config{10.0f, 4.0f};
    PeakDetectionParameters bassline_peak_config{5.0f, 3.5f};
    uint32_t highest_bassnote = 12 * 2 + 4;
    Duration vqt_smoothing_duration_base = Duration::from_millis(70);
    float vqt_smoothing_calmness_min = 0.6f;
    float vqt_smoothing_calmness_max = 2.0f;
    Duration note_calmness_smoothing_duration = Duration::from_millis(3500);
    Duration scene_calmness_smoothing_duration = Duration::from_millis(800);
    Duration tuning_inaccuracy_smoothing_duration = Duration::from_millis(4000);
    float harmonic_threshold = 0.3f;
};

// peak_detection.rs:26-51 — ascending bin indices (the reference collects into a HashSet)
std::vector<uint32_t> find_peaks(const PeakDetectionParameters& cfg, const float* vqt, uint32_t n, uint32_t buckets_per_octave);
// peak_detection.rs:61-148 — sorted by center
std::vector<ContinuousPeak> enhance_peaks_continuous(const std::vector<uint32_t>& peaks, const float* vqt, const VqtRange& range);
// peak_detection.rs:172-241
void promote_bass_peaks_with_harmonics(std::vector<ContinuousPeak>& peaks, const float* vqt, const VqtRange& range,
                                       uint32_t highest_bassnote, float harmonic_threshold);

// analysis.rs:119-410
class AnalysisState {
   public:
    AnalysisState(const VqtRange& range, const FullAnalysisParameters& params);  // analysis.rs:192
    void update_vqt_smoothing_duration(bool has_duration, Duration d);           // analysis.rs:251
    bool preprocess(const float* x_vqt, size_t len, Duration frame_time);        // analysis.rs:288 (false: wrong length)
    float bin_to_frequency(uint32_t bin) const;                                  // analysis.rs:407

    FullAnalysisParameters params;
    VqtRange range;
    std::vector<EmaMeasurement> x_vqt_smoothed;
    std::vector<float> x_vqt_peakfiltered;
    std::vector<float> x_vqt_afterglow;
    std::vector<uint32_t> peaks;  // ascending
    std::vector<ContinuousPeak> peaks_continuous;
    std::vector<float> ml_midi_base_pitches;
    std::vector<EmaMeasurement> calmness;
    std::vector<float> pitch_accuracy;
    std::vector<float> pitch_deviation;
    EmaMeasurement smoothed_scene_calmness;
    EmaMeasurement smoothed_tuning_grid_inaccuracy;

   private:
    std::vector<EmaMeasurement> released_note_calmness_;
};

}  // namespace pvq
