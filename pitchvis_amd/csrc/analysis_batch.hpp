// analysis_batch.hpp — AnalysisState::preprocess for MANY streams on the GPU (the "one wave per stream" half).
//
// preprocess (pitchvis_analysis/src/analysis.rs:288-404) is a recurrence over the frames of ONE stream: the per-bin EMAs, the
// calmness EMAs and the scene calmness of frame t set the smoothing horizons of frame t + 1 (analysis.rs:295-319,
// analysis_modules/calmness.rs:23-95), so a stream cannot be split over time.  Streams are independent of each other, though —
// the reference's trainer analyses many files at once (pitchvis_train/src/train.rs:146-155) — so here one WAVEFRONT owns one
// stream and walks its frames in order with the whole state in registers / LDS, and thousands of streams run side by side
// ("replicas only": no exchange between streams, none between devices).  The host AnalysisState (analysis_host.cpp) stays the
// single-stream face; this is the batch face of the same arithmetic, in the same f32 operation order.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/pvq.h"
#include "analysis_host.hpp"

namespace pvq {

// per-frame outputs, device pointers, any may be null.  Layouts: [stream][frame][n_bins] for the per-bin fields,
// [stream][frame][ceil(n_bins/32)] for the mask, [stream][frame][max_peaks] for center / size, [stream][frame] for the scalars.
struct AnalysisBatchOutputs {
    float* x_vqt_smoothed = nullptr;
    float* x_vqt_peakfiltered = nullptr;
    float* x_vqt_afterglow = nullptr;
    float* calmness = nullptr;
    float* pitch_accuracy = nullptr;
    float* pitch_deviation = nullptr;
    uint32_t* peak_mask = nullptr;
    uint32_t* peak_count = nullptr;
    float* center = nullptr;
    float* size = nullptr;
    uint32_t max_peaks = 0;
    float* scene_calmness = nullptr;
    float* tuning_grid_inaccuracy = nullptr;
};

class AnalysisBatch {
   public:
    static pvq_status create(int device_id, const VqtRange& range, const FullAnalysisParameters& params, uint32_t n_streams,
                             std::unique_ptr<AnalysisBatch>& out);
    ~AnalysisBatch();
    uint32_t n_streams() const { return n_streams_; }
    uint32_t n_bins() const { return range_.n_buckets(); }
    const VqtRange& range() const { return range_; }
    // a grow-only device buffer of the object for the dB frames between the transform and preprocess (pvq_analysis_batch_preprocess_pcm)
    pvq_status frames_buffer(size_t bytes, float** out);
    int device() const { return device_id_; }
    // analysis.rs:251-270, for every stream
    void update_vqt_smoothing_duration(bool has_duration, Duration d);
    // analysis.rs:288 for n_frames frames of every stream: d_db [n_streams][n_frames][n_bins] (device).  frame_time: the same for
    // every frame, or — frame_times_ns != null — one per frame (host array of n_frames).  Asynchronous on `stream`.
    pvq_status preprocess_device(const float* d_db, size_t n_frames, Duration frame_time, const uint64_t* frame_times_ns,
                                 const AnalysisBatchOutputs& outs, hipStream_t stream);
    // the state of one stream after the last call (host copies; synchronises): field as pvq_analysis_field, out [n_bins]
    pvq_status get_field(uint32_t stream_index, int field, float* out);
    pvq_status get_scalars(uint32_t stream_index, float* scene_calmness, float* tuning_grid_inaccuracy);

   private:
    AnalysisBatch() = default;
    int device_id_ = -1;
    VqtRange range_{};
    FullAnalysisParameters params_{};
    bool smooth_has_ = true;   // x_vqt_smoothed[..] has a time horizon (update_vqt_smoothing_duration(None) clears it)
    uint32_t n_streams_ = 0;
    // device state: [n_streams][n_bins] each, then [n_streams]
    float *d_smoothed_ = nullptr, *d_calm_ = nullptr, *d_released_ = nullptr, *d_afterglow_ = nullptr, *d_peakfiltered_ = nullptr;
    float *d_pitch_acc_ = nullptr, *d_pitch_dev_ = nullptr;
    float *d_scene_ = nullptr, *d_tuning_ = nullptr;
    float* d_lnf_ = nullptr;           // ln(f_k), host libm (peak_detection.rs:81-86)
    unsigned long long* d_times_ = nullptr;   // per-frame times of the running call
    size_t times_cap_ = 0;
    void* d_tab_ = nullptr;            // EMA weights of the running call (host libm), then the frames' row indices
    size_t tab_cap_ = 0;
    void* d_frames_ = nullptr;         // dB frames of pvq_analysis_batch_preprocess_pcm
    size_t frames_cap_ = 0;
    void* d_raw_ = nullptr;            // peak masks of the raw frames of the running call (frame-parallel pre-pass), then its scratch flags
    size_t raw_cap_ = 0;
    std::vector<float> tab_host_;      // what d_tab_ holds (constant frame time: reused by the next call without an upload)
};

}  // namespace pvq
