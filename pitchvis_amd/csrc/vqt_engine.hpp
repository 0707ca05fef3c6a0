// vqt_engine.hpp — pvq::Vqt, the GPU-backed mirror of pitchvis_analysis::vqt::Vqt
// (reference pitchvis_analysis/src/vqt.rs:440-513, :866-916) plus the stateless per-frame peak
// pipeline of AnalysisState::preprocess (analysis.rs:332-361).  One instance owns its host plan,
// its device-resident kernel/twiddle tables and a grow-only device workspace; like the
// reference's `&mut self` it is exclusive to one caller at a time.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pvq.h"
#include "multi_host.hpp"
#include "vqt_host.hpp"

namespace pvq {

// Developer knobs are environment variables that only the DEVELOPER build of the library reads (-DPVQ_DEV_KNOBS: libpvq_dev.so,
// used by the tile-shape / fallback tests, the A/B scripts and the phase-stamp tools).  The product library never looks at the
// environment: a stray variable in a consumer's process cannot change what it computes.  Table of knobs: DESIGN.md.
#ifdef PVQ_DEV_KNOBS
inline int dev_knob(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
inline const char* dev_knob_str(const char* name) { return std::getenv(name); }
#else
inline int dev_knob(const char*, int dflt) { return dflt; }
inline const char* dev_knob_str(const char*) { return nullptr; }
#endif

// analysis.rs:72-98 (peak-related fields)
struct AnalysisParameters {
    float peak_min_prominence = 10.0f;
    float peak_min_height = 4.0f;
    float bass_min_prominence = 5.0f;
    float bass_min_height = 3.5f;
    uint32_t highest_bassnote = 28;
    float harmonic_threshold = 0.3f;
};

struct DeviceTables;   // opaque (device_tables.hpp)
struct PeakParamsDev;  // peaks_device.hpp

class Vqt {
   public:
    // Vqt::new, vqt.rs:465.  device_id < 0: host-only plan.
    static pvq_status create(const VqtParameters& p, int device_id, std::unique_ptr<Vqt>& out, VqtError& err);
    ~Vqt();

    const VqtParameters& params() const { return plan_.params; }  // vqt.rs:507
    const VqtKernel& kernel() const { return plan_.kernel; }      // vqt.rs:511
    const HostPlan& plan() const { return plan_; }
    double delay_seconds() const { return plan_.delay_seconds; }  // vqt.rs:449
    uint32_t n_bins() const { return plan_.params.range.n_buckets(); }
    bool has_device() const { return device_id_ >= 0; }
    int device() const { return device_id_; }

    // vqt.rs:866 (host pointers, synchronous)
    pvq_status calculate_vqt_instant_in_db(const float* x, size_t len, float* out_db);
    pvq_status calculate_batch_db(const float* pcm, size_t n_lead, size_t hop, size_t n_frames, float* out_db);
    // device pointers, asynchronous on `stream`
    pvq_status calculate_batch_db_device(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                         float* d_out_db, float* d_out_cplx, hipStream_t stream);
    // MANY streams in one call (the trainer's shape: many files side by side, pitchvis_train/src/train.rs:146-163; a stereo pair
    // as two streams).  Stream s: d_pcm[s] holds n_lead[s] + n_frames[s] * hop samples; its frame f goes to output row
    // s * out_stride_frames + f of d_out_db [n_streams][out_stride_frames][n_bins] and of the peak outputs; rows past a stream's
    // n_frames are zero frames (no peaks).  On the block-DFT path all streams share each stage's launch.  pk may be null.
    pvq_status batch_streams_device(const float* const* d_pcm, const size_t* n_lead, const size_t* n_frames, uint32_t n_streams, size_t hop,
                                    float* d_out_db, size_t out_stride_frames, const AnalysisParameters* a, uint32_t* d_peak_mask,
                                    uint32_t* d_peak_count, float* d_center, float* d_size, uint32_t max_peaks, hipStream_t stream);
    // the whole hot path, peaks fused into the frame kernels
    pvq_status vqt_analyze_batch_device(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                        const AnalysisParameters& a, float* d_out_db, uint32_t* d_peak_mask,
                                        uint32_t* d_peak_count, float* d_center, float* d_size, uint32_t max_peaks,
                                        hipStream_t stream);
    pvq_status analyze_batch_device(const float* d_db, size_t n_frames, const AnalysisParameters& a,
                                    uint32_t* d_peak_mask, uint32_t* d_peak_count, float* d_center,
                                    float* d_size, uint32_t max_peaks, hipStream_t stream);
    pvq_status analyze_batch(const float* db, size_t n_frames, const AnalysisParameters& a, uint32_t* peak_mask,
                             uint32_t* peak_count, float* center, float* size, uint32_t max_peaks);

    void set_algo(pvq_algo a) { algo_ = a; }
    pvq_algo algo() const { return algo_; }
    pvq_algo resolve_algo(size_t hop, size_t n_frames) const;   // the path a batch of this shape takes under the current setting
    // block-DFT GEMM arithmetic: exact fp32 MFMA, or the split-bf16 (3 x bf16, fp32 accumulate) form
    void set_gemm_split_bf16(bool on) { gemm_split_bf16_ = on; }
    void set_workspace_limit(size_t bytes) { workspace_limit_ = bytes ? bytes : ((size_t)1 << 30); }   // block-DFT spectrum workspace
    // rebuilds the device tables with every twiddle factor rounded to fp16 (or back to fp32)
    pvq_status set_twiddle_fp16(bool on);
    bool twiddle_fp16() const { return twiddle_fp16_; }
    bool gemm_split_bf16() const { return gemm_split_bf16_; }
    uint32_t blockdft_columns() const;
    pvq_algo last_algo() const { return last_algo_; }
    // HIP-event timing of every kernel launch (per slot) on the stream it is launched on.
    // Enabling resets the statistics; last_kernel_ms reports the mean per launch since then.
    void set_profiling(int mode);   // 0 off; 1 HIP events around every kernel launch; 2 only around the transform's main kernel (two events per step instead of eight)
    uint32_t last_kernel_ms(float* out, uint32_t cap);
    uint32_t last_kernel_launches(uint32_t* out, uint32_t cap) const;
    uint32_t last_frames_per_launch() const { return last_frames_per_launch_; }
    // flop issued by the matrix instructions of the last block-DFT GEMM launch (0 when the FFT path ran)
    double last_gemm_flop() const { return last_gemm_flop_; }
    // shader clock (MHz) measured inside the GEMM kernel's K loop during the last profiled launch (0: not measured)
    float last_sclk_mhz();
    // NaN / Inf policy.  The reference's callers never hand non-finite samples to the transform (the audio callback drops
    // such chunks, pitchvis_audio/src/audio_desktop.rs:102-105) and its peak stage would panic on the NaNs they cause
    // (peak_detection.rs:145).  Here the dB stages raise a sticky device flag when a frame's spectrum is not finite;
    // this call waits for `stream`, returns PVQ_ERR_NONFINITE_INPUT if the flag is up and clears it.  The synchronous
    // host-buffer entry points call it themselves.
    pvq_status input_status(hipStream_t stream);

    enum KernelSlot { SLOT_FFT_FRAMES = 0, SLOT_BLOCKDFT_GEMM = 1, SLOT_BLOCKDFT_COMBINE = 2, SLOT_BLOCKDFT_DOTS = 3, SLOT_PEAKS = 4, N_SLOTS = 5 };
    static const char* slot_name(uint32_t s);

   private:
    Vqt() = default;
    pvq_status upload_tables();
    pvq_status run_batch(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                         float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream);
    bool make_peak_params(const AnalysisParameters& ap, uint32_t* d_mask, uint32_t* d_count, float* d_center,
                          float* d_size, uint32_t max_peaks, PeakParamsDev& out) const;
    pvq_status launch_fft_path(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                               float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream);
    pvq_status launch_fft_streams(const void* st_table, size_t n_st, const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, size_t rows_total,
                                  float* d_out_db, float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream);   // st_table: FftStream[n_st] (vqt_engine.hip) or null
    bool blockdft_applicable(size_t hop) const;
    bool blockdft_takes_streams(size_t hop) const;
    // One run of frames for the block-DFT path: frame f' (of n_frames) ends at sample first_end + f' * hop of a buffer of n_samples
    // valid samples (zeros before it and after it) and goes to output row out_row0 + f' * row_step.  A stream of a many-streams call is
    // one run (first_end = n_lead + hop, row_step 1); a hop the path cannot take itself but whose r-fold it can (800 -> 1 600) is r
    // interleaved runs of hop r * hop, run i holding the frames i, i + r, ... (first_end = n_lead + (i + 1) hop, row_step r).
    // A run may also read a STAGED buffer that holds many short streams one behind the other, each in a slot of whole 64 r-frame tiles
    // followed by a gap that is the next stream's history (Vqt::batch_streams_device): then `slots` says which output rows the
    // run's frames are — frame t of the run is frame grid_i + row_step * t of the staged buffer — and out_row0 is unused.
    struct Slot { size_t vframe0, n_frames, out_row0; };   // frames [vframe0, vframe0 + n_frames) of the staged buffer -> rows out_row0 ...
    struct StreamIn {
        const float* d_pcm;
        size_t first_end, n_samples, n_frames, out_row0, row_step;
        const Slot* slots = nullptr;
        size_t n_slots = 0, grid_i = 0;
        uint64_t slot_hash = 0;
    };
    size_t blockdft_hop_factor(size_t hop) const;   // smallest r in {1, 2, 4, 8, 16} with blockdft_applicable(r * hop), 0 if none
    size_t auto_block_min_frames(size_t hop, size_t r) const;   // PVQ_ALGO_AUTO: the block-DFT path from this many frames on
    pvq_status launch_blockdft_streams(const StreamIn* st, size_t n_st, size_t hop, float* d_out_db, float* d_out_cplx, size_t rows_total,
                                       const PeakParamsDev* pk, hipStream_t stream);
    pvq_status prepare_blockdft(size_t hop);
    pvq_status launch_blockdft_path(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                    float* d_out_db, float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream);
    pvq_status ensure_workspace(void** ptr, size_t* cap, size_t bytes);
    pvq_status launch_peaks_kernel(const float* d_db, size_t n_frames, const PeakParamsDev& p, hipStream_t s);
    void slot_begin(int slot, hipStream_t s);
    // One handle's calls are ORDERED: the workspaces (X, the group-split rows, staging buffers, tile-list slots, the peak kernels' redo
    // flags) belong to the handle, so a call on another stream than the previous call's — the single-frame route's own stream after an
    // asynchronous batch on the caller's, a pvq_stream's stream, a second user stream — first makes its stream wait (on the device, the
    // host does not block) for everything the previous call queued.  Calls that stay on one stream pay nothing.
    pvq_status order_on(hipStream_t s);
    void slot_end(int slot, hipStream_t s);

    HostPlan plan_;
    int device_id_ = -1;
    DeviceTables* dev_ = nullptr;
    pvq_algo algo_ = PVQ_ALGO_AUTO;
    pvq_algo last_algo_ = PVQ_ALGO_AUTO;
    bool profiling_ = false, profiling_main_only_ = false;
    // host-buffer batches: upload / run / download streams and their events
    bool host_streams_ready_ = false;
    hipStream_t host_streams_[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> host_events_;
    // the single-frame call (calculate_vqt_instant_in_db): page-locked staging for the window union in and the dB values out, one stream
    hipStream_t inst_stream_ = nullptr;
    hipStream_t order_stream_ = nullptr;   // the stream of the handle's previous device call (order_on)
    bool order_valid_ = false;
    hipEvent_t order_ev_ = nullptr;
    float* inst_pin_ = nullptr;   // [window_union + n_bins]
    bool twiddle_fp16_ = false;
    size_t workspace_limit_ = (size_t)1 << 30;   // bytes of X (+ Y) a sub-batch of the block-DFT path may take
    bool gemm_split_bf16_ = false;  // default PVQ_GEMM_F32; PVQ_GEMM_BF16X3 meets the same parity bars and is ~10 % faster end to end
    uint32_t last_frames_per_launch_ = 0;
    double last_gemm_flop_ = 0.0;
    static constexpr int kMaxTimedLaunches = 512;
    std::vector<hipEvent_t> ev_[N_SLOTS][2];  // event pool, grown on demand
    int ev_count_[N_SLOTS] = {};              // launches recorded since profiling was enabled
    // grow-only workspaces for the host-pointer wrappers
    void* ws_pcm_ = nullptr;  size_t ws_pcm_cap_ = 0;
    void* ws_out_ = nullptr;  size_t ws_out_cap_ = 0;
    void* ws_misc_ = nullptr; size_t ws_misc_cap_ = 0;
    void* ws_flags_ = nullptr; size_t ws_flags_cap_ = 0;  // per-frame redo flags of the peak kernels
    void* ws_stage_ = nullptr; size_t ws_stage_cap_ = 0;  // many short streams staged one behind the other (batch_streams_device)
    void* ws_stage_tab_ = nullptr; size_t ws_stage_tab_cap_ = 0;
    void* ws_split_ = nullptr; size_t ws_split_cap_ = 0;   // x_vqt rows of a group-split launch of the FFT path (few frames)
    // the multi-device driver's per-handle shard buffers (PCM, dB, mask, count, center, size) and stream, grow-only
    void* multi_buf_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t multi_cap_[6] = {0, 0, 0, 0, 0, 0};
    hipStream_t multi_stream_ = nullptr;
   public:
    pvq_status multi_buffers(const size_t (&bytes)[6], void* (&out)[6], hipStream_t* st);   // (used by analyze_batch_multi's workers)
   private:
};

// One stream of host PCM analysed on several handles at once (one host thread per handle; handles may sit on different devices
// or share one): contiguous frame ranges with a halo of window_union - hop samples (multi_host.hpp), no collective, outputs written
// straight into the caller's host arrays.  Mirrors the only data-parallel driver of the reference — rayon's map_init with one Vqt
// per worker, pitchvis_train/src/train.rs:146-155 — for ONE long stream.  Any of the peak outputs may be null.
pvq_status analyze_batch_multi(Vqt* const* handles, uint32_t n_handles, const float* pcm, size_t n_lead, size_t hop, size_t n_frames,
                               const AnalysisParameters& a, float* out_db, uint32_t* peak_mask, uint32_t* peak_count, float* center,
                               float* size, uint32_t max_peaks);

// find_peaks over n_frames independent dB rows [n_frames][a.n_bins] (vqt_engine.hip; redo: n_frames bytes of device scratch)
pvq_status launch_peaks_frames(const float* d_db, size_t n_frames, const PeakParamsDev& a, uint8_t* redo, hipStream_t stream);

void set_last_error(const std::string& s);
void set_last_error_noexcept(const char* s) noexcept;   // for exception handlers: never throws (drops the text if it cannot be stored)
const char* get_last_error();

}  // namespace pvq
