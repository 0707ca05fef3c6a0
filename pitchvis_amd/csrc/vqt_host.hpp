// vqt_host.hpp — host side of libpvq: parameter types and construction of the multi-rate sparse
// VQT kernel.  Mirrors the public surface of pitchvis_analysis::vqt (reference file
// pitchvis_analysis/src/vqt.rs): VqtRange / VqtParameters / VqtError / WindowGroup / VqtKernel.
// Construction runs once per parameter set on one CPU thread (as in the reference); everything
// per-frame runs on the GPU (vqt_engine.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace pvq {

// Where the reference panics (assert! / expect in vqt.rs:785-792) the host code throws this; it never leaves the
// library: the extern "C" layer turns it into PVQ_ERR_INVALID_ARG with the reference's panic text (capi.cpp).
struct PanicError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct cf32 {
    float re, im;
};

// vqt.rs:238-262
struct VqtRange {
    float min_freq = 55.0f;
    uint32_t octaves = 7;
    uint32_t buckets_per_octave = 84;
    uint32_t n_buckets() const { return buckets_per_octave * octaves; }
};

// vqt.rs:278-348 (Default: vqt.rs:180-214)
struct VqtParameters {
    float sr = 22050.0f;
    uint32_t n_fft = 2 * 16384;
    VqtRange range{};
    float sparsity_quantile = 0.999f;
    float quality = 1.6f;
    float gamma = 4.8f * 1.6f;
};

// vqt.rs:350-366
struct VqtError {
    enum Kind { None = 0, AboveNyquist = 1, WindowExceedsNFft = 2 } kind = None;
    float a = 0.0f;  // highest_frequency | window_length
    float b = 0.0f;  // nyquist_frequency | n_fft
    std::string to_string() const;  // the reference's Display text
};

// vqt.rs:370-384
struct FilterParams {
    float freq;
    float window_length;
    uint32_t sr_downscaling_factor;
    uint32_t minimum_needed_window_size;
};

// sprs::CsMat<Complex32> as plain arrays
struct CsrMatrix {
    uint32_t rows = 0, cols = 0;
    std::vector<uint32_t> row_ptr;  // rows + 1
    std::vector<uint32_t> col_idx;
    std::vector<cf32> values;
    uint32_t nnz() const { return static_cast<uint32_t>(col_idx.size()); }
};

// vqt.rs:388-410
struct WindowGroup {
    uint32_t window_begin = 0, window_end = 0;
    CsrMatrix filter_bank;
    CsrMatrix negative_filter_bank;  // nnz()==0 <=> None
    uint32_t first_bin = 0;          // row offset into the output vector
    uint32_t window_size() const { return window_end - window_begin; }
};

// vqt.rs:413-415
struct VqtKernel {
    std::vector<WindowGroup> window_groups;
};

struct HostPlan {
    VqtParameters params;
    std::vector<FilterParams> filters;
    VqtKernel kernel;
    float window_center = 0.0f;
    double delay_seconds = 0.0;  // vqt.rs:756
    uint32_t window_union = 0;   // n_fft - min(window_begin)
    // Filter::bandwidth_3db_in_hz (vqt.rs:421, :817-818): the -3 dB band of every filter, read off its decimated frequency response
    // (find_3db_points / calculate_bandwidth, vqt.rs:956-989: "a very crude approximation": the response only spans a few buckets)
    std::vector<float> bandwidth_lo_hz, bandwidth_hi_hz;
    // the kernel construction's warn!() lines (vqt.rs:695-709: coverage gaps between neighbouring filters), in bin order
    std::vector<std::string> warnings;
};

// ln(f_k) per bin exactly as enhance_peaks_continuous evaluates it (peak_detection.rs:81-86):
// ln(min_freq * 2^(k/bpo)) in f32 with the host libm, so the GPU refinement starts from the
// same three abscissae as the CPU reference.
void bin_log_frequencies(const VqtParameters& p, std::vector<float>& lnf);

// vqt.rs:517-587
VqtError filter_bank_params(const VqtParameters& p, std::vector<FilterParams>& out);
// vqt.rs:599-759 (+ :769-852 per filter)
VqtError build_plan(const VqtParameters& p, HostPlan& out);

}  // namespace pvq
