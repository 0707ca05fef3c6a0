// capi.cpp — extern "C" surface of libpvq (include/pvq.h) over pvq::Vqt.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/pvq.h"
#include <cmath>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "analysis_batch.hpp"
#include "analysis_host.hpp"
#include "consumers_host.hpp"
#include "multi_host.hpp"
#include "vqt_engine.hpp"

struct pvq_vqt {
    std::unique_ptr<pvq::Vqt> impl;
};
struct pvq_analysis_state {
    std::unique_ptr<pvq::AnalysisState> impl;
};
struct pvq_analysis_batch {
    std::unique_ptr<pvq::AnalysisBatch> impl;
};
struct pvq_mono_agc {
    pvq::MonoAgc impl;
};
// device-resident ring: the newest buf_size samples are d_ring[w - buf_size, w); compacted when the linear
// buffer (4 x buf_size) runs out
struct pvq_stream {
    ~pvq_stream() {   // also runs when pvq_stream_create fails half way: no device buffer is leaked
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (d_ring) (void)hipFree(d_ring);
        if (d_db) (void)hipFree(d_db);
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
    }
    pvq_vqt* vqt = nullptr;
    size_t buf_size = 0, cap = 0, w = 0;
    float* d_ring = nullptr;
    float* d_db = nullptr;
    bool with_agc = false;
    pvq::MonoAgc agc{0.07f, 0.0001f};   // audio_desktop.rs:93
    float gain = 0.0f;                  // audio_desktop.rs:83
    float chunk_size_ms = 0.0f;
    // page-locked staging and a stream of the object's own: a push is one asynchronous DMA behind the conditioning on the host, a frame
    // the kernels + one asynchronous copy back + ONE wait (pageable buffers cost a staged, synchronous copy each way)
    hipStream_t stream = nullptr;
    float* h_in = nullptr;    // [buf_size]: the chunk being appended
    float* h_out = nullptr;   // [n_bins]
    bool in_flight = false;   // h_in is being read by a copy queued on `stream`
};

namespace {
pvq::VqtParameters to_cpp(const pvq_vqt_params& p) {
    pvq::VqtParameters q;
    q.sr = p.sr;
    q.n_fft = p.n_fft;
    q.range.min_freq = p.min_freq;
    q.range.octaves = p.octaves;
    q.range.buckets_per_octave = p.buckets_per_octave;
    q.sparsity_quantile = p.sparsity_quantile;
    q.quality = p.quality;
    q.gamma = p.gamma;
    return q;
}
pvq::AnalysisParameters to_cpp(const pvq_analysis_params* a) {
    pvq::AnalysisParameters q;
    if (a) {
        q.peak_min_prominence = a->peak_min_prominence;
        q.peak_min_height = a->peak_min_height;
        q.bass_min_prominence = a->bass_min_prominence;
        q.bass_min_height = a->bass_min_height;
        q.highest_bassnote = a->highest_bassnote;
        q.harmonic_threshold = a->harmonic_threshold;
    }
    return q;
}
pvq_status null_handle() {
    pvq::set_last_error("null handle");
    return PVQ_ERR_INVALID_ARG;
}
// include/pvq.h: "no exceptions cross the ABI".  Every extern "C" body below runs inside try { } catch (...) and lands here:
// a PanicError (the reference's assert! / expect texts, vqt.rs:785-792) becomes PVQ_ERR_INVALID_ARG, anything else
// (std::bad_alloc from a table or staging vector, std::length_error, ...) PVQ_ERR_INTERNAL; the text goes to pvq_last_error.
pvq_status translate_exception() noexcept {
    try {
        throw;
    } catch (const pvq::PanicError& e) {
        pvq::set_last_error_noexcept(e.what());
        return PVQ_ERR_INVALID_ARG;
    } catch (const std::bad_alloc&) {
        pvq::set_last_error_noexcept("out of host memory");
        return PVQ_ERR_INTERNAL;
    } catch (const std::exception& e) {
        pvq::set_last_error_noexcept(e.what());
        return PVQ_ERR_INTERNAL;
    } catch (...) {
        pvq::set_last_error_noexcept("unknown exception");
        return PVQ_ERR_INTERNAL;
    }
}
}  // namespace

extern "C" {

const char* pvq_status_string(pvq_status s) {
    try {
        switch (s) {
            case PVQ_OK: return "ok";
            case PVQ_ERR_ABOVE_NYQUIST: return "AboveNyquist";
            case PVQ_ERR_WINDOW_EXCEEDS_NFFT: return "WindowExceedsNFft";
            case PVQ_ERR_BAD_LENGTH: return "input must be exactly n_fft samples";
            case PVQ_ERR_INVALID_ARG: return "invalid argument";
            case PVQ_ERR_NO_DEVICE: return "handle has no GPU context (no CPU fallback)";
            case PVQ_ERR_DEVICE: return "HIP error";
            case PVQ_ERR_UNSUPPORTED: return "unsupported geometry";
            case PVQ_ERR_INTERNAL: return "internal error (out of memory or an unexpected exception)";
            case PVQ_ERR_NONFINITE_INPUT: return "non-finite sample in the input";
        }
        return "unknown";
    } catch (...) { (void)translate_exception(); return nullptr; }
}

const char* pvq_last_error(void) {
    try {
        return pvq::get_last_error();
    } catch (...) { (void)translate_exception(); return nullptr; }
}
uint32_t pvq_abi_version(void) {
    try {
        return PVQ_ABI_VERSION;
    } catch (...) { (void)translate_exception(); return 0; }
}

void pvq_vqt_default_params(pvq_vqt_params* p) {
    try {
        if (!p) return;
        const pvq::VqtParameters d;
        p->sr = d.sr;
        p->n_fft = d.n_fft;
        p->min_freq = d.range.min_freq;
        p->octaves = d.range.octaves;
        p->buckets_per_octave = d.range.buckets_per_octave;
        p->sparsity_quantile = d.sparsity_quantile;
        p->quality = d.quality;
        p->gamma = d.gamma;
    } catch (...) { (void)translate_exception(); }
}

pvq_status pvq_vqt_create(const pvq_vqt_params* params, int device_id, pvq_vqt** out, float err_detail[2]) {
    try {
        if (!params || !out) return null_handle();
        *out = nullptr;
        if (const char* t = pvq::dev_knob_str("PVQ_TEST_THROW")) {   // developer build only: test hook for the exception barrier (tests/test_capi_hardening.py)
            if (!std::strcmp(t, "bad_alloc")) throw std::bad_alloc();
            if (!std::strcmp(t, "length_error")) throw std::length_error("vector::_M_default_append");
            if (!std::strcmp(t, "int")) throw 42;
        }
        pvq::VqtError err;
        std::unique_ptr<pvq::Vqt> impl;
        const pvq_status st = pvq::Vqt::create(to_cpp(*params), device_id, impl, err);
        if (err_detail) {
            err_detail[0] = err.a;
            err_detail[1] = err.b;
        }
        if (st != PVQ_OK) return st;
        *out = new pvq_vqt{std::move(impl)};
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

void pvq_vqt_destroy(pvq_vqt* v) {
    try {
        delete v;
    } catch (...) { (void)translate_exception(); }
}

void pvq_vqt_get_params(const pvq_vqt* v, pvq_vqt_params* out) {
    try {
        if (!v || !out) return;
        const pvq::VqtParameters& d = v->impl->params();
        out->sr = d.sr;
        out->n_fft = d.n_fft;
        out->min_freq = d.range.min_freq;
        out->octaves = d.range.octaves;
        out->buckets_per_octave = d.range.buckets_per_octave;
        out->sparsity_quantile = d.sparsity_quantile;
        out->quality = d.quality;
        out->gamma = d.gamma;
    } catch (...) { (void)translate_exception(); }
}

uint32_t pvq_vqt_n_bins(const pvq_vqt* v) {
    try {
        return v ? v->impl->n_bins() : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
double pvq_vqt_delay_seconds(const pvq_vqt* v) {
    try {
        return v ? v->impl->delay_seconds() : 0.0;
    } catch (...) { (void)translate_exception(); return 0.0; }
}
uint32_t pvq_vqt_window_union(const pvq_vqt* v) {
    try {
        return v ? v->impl->plan().window_union : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
pvq_status pvq_vqt_bandwidths_3db(const pvq_vqt* v, float* lo_hz, float* hi_hz) {
    try {
        if (!v || !lo_hz || !hi_hz) return null_handle();
        const pvq::HostPlan& pl = v->impl->plan();
        std::copy(pl.bandwidth_lo_hz.begin(), pl.bandwidth_lo_hz.end(), lo_hz);
        std::copy(pl.bandwidth_hi_hz.begin(), pl.bandwidth_hi_hz.end(), hi_hz);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
uint32_t pvq_vqt_warning_count(const pvq_vqt* v) {
    try {
        return v ? (uint32_t)v->impl->plan().warnings.size() : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
pvq_status pvq_vqt_warning(const pvq_vqt* v, uint32_t i, char* buf, size_t cap) {
    try {
        if (!v || !buf || cap == 0) return null_handle();
        const pvq::HostPlan& pl = v->impl->plan();
        if (i >= pl.warnings.size()) {
            pvq::set_last_error("warning index out of range");
            return PVQ_ERR_INVALID_ARG;
        }
        const std::string& w = pl.warnings[i];
        const size_t nb = std::min(cap - 1, w.size());
        std::memcpy(buf, w.data(), nb);
        buf[nb] = '\0';
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
uint32_t pvq_vqt_n_groups(const pvq_vqt* v) {
    try {
        return v ? static_cast<uint32_t>(v->impl->kernel().window_groups.size()) : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}

pvq_status pvq_vqt_group_info(const pvq_vqt* v, uint32_t g, uint32_t info[5]) {
    try {
        if (!v || !info) return null_handle();
        const auto& groups = v->impl->kernel().window_groups;
        if (g >= groups.size()) {
            pvq::set_last_error("group index out of range");
            return PVQ_ERR_INVALID_ARG;
        }
        info[0] = groups[g].window_begin;
        info[1] = groups[g].window_end;
        info[2] = groups[g].filter_bank.rows;
        info[3] = groups[g].filter_bank.nnz();
        info[4] = groups[g].negative_filter_bank.nnz();
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_group_csr(const pvq_vqt* v, uint32_t g, int negative, uint32_t* row_ptr, uint32_t* col_idx,
                             float* values) {
    try {
        if (!v || !row_ptr || !col_idx || !values) return null_handle();
        const auto& groups = v->impl->kernel().window_groups;
        if (g >= groups.size()) {
            pvq::set_last_error("group index out of range");
            return PVQ_ERR_INVALID_ARG;
        }
        const pvq::CsrMatrix& m = negative ? groups[g].negative_filter_bank : groups[g].filter_bank;
        std::memcpy(row_ptr, m.row_ptr.data(), sizeof(uint32_t) * m.row_ptr.size());
        std::memcpy(col_idx, m.col_idx.data(), sizeof(uint32_t) * m.col_idx.size());
        std::memcpy(values, m.values.data(), sizeof(pvq::cf32) * m.values.size());
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_filter_params(const pvq_vqt* v, float* freq, float* window_length, uint32_t* factor,
                                 uint32_t* minwin) {
    try {
        if (!v) return null_handle();
        const auto& f = v->impl->plan().filters;
        for (size_t k = 0; k < f.size(); ++k) {
            if (freq) freq[k] = f[k].freq;
            if (window_length) window_length[k] = f[k].window_length;
            if (factor) factor[k] = f[k].sr_downscaling_factor;
            if (minwin) minwin[k] = f[k].minimum_needed_window_size;
        }
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_calculate_instant_db(pvq_vqt* v, const float* x, size_t len, float* out_db) {
    try {
        if (!v) return null_handle();
        if (!x || !out_db) {
            pvq::set_last_error("null pointer");
            return PVQ_ERR_INVALID_ARG;
        }
        return v->impl->calculate_vqt_instant_in_db(x, len, out_db);
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_calculate_batch_db(pvq_vqt* v, const float* pcm, size_t n_lead, size_t hop, size_t n_frames,
                                      float* out_db) {
    try {
        if (!v) return null_handle();
        return v->impl->calculate_batch_db(pcm, n_lead, hop, n_frames, out_db);
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_calculate_batch_db_device(pvq_vqt* v, const float* d_pcm, size_t n_lead, size_t hop,
                                             size_t n_frames, float* d_out_db, float* d_out_cplx, void* stream) {
    try {
        if (!v) return null_handle();
        return v->impl->calculate_batch_db_device(d_pcm, n_lead, hop, n_frames, d_out_db, d_out_cplx,
                                                  static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_set_algo(pvq_vqt* v, pvq_algo algo) {
    try {
        if (!v) return null_handle();
        if (algo != PVQ_ALGO_AUTO && algo != PVQ_ALGO_FFT && algo != PVQ_ALGO_BLOCKDFT) {
            pvq::set_last_error("unknown algorithm");
            return PVQ_ERR_INVALID_ARG;
        }
        v->impl->set_algo(algo);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
pvq_algo pvq_vqt_last_algo(const pvq_vqt* v) {
    try {
        return v ? v->impl->last_algo() : PVQ_ALGO_AUTO;
    } catch (...) { (void)translate_exception(); return PVQ_ALGO_AUTO; }
}
pvq_algo pvq_vqt_resolve_algo(pvq_vqt* v, size_t hop, size_t n_frames) {
    try {
        if (!v || hop == 0) return PVQ_ALGO_AUTO;
        return v->impl->resolve_algo(hop, n_frames);
    } catch (...) { (void)translate_exception(); return PVQ_ALGO_AUTO; }
}

pvq_status pvq_vqt_set_twiddle_fp16(pvq_vqt* v, int enable) {
    try {
        if (!v) return null_handle();
        return v->impl->set_twiddle_fp16(enable != 0);
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_set_workspace_limit(pvq_vqt* v, uint64_t bytes) {
    try {
        if (!v) return null_handle();
        v->impl->set_workspace_limit((size_t)bytes);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

uint32_t pvq_vqt_blockdft_columns(const pvq_vqt* v) {
    try {
        return v ? v->impl->blockdft_columns() : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}

pvq_status pvq_vqt_set_gemm_precision(pvq_vqt* v, pvq_gemm_precision p) {
    try {
        if (!v) return null_handle();
        if (p != PVQ_GEMM_F32 && p != PVQ_GEMM_BF16X3) {
            pvq::set_last_error("unknown GEMM precision");
            return PVQ_ERR_INVALID_ARG;
        }
        v->impl->set_gemm_split_bf16(p == PVQ_GEMM_BF16X3);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

void pvq_analysis_default_params(pvq_analysis_params* a) {
    try {
        if (!a) return;
        const pvq::AnalysisParameters d;
        a->peak_min_prominence = d.peak_min_prominence;
        a->peak_min_height = d.peak_min_height;
        a->bass_min_prominence = d.bass_min_prominence;
        a->bass_min_height = d.bass_min_height;
        a->highest_bassnote = d.highest_bassnote;
        a->harmonic_threshold = d.harmonic_threshold;
    } catch (...) { (void)translate_exception(); }
}

pvq_status pvq_analyze_batch_device(pvq_vqt* v, const float* d_db, size_t n_frames, const pvq_analysis_params* a,
                                    uint32_t* d_peak_mask, uint32_t* d_peak_count, float* d_center, float* d_size,
                                    uint32_t max_peaks, void* stream) {
    try {
        if (!v) return null_handle();
        return v->impl->analyze_batch_device(d_db, n_frames, to_cpp(a), d_peak_mask, d_peak_count, d_center, d_size,
                                             max_peaks, static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_analyze_batch(pvq_vqt* v, const float* db, size_t n_frames, const pvq_analysis_params* a,
                             uint32_t* peak_mask, uint32_t* peak_count, float* center, float* size,
                             uint32_t max_peaks) {
    try {
        if (!v) return null_handle();
        return v->impl->analyze_batch(db, n_frames, to_cpp(a), peak_mask, peak_count, center, size, max_peaks);
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_analyze_batch_device(pvq_vqt* v, const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                        const pvq_analysis_params* a, float* d_out_db, uint32_t* d_peak_mask,
                                        uint32_t* d_peak_count, float* d_center, float* d_size, uint32_t max_peaks,
                                        void* stream) {
    try {
        if (!v) return null_handle();
        return v->impl->vqt_analyze_batch_device(d_pcm, n_lead, hop, n_frames, to_cpp(a), d_out_db, d_peak_mask,
                                                 d_peak_count, d_center, d_size, max_peaks,
                                                 static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_calculate_batch_db_streams(pvq_vqt* v, const float* const* d_pcm, const size_t* n_lead, const size_t* n_frames,
                                              uint32_t n_streams, size_t hop, float* d_out_db, size_t out_stride_frames, void* stream) {
    try {
        if (!v) return null_handle();
        return v->impl->batch_streams_device(d_pcm, n_lead, n_frames, n_streams, hop, d_out_db, out_stride_frames, nullptr, nullptr, nullptr,
                                             nullptr, nullptr, 0, static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_analyze_batch_streams(pvq_vqt* v, const float* const* d_pcm, const size_t* n_lead, const size_t* n_frames,
                                         uint32_t n_streams, size_t hop, const pvq_analysis_params* a, float* d_out_db,
                                         size_t out_stride_frames, uint32_t* d_peak_mask, uint32_t* d_peak_count, float* d_center,
                                         float* d_size, uint32_t max_peaks, void* stream) {
    try {
        if (!v) return null_handle();
        const pvq::AnalysisParameters ap = to_cpp(a);
        return v->impl->batch_streams_device(d_pcm, n_lead, n_frames, n_streams, hop, d_out_db, out_stride_frames, &ap, d_peak_mask,
                                             d_peak_count, d_center, d_size, max_peaks, static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_plan_shard(uint64_t n_frames_total, uint64_t hop, uint64_t window_union, uint32_t rank, uint32_t world, pvq_shard* out) {
    try {
        pvq::ShardPlan sp;
        if (!out || !pvq::plan_shard(n_frames_total, hop, window_union, rank, world, &sp)) {
            pvq::set_last_error("pvq_plan_shard: null output or rank >= world");
            return PVQ_ERR_INVALID_ARG;
        }
        out->first_frame = sp.first_frame;
        out->n_frames = sp.n_frames;
        out->sample_begin = sp.sample_begin;
        out->sample_end = sp.sample_end;
        out->n_lead = sp.n_lead;
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_analyze_batch_multi(pvq_vqt* const* handles, uint32_t n_handles, const float* pcm, size_t n_lead, size_t hop,
                                       size_t n_frames, const pvq_analysis_params* a, float* out_db, uint32_t* peak_mask,
                                       uint32_t* peak_count, float* center, float* size, uint32_t max_peaks) {
    try {
        if (!handles || n_handles == 0) return null_handle();
        std::vector<pvq::Vqt*> hs(n_handles, nullptr);
        for (uint32_t i = 0; i < n_handles; ++i) {
            if (!handles[i]) return null_handle();
            hs[i] = handles[i]->impl.get();
        }
        return pvq::analyze_batch_multi(hs.data(), n_handles, pcm, n_lead, hop, n_frames, to_cpp(a), out_db, peak_mask, peak_count,
                                        center, size, max_peaks);
    } catch (...) { return translate_exception(); }
}

void pvq_analysis_full_default_params(pvq_analysis_full_params* p) {
    try {
        if (!p) return;
        const pvq::FullAnalysisParameters d;
        p->spectrogram_length = d.spectrogram_length;
        p->peak_min_prominence = d.peak_config.min_prominence;
        p->peak_min_height = d.peak_config.min_height;
        p->bass_min_prominence = d.bassline_peak_config.min_prominence;
        p->bass_min_height = d.bassline_peak_config.min_height;
        p->highest_bassnote = d.highest_bassnote;
        p->vqt_smoothing_duration_base_ns = d.vqt_smoothing_duration_base.ns;
        p->vqt_smoothing_calmness_min = d.vqt_smoothing_calmness_min;
        p->vqt_smoothing_calmness_max = d.vqt_smoothing_calmness_max;
        p->note_calmness_smoothing_duration_ns = d.note_calmness_smoothing_duration.ns;
        p->scene_calmness_smoothing_duration_ns = d.scene_calmness_smoothing_duration.ns;
        p->tuning_inaccuracy_smoothing_duration_ns = d.tuning_inaccuracy_smoothing_duration.ns;
        p->harmonic_threshold = d.harmonic_threshold;
    } catch (...) { (void)translate_exception(); }
}

static pvq::FullAnalysisParameters full_to_cpp(const pvq_analysis_full_params* params) {
    pvq::FullAnalysisParameters q;
    if (params) {
        q.spectrogram_length = params->spectrogram_length;
        q.peak_config = {params->peak_min_prominence, params->peak_min_height};
        q.bassline_peak_config = {params->bass_min_prominence, params->bass_min_height};
        q.highest_bassnote = params->highest_bassnote;
        q.vqt_smoothing_duration_base = pvq::Duration{params->vqt_smoothing_duration_base_ns};
        q.vqt_smoothing_calmness_min = params->vqt_smoothing_calmness_min;
        q.vqt_smoothing_calmness_max = params->vqt_smoothing_calmness_max;
        q.note_calmness_smoothing_duration = pvq::Duration{params->note_calmness_smoothing_duration_ns};
        q.scene_calmness_smoothing_duration = pvq::Duration{params->scene_calmness_smoothing_duration_ns};
        q.tuning_inaccuracy_smoothing_duration = pvq::Duration{params->tuning_inaccuracy_smoothing_duration_ns};
        q.harmonic_threshold = params->harmonic_threshold;
    }
    return q;
}

pvq_status pvq_analysis_state_create(float min_freq, uint32_t octaves, uint32_t buckets_per_octave,
                                     const pvq_analysis_full_params* params, pvq_analysis_state** out) {
    try {
        if (!out) return null_handle();
        *out = nullptr;
        if (!(min_freq > 0.0f) || octaves == 0 || buckets_per_octave == 0) {
            pvq::set_last_error("invalid VqtRange");
            return PVQ_ERR_INVALID_ARG;
        }
        const pvq::FullAnalysisParameters q = full_to_cpp(params);
        pvq::VqtRange r;
        r.min_freq = min_freq;
        r.octaves = octaves;
        r.buckets_per_octave = buckets_per_octave;
        *out = new pvq_analysis_state{std::unique_ptr<pvq::AnalysisState>(new pvq::AnalysisState(r, q))};
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

void pvq_analysis_state_destroy(pvq_analysis_state* s) {
    try {
        delete s;
    } catch (...) { (void)translate_exception(); }
}

pvq_status pvq_analysis_state_update_vqt_smoothing_duration(pvq_analysis_state* s, int has_duration, uint64_t duration_ns) {
    try {
        if (!s) return null_handle();
        s->impl->update_vqt_smoothing_duration(has_duration != 0, pvq::Duration{duration_ns});
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_analysis_state_preprocess(pvq_analysis_state* s, const float* x_vqt, size_t len, uint64_t frame_time_ns) {
    try {
        if (!s || !x_vqt) return null_handle();
        if (!s->impl->preprocess(x_vqt, len, pvq::Duration{frame_time_ns})) {
            pvq::set_last_error("x_vqt.len() == range.n_buckets()");
            return PVQ_ERR_BAD_LENGTH;
        }
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

float pvq_analysis_state_bin_to_frequency(const pvq_analysis_state* s, uint32_t bin) {
    try {
        return s ? s->impl->bin_to_frequency(bin) : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
uint32_t pvq_analysis_state_n_buckets(const pvq_analysis_state* s) {
    try {
        return s ? s->impl->range.n_buckets() : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}

pvq_status pvq_analysis_state_get_field(const pvq_analysis_state* s, pvq_analysis_field f, float* out) {
    try {
        if (!s || !out) return null_handle();
        const pvq::AnalysisState& a = *s->impl;
        const uint32_t n = a.range.n_buckets();
        switch (f) {
            case PVQ_FIELD_X_VQT_SMOOTHED:
                for (uint32_t i = 0; i < n; ++i) out[i] = a.x_vqt_smoothed[i].get();
                return PVQ_OK;
            case PVQ_FIELD_X_VQT_PEAKFILTERED: std::memcpy(out, a.x_vqt_peakfiltered.data(), n * sizeof(float)); return PVQ_OK;
            case PVQ_FIELD_X_VQT_AFTERGLOW: std::memcpy(out, a.x_vqt_afterglow.data(), n * sizeof(float)); return PVQ_OK;
            case PVQ_FIELD_CALMNESS:
                for (uint32_t i = 0; i < n; ++i) out[i] = a.calmness[i].get();
                return PVQ_OK;
            case PVQ_FIELD_PITCH_ACCURACY: std::memcpy(out, a.pitch_accuracy.data(), n * sizeof(float)); return PVQ_OK;
            case PVQ_FIELD_PITCH_DEVIATION: std::memcpy(out, a.pitch_deviation.data(), n * sizeof(float)); return PVQ_OK;
        }
        pvq::set_last_error("unknown field");
        return PVQ_ERR_INVALID_ARG;
    } catch (...) { return translate_exception(); }
}

uint32_t pvq_analysis_state_get_peaks(const pvq_analysis_state* s, uint32_t* out, uint32_t capacity) {
    try {
        if (!s) return 0;
        const auto& p = s->impl->peaks;
        for (uint32_t i = 0; i < p.size() && i < capacity && out; ++i) out[i] = p[i];
        return static_cast<uint32_t>(p.size());
    } catch (...) { (void)translate_exception(); return 0; }
}
uint32_t pvq_analysis_state_get_peaks_continuous(const pvq_analysis_state* s, float* center, float* size, uint32_t capacity) {
    try {
        if (!s) return 0;
        const auto& p = s->impl->peaks_continuous;
        for (uint32_t i = 0; i < p.size() && i < capacity; ++i) {
            if (center) center[i] = p[i].center;
            if (size) size[i] = p[i].size;
        }
        return static_cast<uint32_t>(p.size());
    } catch (...) { (void)translate_exception(); return 0; }
}
float pvq_analysis_state_scene_calmness(const pvq_analysis_state* s) {
    try {
        return s ? s->impl->smoothed_scene_calmness.get() : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
float pvq_analysis_state_tuning_grid_inaccuracy(const pvq_analysis_state* s) {
    try {
        return s ? s->impl->smoothed_tuning_grid_inaccuracy.get() : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}

pvq_status pvq_analysis_batch_create(int device_id, float min_freq, uint32_t octaves, uint32_t buckets_per_octave,
                                     const pvq_analysis_full_params* params, uint32_t n_streams, pvq_analysis_batch** out) {
    try {
        if (!out) return null_handle();
        *out = nullptr;
        pvq::VqtRange r;
        r.min_freq = min_freq;
        r.octaves = octaves;
        r.buckets_per_octave = buckets_per_octave;
        std::unique_ptr<pvq::AnalysisBatch> impl;
        const pvq_status st = pvq::AnalysisBatch::create(device_id, r, full_to_cpp(params), n_streams, impl);
        if (st != PVQ_OK) return st;
        *out = new pvq_analysis_batch{std::move(impl)};
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
void pvq_analysis_batch_destroy(pvq_analysis_batch* b) {
    try {
        delete b;
    } catch (...) { (void)translate_exception(); }
}
pvq_status pvq_analysis_batch_update_vqt_smoothing_duration(pvq_analysis_batch* b, int has_duration, uint64_t duration_ns) {
    try {
        if (!b) return null_handle();
        b->impl->update_vqt_smoothing_duration(has_duration != 0, pvq::Duration{duration_ns});
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_analysis_batch_preprocess_device(pvq_analysis_batch* b, const float* d_db, size_t n_frames, uint64_t frame_time_ns,
                                                const uint64_t* frame_times_ns, const pvq_analysis_batch_outputs* outs, void* stream) {
    try {
        if (!b) return null_handle();
        pvq::AnalysisBatchOutputs o;
        if (outs) {
            o.x_vqt_smoothed = outs->x_vqt_smoothed; o.x_vqt_peakfiltered = outs->x_vqt_peakfiltered; o.x_vqt_afterglow = outs->x_vqt_afterglow;
            o.calmness = outs->calmness; o.pitch_accuracy = outs->pitch_accuracy; o.pitch_deviation = outs->pitch_deviation;
            o.peak_mask = outs->peak_mask; o.peak_count = outs->peak_count; o.center = outs->center; o.size = outs->size;
            o.max_peaks = outs->max_peaks; o.scene_calmness = outs->scene_calmness; o.tuning_grid_inaccuracy = outs->tuning_grid_inaccuracy;
        }
        return b->impl->preprocess_device(d_db, n_frames, pvq::Duration{frame_time_ns}, frame_times_ns, o, static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_analysis_batch_preprocess_pcm(pvq_analysis_batch* b, pvq_vqt* v, const float* const* d_pcm, const size_t* n_lead, size_t n_frames,
                                             size_t hop, uint64_t frame_time_ns, float* d_db, const pvq_analysis_batch_outputs* outs, void* stream) {
    try {
        if (!b || !v) return null_handle();
        const pvq::VqtRange& r = v->impl->params().range;
        if (b->impl->device() != v->impl->device() || b->impl->n_bins() != v->impl->n_bins() || b->impl->range().min_freq != r.min_freq ||
            b->impl->range().buckets_per_octave != r.buckets_per_octave) {
            pvq::set_last_error("the transform handle and the analysis batch must sit on the same device and share the VqtRange");
            return PVQ_ERR_INVALID_ARG;
        }
        if (n_frames == 0) return PVQ_OK;
        const uint32_t ns = b->impl->n_streams();
        float* db = d_db;
        if (!db) {
            pvq_status st = b->impl->frames_buffer((size_t)ns * n_frames * b->impl->n_bins() * sizeof(float), &db);
            if (st != PVQ_OK) return st;
        }
        std::vector<size_t> nf(ns, n_frames);
        pvq_status st = v->impl->batch_streams_device(d_pcm, n_lead, nf.data(), ns, hop, db, n_frames, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                                      static_cast<hipStream_t>(stream));
        if (st != PVQ_OK) return st;
        pvq::AnalysisBatchOutputs o;
        if (outs) {
            o.x_vqt_smoothed = outs->x_vqt_smoothed; o.x_vqt_peakfiltered = outs->x_vqt_peakfiltered; o.x_vqt_afterglow = outs->x_vqt_afterglow;
            o.calmness = outs->calmness; o.pitch_accuracy = outs->pitch_accuracy; o.pitch_deviation = outs->pitch_deviation;
            o.peak_mask = outs->peak_mask; o.peak_count = outs->peak_count; o.center = outs->center; o.size = outs->size;
            o.max_peaks = outs->max_peaks; o.scene_calmness = outs->scene_calmness; o.tuning_grid_inaccuracy = outs->tuning_grid_inaccuracy;
        }
        return b->impl->preprocess_device(db, n_frames, pvq::Duration{frame_time_ns}, nullptr, o, static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_analysis_batch_get_field(pvq_analysis_batch* b, uint32_t stream_index, pvq_analysis_field f, float* out) {
    try {
        if (!b) return null_handle();
        return b->impl->get_field(stream_index, (int)f, out);
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_analysis_batch_get_scalars(pvq_analysis_batch* b, uint32_t stream_index, float* scene_calmness, float* tuning_grid_inaccuracy) {
    try {
        if (!b) return null_handle();
        return b->impl->get_scalars(stream_index, scene_calmness, tuning_grid_inaccuracy);
    } catch (...) { return translate_exception(); }
}

pvq_status pvq_vqt_set_profiling(pvq_vqt* v, int enable) {
    try {
        if (!v) return null_handle();
        v->impl->set_profiling(enable == 2 ? 2 : (enable != 0 ? 1 : 0));
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
uint32_t pvq_vqt_last_kernel_ms(pvq_vqt* v, float* out_ms, uint32_t capacity) {
    try {
        if (!v || !out_ms) return 0;
        return v->impl->last_kernel_ms(out_ms, capacity);
    } catch (...) { (void)translate_exception(); return 0; }
}
uint32_t pvq_vqt_last_kernel_launches(pvq_vqt* v, uint32_t* out_n, uint32_t capacity) {
    try {
        if (!v || !out_n) return 0;
        return v->impl->last_kernel_launches(out_n, capacity);
    } catch (...) { (void)translate_exception(); return 0; }
}
pvq_status pvq_vqt_input_status(pvq_vqt* v, void* stream) {
    try {
        if (!v) return null_handle();
        return v->impl->input_status(static_cast<hipStream_t>(stream));
    } catch (...) { return translate_exception(); }
}
double pvq_vqt_last_gemm_flop(const pvq_vqt* v) {
    try {
        return v ? v->impl->last_gemm_flop() : 0.0;
    } catch (...) { (void)translate_exception(); return 0.0; }
}
float pvq_vqt_last_sclk_mhz(pvq_vqt* v) {
    try {
        return v ? v->impl->last_sclk_mhz() : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
uint32_t pvq_vqt_last_frames_per_launch(const pvq_vqt* v) {
    try {
        return v ? v->impl->last_frames_per_launch() : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
const char* pvq_vqt_kernel_name(uint32_t slot) {
    try {
        return pvq::Vqt::slot_name(slot);
    } catch (...) { (void)translate_exception(); return nullptr; }
}

// ------------------------------------------------------------------------------------------------
// callers either side of the path
// ------------------------------------------------------------------------------------------------
pvq_status pvq_mono_agc_create(float desired_output_rms, float distortion_factor, pvq_mono_agc** out) {
    try {
        if (!out) return null_handle();
        *out = nullptr;
        std::string why;
        if (!pvq::MonoAgc::valid(desired_output_rms, distortion_factor, &why)) {
            pvq::set_last_error(why);
            return PVQ_ERR_INVALID_ARG;
        }
        *out = new pvq_mono_agc{pvq::MonoAgc(desired_output_rms, distortion_factor)};
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
void pvq_mono_agc_destroy(pvq_mono_agc* a) {
    try {
        delete a;
    } catch (...) { (void)translate_exception(); }
}
void pvq_mono_agc_freeze_gain(pvq_mono_agc* a, int freeze) {
    try {
        if (a) a->impl.freeze_gain(freeze != 0);
    } catch (...) { (void)translate_exception(); }
}
int pvq_mono_agc_is_gain_frozen(const pvq_mono_agc* a) {
    try {
        return a && a->impl.is_gain_frozen() ? 1 : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
float pvq_mono_agc_gain(const pvq_mono_agc* a) {
    try {
        return a ? a->impl.gain() : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
void pvq_mono_agc_process(pvq_mono_agc* a, float* samples, size_t n) {
    try {
        if (a && samples) a->impl.process(samples, n);
    } catch (...) { (void)translate_exception(); }
}

size_t pvq_train_chunk_samples(const pvq_vqt* v) {
    try {
        return v ? pvq::train_chunk_samples(v->impl->delay_seconds(), v->impl->params().sr) : 0;
    } catch (...) { (void)translate_exception(); return 0; }
}
pvq_status pvq_train_condition_stream(pvq_mono_agc* a, const float* left, const float* right, size_t n_chunks, size_t chunk,
                                      float* mono_out, float* gain_out) {
    try {
        if (!a || !left || !mono_out) return null_handle();
        pvq::train_condition_stream(a->impl, left, right, n_chunks, chunk, mono_out, gain_out);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_train_frames_db(pvq_vqt* v, const float* mono, size_t n_chunks, size_t chunk, size_t step, float* out_db) {
    try {
        if (!v || !mono || !out_db) return null_handle();
        if (chunk == 0 || step == 0) {
            pvq::set_last_error("chunk and step must be positive");
            return PVQ_ERR_INVALID_ARG;
        }
        const size_t n_frames = n_chunks / step;
        if (n_frames == 0) return PVQ_OK;
        return v->impl->calculate_batch_db(mono, 0, chunk * step, n_frames, out_db);
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_train_rows(const float* db, size_t n_frames, uint32_t n_bins, const uint32_t* voice_ptr, const int32_t* voice_key,
                          const float* voice_gain_left, const float* voice_gain_right, const float* agc_gain, float* out_rows) {
    try {
        if (!db || !voice_ptr || !agc_gain || !out_rows) return null_handle();
        std::string why;
        if (!pvq::train_rows(db, n_frames, n_bins, voice_ptr, voice_key, voice_gain_left, voice_gain_right, agc_gain, out_rows, &why)) {
            pvq::set_last_error(why);
            return PVQ_ERR_INVALID_ARG;
        }
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_npy_write_f32(const char* path, const float* data, uint64_t n) {
    try {
        if (!path || (!data && n)) return null_handle();
        std::string why;
        if (!pvq::npy_write_f32(path, data, n, &why)) {
            pvq::set_last_error(why);
            return PVQ_ERR_INVALID_ARG;
        }
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

#define PVQ_CAPI_HIP(call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            pvq::set_last_error(std::string(#call) + " failed: " + hipGetErrorString(e_));  \
            return PVQ_ERR_DEVICE;                                                          \
        }                                                                                   \
    } while (0)

pvq_status pvq_stream_create(pvq_vqt* v, size_t buf_size, int with_agc, pvq_stream** out) {
    try {
        if (!v || !out) return null_handle();
        *out = nullptr;
        const size_t n_fft = v->impl->params().n_fft;
        if (buf_size < n_fft) {
            pvq::set_last_error("ring buffer shorter than n_fft");
            return PVQ_ERR_BAD_LENGTH;
        }
        if (!v->impl->has_device()) {
            pvq::set_last_error("the streaming front end needs a GPU handle");
            return PVQ_ERR_NO_DEVICE;
        }
        auto s = std::make_unique<pvq_stream>();
        s->vqt = v;
        s->buf_size = buf_size;
        s->cap = 4 * buf_size;
        s->with_agc = with_agc != 0;
        PVQ_CAPI_HIP(hipSetDevice(v->impl->device()));
        PVQ_CAPI_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_ring), s->cap * sizeof(float)));
        PVQ_CAPI_HIP(hipMemset(s->d_ring, 0, s->cap * sizeof(float)));
        PVQ_CAPI_HIP(hipStreamSynchronize(nullptr));   // (the object's own stream does not wait for the null stream)
        PVQ_CAPI_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_db), v->impl->n_bins() * sizeof(float)));
        PVQ_CAPI_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        PVQ_CAPI_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_in), buf_size * sizeof(float), hipHostMallocDefault));
        PVQ_CAPI_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_out), v->impl->n_bins() * sizeof(float), hipHostMallocDefault));
        s->w = buf_size;
        *out = s.release();
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
void pvq_stream_destroy(pvq_stream* s) {
    try {
        delete s;
    } catch (...) { (void)translate_exception(); }
}
pvq_status pvq_stream_push(pvq_stream* s, const float* data, size_t n) {
    try {
        if (!s || (!data && n)) return null_handle();
        if (n == 0) return PVQ_OK;
        for (size_t i = 0; i < n; ++i)
            if (!std::isfinite(data[i])) return PVQ_OK;   // audio_desktop.rs:97-100: the chunk is dropped
        if (n > s->buf_size) {                            // Vec::drain(..n) would panic
            pvq::set_last_error("chunk longer than the ring buffer");
            return PVQ_ERR_BAD_LENGTH;
        }
        PVQ_CAPI_HIP(hipSetDevice(s->vqt->impl->device()));
        if (s->in_flight) {   // a second push before the previous one's copy was waited for (no frame in between)
            PVQ_CAPI_HIP(hipStreamSynchronize(s->stream));
            s->in_flight = false;
        }
        std::copy(data, data + n, s->h_in);
        if (s->with_agc) {
            float sq = 0.0f;
            for (size_t i = 0; i < n; ++i) sq += data[i] * data[i];     // audio_desktop.rs:101
            s->agc.freeze_gain(sq < 1e-6f);                             // :102
            s->agc.process(s->h_in, n);                                 // :111 (over the newest samples of the ring)
            s->gain = s->agc.gain();                                    // :112
        }
        if (s->w + n > s->cap) {   // compact: newest buf_size samples to the front (ranges do not overlap: cap = 4 buf_size)
            PVQ_CAPI_HIP(hipMemcpyAsync(s->d_ring, s->d_ring + s->w - s->buf_size, s->buf_size * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
            s->w = s->buf_size;
        }
        PVQ_CAPI_HIP(hipMemcpyAsync(s->d_ring + s->w, s->h_in, n * sizeof(float), hipMemcpyHostToDevice, s->stream));
        s->in_flight = true;
        s->w += n;
        s->chunk_size_ms = static_cast<float>(n) / s->vqt->impl->params().sr * 1000.0f;   // :118
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
float pvq_stream_gain(const pvq_stream* s) {
    try {
        return s ? s->gain : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
float pvq_stream_chunk_size_ms(const pvq_stream* s) {
    try {
        return s ? s->chunk_size_ms : 0.0f;
    } catch (...) { (void)translate_exception(); return 0.0f; }
}
pvq_status pvq_stream_frame_db(pvq_stream* s, float* out_db) {
    try {
        if (!s || !out_db) return null_handle();
        const size_t n_fft = s->vqt->impl->params().n_fft;
        // one frame over the newest n_fft samples: n_lead = n_fft - 1 samples of history + a hop of 1
        pvq_status st = s->vqt->impl->calculate_batch_db_device(s->d_ring + s->w - n_fft, n_fft - 1, 1, 1, s->d_db, nullptr, s->stream);
        if (st != PVQ_OK) return st;
        const size_t nb = s->vqt->impl->n_bins();
        PVQ_CAPI_HIP(hipMemcpyAsync(s->h_out, s->d_db, nb * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        PVQ_CAPI_HIP(hipStreamSynchronize(s->stream));
        s->in_flight = false;
        std::copy(s->h_out, s->h_out + nb, out_db);
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}
pvq_status pvq_stream_read(pvq_stream* s, float* out, size_t n_last) {
    try {
        if (!s || !out) return null_handle();
        if (n_last > s->buf_size) {
            pvq::set_last_error("n_last exceeds the ring buffer");
            return PVQ_ERR_BAD_LENGTH;
        }
        PVQ_CAPI_HIP(hipSetDevice(s->vqt->impl->device()));
        PVQ_CAPI_HIP(hipStreamSynchronize(s->stream));   // the appends queued on the object's stream
        s->in_flight = false;
        PVQ_CAPI_HIP(hipMemcpy(out, s->d_ring + s->w - n_last, n_last * sizeof(float), hipMemcpyDeviceToHost));
        return PVQ_OK;
    } catch (...) { return translate_exception(); }
}

void* pvq_host_alloc(size_t bytes) {
    try {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
        if (e != hipSuccess) {
            pvq::set_last_error(std::string("hipHostMalloc failed: ") + hipGetErrorString(e));
            return nullptr;
        }
        return p;
    } catch (...) { (void)translate_exception(); return nullptr; }
}
void pvq_host_free(void* p) {
    try {
        if (p) (void)hipHostFree(p);
    } catch (...) { (void)translate_exception(); }
}

void pvq_calculate_color(uint16_t buckets_per_octave, float bucket, const float* colors, float gray_level, float easing_pow,
                         float out_rgb[3]) {
    try {
        pvq::calculate_color(buckets_per_octave, bucket, reinterpret_cast<const float(*)[3]>(colors), gray_level, easing_pow, out_rgb);
    } catch (...) { (void)translate_exception(); }
}
size_t pvq_led_frame(uint32_t n_buckets, uint16_t buckets_per_octave, const float* center, const float* size, uint32_t n_peaks,
                     const float* colors, float gray_level, float easing_pow, uint8_t* out) {
    try {
        return pvq::led_frame(n_buckets, buckets_per_octave, center, size, n_peaks, reinterpret_cast<const float(*)[3]>(colors),
                              gray_level, easing_pow, out);
    } catch (...) { (void)translate_exception(); return 0; }
}

}  // extern "C"
