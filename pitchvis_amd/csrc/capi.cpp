// capi.cpp — extern "C" surface of libpvq (include/pvq.h) over pvq::Vqt.
#include <cstring>
#include <memory>

#include "../../include/pvq.h"
#include "vqt_engine.hpp"

struct pvq_vqt {
    std::unique_ptr<pvq::Vqt> impl;
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
}  // namespace

extern "C" {

const char* pvq_status_string(pvq_status s) {
    switch (s) {
        case PVQ_OK: return "ok";
        case PVQ_ERR_ABOVE_NYQUIST: return "AboveNyquist";
        case PVQ_ERR_WINDOW_EXCEEDS_NFFT: return "WindowExceedsNFft";
        case PVQ_ERR_BAD_LENGTH: return "input must be exactly n_fft samples";
        case PVQ_ERR_INVALID_ARG: return "invalid argument";
        case PVQ_ERR_NO_DEVICE: return "handle has no GPU context (no CPU fallback)";
        case PVQ_ERR_DEVICE: return "HIP error";
        case PVQ_ERR_UNSUPPORTED: return "unsupported geometry";
    }
    return "unknown";
}

const char* pvq_last_error(void) { return pvq::get_last_error(); }
uint32_t pvq_abi_version(void) { return PVQ_ABI_VERSION; }

void pvq_vqt_default_params(pvq_vqt_params* p) {
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
}

pvq_status pvq_vqt_create(const pvq_vqt_params* params, int device_id, pvq_vqt** out, float err_detail[2]) {
    if (!params || !out) return null_handle();
    *out = nullptr;
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
}

void pvq_vqt_destroy(pvq_vqt* v) { delete v; }

void pvq_vqt_get_params(const pvq_vqt* v, pvq_vqt_params* out) {
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
}

uint32_t pvq_vqt_n_bins(const pvq_vqt* v) { return v ? v->impl->n_bins() : 0; }
double pvq_vqt_delay_seconds(const pvq_vqt* v) { return v ? v->impl->delay_seconds() : 0.0; }
uint32_t pvq_vqt_window_union(const pvq_vqt* v) { return v ? v->impl->plan().window_union : 0; }
uint32_t pvq_vqt_n_groups(const pvq_vqt* v) {
    return v ? static_cast<uint32_t>(v->impl->kernel().window_groups.size()) : 0;
}

pvq_status pvq_vqt_group_info(const pvq_vqt* v, uint32_t g, uint32_t info[5]) {
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
}

pvq_status pvq_vqt_group_csr(const pvq_vqt* v, uint32_t g, int negative, uint32_t* row_ptr, uint32_t* col_idx,
                             float* values) {
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
}

pvq_status pvq_vqt_filter_params(const pvq_vqt* v, float* freq, float* window_length, uint32_t* factor,
                                 uint32_t* minwin) {
    if (!v) return null_handle();
    const auto& f = v->impl->plan().filters;
    for (size_t k = 0; k < f.size(); ++k) {
        if (freq) freq[k] = f[k].freq;
        if (window_length) window_length[k] = f[k].window_length;
        if (factor) factor[k] = f[k].sr_downscaling_factor;
        if (minwin) minwin[k] = f[k].minimum_needed_window_size;
    }
    return PVQ_OK;
}

pvq_status pvq_vqt_calculate_instant_db(pvq_vqt* v, const float* x, size_t len, float* out_db) {
    if (!v) return null_handle();
    if (!x || !out_db) {
        pvq::set_last_error("null pointer");
        return PVQ_ERR_INVALID_ARG;
    }
    return v->impl->calculate_vqt_instant_in_db(x, len, out_db);
}

pvq_status pvq_vqt_calculate_batch_db(pvq_vqt* v, const float* pcm, size_t n_lead, size_t hop, size_t n_frames,
                                      float* out_db) {
    if (!v) return null_handle();
    return v->impl->calculate_batch_db(pcm, n_lead, hop, n_frames, out_db);
}

pvq_status pvq_vqt_calculate_batch_db_device(pvq_vqt* v, const float* d_pcm, size_t n_lead, size_t hop,
                                             size_t n_frames, float* d_out_db, float* d_out_cplx, void* stream) {
    if (!v) return null_handle();
    return v->impl->calculate_batch_db_device(d_pcm, n_lead, hop, n_frames, d_out_db, d_out_cplx,
                                              static_cast<hipStream_t>(stream));
}

pvq_status pvq_vqt_set_algo(pvq_vqt* v, pvq_algo algo) {
    if (!v) return null_handle();
    if (algo != PVQ_ALGO_AUTO && algo != PVQ_ALGO_FFT && algo != PVQ_ALGO_BLOCKDFT) {
        pvq::set_last_error("unknown algorithm");
        return PVQ_ERR_INVALID_ARG;
    }
    v->impl->set_algo(algo);
    return PVQ_OK;
}
pvq_algo pvq_vqt_last_algo(const pvq_vqt* v) { return v ? v->impl->last_algo() : PVQ_ALGO_AUTO; }

uint32_t pvq_vqt_blockdft_columns(const pvq_vqt* v) { return v ? v->impl->blockdft_columns() : 0; }

pvq_status pvq_vqt_set_gemm_precision(pvq_vqt* v, pvq_gemm_precision p) {
    if (!v) return null_handle();
    if (p != PVQ_GEMM_F32 && p != PVQ_GEMM_BF16X3) {
        pvq::set_last_error("unknown GEMM precision");
        return PVQ_ERR_INVALID_ARG;
    }
    v->impl->set_gemm_split_bf16(p == PVQ_GEMM_BF16X3);
    return PVQ_OK;
}

void pvq_analysis_default_params(pvq_analysis_params* a) {
    if (!a) return;
    const pvq::AnalysisParameters d;
    a->peak_min_prominence = d.peak_min_prominence;
    a->peak_min_height = d.peak_min_height;
    a->bass_min_prominence = d.bass_min_prominence;
    a->bass_min_height = d.bass_min_height;
    a->highest_bassnote = d.highest_bassnote;
    a->harmonic_threshold = d.harmonic_threshold;
}

pvq_status pvq_analyze_batch_device(pvq_vqt* v, const float* d_db, size_t n_frames, const pvq_analysis_params* a,
                                    uint32_t* d_peak_mask, uint32_t* d_peak_count, float* d_center, float* d_size,
                                    uint32_t max_peaks, void* stream) {
    if (!v) return null_handle();
    return v->impl->analyze_batch_device(d_db, n_frames, to_cpp(a), d_peak_mask, d_peak_count, d_center, d_size,
                                         max_peaks, static_cast<hipStream_t>(stream));
}

pvq_status pvq_analyze_batch(pvq_vqt* v, const float* db, size_t n_frames, const pvq_analysis_params* a,
                             uint32_t* peak_mask, uint32_t* peak_count, float* center, float* size,
                             uint32_t max_peaks) {
    if (!v) return null_handle();
    return v->impl->analyze_batch(db, n_frames, to_cpp(a), peak_mask, peak_count, center, size, max_peaks);
}

pvq_status pvq_vqt_analyze_batch_device(pvq_vqt* v, const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                        const pvq_analysis_params* a, float* d_out_db, uint32_t* d_peak_mask,
                                        uint32_t* d_peak_count, float* d_center, float* d_size, uint32_t max_peaks,
                                        void* stream) {
    if (!v) return null_handle();
    return v->impl->vqt_analyze_batch_device(d_pcm, n_lead, hop, n_frames, to_cpp(a), d_out_db, d_peak_mask,
                                             d_peak_count, d_center, d_size, max_peaks,
                                             static_cast<hipStream_t>(stream));
}

pvq_status pvq_vqt_set_profiling(pvq_vqt* v, int enable) {
    if (!v) return null_handle();
    v->impl->set_profiling(enable != 0);
    return PVQ_OK;
}
uint32_t pvq_vqt_last_kernel_ms(pvq_vqt* v, float* out_ms, uint32_t capacity) {
    if (!v || !out_ms) return 0;
    return v->impl->last_kernel_ms(out_ms, capacity);
}
uint32_t pvq_vqt_last_kernel_launches(pvq_vqt* v, uint32_t* out_n, uint32_t capacity) {
    if (!v || !out_n) return 0;
    return v->impl->last_kernel_launches(out_n, capacity);
}
uint32_t pvq_vqt_last_frames_per_launch(const pvq_vqt* v) { return v ? v->impl->last_frames_per_launch() : 0; }
const char* pvq_vqt_kernel_name(uint32_t slot) { return pvq::Vqt::slot_name(slot); }

}  // extern "C"
