// vqt_host.cpp — host-side construction of the VQT spectral kernel (runs once per parameter set).
//
// Follows pitchvis_analysis/src/vqt.rs: filter_bank_params (:517-587), vqt_kernel (:599-759),
// calculate_filter (:769-852).  The reference evaluates everything here in f32 (including the
// wavelet phase, whose f32 rounding at angles of thousands of radians is part of the kernel it
// ships), so this file keeps f32 and the reference's operation order; it MUST be compiled with
// -ffp-contract=off.  The small complex FFTs (64..1024 points) are a plain iterative radix-2.
#include "vqt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace pvq {

std::string VqtError::to_string() const {
    char buf[320];
    switch (kind) {
        case AboveNyquist:
            std::snprintf(buf, sizeof buf,
                          "the highest VQT bin frequency (%g Hz) exceeds the Nyquist frequency (%g Hz); "
                          "reduce octaves or increase the sample rate",
                          a, b);
            return buf;
        case WindowExceedsNFft:
            std::snprintf(buf, sizeof buf,
                          "the longest filter window (%g samples) exceeds n_fft (%u samples); "
                          "increase n_fft or gamma, or decrease quality",
                          a, static_cast<unsigned>(b));
            return buf;
        default:
            return "ok";
    }
}

namespace {

// Rust `as usize` on an f32: truncate toward zero, saturate, NaN -> 0.
uint32_t trunc_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(x);
}

// Unnormalised forward complex FFT, power-of-two length, decimation in time.
class SmallFft {
   public:
    explicit SmallFft(uint32_t n) : n_(n), rev_(n), tw_(n / 2 ? n / 2 : 1) {
        uint32_t lg = 0;
        while ((1u << lg) < n) ++lg;
        if ((1u << lg) != n) throw PanicError("FFT length must be a power of two");
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t r = 0;
            for (uint32_t b = 0; b < lg; ++b)
                if (i & (1u << b)) r |= 1u << (lg - 1 - b);
            rev_[i] = r;
        }
        const double pi = 3.14159265358979323846;
        for (uint32_t k = 0; k < n / 2; ++k) {
            double ang = -2.0 * pi * static_cast<double>(k) / static_cast<double>(n);
            tw_[k] = cf32{static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang))};
        }
    }
    void forward(std::vector<cf32>& a) const {
        for (uint32_t i = 0; i < n_; ++i)
            if (rev_[i] > i) std::swap(a[i], a[rev_[i]]);
        for (uint32_t len = 2; len <= n_; len <<= 1) {
            const uint32_t half = len >> 1, step = n_ / len;
            for (uint32_t base = 0; base < n_; base += len) {
                for (uint32_t j = 0; j < half; ++j) {
                    const cf32 w = tw_[j * step];
                    const cf32 u = a[base + j];
                    const cf32 v = a[base + j + half];
                    const float tr = v.re * w.re - v.im * w.im;
                    const float ti = v.re * w.im + v.im * w.re;
                    a[base + j] = cf32{u.re + tr, u.im + ti};
                    a[base + j + half] = cf32{u.re - tr, u.im - ti};
                }
            }
        }
    }

   private:
    uint32_t n_;
    std::vector<uint32_t> rev_;
    std::vector<cf32> tw_;
};

struct RateGroup {  // vqt.rs:608-612
    uint32_t factor;
    uint32_t w0, w1;
    uint32_t first, count;
};

// vqt.rs:769-852: one filter's sparsified conjugate spectrum at its decimated rate.
void calculate_filter(float sr, float sparsity_quantile, uint32_t sr_scaling, const FilterParams& fp,
                      uint32_t w0, uint32_t w1, float window_center, const SmallFft& fft,
                      std::vector<cf32>& v, float& band_lo_hz, float& band_hi_hz) {
    const float scaling = static_cast<float>(sr_scaling);
    const float scaled_freq = fp.freq * scaling;
    const float scaled_window_length = fp.window_length / scaling;
    const uint32_t len = trunc_sat(std::round(scaled_window_length));
    const float scaled_center = (window_center - static_cast<float>(w0)) / scaling;
    const uint32_t center = trunc_sat(std::floor(scaled_center));
    const uint32_t scaled_n_fft = (w1 - w0) / sr_scaling;
    // the reference panics here (vqt.rs:785-792); the C ABI reports the same text with PVQ_ERR_INVALID_ARG
    if (!(len <= scaled_n_fft)) throw PanicError("assertion failed: scaled_window_length_rounded <= scaled_n_fft");
    if (!(center >= len / 2))
        throw PanicError("filter window must fit between the start of its group window and the common window center");
    const uint32_t begin = center - len / 2;
    if (!(begin + len <= scaled_n_fft)) throw PanicError("filter window must end before the end of its group window");

    v.assign(scaled_n_fft, cf32{0.0f, 0.0f});
    const double pi = 3.14159265358979323846;
    const float two_pi_f32 = 2.0f * 3.14159274101257324f;
    for (uint32_t i = 0; i < len; ++i) {
        // apodize::hanning_iter (f64), then `as f32`
        const double xx = (pi * static_cast<double>(i)) / static_cast<double>(len - 1);
        const float hann = static_cast<float>(0.5 - 0.5 * std::cos(2.0 * xx));
        // Complex32::i() * 2.0 * PI * (i as f32) * scaled_freq / sr, evaluated left to right in f32
        float angle = two_pi_f32 * static_cast<float>(i);
        angle = angle * scaled_freq;
        angle = angle / sr;
        v[begin + i] = cf32{hann * std::cos(angle), hann * std::sin(angle)};
    }
    float norm_1 = 0.0f;
    for (const cf32& z : v) norm_1 += std::hypot(z.re, z.im);
    for (cf32& z : v) {
        z.re /= norm_1;
        z.im /= norm_1;
    }
    fft.forward(v);
    for (cf32& z : v) z.im = -z.im;

    std::vector<float> mags(scaled_n_fft);
    for (uint32_t i = 0; i < scaled_n_fft; ++i) mags[i] = std::hypot(v[i].re, v[i].im);
    {   // calculate_bandwidth (vqt.rs:981-989) over find_3db_points (vqt.rs:962-977), at the decimated rate
        uint32_t peak = 0;   // util::arg_max: the first maximum
        for (uint32_t i = 1; i < scaled_n_fft; ++i)
            if (mags[i] > mags[peak]) peak = i;
        const float threshold = mags[peak] / std::sqrt(2.0f);   // -3 dB
        uint32_t lower = peak, upper = peak;
        while (lower > 0 && mags[lower] > threshold) --lower;
        while (upper < scaled_n_fft - 1 && mags[upper] > threshold) ++upper;
        const float scaled_sr = sr / scaling;
        band_lo_hz = static_cast<float>(lower) * scaled_sr / static_cast<float>(scaled_n_fft);
        band_hi_hz = static_cast<float>(upper) * scaled_sr / static_cast<float>(scaled_n_fft);
    }
    std::sort(mags.begin(), mags.end());
    float total = 0.0f;
    for (float m : mags) total += m;
    const float target = (1.0f - sparsity_quantile) * total;
    float accum = 0.0f;
    uint32_t cutoff_idx = 0;
    while (accum < target && cutoff_idx < scaled_n_fft) {
        accum += mags[cutoff_idx];
        ++cutoff_idx;
    }
    const float cutoff = cutoff_idx == 0 ? 0.0f : mags[cutoff_idx - 1];
    for (cf32& z : v)
        if (std::hypot(z.re, z.im) < cutoff) z = cf32{0.0f, 0.0f};
}

struct Entry {
    uint32_t row, col;
    cf32 v;
};

void to_csr(std::vector<Entry>& e, uint32_t rows, uint32_t cols, CsrMatrix& m) {
    std::stable_sort(e.begin(), e.end(), [](const Entry& x, const Entry& y) {
        return x.row != y.row ? x.row < y.row : x.col < y.col;
    });
    m.rows = rows;
    m.cols = cols;
    m.row_ptr.assign(rows + 1, 0u);
    m.col_idx.clear();
    m.values.clear();
    m.col_idx.reserve(e.size());
    m.values.reserve(e.size());
    for (const Entry& t : e) {
        ++m.row_ptr[t.row + 1];
        m.col_idx.push_back(t.col);
        m.values.push_back(t.v);
    }
    for (uint32_t r = 0; r < rows; ++r) m.row_ptr[r + 1] += m.row_ptr[r];
}

}  // namespace

void bin_log_frequencies(const VqtParameters& p, std::vector<float>& lnf) {
    const uint32_t n_bins = p.range.n_buckets();
    const float bpo = static_cast<float>(p.range.buckets_per_octave);
    lnf.resize(n_bins);
    for (uint32_t k = 0; k < n_bins; ++k) {
        const float f = p.range.min_freq * std::pow(2.0f, static_cast<float>(k) / bpo);
        lnf[k] = std::log(f);
    }
}

VqtError filter_bank_params(const VqtParameters& p, std::vector<FilterParams>& out) {
    const uint32_t n_bins = p.range.n_buckets();
    const float bpo = static_cast<float>(p.range.buckets_per_octave);
    VqtError err;
    const float highest = p.range.min_freq * std::pow(2.0f, static_cast<float>(n_bins - 1) / bpo);
    const float nyquist = p.sr / 2.0f;
    if (highest > nyquist) {
        err.kind = VqtError::AboveNyquist;
        err.a = highest;
        err.b = nyquist;
        return err;
    }
    const float r = std::pow(2.0f, 1.0f / bpo);
    const float alpha = (r * r - 1.0f) / (r * r + 1.0f);
    out.resize(n_bins);
    for (uint32_t k = 0; k < n_bins; ++k) {
        FilterParams f;
        f.freq = p.range.min_freq * std::pow(2.0f, static_cast<float>(k) / bpo);
        f.window_length = p.quality * p.sr / (alpha * f.freq + p.gamma);
        const float minimum_scaled_sr = std::ceil(f.freq * 2.0f * 1.15f);
        const uint32_t k_rate = trunc_sat(std::floor(std::log2(p.sr / minimum_scaled_sr)));
        f.sr_downscaling_factor = 1u << k_rate;
        const uint32_t k_win = trunc_sat(std::floor(std::log2(static_cast<float>(p.n_fft) / f.window_length)));
        f.minimum_needed_window_size = p.n_fft >> k_win;
        out[k] = f;
    }
    if (out[0].window_length > static_cast<float>(p.n_fft)) {
        err.kind = VqtError::WindowExceedsNFft;
        err.a = out[0].window_length;
        err.b = static_cast<float>(p.n_fft);
    }
    return err;
}

VqtError build_plan(const VqtParameters& p, HostPlan& plan) {
    plan = HostPlan{};
    plan.params = p;
    VqtError err = filter_bank_params(p, plan.filters);
    if (err.kind != VqtError::None) return err;
    const std::vector<FilterParams>& filters = plan.filters;
    const uint32_t n_bins = p.range.n_buckets();
    const float n_fft_f = static_cast<float>(p.n_fft);
    const float window_center = n_fft_f - filters[0].window_length / 2.0f;
    plan.window_center = window_center;

    std::vector<RateGroup> rate_groups;
    for (uint32_t k = 0; k < n_bins;) {
        uint32_t e = k + 1;
        while (e < n_bins && filters[e].sr_downscaling_factor == filters[e - 1].sr_downscaling_factor) ++e;
        uint32_t window_size = 0;
        for (uint32_t i = k; i < e; ++i) window_size = std::max(window_size, filters[i].minimum_needed_window_size);
        RateGroup rg;
        rg.factor = filters[k].sr_downscaling_factor;
        const float half = static_cast<float>(window_size) / 2.0f;
        if ((window_center + half) < n_fft_f) {
            rg.w0 = trunc_sat(window_center - half);
            rg.w1 = trunc_sat(window_center + half);
        } else {
            rg.w0 = p.n_fft - window_size;
            rg.w1 = p.n_fft;
        }
        rg.first = k;
        rg.count = e - k;
        rate_groups.push_back(rg);
        k = e;
    }

    const float kernel_gain = std::sqrt(p.sr);
    uint32_t min_begin = p.n_fft;
    std::vector<cf32> spectrum;
    float last_upper_bandwidth = 0.0f;   // vqt.rs:649
    plan.bandwidth_lo_hz.assign(filters.size(), 0.0f);
    plan.bandwidth_hi_hz.assign(filters.size(), 0.0f);
    plan.warnings.clear();
    for (size_t a = 0; a < rate_groups.size();) {
        size_t b = a + 1;
        while (b < rate_groups.size() && rate_groups[b].w0 == rate_groups[a].w0 && rate_groups[b].w1 == rate_groups[a].w1) ++b;
        WindowGroup wg;
        wg.window_begin = rate_groups[a].w0;
        wg.window_end = rate_groups[a].w1;
        wg.first_bin = rate_groups[a].first;
        const uint32_t window_size = wg.window_size();
        const uint32_t n_spectrum = window_size / 2 + 1;
        uint32_t n_filters = 0;
        for (size_t i = a; i < b; ++i) n_filters += rate_groups[i].count;

        std::vector<Entry> pos, neg;
        uint32_t row = 0;
        for (size_t i = a; i < b; ++i) {
            const uint32_t m = rate_groups[i].factor;
            const uint32_t scaled_n_fft = window_size / m;
            const SmallFft fft(scaled_n_fft);
            for (uint32_t f = 0; f < rate_groups[i].count; ++f, ++row) {
                const FilterParams& fpar = filters[rate_groups[i].first + f];
                float blo = 0.0f, bhi = 0.0f;
                calculate_filter(p.sr, p.sparsity_quantile, m, fpar, wg.window_begin, wg.window_end, window_center, fft, spectrum, blo, bhi);
                plan.bandwidth_lo_hz[rate_groups[i].first + f] = blo;
                plan.bandwidth_hi_hz[rate_groups[i].first + f] = bhi;
                if (last_upper_bandwidth > 0.0f && blo > last_upper_bandwidth) {   // vqt.rs:695-709, the reference's warn!() text
                    char msg[320];
                    std::snprintf(msg, sizeof msg,
                                  "coverage gap below the filter at %.1f Hz: its -3 dB band starts at %.2f Hz but the previous filter's band ends at "
                                  "%.2f Hz (%.1f%% of this filter's bandwidth); decrease quality to close the gap",
                                  fpar.freq, blo, last_upper_bandwidth, 100.0f * (blo - last_upper_bandwidth) / (bhi - blo));
                    plan.warnings.emplace_back(msg);
                }
                last_upper_bandwidth = bhi;
                for (uint32_t j = 0; j < scaled_n_fft; ++j) {
                    const cf32 z = spectrum[j];
                    if (z.re == 0.0f && z.im == 0.0f) continue;
                    cf32 value{z.re * kernel_gain, z.im * kernel_gain};
                    value.re = value.re / static_cast<float>(window_size);
                    value.im = value.im / static_cast<float>(window_size);
                    if (j <= scaled_n_fft / 2)
                        pos.push_back(Entry{row, j, value});
                    else
                        neg.push_back(Entry{row, scaled_n_fft - j, cf32{value.re, -value.im}});
                }
            }
        }
        to_csr(pos, n_filters, n_spectrum, wg.filter_bank);
        to_csr(neg, n_filters, n_spectrum, wg.negative_filter_bank);
        min_begin = std::min(min_begin, wg.window_begin);
        plan.kernel.window_groups.push_back(std::move(wg));
        a = b;
    }
    plan.delay_seconds = static_cast<double>((n_fft_f - window_center) / p.sr);
    plan.window_union = p.n_fft - min_begin;
    return err;
}

}  // namespace pvq
