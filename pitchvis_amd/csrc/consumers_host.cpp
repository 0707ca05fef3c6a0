// consumers_host.cpp — see consumers_host.hpp.  Built with -ffp-contract=off: every expression below is the
// reference's f32 expression, operation for operation.
#include "consumers_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace pvq {

// ---------------------------------------------------------------------------------------------------------
// dagc_fork/src/lib.rs
// ---------------------------------------------------------------------------------------------------------
bool MonoAgc::valid(float desired_output_rms, float distortion_factor, std::string* why) {
    if (!(desired_output_rms > 0.0f && std::isfinite(desired_output_rms))) {   // lib.rs:37-41
        if (why) *why = "`desired_output_rms` must be a finite positive number, but got " + std::to_string(desired_output_rms);
        return false;
    }
    if (!(distortion_factor >= 0.0f && distortion_factor <= 1.0f)) {           // lib.rs:42-46
        if (why) *why = "`distortion_factor` must be a number within `0.0 ..= 1.0`, but got " + std::to_string(distortion_factor);
        return false;
    }
    return true;
}

void MonoAgc::process(float* samples, size_t n) {   // lib.rs:76-86
    for (size_t i = 0; i < n; ++i) {
        float x = samples[i] * gain_;
        samples[i] = x;
        if (!frozen_) {
            const float y = (x * x) / desired_output_rms_;
            float g = 1.0f + (distortion_factor_ * (1.0f - y));
            g = std::fmax(g, distortion_factor_);   // f32::max (a NaN operand is ignored, like fmaxf)
            gain_ *= g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pitchvis_train/src/train.rs
// ---------------------------------------------------------------------------------------------------------
size_t train_chunk_samples(double delay_seconds, float sr) {
    // vqt.delay is Duration::from_secs_f32(..) (vqt.rs:756); as_millis() truncates
    const uint64_t delay_ms = static_cast<uint64_t>(delay_seconds * 1000.0);
    const size_t s = static_cast<size_t>(delay_ms) * static_cast<size_t>(sr) / 1000;   // train.rs:128 (SR is an integer constant)
    return (s / 64) * 64;                                                              // train.rs:129
}

void train_condition_stream(MonoAgc& agc, const float* left, const float* right, size_t n_chunks, size_t chunk, float* mono_out,
                            float* gain_out) {
    for (size_t c = 0; c < n_chunks; ++c) {
        float* dst = mono_out + c * chunk;
        const float* l = left + c * chunk;
        const float* r = right ? right + c * chunk : nullptr;
        float sq = 0.0f;
        for (size_t i = 0; i < chunk; ++i) {
            const float m = r ? (l[i] + r[i]) / 2.0f : l[i];   // train.rs:286-289
            dst[i] = m;
            sq += m * m;                                        // train.rs:292 (sequential f32 sum of powi(2))
        }
        agc.freeze_gain(sq < 1e-6f);                            // train.rs:293
        agc.process(dst, chunk);                                // train.rs:298-301 (the ring buffer's newest samples)
        if (gain_out) gain_out[c] = agc.gain();
    }
}

bool train_rows(const float* db, size_t n_frames, uint32_t n_bins, const uint32_t* voice_ptr, const int32_t* voice_key,
                const float* voice_gain_left, const float* voice_gain_right, const float* agc_gain, float* out_rows, std::string* why) {
    const size_t row_len = static_cast<size_t>(n_bins) + 128;
    std::map<int32_t, float> prev, cur;
    for (size_t f = 0; f < n_frames; ++f) {
        prev.swap(cur);   // train.rs:314
        cur.clear();
        for (uint32_t v = voice_ptr[f]; v < voice_ptr[f + 1]; ++v) {   // train.rs:317-337
            const float gain = (voice_gain_left[v] + voice_gain_right[v]) / 2.0f * agc_gain[f];
            auto it = cur.find(voice_key[v]);
            if (it != cur.end()) {
                if (gain > it->second) it->second = gain;
            } else {
                cur.emplace(voice_key[v], gain);
            }
        }
        float* row = out_rows + f * row_len;
        std::memcpy(row, db + f * static_cast<size_t>(n_bins), sizeof(float) * n_bins);   // train.rs:451
        float* targets = row + n_bins;
        for (int k = 0; k < 128; ++k) targets[k] = 0.0f;
        for (const auto& kv : prev) {                                                      // train.rs:456-458
            if (kv.first < 0 || kv.first >= 128) {
                if (why) *why = "midi key " + std::to_string(kv.first) + " outside 0..127";
                return false;
            }
            targets[kv.first] = (kv.second > 0.5f) ? 1.0f : 0.0f;
        }
    }
    return true;
}

bool npy_write_f32(const char* path, const float* data, uint64_t n, std::string* why) {
    // NumPy format 1.0: magic, version, u16 header length, ASCII dict padded with spaces to a multiple of 64, '\n'
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(n) + ",), }";
    size_t total = 10 + dict.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    FILE* fp = std::fopen(path, "wb");
    if (!fp) {
        if (why) *why = std::string("cannot open ") + path;
        return false;
    }
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const uint16_t hl = static_cast<uint16_t>(dict.size());
    const unsigned char hlen[2] = {static_cast<unsigned char>(hl & 0xff), static_cast<unsigned char>(hl >> 8)};
    bool ok = std::fwrite(magic, 1, 8, fp) == 8 && std::fwrite(hlen, 1, 2, fp) == 2 &&
              std::fwrite(dict.data(), 1, dict.size(), fp) == dict.size() && (n == 0 || std::fwrite(data, sizeof(float), n, fp) == n);
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok && why) *why = std::string("short write to ") + path;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------
// pitchvis_colors/src/lib.rs with the conversions of the `lab` crate (0.11.0, Cargo.lock:3985) restated:
// sRGB (u8) <-> CIE XYZ (D65) <-> L*a*b* <-> LCh.  Not vendored in the reference tree; parity with the crate's
// exact constants is unpinned.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr float KAPPA = 24389.0f / 27.0f;
constexpr float EPSILON = 216.0f / 24389.0f;
constexpr float CBRT_EPSILON = 6.0f / 29.0f;
constexpr float S_0 = 0.003130668442500564f;
constexpr float E_0_255 = 3294.6f * S_0;
constexpr float WHITE_X = 0.9504492182750991f;
constexpr float WHITE_Z = 1.0889166484304715f;

inline float srgb_expand(float c) {   // c in 0..255
    if (c > E_0_255) return std::pow((c + 0.055f * 255.0f) / (1.055f * 255.0f), 2.4f);
    return c / (12.92f * 255.0f);
}
inline float srgb_compress(float c) {
    const float v = (c > S_0) ? 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f : 12.92f * c;
    return std::fmax(std::fmin(v, 1.0f), 0.0f);
}
inline float lab_map(float c) { return (c > EPSILON) ? std::pow(c, 1.0f / 3.0f) : (KAPPA * c + 16.0f) / 116.0f; }

void rgb_to_lch(const uint8_t rgb[3], float& l, float& c, float& h) {
    const float r = srgb_expand(static_cast<float>(rgb[0])), g = srgb_expand(static_cast<float>(rgb[1])),
                b = srgb_expand(static_cast<float>(rgb[2]));
    const float x = r * 0.4124108464885388f + g * 0.3575845678529519f + b * 0.18045380393360833f;
    const float y = r * 0.21264934272065283f + g * 0.7151691357059038f + b * 0.07218152157344333f;
    const float z = r * 0.019331758429150258f + g * 0.11919485595098397f + b * 0.9503900340503373f;
    const float fx = lab_map(x / WHITE_X), fy = lab_map(y), fz = lab_map(z / WHITE_Z);
    l = (116.0f * fy) - 16.0f;
    const float a = 500.0f * (fx - fy), bb = 200.0f * (fy - fz);
    c = std::hypot(a, bb);
    h = std::atan2(bb, a);
}
void lch_to_rgb(float l, float c, float h, uint8_t rgb[3]) {
    const float a = c * std::cos(h), bb = c * std::sin(h);
    const float fy = (l + 16.0f) / 116.0f;
    const float fx = (a / 500.0f) + fy;
    const float fz = fy - (bb / 200.0f);
    const float xr = (fx > CBRT_EPSILON) ? fx * fx * fx : ((fx * 116.0f) - 16.0f) / KAPPA;
    const float yr = (l > EPSILON * KAPPA) ? fy * fy * fy : l / KAPPA;
    const float zr = (fz > CBRT_EPSILON) ? fz * fz * fz : ((fz * 116.0f) - 16.0f) / KAPPA;
    const float x = xr * WHITE_X, y = yr, z = zr * WHITE_Z;
    const float r = x * 3.240812398895283f - y * 1.5373084456298136f - z * 0.4985865229069666f;
    const float g = x * -0.9692430170086407f + y * 1.8759663029085742f + z * 0.04155503085668564f;
    const float b = x * 0.055638398436112804f - y * 0.20400746093241362f + z * 1.0571295702861434f;
    rgb[0] = static_cast<uint8_t>(std::round(srgb_compress(r) * 255.0f));
    rgb[1] = static_cast<uint8_t>(std::round(srgb_compress(g) * 255.0f));
    rgb[2] = static_cast<uint8_t>(std::round(srgb_compress(b) * 255.0f));
}
inline uint8_t sat_u8(float v) {   // Rust `as u8`: saturating, NaN -> 0, truncation toward zero
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return static_cast<uint8_t>(v);
}
}  // namespace

void calculate_color(uint16_t buckets_per_octave, float bucket, const float colors[12][3], float gray_level, float easing_pow,
                     float out_rgb[3]) {
    const float pitch_continuous = 12.0f * bucket / static_cast<float>(buckets_per_octave);   // lib.rs:93
    const float rounded = std::round(pitch_continuous);
    const size_t idx = static_cast<size_t>(rounded < 0.0f ? 0.0f : rounded) % 12;             // `as usize` saturates at 0
    uint8_t base[3];
    for (int i = 0; i < 3; ++i) base[i] = sat_u8(colors[idx][i] * 255.0f);                     // lib.rs:94-95
    const float inaccuracy_cents = std::fabs(pitch_continuous - rounded);                      // lib.rs:96
    float l, c, h;
    rgb_to_lch(base, l, c, h);                                                                 // lib.rs:98
    const float saturation = 1.0f - std::pow(2.0f * inaccuracy_cents, easing_pow);             // lib.rs:104
    c *= saturation;                                                                           // lib.rs:105
    l = saturation * l + (1.0f - saturation) * gray_level;                                     // lib.rs:106
    uint8_t rgb[3];
    lch_to_rgb(l, c, h, rgb);                                                                  // lib.rs:108
    for (int i = 0; i < 3; ++i) out_rgb[i] = static_cast<float>(rgb[i]) / 255.0f;
}

size_t led_frame(uint32_t n_buckets, uint16_t buckets_per_octave, const float* center, const float* size, uint32_t n_peaks,
                 const float colors[12][3], float gray_level, float easing_pow, uint8_t* out) {
    std::vector<float> x(n_buckets, 0.0f);                                      // main.rs:130
    for (uint32_t p = 0; p < n_peaks; ++p) {                                    // main.rs:131-140
        const float fl = std::floor(center[p]);
        if (!(fl >= 0.0f) || fl >= static_cast<float>(n_buckets)) continue;     // the reference would panic on an index out of range
        const size_t lower = static_cast<size_t>(fl);
        const float fract = center[p] - std::trunc(center[p]);                  // f32::fract
        x[lower] = size[p] * (1.0f - std::pow(fract, 1.9f));
        if (lower < n_buckets - 1) x[lower + 1] = size[p] * std::pow(fract, 1.9f);
    }
    // util::arg_max (util.rs:34-45): fold from f32::MIN, first maximum wins
    size_t k_max = 0;
    float best = -3.40282347e+38f;
    for (size_t i = 0; i < x.size(); ++i)
        if (x[i] > best) {
            best = x[i];
            k_max = i;
        }
    const float max_size = x.empty() ? 0.0f : x[k_max];
    size_t o = 0;
    out[o++] = 0xFF;                                                            // main.rs:146
    const uint16_t num_triples = static_cast<uint16_t>(n_buckets);              // main.rs:148 (x_vqt_peakfiltered.len())
    out[o++] = static_cast<uint8_t>(num_triples / 256);
    out[o++] = static_cast<uint8_t>(num_triples % 256);
    const uint32_t shift = buckets_per_octave - 3u * (buckets_per_octave / 12u);   // main.rs:154
    for (uint32_t idx = 0; idx < n_buckets; ++idx) {
        float rgb[3];
        const float bucket = std::fmod(static_cast<float>(idx + shift), static_cast<float>(buckets_per_octave));
        calculate_color(buckets_per_octave, bucket, colors, gray_level, easing_pow, rgb);
        const float color_coefficient = 1.0f - (1.0f - x[idx] / max_size);       // main.rs:162
        for (int i = 0; i < 3; ++i) out[o++] = sat_u8((rgb[i] * color_coefficient) * 254.0f);   // main.rs:163-167
    }
    return o;
}

}  // namespace pvq
