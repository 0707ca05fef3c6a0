// consumers_host.hpp — host-side callers either side of the VQT path.
//
//  * MonoAgc            — dagc_fork/src/lib.rs:19-87 (sample-sequential gain control, upstream of the path)
//  * train stream / rows — pitchvis_train/src/train.rs:252-351 (chunked downmix + AGC + ring buffer, a VQT frame
//                          every `step` chunks) and :443-460 (row = n_bins dB values + 128 key targets), .npy
//                          writer :192-208.  The frames themselves come from the GPU batch path.
//  * LED frame          — pitchvis_serial/src/main.rs:122-175 (0xFF, 16-bit count, RGB triples <= 0xFE) with the
//                          colour mapping of pitchvis_colors/src/lib.rs:86-117.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pvq {

// dagc_fork/src/lib.rs:19-87
class MonoAgc {
   public:
    // lib.rs:36-49: desired_output_rms finite and > 0, distortion_factor in [0, 1]
    static bool valid(float desired_output_rms, float distortion_factor, std::string* why);
    MonoAgc(float desired_output_rms, float distortion_factor)
        : desired_output_rms_(desired_output_rms), distortion_factor_(distortion_factor) {}
    void freeze_gain(bool freeze) { frozen_ = freeze; }
    bool is_gain_frozen() const { return frozen_; }
    float gain() const { return gain_; }
    void process(float* samples, size_t n);   // lib.rs:76-86

   private:
    float desired_output_rms_, distortion_factor_;
    float gain_ = 1.0f;
    bool frozen_ = false;
};

// train.rs:128-129: chunk = (delay_ms * sr / 1000) / 64 * 64 with delay_ms = Duration::as_millis() (truncating)
size_t train_chunk_samples(double delay_seconds, float sr);

// train.rs:286-310 over n_chunks rendered chunks: left := (left + right) / 2; the gain is frozen for a chunk whose
// sum of squares is < 1e-6; the AGC runs in place over the chunk.  mono_out [n_chunks * chunk];
// gain_out [n_chunks] = agc.gain() after each chunk.  right may be null (mono render: left is taken as is).
void train_condition_stream(MonoAgc& agc, const float* left, const float* right, size_t n_chunks, size_t chunk, float* mono_out,
                            float* gain_out);

// train.rs:317-337 + 347 + 443-460.  Frame f carries the voices sounding at its chunk; its row is labelled with the
// active keys of the PREVIOUS frame (train.rs:314,347: prev_active_keys), empty for frame 0.  A key's value is the
// largest (mix_left + mix_right) / 2 * agc_gain over its voices; target[key] = value > 0.5.
// Returns false (message in *why) on a key outside 0..127 (the reference indexes a [f32; 128] and would panic).
bool train_rows(const float* db, size_t n_frames, uint32_t n_bins, const uint32_t* voice_ptr, const int32_t* voice_key,
                const float* voice_gain_left, const float* voice_gain_right, const float* agc_gain, float* out_rows, std::string* why);

// train.rs:192-208: a flat little-endian f32 .npy (version 1.0 header, shape (n,))
bool npy_write_f32(const char* path, const float* data, uint64_t n, std::string* why);

// pitchvis_colors/src/lib.rs:86-117 (LCh via CIE L*a*b*, D65, sRGB; the `lab` crate's conversions restated)
void calculate_color(uint16_t buckets_per_octave, float bucket, const float colors[12][3], float gray_level, float easing_pow,
                     float out_rgb[3]);

// pitchvis_serial/src/main.rs:122-175.  center/size: AnalysisState::peaks_continuous; n_buckets = range.n_buckets().
// out must hold 3 + 3 * n_buckets bytes; returns the number written.
size_t led_frame(uint32_t n_buckets, uint16_t buckets_per_octave, const float* center, const float* size, uint32_t n_peaks,
                 const float colors[12][3], float gray_level, float easing_pow, uint8_t* out);

}  // namespace pvq
