// analysis_batch.hip — AnalysisState::preprocess for many streams, one wavefront per stream (analysis_batch.hpp).
//
// Every arithmetic step below is the host AnalysisState's (analysis_host.cpp, itself the reference's operation order, file:line
// cited there), in f32 with FMA contraction off.  Two things make the GPU values follow the host's through the recurrence:
//   * the libm calls of the reference (exp, ln, log2, log10, powf) are evaluated in double and rounded once — the correctly rounded
//     f32 result in all but ~1e-8 of the calls, which is what glibc's expf / logf / log2f / powf deliver too;
//   * the sums the reference accumulates bin by bin (scene calmness: calmness.rs:62-92; tuning inaccuracy: pitch_analysis.rs:55-66)
//     are accumulated in that same order here, not as a tree.
// One 64-lane wave owns a stream: its bins sit at lane + 64 k (k < NK), the per-bin EMA states live in registers for the whole
// call, the three find_peaks passes per frame (bass / general split on the smoothed frame, general on the raw frame) reuse the
// wave routine of the batched peak kernels (peaks_device.hpp: bit-identical peak sets).
#include "analysis_batch.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "peaks_device.hpp"
#include "vqt_engine.hpp"

namespace pvq {

#define PVQ_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_last_error(std::string(#call) + " failed: " + hipGetErrorString(e_));              \
            return PVQ_ERR_DEVICE;                                                                 \
        }                                                                                          \
    } while (0)

struct AbArgs {
    const float* db;   // [n_streams][n_frames][n_bins]
    int n_streams, n_frames, n_bins, bpo, octaves;
    float min_freq;
    float log2_min_freq;   // log2f(min_freq), host libm (peak_detection.rs:205)
    float peak_prom, peak_h, bass_prom, bass_h;
    int highest_bassnote;
    float harm_thr;
    unsigned long long base_ms;
    int smooth_has;
    float calm_min, calm_max;
    unsigned long long note_ns, scene_ns, tuning_ns;
    unsigned long long frame_ns;
    const unsigned long long* frame_times;   // optional, [n_frames]
    // EMA weights 1 - exp(-2 dt / horizon) from the HOST's libm, one row per distinct frame time of the call (alpha_tab): the row is
    // (alpha_note, alpha_scene, alpha_tuning, -, alpha[horizon = 0 .. tab_n - 1 whole ms]): analysis.rs:301-323 truncates a bin's
    // horizon to whole milliseconds, so a few hundred values cover every bin of every frame.  frame_row: the row of each frame
    // (null: row 0 for all).  alpha_tab null: the weights are evaluated here (double exp, rounded once).
    const float* alpha_tab;
    const uint32_t* frame_row;
    int tab_n, tab_stride;
    int tab_lds;   // ab_recurrence: floats of the (single) table row staged in LDS, 0: read from memory
    const float* lnf;
    int dist, min_bin, radius;
    float *smoothed, *calm, *released, *afterglow, *peakfiltered, *pitch_acc, *pitch_dev, *scene, *tuning;   // state
    AnalysisBatchOutputs o;
    unsigned scratch_bytes;   // per-wave scratch of the peak routines: max(peaks_scratch_bytes, peaks_lean_scratch_bytes)
    unsigned wave_bytes;      // LDS bytes per wave
    int fold_min;             // ab_recurrence: a chunk with more contributing bins than this sums them by the DPP fold (developer build: PVQ_AB_FOLD)
    int generic_peaks;        // developer build: 1 = the generic peak routine for every frame (A/B against the lean one)
    // calmness.rs:40: the peaks of the RAW frames depend on the input alone, not on the recurrence: found for all frames of all streams at
    // once by the frame kernels before this kernel starts (launch_peaks_frames), [stream][frame][words] bit masks
    const uint32_t* raw_mask;
    float* sm_rows;     // [stream][frame][bin]: the smoothed frames (ab_recurrence -> ab_frames): the caller's x_vqt_smoothed, or a workspace of the object
    float* tuning_in;   // [stream][frame]: 100 x the frame's power-weighted average deviation (ab_frames -> ab_tuning)
};

namespace {
// the reference's libm calls: double evaluation, one rounding
__device__ __forceinline__ float ab_exp(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float ab_log2(float x) { return (float)log2((double)x); }
__device__ __forceinline__ float ab_log10(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float ab_exp2(float x) { return (float)exp2((double)x); }   // 2.0_f32.powf(x)
// 10.0_f32.powf(y): 2^(y log2 10) in double (the product's rounding error, ~1e-15 relative after the exp2, is eight orders below
// half an f32 ulp) — a double exp2 costs a quarter of a double pow
__device__ __forceinline__ float ab_pow10(float y) { return (float)exp2((double)y * 3.321928094887362347870319429489390175865); }
// core::time::Duration::as_secs_f32 (analysis_host.hpp)
__device__ __forceinline__ float ab_secs(unsigned long long ns) {
    return (float)(ns / 1000000000ull) + (float)(ns % 1000000000ull) / 1000000000.0f;
}
__device__ __forceinline__ unsigned long long ab_trunc_u64(float x) {
    if (!(x > 0.0f)) return 0ull;
    if (x >= 18446744073709551616.0f) return ~0ull;
    return (unsigned long long)x;
}
__device__ __forceinline__ void ab_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the same for LDS traffic only: the wave's LDS writes have landed (vector-memory loads — a frame's prefetches — stay in flight)
__device__ __forceinline__ void ab_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float ab_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// Two running sums continued over the wave's 64 values IN LANE ORDER — ((s + v0) + v1) + ... + v63, the reference's sequential f32 sums —
// without a readlane per element: with x' = (s + v0, v1, ..., v63), t rounds of  a[l] <- a[l - 1] + x'[l]  (one add whose first operand comes
// through a DPP wave shift) leave the exact left fold of lanes 0 ... t in lane t, whatever the other lanes hold meanwhile.  63 dependent
// adds per sum, the two chains interleaved.  Lane 0 adds the +0.0 an out-of-range DPP read delivers: exact for the non-negative values here.
__device__ __forceinline__ void ab_fold2(float& sa, float& sb, float va, float vb, int lane) {
    const float xa = lane == 0 ? sa + va : va, xb = lane == 0 ? sb + vb : vb;
    float a = xa, b = xb;
#pragma unroll
    for (int t = 1; t < 64; ++t) {
        a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + xa;
        b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x138, 0xf, 0xf, true)) + xb;
    }
    sa = ab_readlane(a, 63);
    sb = ab_readlane(b, 63);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// Round 5: the frame loop is split by what actually recurs.  Of preprocess's steps only three carry state from frame to frame:
// the per-bin EMA (its horizon follows the scene calmness of the frame before: analysis.rs:295-323), the afterglow (afterglow.rs:27-36)
// and the calmness step (calmness.rs:23-95: per-bin calmness / released EMAs, the scene calmness) — and none of them looks at the
// smoothed frame's PEAKS: the calmness step takes the raw frame's peaks (found before, frame-parallel: raw_mask) and the smoothed
// values.  find_peaks on the smoothed frame, enhance_peaks_continuous, promote_bass_peaks_with_harmonics, the peak-filtered frame, the
// pitch accuracy / deviation rows and the frame's power-weighted tuning inaccuracy depend on that frame's smoothed row alone.  So:
//   ab_recurrence   one wave per STREAM walks the frames in order: EMA -> smoothed row (kept: it is an output, or a workspace),
//                   afterglow, calmness, scene calmness.  ~300 instructions per frame instead of ~1 500 (2 700 at 588 bins).
//   ab_frames       one wave per (stream, FRAME), every frame of every stream at once: the smoothed row's peaks and everything
//                   derived from them; leaves the frame's tuning-grid inaccuracy (100 x average) for
//   ab_tuning       one wave per stream: the scalar EMA of pitch_analysis.rs:55-66 over the call's frames.
// Same operations in the same order as the single kernel of rounds 3-4 (and as the host AnalysisState): the recurrence state stays
// bit-identical to the oracle (tests/test_analysis_batch_gpu.py).  64 streams no longer mean 64 busy waves for the whole call.
// ------------------------------------------------------------------------------------------------
// LDS of one wave of ab_recurrence: the smoothed frame's row, the amplitude weights of the flagged bins, the compacted list of
// flagged bins
__host__ __device__ inline unsigned ab_rec_wave_bytes(int n_bins) {
    const unsigned npad = (unsigned)((n_bins + 63) / 64 * 64);
    return (unsigned)((sizeof(float) * 2 * npad + 2 * npad + 15) / 16 * 16);
}
// ... of ab_frames: the smoothed row with PK_PAD samples of +INF on both sides (the lean peak routine walks off the frame into them), the
// frame's continuous peaks (centre, size: npad / 2 each), flags, a u16 row (the compacted bass list, later the pitch rows' peak indices) and
// a region that holds the peak routines' scratch + the lean routine's peak list while the peaks are found and refined, the peaks' pitch
// accuracy / deviation (npad / 2 floats each) afterwards.  (Each in a place of its own, a wave took 16 KB at 588 bins: two workgroups per CU.)
__host__ __device__ inline unsigned ab_union_bytes(int n_bins, unsigned scratch_bytes) {
    const unsigned npad = (unsigned)((n_bins + 63) / 64 * 64);
    const unsigned a = scratch_bytes + npad /*lean peak list: npad / 2 u16*/, b = 4u * npad;
    return a > b ? a : b;
}
__host__ __device__ inline unsigned ab_wave_bytes(int n_bins, unsigned scratch_bytes) {
    const unsigned npad = (unsigned)((n_bins + 63) / 64 * 64);
    return (unsigned)((sizeof(float) * ((npad + 2 * PK_PAD) + 2 * (npad / 2)) + npad /*flags*/ + 2 * npad /*u16 row*/ + ab_union_bytes(n_bins, scratch_bytes) + 15) / 16 * 16);
}

template <int NK, bool DENSE>   // DENSE: 64 (NK - 1) < n_bins <= 64 NK (the host's promise: the chunk tests and the padded length fold away)
__global__ __launch_bounds__(256, NK <= 4 ? 4 : (NK <= 8 ? 3 : 2)) void ab_recurrence(AbArgs a) {   // (the per-bin state lives in registers: 7 NK values per lane)
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * 4 + wave;
    if constexpr (DENSE) __builtin_assume(a.n_bins > 64 * (NK - 1) && a.n_bins <= 64 * NK);
    const int n = a.n_bins, npad = (n + 63) / 64 * 64, words = (n + 31) / 32;
    // one frame time for the whole call (the usual case): the weights' row sits in LDS — the look-up is a link of the frame-to-frame chain
    // (scene calmness -> horizon -> weight -> EMA -> ... -> scene calmness), and an LDS read is the shortest one
    const float* tabL = reinterpret_cast<const float*>(ab_lds);
    if (a.tab_lds) {
        float* t = reinterpret_cast<float*>(ab_lds);
        for (int i = tid; i < a.tab_lds; i += 256) t[i] = a.alpha_tab[i];
        __syncthreads();
    }
    if (s >= a.n_streams) return;   // (no workgroup barrier past this point: waves are independent)
    unsigned char* base = ab_lds + (size_t)((a.tab_lds * 4 + 15) / 16 * 16) + (size_t)wave * ab_rec_wave_bytes(n);
    float* rowA = reinterpret_cast<float*>(base);                 // the smoothed frame
    float* pw = rowA + npad;                                      // 10^(dB / 10) of the bins around a raw peak
    uint16_t* flist = reinterpret_cast<uint16_t*>(pw + npad);     // compacted list of the flagged bins

    float y_sm[NK], y_calm[NK], y_rel[NK], y_glow[NK];
    float bm[NK], glow_k[NK];   // analysis.rs:310-316 base * frequency_multiplier; afterglow.rs:31 decay per bin
    const float bpo_f = (float)a.bpo, n_f = (float)n;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int bin = lane + 64 * k;
        const bool in = bin < n;
        y_sm[k] = in ? a.smoothed[(size_t)s * n + bin] : 0.0f;
        y_calm[k] = in ? a.calm[(size_t)s * n + bin] : 0.0f;
        y_rel[k] = in ? a.released[(size_t)s * n + bin] : 0.0f;
        y_glow[k] = in ? a.afterglow[(size_t)s * n + bin] : 0.0f;
        const float octave_fraction = (float)bin / bpo_f / (float)a.octaves;
        const float frequency_multiplier = 1.5f - 0.5f * octave_fraction;
        bm[k] = (float)a.base_ms * frequency_multiplier;
        glow_k[k] = 0.85f - 0.15f * ((float)bin / n_f);
    }
    float scene = a.scene[s];
    const float note_s = ab_secs(a.note_ns), scene_s = ab_secs(a.scene_ns);
    const float* db_s = a.db + (size_t)s * a.n_frames * n;
    float* sm_s = a.sm_rows + (size_t)s * a.n_frames * n;   // the smoothed rows: the caller's x_vqt_smoothed, or the object's workspace
    // The next frame's values and raw-peak mask word are fetched a frame ahead, and a frame's EMA weights all at once — with UNCONDITIONAL loads
    // (clamped addresses, the lanes past the frame's end select afterwards): behind a per-lane `if` the compiler waits for every load right
    // where it issues it (s_waitcnt vmcnt(0) inside the branch), and the wave stood through 2 NK global-memory round trips per frame.
    int binc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) binc[k] = min(lane + 64 * k, n - 1);
    float xn[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) xn[k] = db_s[binc[k]];
    const uint32_t* rm_s = a.raw_mask + (size_t)s * a.n_frames * words;
    const int wlane = min(lane, words - 1);
    uint32_t mw_next = rm_s[wlane];   // (lane < words <= 32 holds a word of the mask)
    unsigned long long vmask[NK];     // the chunk's bins that lie in the frame
#pragma unroll
    for (int k = 0; k < NK; ++k) vmask[k] = n - 64 * k >= 64 ? ~0ull : (n - 64 * k <= 0 ? 0ull : (1ull << (n - 64 * k)) - 1ull);

    for (int f = 0; f < a.n_frames; ++f) {
        const size_t fr = (size_t)s * a.n_frames + f;
        // (the frame time itself is only needed where a weight is not in the table: evaluated there)
        auto dt_s_of = [&]() { return ab_secs(a.frame_times ? a.frame_times[f] : a.frame_ns); };
        const float* tab = a.alpha_tab ? a.alpha_tab + (size_t)(a.frame_row ? a.frame_row[f] : 0u) * a.tab_stride : nullptr;
        const float alpha_c = a.tab_lds ? tabL[0] : (tab ? tab[0] : 1.0f - ab_exp(-2.0f * dt_s_of() / note_s));
        const float alpha_s = a.tab_lds ? tabL[1] : (tab ? tab[1] : 1.0f - ab_exp(-2.0f * dt_s_of() / scene_s));
        // ---- analysis.rs:295-323: per-bin EMA with a frequency- and calmness-dependent horizon; afterglow.rs:27-36
        const float cm = a.calm_min + (a.calm_max - a.calm_min) * scene;
        float alpha[NK];
        if (a.smooth_has) {   // (uniform)
            // the horizon in whole ms, trunc(base * multiplier * cm) as u64 (analysis.rs:316): it is below the table's length exactly when the
            // float is, so the usual case needs no 64-bit conversion
            int hidx[NK];
            bool miss = false;
            const float tab_n_f = (float)a.tab_n;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float xk = bm[k] * cm;
                const bool pos = a.base_ms > 0 && xk > 0.0f;
                const bool in = tab && (pos ? xk < tab_n_f : a.tab_n > 0);
                hidx[k] = (pos && in) ? (int)xk : 0;
                miss |= !in;
            }
            if (__ballot(miss) == 0) {   // the usual case: every weight is in the host's table — all NK loads in flight together
                if (a.tab_lds) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) alpha[k] = tabL[4 + hidx[k]];
                } else {
#pragma unroll
                    for (int k = 0; k < NK; ++k) alpha[k] = tab[4 + hidx[k]];
                }
            } else {
                const float dt_s = dt_s_of();
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const unsigned long long hms = a.base_ms > 0 ? ab_trunc_u64(bm[k] * cm) : 0ull;
                    if (tab && hms < (unsigned long long)a.tab_n) alpha[k] = tab[4 + (int)hms];
                    else alpha[k] = 1.0f - ab_exp(-2.0f * dt_s / ab_secs(hms * 1000000ull));
                }
            }
        }
        // the next frame's inputs (the last frame fetches itself again), issued BEHIND the weights' loads: vector-memory loads return in order,
        // so the wait for the weights leaves these in flight for the whole frame
        const size_t fnext = (size_t)(f + 1 < a.n_frames ? f + 1 : f);
        float xnext[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) xnext[k] = db_s[fnext * n + binc[k]];
        const uint32_t mw_after = rm_s[fnext * words + wlane];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            const float x = xn[k];
            if (!a.smooth_has) y_sm[k] = x;   // util.rs:117-120
            else y_sm[k] = y_sm[k] + alpha[k] * (x - y_sm[k]);
            float g = y_glow[k];
            g *= glow_k[k];
            if (g < y_sm[k]) g = y_sm[k];
            y_glow[k] = g;
            if (bin < n) {
                rowA[bin] = y_sm[k];
                sm_s[(size_t)f * n + bin] = y_sm[k];
                if (a.o.x_vqt_afterglow) a.o.x_vqt_afterglow[fr * n + bin] = g;
            }
        }
        // ---- calmness.rs:23-95: the peaks of the RAW frame (found before this kernel started: raw_mask) mark the bins "around a note":
        //      peak p flags [max(0, p - radius), min(n, p + radius)), i.e. bin i is flagged iff one of the bins i - radius + 1 ... i + radius is a peak.
        //      On the scalar unit: the mask's words are wave-uniform, so the 64 bins of chunk k are one 64-bit word, and the flags are that word
        //      OR-ed with its shifts by 1 ... radius down and 1 ... radius - 1 up (bits carried in from the neighbouring chunks) — which is
        //      at once the ballot the compaction below needs.  (Round 4 read the mask bit by bit from LDS: 2 radius dependent LDS round trips per
        //      chunk; with one wave on its SIMD nothing hides them: 1.5 of a frame's 4 us at 252 bins.)
        unsigned long long pm[NK + 2];   // pm[k + 1]: chunk k of the raw mask; zero words on both sides
        pm[0] = 0ull;
        pm[NK + 1] = 0ull;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)mw_next, 2 * k), hi = (uint32_t)__builtin_amdgcn_readlane((int)mw_next, 2 * k + 1);
            pm[k + 1] = (2 * k < words ? (unsigned long long)lo : 0ull) | (2 * k + 1 < words ? (unsigned long long)hi << 32 : 0ull);
        }
        // amplitude weights 10^(dB / 10) of the flagged bins only (a handful per frame): compacted, one lane per flagged bin
        bool fl[NK];
        {
            uint32_t n_fl = 0;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int bin = lane + 64 * k;
                unsigned long long fm = a.radius > 0 ? pm[k + 1] : 0ull;
                for (int d = 1; d <= a.radius; ++d) fm |= (pm[k + 1] >> d) | (pm[k + 2] << (64 - d));
                for (int d = 1; d < a.radius; ++d) fm |= (pm[k + 1] << d) | (pm[k] >> (64 - d));
                fm &= vmask[k];   // bins of the frame
                fl[k] = ((fm >> lane) & 1ull) != 0ull;
                if (fl[k]) flist[n_fl + pk_rank(fm, 0)] = (uint16_t)bin;
                n_fl += __popcll(fm);
            }
            ab_lds_sync();
            for (uint32_t idx = lane; idx < n_fl; idx += 64) {
                const int bin = flist[idx];
                pw[bin] = ab_pow10(rowA[bin] / 10.0f);
            }
            ab_lds_sync();
        }
        {
            float weighted_sum = 0.0f, weight_sum = 0.0f;   // in bin order, as the reference's loop; bins that contribute nothing are skipped
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int bin = lane + 64 * k;
                float ws = 0.0f, w = 0.0f;
                if (bin < n) {
                    if (fl[k]) {
                        y_calm[k] = y_calm[k] + alpha_c * (1.0f - y_calm[k]);
                        y_rel[k] = y_calm[k];
                        const float power = pw[bin];
                        ws = y_calm[k] * power;
                        w = power;
                    } else {
                        y_calm[k] = y_calm[k] + alpha_c * (0.0f - y_calm[k]);
                        y_rel[k] = y_rel[k] + alpha_c * (0.0f - y_rel[k]);
                        if (y_rel[k] > 0.01f) {
                            w = y_rel[k] * 0.3f;
                            ws = y_rel[k] * w;
                        }
                    }
                    if (a.o.calmness) a.o.calmness[fr * n + bin] = y_calm[k];
                }
                unsigned long long m = __ballot(ws != 0.0f || w != 0.0f);   // (adding +0.0 to a non-negative sum changes nothing)
                if (__popcll(m) > a.fold_min) {   // (uniform) many contributing bins (a peak-rich frame): the fold over all 64 lanes is the shorter chain
                    ab_fold2(weighted_sum, weight_sum, ws, w, lane);
                } else {
                    while (m) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1;
                        weighted_sum += ab_readlane(ws, b);
                        weight_sum += ab_readlane(w, b);
                    }
                }
            }
            if (weight_sum > 0.0f) scene = scene + alpha_s * (weighted_sum / weight_sum - scene);
        }
        if (lane == 0 && a.o.scene_calmness) a.o.scene_calmness[fr] = scene;
#pragma unroll
        for (int k = 0; k < NK; ++k) xn[k] = xnext[k];
        mw_next = mw_after;
        ab_lds_sync();
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int bin = lane + 64 * k;
        if (bin < n) {
            a.smoothed[(size_t)s * n + bin] = y_sm[k];
            a.calm[(size_t)s * n + bin] = y_calm[k];
            a.released[(size_t)s * n + bin] = y_rel[k];
            a.afterglow[(size_t)s * n + bin] = y_glow[k];
        }
    }
    if (lane == 0) a.scene[s] = scene;
}

template <int NK, bool DIST, bool DENSE>
__global__ __launch_bounds__(256, NK <= 6 ? 4 : (NK <= 12 ? 3 : 2)) void ab_frames(AbArgs a) {   // (latency-bound peak logic: occupancy is what makes it fast)
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (DENSE) __builtin_assume(a.n_bins > 64 * (NK - 1) && a.n_bins <= 64 * NK);
    const int n = a.n_bins, npad = (n + 63) / 64 * 64, words = (n + 31) / 32;
    const float INF = __builtin_huge_valf();
    const long long rows = (long long)a.n_streams * a.n_frames;

    PeakParamsDev ps{};   // analysis.rs:332-349: bass config at or below highest_bassnote, general config above
    ps.n_bins = n; ps.bpo = a.bpo; ps.min_freq = a.min_freq; ps.lnf = a.lnf;
    ps.peak_min_prominence = a.peak_prom; ps.peak_min_height = a.peak_h;
    ps.bass_min_prominence = a.bass_prom; ps.bass_min_height = a.bass_h;
    ps.highest_bassnote = a.highest_bassnote; ps.harmonic_threshold = a.harm_thr;
    ps.dist = a.dist; ps.min_bin = a.min_bin;
    ps.mask = a.o.peak_mask;     // [stream][frame][words] = [row][words]
    ps.count = a.o.peak_count;
    ps.center = nullptr; ps.size = nullptr; ps.max_peaks = 0;

    // workgroup-shared: the per-bin thresholds of the lean routine's candidate test (the bass / general split configuration)
    float* thr = reinterpret_cast<float*>(ab_lds);   // [2][npad]: H, P
    peaks_lean_thresholds(thr, thr + npad, ps, tid, 256);
    __syncthreads();

    unsigned char* base = ab_lds + (size_t)2 * npad * sizeof(float) + (size_t)wave * a.wave_bytes;
    float* rowA = reinterpret_cast<float*>(base) + PK_PAD;      // the smoothed frame (find_peaks input), +INF on both sides
    float* pc_c = rowA + npad + PK_PAD;                         // peaks_continuous of the frame: center, size (npad / 2 each)
    float* pc_s = pc_c + npad / 2;
    unsigned char* flag = reinterpret_cast<unsigned char*>(pc_s + npad / 2);     // is-peak flags
    uint16_t* flist = reinterpret_cast<uint16_t*>(flag + npad);                  // compacted list of the bass peaks ...
    uint16_t* rowB = flist;                                                      // ... and, once they are promoted, the pitch rows' peak indices (1 + index, 0: none)
    unsigned char* scratch = reinterpret_cast<unsigned char*>(flist + npad);     // the peak routines' scratch + the lean routine's list ...
    uint16_t* plist_lean = reinterpret_cast<uint16_t*>(scratch + a.scratch_bytes);
    float* pk_acc = reinterpret_cast<float*>(scratch);                           // ... and, once the peaks are refined, their pitch accuracy / deviation
    float* pk_dev = pk_acc + npad / 2;
    const uint16_t* plist_gen = reinterpret_cast<const uint16_t*>(scratch + npad);   // where peaks_wave_nk leaves its list
    for (int i = lane; i < PK_PAD; i += 64) {
        rowA[-PK_PAD + i] = INF;
        rowA[npad + i] = INF;
    }
    for (int i = n + lane; i < npad; i += 64) rowA[i] = INF;
    const float bpo_f = (float)a.bpo, n_f = (float)n;

    // one (stream, frame) row per wave, no loop over rows (the host launches a workgroup per 4 rows): a loop's hoisted invariants cost the
    // lean peak kernel half its registers (vqt_engine.hip, peaks_frames_lean)
    const long long fr = (long long)blockIdx.x * 4 + wave;
    if (fr < rows) {
        const int s = (int)(fr / a.n_frames), f = (int)(fr - (long long)s * a.n_frames);
        float y_sm[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            y_sm[k] = a.sm_rows[(size_t)fr * n + min(bin, n - 1)];   // (unconditional, clamped: all NK loads in flight together)
            if (bin < n) rowA[bin] = y_sm[k];
        }
        ab_wave_sync();
        // ---- analysis.rs:332-349: peaks of the smoothed frame (mask / count go straight to the outputs)
        uint32_t total = 0;
        const uint16_t* plist;
        {
            uint32_t n_cand[1] = {0}, np[1] = {0};
            bool done = false;
            if (!a.generic_peaks) done = peaks_lean_scan<NK, DIST>(rowA, scratch, thr, thr + npad, n_cand[0], ps, lane);
            if (done) {
                const bool wr[1] = {true};
                const size_t fr_[1] = {(size_t)fr};
                peaks_lean_walk<NK, 1>(rowA, 0, scratch, 0, plist_lean, npad / 2, n_cand, np, wr, fr_, ps, lane);
                total = np[0];
                plist = plist_lean;
            } else {
                ab_wave_sync();
                peaks_wave_nk<NK>(rowA, scratch, (size_t)fr, ps, lane, &total);
                plist = plist_gen;
            }
        }
        ab_wave_sync();
        // ---- peak_detection.rs:61-148: one lane per peak (ascending bins = ascending centres: a centre stays between its peak's
        //      neighbours and peaks are never adjacent)
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < npad) flag[lane + 64 * k] = 0;
        uint32_t n_bass = 0;
        for (uint32_t i0 = 0; i0 < total; i0 += 64) {
            const uint32_t idx = i0 + lane;
            const bool have = idx < total;
            const int p = have ? plist[idx] : 1;
            float ctr, sz;
            if (p < 1 || p > n - 2) {
                ctr = (float)p;
                sz = rowA[p];
            } else {
                const float l0 = a.lnf[p - 1], l1 = a.lnf[p], l2 = a.lnf[p + 1];
                const float a0 = rowA[p - 1], a1 = rowA[p], a2 = rowA[p + 1];
                const float denom = (l0 - l1) * (l0 - l2) * (l1 - l2);
                if (fabsf(denom) < 1.1920929e-07f) {
                    ctr = (float)p;
                    sz = rowA[p];
                } else {
                    const float qa = (l2 * (a1 - a0) + l0 * (a2 - a1) + l1 * (a0 - a2)) / denom;
                    const float qb = ((l2 * l2) * (a0 - a1) + (l0 * l0) * (a1 - a2) + (l1 * l1) * (a2 - a0)) / denom;
                    const float lfp = (fabsf(qa) < 1.1920929e-07f) ? l1 : pk_clampf(-qb / (2.0f * qa), l0, l2);
                    const float f_peak = ab_exp(lfp);
                    const float center = bpo_f * ab_log2(f_peak / a.min_freq);
                    const float cc = pk_clampf(center, 0.0f, n_f - 1.0f);
                    const int lower = (int)floorf(cc);
                    const int upper = min(lower + 1, n - 1);
                    const float fract = cc - truncf(cc);
                    ctr = cc;
                    sz = fmaxf(rowA[lower] * (1.0f - fract) + rowA[upper] * fract, 0.0f);
                }
            }
            if (have) {
                pc_c[idx] = ctr;
                pc_s[idx] = sz;
                flag[p] = 1;
            }
            // promote_bass_peaks_with_harmonics applies to the peaks whose centre is not above highest_bassnote: their list
            // positions go to flist
            const bool is_bass = have && !(ctr > (float)a.highest_bassnote);
            const unsigned long long bmk = __ballot(is_bass);
            if (is_bass) flist[n_bass + __popcll(bmk & ((1ull << lane) - 1ull))] = (uint16_t)idx;
            n_bass += __popcll(bmk);
        }
        ab_wave_sync();
        // ---- peak_detection.rs:172-241: FOUR lanes per bass peak, one per harmonic h = 2 .. 5 (their log2 / powf evaluations run
        //      side by side); the score is then added up in the reference's order h = 2, 3, 4, 5 by every lane of the quad
        for (uint32_t q0 = 0; q0 < n_bass; q0 += 16) {
            const uint32_t q = q0 + (lane >> 2);
            const bool have = q < n_bass;
            const uint32_t idx = have ? flist[q] : 0;
            const int h = 2 + (lane & 3);
            const float ctr = pc_c[idx];
            float sz = pc_s[idx];
            const float f0 = a.min_freq * ab_exp2(ctr / bpo_f);
            const float p0 = ab_pow10(sz / 10.0f);
            const float wts[4] = {0.5f, 0.3f, 0.15f, 0.05f};
            float term = 0.0f;   // this harmonic's contribution (+0.0: adding it changes nothing, as not adding it)
            const float hf = f0 * (float)h;
            if (have && hf >= a.min_freq) {
                const float hb = (ab_log2(hf) - a.log2_min_freq) * bpo_f;
                if (hb >= 0.0f && hb < n_f) {
                    const int lo = (int)floorf(hb);
                    const int hi = min((int)ceilf(hb), n - 1);
                    const float frac = hb - truncf(hb);
                    const float adb = (lo == hi) ? rowA[lo] : (rowA[lo] * (1.0f - frac) + rowA[hi] * frac);
                    const float hp = ab_pow10(adb / 10.0f);
                    if (hp > p0 * a.harm_thr) term = hp * wts[h - 2];
                }
            }
            float score = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) score += __shfl(term, (lane & ~3) + j);
            if (have && (lane & 3) == 0 && score > 0.0f) {
                const float boost = fminf(1.0f + 0.5f * (score / fmaxf(p0, 1e-6f)), 1.5f);
                sz += 10.0f * ab_log10(boost);
                pc_s[idx] = sz;
            }
        }
        ab_wave_sync();
        if (a.o.center) {
            for (uint32_t idx = lane; idx < total && idx < a.o.max_peaks; idx += 64) {
                a.o.center[(size_t)fr * a.o.max_peaks + idx] = pc_c[idx];
                a.o.size[(size_t)fr * a.o.max_peaks + idx] = pc_s[idx];
            }
        }
        // ---- afterglow.rs:10-21: the peak-filtered frame
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) {
                const float pf = flag[bin] ? y_sm[k] : 0.0f;
                if (a.o.x_vqt_peakfiltered) a.o.x_vqt_peakfiltered[(size_t)fr * n + bin] = pf;
                if (f == a.n_frames - 1) a.peakfiltered[(size_t)s * n + bin] = pf;   // (the call's last frame is part of the state: the getters)
            }
        }
        // ---- pitch_analysis.rs:55-66: tuning-grid inaccuracy, power-weighted, accumulated in peak order (its EMA over the frames: ab_tuning);
        //      pitch_analysis.rs:12-42: per-bin accuracy / deviation at the peaks' nearest bins, later peaks overwrite earlier ones
        //      (centres ascend, so the peaks that round to one bin are neighbours in the list: the last of them writes)
        {
            float inaccuracy_sum = 0.0f, power_sum = 0.0f;
            for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                const uint32_t idx = i0 + lane;
                const bool have = idx < total;
                const float c = have ? pc_c[idx] : 0.0f;
                const float power = have ? ab_pow10(pc_s[idx] / 10.0f) : 0.0f;
                const float semis = c * 12.0f / bpo_f;
                const float deviation = semis - roundf(semis);
                const float wi = fabsf(deviation) * power;
                const uint32_t cnt = min(64u, total - i0);
                for (uint32_t j = 0; j < cnt; ++j) {   // (uniform: every lane adds the same values in list order)
                    power_sum += ab_readlane(power, (int)j);
                    inaccuracy_sum += ab_readlane(wi, (int)j);
                }
                if (have) {
                    pk_acc[idx] = fmaxf(1.0f - 2.0f * fabsf(deviation), 0.0f);
                    pk_dev[idx] = deviation;
                }
            }
            const float avg = power_sum > 0.0f ? inaccuracy_sum / power_sum : 0.0f;
            if (lane == 0) a.tuning_in[fr] = 100.0f * avg;
        }
        // ---- pitch accuracy / deviation rows: bin <- 1 + index of the peak that writes it
        ab_wave_sync();
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) rowB[bin] = 0;
        }
        ab_wave_sync();
        for (uint32_t idx = lane; idx < total; idx += 64) {
            const float rc = roundf(pc_c[idx]);
            const bool last = idx + 1 >= total || roundf(pc_c[idx + 1]) != rc;
            if (last && rc >= 0.0f && rc < n_f) rowB[(int)rc] = (uint16_t)(idx + 1);
        }
        ab_wave_sync();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int bin = lane + 64 * k;
            if (bin < n) {
                const int w = (int)rowB[bin];
                const float acc = w ? pk_acc[w - 1] : 0.0f, dev = w ? pk_dev[w - 1] : 0.0f;
                if (a.o.pitch_accuracy) a.o.pitch_accuracy[(size_t)fr * n + bin] = acc;
                if (a.o.pitch_deviation) a.o.pitch_deviation[(size_t)fr * n + bin] = dev;
                if (f == a.n_frames - 1) {
                    a.pitch_acc[(size_t)s * n + bin] = acc;
                    a.pitch_dev[(size_t)s * n + bin] = dev;
                }
            }
        }
        ab_wave_sync();
    }
}

// pitch_analysis.rs:55-66: smoothed_tuning_grid_inaccuracy — a scalar EMA over the call's frames.  A wave per stream: the lanes fetch 64
// frames' inputs and weights at once (a chunk ahead), then the recurrence runs over the lanes by readlane — two dependent operations per
// frame.  (One thread per stream walked 1 000 frames through 1 000 exposed memory round trips: 0.42 ms beside a 2.5 ms recurrence.)
__global__ __launch_bounds__(256) void ab_tuning(AbArgs a) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= a.n_streams) return;
    float tuning = a.tuning[s];
    const float tuning_s = ab_secs(a.tuning_ns);
    const float* in_s = a.tuning_in + (size_t)s * a.n_frames;
    auto fetch = [&](int f0, float& x, float& al) {
        const int f = min(f0 + lane, a.n_frames - 1);
        x = in_s[f];
        const float* tab = a.alpha_tab ? a.alpha_tab + (size_t)(a.frame_row ? a.frame_row[f] : 0u) * a.tab_stride : nullptr;
        al = tab ? tab[2] : 1.0f - ab_exp(-2.0f * ab_secs(a.frame_times ? a.frame_times[f] : a.frame_ns) / tuning_s);
    };
    float x, al;
    fetch(0, x, al);
    for (int f0 = 0; f0 < a.n_frames; f0 += 64) {
        float xn, aln;
        fetch(f0 + 64 < a.n_frames ? f0 + 64 : f0, xn, aln);
        const int cnt = min(64, a.n_frames - f0);
        float res = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (j >= cnt) break;   // (uniform)
            tuning = tuning + ab_readlane(al, j) * (ab_readlane(x, j) - tuning);
            res = lane == j ? tuning : res;
        }
        if (a.o.tuning_grid_inaccuracy && lane < cnt) a.o.tuning_grid_inaccuracy[(size_t)s * a.n_frames + f0 + lane] = res;
        x = xn;
        al = aln;
    }
    if (lane == 0) a.tuning[s] = tuning;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
AnalysisBatch::~AnalysisBatch() {
    if (device_id_ >= 0) (void)hipSetDevice(device_id_);
    for (float* p : {d_smoothed_, d_calm_, d_released_, d_afterglow_, d_peakfiltered_, d_pitch_acc_, d_pitch_dev_, d_scene_, d_tuning_, d_lnf_})
        if (p) (void)hipFree(p);
    if (d_times_) (void)hipFree(d_times_);
    if (d_tab_) (void)hipFree(d_tab_);
    if (d_frames_) (void)hipFree(d_frames_);
    if (d_raw_) (void)hipFree(d_raw_);
}

pvq_status AnalysisBatch::frames_buffer(size_t bytes, float** out) {
    PVQ_HIP(hipSetDevice(device_id_));
    if (frames_cap_ < bytes) {
        if (d_frames_) PVQ_HIP(hipFree(d_frames_));   // (synchronises the device: nothing still reads the old buffer)
        d_frames_ = nullptr;
        frames_cap_ = 0;
        PVQ_HIP(hipMalloc(&d_frames_, bytes));
        frames_cap_ = bytes;
    }
    *out = static_cast<float*>(d_frames_);
    return PVQ_OK;
}

pvq_status AnalysisBatch::create(int device_id, const VqtRange& range, const FullAnalysisParameters& params, uint32_t n_streams,
                                 std::unique_ptr<AnalysisBatch>& out) {
    out.reset();
    if (device_id < 0) {
        set_last_error("the batched AnalysisState runs on a GPU; there is no CPU fallback (the host AnalysisState is the single-stream face)");
        return PVQ_ERR_NO_DEVICE;
    }
    const uint32_t n = range.n_buckets();
    if (!(range.min_freq > 0.0f) || range.octaves == 0 || range.buckets_per_octave == 0 || n_streams == 0) {
        set_last_error("invalid VqtRange or zero streams");
        return PVQ_ERR_INVALID_ARG;
    }
    if (n < 3 || n > 1024) {
        set_last_error("unsupported: the batched AnalysisState takes 3 .. 1024 bins");
        return PVQ_ERR_UNSUPPORTED;
    }
    PVQ_HIP(hipSetDevice(device_id));
    std::unique_ptr<AnalysisBatch> b(new AnalysisBatch());
    b->device_id_ = device_id;
    b->range_ = range;
    b->params_ = params;
    b->n_streams_ = n_streams;
    const size_t per = (size_t)n_streams * n * sizeof(float);
    for (float** p : {&b->d_smoothed_, &b->d_calm_, &b->d_released_, &b->d_afterglow_, &b->d_peakfiltered_, &b->d_pitch_acc_, &b->d_pitch_dev_}) {
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(p), per));
        PVQ_HIP(hipMemset(*p, 0, per));   // analysis.rs:192-241: every EMA starts at 0
    }
    for (float** p : {&b->d_scene_, &b->d_tuning_}) {
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(p), n_streams * sizeof(float)));
        PVQ_HIP(hipMemset(*p, 0, n_streams * sizeof(float)));
    }
    VqtParameters vp;
    vp.range = range;
    std::vector<float> lnf;
    bin_log_frequencies(vp, lnf);
    PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_lnf_), lnf.size() * sizeof(float)));
    PVQ_HIP(hipMemcpy(b->d_lnf_, lnf.data(), lnf.size() * sizeof(float), hipMemcpyHostToDevice));
    out = std::move(b);
    return PVQ_OK;
}

void AnalysisBatch::update_vqt_smoothing_duration(bool has_duration, Duration d) {
    params_.vqt_smoothing_duration_base = has_duration ? d : Duration::from_millis(0);
    smooth_has_ = has_duration;
}

pvq_status AnalysisBatch::preprocess_device(const float* d_db, size_t n_frames, Duration frame_time, const uint64_t* frame_times_ns,
                                            const AnalysisBatchOutputs& outs, hipStream_t stream) {
    if (n_frames == 0) return PVQ_OK;
    if (!d_db) {
        set_last_error("null frame pointer");
        return PVQ_ERR_INVALID_ARG;
    }
    if ((outs.center != nullptr) != (outs.size != nullptr) || (outs.center && outs.max_peaks == 0)) {
        set_last_error("center and size go together, with max_peaks > 0");
        return PVQ_ERR_INVALID_ARG;
    }
    if (n_frames > 0x7FFFFFFFull) {
        set_last_error("too many frames in one call");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    AbArgs a{};
    a.db = d_db;
    a.n_streams = (int)n_streams_;
    a.n_frames = (int)n_frames;
    a.n_bins = (int)range_.n_buckets();
    a.bpo = (int)range_.buckets_per_octave;
    a.octaves = (int)range_.octaves;
    a.min_freq = range_.min_freq;
    a.peak_prom = params_.peak_config.min_prominence;
    a.peak_h = params_.peak_config.min_height;
    a.bass_prom = params_.bassline_peak_config.min_prominence;
    a.bass_h = params_.bassline_peak_config.min_height;
    a.highest_bassnote = (int)params_.highest_bassnote;
    a.harm_thr = params_.harmonic_threshold;
    a.base_ms = params_.vqt_smoothing_duration_base.as_millis();
    a.smooth_has = smooth_has_ ? 1 : 0;
    a.calm_min = params_.vqt_smoothing_calmness_min;
    a.calm_max = params_.vqt_smoothing_calmness_max;
    a.note_ns = params_.note_calmness_smoothing_duration.ns;
    a.scene_ns = params_.scene_calmness_smoothing_duration.ns;
    a.tuning_ns = params_.tuning_inaccuracy_smoothing_duration.ns;
    a.frame_ns = frame_time.ns;
    a.frame_times = nullptr;
    if (frame_times_ns) {
        if (times_cap_ < n_frames) {
            if (d_times_) PVQ_HIP(hipFree(d_times_));
            d_times_ = nullptr;
            times_cap_ = 0;
            PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&d_times_), n_frames * sizeof(unsigned long long)));
            times_cap_ = n_frames;
        }
        PVQ_HIP(hipMemcpyAsync(d_times_, frame_times_ns, n_frames * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
        a.frame_times = d_times_;
    }
    // EMA weights from the host's libm (util.rs:106-110: alpha = 1 - exp(-2 timestep / time_horizon), all f32).  The per-bin
    // horizons of analysis.rs:301-323 are whole milliseconds between 0 and base * 1.5 * max(calmness multiplier): one table row per
    // distinct frame time of the call covers every bin of every frame, built with the very expf the host AnalysisState calls — the
    // device's own evaluation (a double exp rounded once) differs from it by an ulp now and then.
    a.alpha_tab = nullptr;
    a.frame_row = nullptr;
    a.tab_n = a.tab_stride = 0;
    a.tab_lds = 0;
    {
        const float cmax = std::max(std::max(a.calm_min, a.calm_max), 0.0f);
        const double hmax_d = (double)a.base_ms * 1.5 * (double)cmax + 2.0;
        const size_t tab_n = smooth_has_ && hmax_d < 65536.0 ? (size_t)hmax_d + 1 : 1;
        std::vector<uint64_t> distinct;
        std::vector<uint32_t> rows;
        if (frame_times_ns) {
            rows.resize(n_frames);
            std::unordered_map<uint64_t, uint32_t> seen;
            for (size_t f = 0; f < n_frames; ++f) {
                auto it = seen.find(frame_times_ns[f]);
                if (it == seen.end()) {
                    it = seen.emplace(frame_times_ns[f], (uint32_t)distinct.size()).first;
                    distinct.push_back(frame_times_ns[f]);
                }
                rows[f] = it->second;
            }
        } else {
            distinct.push_back(frame_time.ns);
        }
        const size_t stride = 4 + tab_n;
        if (distinct.size() * stride <= ((size_t)1 << 22)) {   // at most 16 MB of weights; beyond that the kernel evaluates them itself
            std::vector<float> tab(distinct.size() * stride);
            for (size_t r = 0; r < distinct.size(); ++r) {
                const Duration ts{distinct[r]};
                float* row = tab.data() + r * stride;
                auto alpha = [&](Duration horizon) { return ema_alpha(ts, horizon); };   // (analysis_host.cpp: what EmaMeasurement::update_with_timestep calls)
                row[0] = alpha(params_.note_calmness_smoothing_duration);
                row[1] = alpha(params_.scene_calmness_smoothing_duration);
                row[2] = alpha(params_.tuning_inaccuracy_smoothing_duration);
                row[3] = 0.0f;
                for (size_t h = 0; h < tab_n; ++h) row[4 + h] = alpha(Duration::from_millis(h));
            }
            // the table of the last call is reused as it stands when nothing it depends on changed (the usual case: one frame time for
            // every call) — no upload, no wait; otherwise the stream is drained first: an earlier call may still read the buffer
            const size_t need = tab.size() * sizeof(float) + rows.size() * sizeof(uint32_t);
            const bool same = rows.empty() && d_tab_ && tab == tab_host_;
            if (!same) {
                PVQ_HIP(hipStreamSynchronize(stream));
                if (tab_cap_ < need) {
                    PVQ_HIP(hipDeviceSynchronize());   // (a call on another stream may read it too)
                    if (d_tab_) PVQ_HIP(hipFree(d_tab_));
                    d_tab_ = nullptr;
                    tab_cap_ = 0;
                    tab_host_.clear();
                    PVQ_HIP(hipMalloc(&d_tab_, need));
                    tab_cap_ = need;
                }
                // (synchronous copies from pageable host memory)
                PVQ_HIP(hipMemcpy(d_tab_, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
                tab_host_ = rows.empty() ? tab : std::vector<float>();
            }
            a.alpha_tab = static_cast<const float*>(d_tab_);
            a.tab_n = (int)tab_n;
            a.tab_stride = (int)stride;
            a.tab_lds = rows.empty() && stride <= 4096 ? (int)stride : 0;   // (16 KB of LDS at most)
            if (!rows.empty()) {
                uint32_t* d_rows = reinterpret_cast<uint32_t*>(static_cast<char*>(d_tab_) + tab.size() * sizeof(float));
                PVQ_HIP(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                a.frame_row = d_rows;
            }
        }
    }
    a.lnf = d_lnf_;
    a.log2_min_freq = std::log2(range_.min_freq);
    {   // peak_detection.rs:37, :45 and calmness.rs:37, as the batched peak kernels derive them
        const float dist_f = std::round((float)a.bpo * 0.4f / 12.0f);
        a.dist = dist_f > 0.0f ? (int)dist_f : 0;
        a.min_bin = ((a.bpo / 12) + 1) / 2;
        a.radius = a.bpo / 12 / 3;
    }
    a.smoothed = d_smoothed_; a.calm = d_calm_; a.released = d_released_; a.afterglow = d_afterglow_;
    a.peakfiltered = d_peakfiltered_; a.pitch_acc = d_pitch_acc_; a.pitch_dev = d_pitch_dev_;
    a.scene = d_scene_; a.tuning = d_tuning_;
    a.o = outs;
    const int npad = (a.n_bins + 63) / 64 * 64;
    a.scratch_bytes = (unsigned)((std::max(peaks_scratch_bytes(a.n_bins, a.dist), peaks_lean_scratch_bytes(a.n_bins, a.dist)) + 15) / 16 * 16);
    a.wave_bytes = ab_wave_bytes(a.n_bins, a.scratch_bytes);
    a.generic_peaks = dev_knob("PVQ_AB_GENERIC", 0);
    a.fold_min = dev_knob("PVQ_AB_FOLD", 24);
    {   // the raw frames' peaks (calmness.rs:40: the general configuration on the whole frame), every frame of every stream at once;
        // workspace of the call: the masks, the frames' tuning inputs, the smoothed rows unless the caller takes them, the frame kernels' redo flags
        const size_t rows = (size_t)n_streams_ * n_frames, words = (size_t)(a.n_bins + 31) / 32;
        const size_t b_mask = rows * words * sizeof(uint32_t), b_tun = rows * sizeof(float), b_sm = outs.x_vqt_smoothed ? 0 : rows * (size_t)a.n_bins * sizeof(float);
        const size_t need = b_mask + b_tun + b_sm + rows;
        if (raw_cap_ < need) {
            if (d_raw_) PVQ_HIP(hipFree(d_raw_));   // (synchronises the device: nothing still reads the old buffer)
            d_raw_ = nullptr;
            raw_cap_ = 0;
            PVQ_HIP(hipMalloc(&d_raw_, need));
            raw_cap_ = need;
        }
        char* wsb = static_cast<char*>(d_raw_);
        a.raw_mask = reinterpret_cast<const uint32_t*>(wsb);
        a.tuning_in = reinterpret_cast<float*>(wsb + b_mask);
        a.sm_rows = outs.x_vqt_smoothed ? outs.x_vqt_smoothed : reinterpret_cast<float*>(wsb + b_mask + b_tun);
        PeakParamsDev pg{};
        pg.n_bins = a.n_bins; pg.bpo = a.bpo; pg.min_freq = a.min_freq; pg.lnf = a.lnf;
        pg.peak_min_prominence = a.peak_prom; pg.peak_min_height = a.peak_h;
        pg.bass_min_prominence = a.peak_prom; pg.bass_min_height = a.peak_h;
        pg.highest_bassnote = a.highest_bassnote; pg.harmonic_threshold = a.harm_thr;
        pg.dist = a.dist; pg.min_bin = a.min_bin;
        pg.mask = reinterpret_cast<uint32_t*>(wsb);
        pg.count = nullptr; pg.center = nullptr; pg.size = nullptr; pg.max_peaks = 0;
        pvq_status ps = launch_peaks_frames(d_db, rows, pg, reinterpret_cast<uint8_t*>(wsb + b_mask + b_tun + b_sm), stream);
        if (ps != PVQ_OK) return ps;
    }
    const bool dist = a.dist > 1;
    // 1. the recurrence, a wave per stream
    {
        const size_t lds = (size_t)ab_rec_wave_bytes(a.n_bins) * 4 + (size_t)((a.tab_lds * 4 + 15) / 16 * 16);
        auto launch = [&](auto kern) -> pvq_status {
            PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((n_streams_ + 3) / 4), dim3(256), lds, stream, a);
            return PVQ_OK;
        };
        // bins per lane: the smallest instantiation that holds the frame
#define PVQ_AB_REC(NK) (a.n_bins > 64 * (NK - 1) ? launch(ab_recurrence<NK, true>) : launch(ab_recurrence<NK, false>))
        pvq_status lst = a.n_bins <= 256 ? PVQ_AB_REC(4) : a.n_bins <= 384 ? PVQ_AB_REC(6) : a.n_bins <= 512 ? PVQ_AB_REC(8)
                         : a.n_bins <= 640 ? PVQ_AB_REC(10) : a.n_bins <= 768 ? PVQ_AB_REC(12) : PVQ_AB_REC(16);
#undef PVQ_AB_REC
        if (lst != PVQ_OK) return lst;
    }
    // 2. everything that hangs on a frame's smoothed row alone, a wave per (stream, frame)
    {
        const size_t rows = (size_t)n_streams_ * n_frames;
        const size_t lds = (size_t)2 * npad * sizeof(float) + (size_t)a.wave_bytes * 4;
        if ((rows + 3) / 4 > 0x7fffffffull) {
            set_last_error("analysis batch: more than 2^33 frames in one call");
            return PVQ_ERR_INVALID_ARG;
        }
        const unsigned grid = (unsigned)((rows + 3) / 4);
        auto launch = [&](auto kern) -> pvq_status {
            PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
            return PVQ_OK;
        };
#define PVQ_AB_LAUNCH(NK) (a.n_bins > 64 * (NK - 1) ? (dist ? launch(ab_frames<NK, true, true>) : launch(ab_frames<NK, false, true>)) \
                                                   : (dist ? launch(ab_frames<NK, true, false>) : launch(ab_frames<NK, false, false>)))
        pvq_status lst = a.n_bins <= 256 ? PVQ_AB_LAUNCH(4) : a.n_bins <= 384 ? PVQ_AB_LAUNCH(6) : a.n_bins <= 512 ? PVQ_AB_LAUNCH(8)
                         : a.n_bins <= 640 ? PVQ_AB_LAUNCH(10) : a.n_bins <= 768 ? PVQ_AB_LAUNCH(12) : PVQ_AB_LAUNCH(16);
#undef PVQ_AB_LAUNCH
        if (lst != PVQ_OK) return lst;
    }
    // 3. the tuning inaccuracy's EMA, a thread per stream
    hipLaunchKernelGGL(ab_tuning, dim3((n_streams_ + 3) / 4), dim3(256), 0, stream, a);
    pvq_status lst = PVQ_OK;
    if (lst != PVQ_OK) return lst;
    PVQ_HIP(hipGetLastError());
    return PVQ_OK;
}

pvq_status AnalysisBatch::get_field(uint32_t stream_index, int field, float* out) {
    if (stream_index >= n_streams_ || !out) {
        set_last_error("stream index out of range or null output");
        return PVQ_ERR_INVALID_ARG;
    }
    const float* src = nullptr;
    switch (field) {
        case PVQ_FIELD_X_VQT_SMOOTHED: src = d_smoothed_; break;
        case PVQ_FIELD_X_VQT_PEAKFILTERED: src = d_peakfiltered_; break;
        case PVQ_FIELD_X_VQT_AFTERGLOW: src = d_afterglow_; break;
        case PVQ_FIELD_CALMNESS: src = d_calm_; break;
        case PVQ_FIELD_PITCH_ACCURACY: src = d_pitch_acc_; break;
        case PVQ_FIELD_PITCH_DEVIATION: src = d_pitch_dev_; break;
        default: set_last_error("unknown field"); return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    PVQ_HIP(hipDeviceSynchronize());
    const size_t n = range_.n_buckets();
    PVQ_HIP(hipMemcpy(out, src + (size_t)stream_index * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return PVQ_OK;
}

pvq_status AnalysisBatch::get_scalars(uint32_t stream_index, float* scene_calmness, float* tuning_grid_inaccuracy) {
    if (stream_index >= n_streams_) {
        set_last_error("stream index out of range");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    PVQ_HIP(hipDeviceSynchronize());
    if (scene_calmness) PVQ_HIP(hipMemcpy(scene_calmness, d_scene_ + stream_index, sizeof(float), hipMemcpyDeviceToHost));
    if (tuning_grid_inaccuracy) PVQ_HIP(hipMemcpy(tuning_grid_inaccuracy, d_tuning_ + stream_index, sizeof(float), hipMemcpyDeviceToHost));
    return PVQ_OK;
}

}  // namespace pvq
