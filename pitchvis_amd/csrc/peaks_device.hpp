// peaks_device.hpp — per-wavefront peak / note detection on one dB frame held in LDS.
//
// Restates, for one frame and one 64-lane wave:
//   find_peaks (analysis_modules/peak_detection.rs:26-51; find_peaks 0.1.5 PeakFinder: plateau-aware
//     strict local maxima, inclusive height / prominence bounds, optional min_distance),
//   the bass / general split of AnalysisState::preprocess (analysis.rs:332-349),
//   enhance_peaks_continuous (peak_detection.rs:61-148) and
//   promote_bass_peaks_with_harmonics (peak_detection.rs:172-241).
// Peak membership is decided with exactly the reference's f32 comparisons, so the index set is
// bit-identical to the CPU path for the same dB frame.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace pvq {

struct PeakParamsDev {
    int n_bins;
    int bpo;
    float min_freq;
    const float* lnf;  // ln(f_k) per bin, host libm (peak_detection.rs:81-86)
    float peak_min_prominence, peak_min_height;
    float bass_min_prominence, bass_min_height;
    int highest_bassnote;
    float harmonic_threshold;
    int dist;     // round(bpo*0.4/12), peak_detection.rs:37
    int min_bin;  // ceil((bpo/12)/2), peak_detection.rs:45
    // outputs (any may be null)
    uint32_t* mask;   // [n_frames][ceil(n_bins/32)]
    uint32_t* count;  // [n_frames]
    float* center;    // [n_frames][max_peaks]
    float* size;
    uint32_t max_peaks;
};

// bytes of LDS scratch one wave needs besides the frame itself
__host__ __device__ inline size_t peaks_scratch_bytes(int n_bins, int dist) {
    const size_t n = (size_t)((n_bins + 63) / 64 * 64);
    return n /*cand*/ + n /*plist: n/2 u16*/ + (dist > 1 ? n /*keep*/ + 2 * n /*compacted candidate list: n/2 + 1 u16*/ : 0);
}

__device__ __forceinline__ float pk_clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// scipy-style prominence test for ONE candidate peak, evaluated by the whole wave (the candidate
// bin ci, its height h and the bound P are wave-uniform).  With the reference's exact f32 test: a
// side passes when, walking away from the peak, a sample v with fl(h - v) >= P is met before the
// first sample strictly higher than h (the subtraction is monotone in v, so this is equivalent to
// fl(h - min_over_the_walk) >= P).  Each 64-bin chunk is classified with two ballots and resolved
// with bit arithmetic on the masks: no per-lane serial walk, no LDS latency chain.
template <int NK>
__device__ __forceinline__ float pk_sel(const float (&v)[NK], int k) {
    float r = v[0];
#pragma unroll
    for (int j = 1; j < NK; ++j) r = (k == j) ? v[j] : r;
    return r;
}

// v[k] = x[lane + 64 k]: the whole frame lives in registers, so the candidate loop below has no
// memory access at all; each step is two compares, two ballots and a few scalar bit operations.
template <int NK>
__device__ __forceinline__ bool pk_prom_ok_wave(const float (&v)[NK], int n, int ci, float h, float P, int lane) {
    bool ok = false;
    for (int k = ci >> 6; k >= 0; --k) {  // left side: nearest = highest set bit
        const int q = (k << 6) + lane;
        const float vv = pk_sel<NK>(v, k);
        const bool in = q < ci;
        const unsigned long long mH = __ballot(in && vv > h);
        const unsigned long long mL = __ballot(in && (h - vv >= P));
        if (mH | mL) {
            ok = mL > mH;  // the low sample is nearer than the higher one <=> its top bit is higher
            break;
        }
    }
    if (!ok) return false;
    ok = false;
    const int kmax = (n + 63) >> 6;
    for (int k = ci >> 6; k < kmax; ++k) {  // right side: nearest = lowest set bit
        const int q = (k << 6) + lane;
        const float vv = pk_sel<NK>(v, k);
        const bool in = q > ci && q < n;
        const unsigned long long mH = __ballot(in && vv > h);
        const unsigned long long mL = __ballot(in && (h - vv >= P));
        if (mH | mL) {
            ok = mL != 0 && (mH == 0 || (mL & (0 - mL)) < (mH & (0 - mH)));
            break;
        }
    }
    return ok;
}

// Windowed, lane-parallel form of the same test: every candidate lane inspects the S samples next to
// its peak on one side (independent LDS reads, no chain) and records where a higher / a low-enough
// sample sits; the nearest decisive sample settles the side.  Returns 1 = side passes, 2 = side
// fails (higher sample or frame edge first), 0 = undecided within S samples (resolved by the
// wave-cooperative test above, which is exact for any distance).
template <int S>
__device__ __forceinline__ int pk_side_window(const float* x, int i, int n, int dir, float h, float P) {
    unsigned mh = 0u, ml = 0u;
#pragma unroll
    for (int s = 1; s <= S; ++s) {
        const int q = i + dir * s;
        const bool in = (q >= 0) && (q < n);
        const float v = x[in ? q : i];
        const bool hi = !in || v > h;
        const bool lo = in && (h - v >= P);
        mh |= (hi ? 1u : 0u) << s;
        ml |= (lo ? 1u : 0u) << s;
    }
    const unsigned any = mh | ml;
    if (!any) return 0;
    const unsigned first = any & (0u - any);
    return (ml & first) ? 1 : 2;
}

// enhance_peaks_continuous for one peak (peak_detection.rs:61-148): log-frequency parabola through the peak and its
// neighbours, centre clamped between them, size interpolated at the centre
__device__ __forceinline__ void pk_enhance(const float* x, int p, const PeakParamsDev& a, float& ctr, float& sz) {
#pragma clang fp contract(off)
    const int nb = a.n_bins;
    const float bpo = (float)a.bpo;
    if (p < 1 || p > nb - 2) {
        ctr = (float)p;
        sz = x[p];
    } else {
        const float l0 = a.lnf[p - 1], l1 = a.lnf[p], l2 = a.lnf[p + 1];
        const float a0 = x[p - 1], a1 = x[p], a2 = x[p + 1];
        const float denom = (l0 - l1) * (l0 - l2) * (l1 - l2);
        if (fabsf(denom) < 1.1920929e-07f) {
            ctr = (float)p;
            sz = x[p];
        } else {
            const float qa = (l2 * (a1 - a0) + l0 * (a2 - a1) + l1 * (a0 - a2)) / denom;
            const float qb = ((l2 * l2) * (a0 - a1) + (l0 * l0) * (a1 - a2) + (l1 * l1) * (a2 - a0)) / denom;
            const float lfp = (fabsf(qa) < 1.1920929e-07f) ? l1 : pk_clampf(-qb / (2.0f * qa), l0, l2);
            const float f_peak = expf(lfp);
            const float est = bpo * log2f(f_peak / a.min_freq);
            const float cc = pk_clampf(est, 0.0f, (float)nb - 1.0f);
            const int lower = (int)floorf(cc);
            const int upper = min(lower + 1, nb - 1);
            const float fract = cc - truncf(cc);
            ctr = cc;
            sz = fmaxf(x[lower] * (1.0f - fract) + x[upper] * fract, 0.0f);
        }
    }
}

// promote_bass_peaks_with_harmonics for one peak whose centre is not above highest_bassnote (peak_detection.rs:172-241)
__device__ __forceinline__ void pk_promote(const float* x, const PeakParamsDev& a, float ctr, float& sz) {
#pragma clang fp contract(off)
    const int nb = a.n_bins;
    const float bpo = (float)a.bpo;
    const float f0 = a.min_freq * exp2f(ctr / bpo);            // 2^(c/bpo)
    const float p0 = exp2f((sz / 10.0f) * 3.32192809488736f);  // 10^(dB/10)
    float score = 0.0f;
    const float wts[4] = {0.5f, 0.3f, 0.15f, 0.05f};
#pragma unroll
    for (int h = 2; h <= 5; ++h) {
        const float hf = f0 * (float)h;
        if (hf >= a.min_freq) {
            const float hb = (log2f(hf) - log2f(a.min_freq)) * bpo;
            if (hb >= 0.0f && hb < (float)nb) {
                const int lo = (int)floorf(hb);
                const int hi = min((int)ceilf(hb), nb - 1);
                const float frac = hb - truncf(hb);
                const float adb = (lo == hi) ? x[lo] : (x[lo] * (1.0f - frac) + x[hi] * frac);
                const float hp = exp2f((adb / 10.0f) * 3.32192809488736f);
                if (hp > p0 * a.harmonic_threshold) score += hp * wts[h - 2];
            }
        }
    }
    if (score > 0.0f) {
        const float boost = fminf(1.0f + 0.5f * (score / fmaxf(p0, 1e-6f)), 1.5f);
        sz += 10.0f * log10f(boost);
    }
}

// enhance + promote for one peak
__device__ __forceinline__ void pk_refine(const float* x, int p, const PeakParamsDev& a, float& ctr, float& sz) {
    pk_enhance(x, p, a, ctr, sz);
    if (!(ctr > (float)a.highest_bassnote)) pk_promote(x, a, ctr, sz);
}

// scipy-style greedy distance suppression among the candidates with x >= min_height (only for dist > 1, e.g. 84 bins per
// octave), evaluated by the whole wave.  Sequentially (find_peaks / scipy _select_by_peak_distance), candidates
// are visited from the highest to the lowest (ties: the later position first) and a visited candidate that is still
// kept removes every other candidate closer than `dist`.  Equivalently, a candidate is kept iff no higher-priority
// candidate within `dist` is kept: decided in rounds — a candidate whose higher-priority neighbours are all decided
// becomes kept (none of them kept) or removed (one of them kept).  Each round settles at least the highest
// undecided candidate of every neighbourhood; a noisy frame needs a handful of rounds.
// keep[i]: 0 = no candidate / removed, 1 = kept, 2 = undecided (on return only 0 / 1).
// The candidates (about a third of the bins) are compacted into `list` first, so a round costs ceil(candidates / 64)
// steps instead of ceil(bins / 64).  States are updated in place: they only ever go 2 -> 0 or 2 -> 1, "removed" needs a
// kept higher-priority neighbour and "kept" needs all of them removed — both verdicts stay true whatever happens later, so
// it does not matter whether a lane sees a neighbour's state from before or after this step.
// list: room for n / 2 + 1 u16 (strict local maxima and plateau tops are never adjacent).
__device__ __forceinline__ void pk_distance_rounds(const float* x, int n, int dist, uint8_t* keep, const uint16_t* list, int n_list, int lane);
__device__ __forceinline__ void pk_distance_wave(const float* x, int n, const uint8_t* cand, float min_height, int dist, uint8_t* keep,
                                                 uint16_t* list, int lane) {
    int n_list = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool c = i < n && cand[i] && x[i] >= min_height;
        if (i < n) keep[i] = c ? 2 : 0;
        const unsigned long long bm = __ballot(c);
        if (c) list[n_list + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)i;
        n_list += __popcll(bm);
    }
    pk_distance_rounds(x, n, dist, keep, list, n_list, lane);
}
// the rounds alone: keep[i] = 2 for the candidates (compacted in `list`, ascending), 0 elsewhere
__device__ __forceinline__ void pk_distance_rounds(const float* x, int n, int dist, uint8_t* keep, const uint16_t* list, int n_list, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // branch-free rounds: the candidates' lanes run the same instruction stream whatever their neighbours are (per-lane
    // `continue`s cost a pair of exec-mask updates each: the scalar unit was the busiest part of the peak kernel at 84 bins
    // per octave)
    for (int round = 0; round < n; ++round) {   // terminates long before: every round decides at least one candidate
        bool any = false;
        for (int k0 = 0; k0 < n_list; k0 += 64) {
            const int k = k0 + lane;
            const bool valid = k < n_list;
            const int i = valid ? (int)list[k] : 0;
            const bool und = valid && keep[i] == 2;
            const float h = x[i];
            bool blocked = false, killed = false;
            for (int d = 1; d < dist; ++d) {
#pragma unroll
                for (int sgn = -1; sgn <= 1; sgn += 2) {
                    const int j = i + sgn * d;
                    const bool inb = j >= 0 && j < n;
                    const int jc = inb ? j : i;
                    const uint8_t sj = keep[jc];
                    const float hj = x[jc];
                    const bool hp = inb && sj != 0 && (hj > h || (hj == h && j > i));   // a live candidate of higher priority
                    killed |= hp && sj == 1;
                    blocked |= hp && sj == 2;
                }
            }
            const uint8_t verdict = killed ? 0 : (blocked ? 2 : 1);
            if (und) keep[i] = verdict;
            any |= und && verdict == 2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!__ballot(any)) break;
    }
}

// The same rule for dist <= 4 (every geometry up to 134 bins per octave; 84 gives 3), in registers.  Peak positions are never adjacent,
// so within a distance below 4 a candidate has at most ONE other candidate on each side — its neighbours in the compacted (ascending)
// list.  A lane owns list entry 64 c + lane of every chunk c; the neighbours' positions and heights are fetched once by two DPP
// wave shifts (the wave's end lanes from the next chunk by readlane), "that neighbour is within reach and outranks me" becomes two
// booleans, and a round is two shifts of the states and a handful of compares per chunk — no LDS traffic, no fences.  (The rounds
// over LDS cost 4 probes x 2 reads per candidate and round.)  The candidate test that follows the rule (height, and the prominence the
// frame's minimum allows at all) runs on the same registers: per list chunk, not per 64 bins of the frame.
__device__ __forceinline__ int pk_from_below(int v, int lane, int carry) {   // lane l <- lane l - 1; lane 0 <- carry (uniform)
    const int s = __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return lane == 0 ? carry : s;
}
__device__ __forceinline__ int pk_from_above(int v, int lane, int carry) {   // lane l <- lane l + 1; lane 63 <- carry (uniform)
    const int s = __builtin_amdgcn_update_dpp(0, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    return lane == 63 ? carry : s;
}
template <int NC, typename Thr>   // chunks of 64 list entries: n_list <= 64 NC
__device__ __forceinline__ uint32_t pk_distance_regs(const float* x, int dist, const uint16_t* list, int n_list, int lane, Thr thr, float fmin, uint16_t* out,
                                                     int dump) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float NINF = -__builtin_huge_valf();
    int p[NC], st[NC];
    float h[NC];
    const int nc = (n_list + 63) >> 6;   // chunks that hold a candidate at all (uniform): the others are skipped
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        p[c] = 1 << 20;
        h[c] = NINF;
        st[c] = 0;
        if (c >= nc) continue;
        const int k = (c << 6) + lane;
        const bool valid = k < n_list;
        p[c] = valid ? (int)list[valid ? k : 0] : (1 << 20);
        h[c] = valid ? x[valid ? p[c] : 0] : NINF;
        st[c] = valid ? 2 : 0;
    }
    bool L[NC], R[NC];   // the list neighbour below / above lies within reach and has the higher priority (ties: the later position)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        L[c] = R[c] = false;
        if (c >= nc) continue;
        const int pl = pk_from_below(p[c], lane, c > 0 ? __builtin_amdgcn_readlane(p[c > 0 ? c - 1 : 0], 63) : -(1 << 20));
        const int pr = pk_from_above(p[c], lane, c + 1 < NC ? __builtin_amdgcn_readlane(p[c + 1 < NC ? c + 1 : c], 0) : (1 << 21));
        const float hl = __int_as_float(pk_from_below(__float_as_int(h[c]), lane,
                                                      c > 0 ? __builtin_amdgcn_readlane(__float_as_int(h[c > 0 ? c - 1 : 0]), 63) : __float_as_int(NINF)));
        const float hr = __int_as_float(pk_from_above(__float_as_int(h[c]), lane,
                                                      c + 1 < NC ? __builtin_amdgcn_readlane(__float_as_int(h[c + 1 < NC ? c + 1 : c]), 0) : __float_as_int(NINF)));
        L[c] = (p[c] - pl < dist) & (hl > h[c]);
        R[c] = (pr - p[c] < dist) & (hr >= h[c]);
    }
    for (int round = 0; round < 64 * NC; ++round) {   // (every round decides at least the highest undecided candidate)
        bool any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c >= nc) continue;
            const int sl = pk_from_below(st[c], lane, c > 0 ? __builtin_amdgcn_readlane(st[c > 0 ? c - 1 : 0], 63) : 0);
            const int sr = pk_from_above(st[c], lane, c + 1 < NC ? __builtin_amdgcn_readlane(st[c + 1 < NC ? c + 1 : c], 0) : 0);
            const bool killed = (L[c] & (sl == 1)) | (R[c] & (sr == 1));
            const bool blocked = (L[c] & (sl == 2)) | (R[c] & (sr == 2));
            const int verdict = killed ? 0 : (blocked ? 2 : 1);
            st[c] = st[c] == 2 ? verdict : st[c];
            any |= st[c] == 2;
        }
        if (!__ballot(any)) break;
    }
    // the survivors that also pass the candidate test (thr(bin, H, P): the bin's height / prominence bounds; fmin: the frame's minimum)
    // go back into the list, compacted in place (every entry sits in a register by now); the rest of the frame is never looked at again
    uint32_t n_out = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c >= nc) continue;
        float H, P;
        thr(p[c] < (1 << 20) ? p[c] : 0, H, P);
        const bool pre = (st[c] == 1) & (h[c] >= H) & (h[c] - fmin >= P);
        const unsigned long long bm = __ballot(pre);
        out[pre ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, n_out)) : dump] = (uint16_t)p[c];
        n_out += __popcll(bm);
    }
    return n_out;
}

// x: the frame's dB values in LDS (n_bins <= 64*NK floats, already visible to the whole wave);
// scratch: peaks_scratch_bytes() bytes of LDS private to this wave.  Called by all 64 lanes.
template <int NK>
__device__ __forceinline__ void peaks_wave_nk(const float* x, unsigned char* scratch, size_t frame, const PeakParamsDev& a,
                                              int lane, uint32_t* total_out = nullptr) {   // total_out: the frame's number of peaks (their bins, ascending, are left in scratch + npad as u16)
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    const int words = (n + 31) / 32;
    uint8_t* cand = scratch;
    uint16_t* plist = reinterpret_cast<uint16_t*>(scratch + npad);  // compacted peak bins, ascending
    uint8_t* keep0 = scratch + 2 * npad;

    // frame minimum: a peak of height h can only reach prominence P if fl(h - min) >= P, which
    // rejects the many low local maxima of a noisy frame without walking at all
    float fmin_ = 3.40282347e+38f;
    for (int i = lane; i < npad; i += 64) {
        cand[i] = 0;
        if (i < n) fmin_ = fminf(fmin_, x[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fmin_ = fminf(fmin_, __shfl_xor(fmin_, o));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // plateau-aware strict local maxima; the first and last sample are never peaks
    for (int i = lane; i < n; i += 64) {
        if (i >= 1 && i < n - 1 && x[i - 1] < x[i]) {
            int ia = i + 1;
            while (ia < n - 1 && x[ia] == x[i]) ++ia;
            if (x[ia] < x[i]) cand[(i + ia) >> 1] = 1;  // Peak.position = i..ia, middle_position()
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (a.dist > 1) {
        uint16_t* dl = reinterpret_cast<uint16_t*>(keep0 + npad);   // compacted candidate list
        // find_peaks runs its distance rule once per height threshold (bass / general), but one evaluation at the lower threshold
        // serves both: a candidate's fate depends only on candidates of higher priority (greater height), so the candidates
        // between the two thresholds — the lowest of all — never change the verdict of one at or above the higher threshold,
        // and the height test below removes them where they do not belong
        pk_distance_wave(x, n, cand, fminf(a.bass_min_height, a.peak_min_height), a.dist, keep0, dl, lane);
    }
    float v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) v[k] = ((k << 6) + lane < n) ? x[(k << 6) + lane] : 0.0f;
    uint32_t total = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const float xv = pk_sel<NK>(v, base >> 6);
        bool pre = false;  // local maximum that passes every per-bin filter except the prominence walk
        if (i < n && cand[i] && i >= a.min_bin) {
            const bool bass = i <= a.highest_bassnote;  // analysis.rs:338,346
            const float H = bass ? a.bass_min_height : a.peak_min_height;
            const float P = bass ? a.bass_min_prominence : a.peak_min_prominence;
            pre = xv >= H && (a.dist <= 1 || keep0[i]) && (!(P > 0.0f) || (xv - fmin_ >= P));
        }
        // most candidates are settled by the 16 samples on either side; the rest go to the exact wave test
        int st = 2;
        if (pre) {
            const float P = (i <= a.highest_bassnote) ? a.bass_min_prominence : a.peak_min_prominence;
            if (!(P > 0.0f)) {
                st = 1;
            } else {
                const int l = pk_side_window<16>(x, i, n, -1, xv, P);
                if (l != 2) {
                    const int r = pk_side_window<16>(x, i, n, 1, xv, P);
                    st = (r == 2) ? 2 : ((l == 1 && r == 1) ? 1 : 0);
                }
            }
        }
        unsigned long long pm = __ballot(st == 1);  // peaks of this chunk (wave-uniform)
        unsigned long long cm = __ballot(st == 0);
        while (cm) {
            const int b = __builtin_ctzll(cm);
            cm &= cm - 1;
            const int ci = base + b;
            const float h = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b));
            const float P = (ci <= a.highest_bassnote) ? a.bass_min_prominence : a.peak_min_prominence;
            if (!(P > 0.0f) || pk_prom_ok_wave<NK>(v, n, ci, h, P, lane)) pm |= 1ull << b;
        }
        const bool is_peak = (pm >> lane) & 1ull;
        if (a.mask) {
            if (lane == 0 && (base >> 5) < words) a.mask[frame * words + (base >> 5)] = (uint32_t)pm;
            if (lane == 1 && (base >> 5) + 1 < words) a.mask[frame * words + (base >> 5) + 1] = (uint32_t)(pm >> 32);
        }
        const uint32_t slot = total + __popcll(pm & ((1ull << lane) - 1ull));
        total += __popcll(pm);
        if (is_peak) plist[slot] = (uint16_t)i;
    }
    if (a.count && lane == 0) a.count[frame] = total;
    if (total_out) *total_out = total;
    if (a.center) {
        // refine all peaks of the frame side by side: one lane per peak
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t lim = total < a.max_peaks ? total : a.max_peaks;
        for (uint32_t sidx = lane; sidx < lim; sidx += 64) {
            float ctr, sz;
            pk_refine(x, (int)plist[sidx], a, ctr, sz);
            a.center[frame * a.max_peaks + sidx] = ctr;
            a.size[frame * a.max_peaks + sidx] = sz;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Lean common-case routine (dist <= 1, no plateau peak in the frame).
// x points at the frame inside an LDS row that carries PK_PAD samples of +INF on both sides, so the
// walk needs no range checks: running off the frame looks like meeting a higher sample, which is the
// reference's edge behaviour.  Each side is scanned over a fixed window with a "poisoned running
// minimum": M = running max; once M exceeds the peak height every later sample counts as +INF;
// m = min of the surviving samples.  The side passes iff fl(h - m) >= P — exactly the reference test
// on the minimum of the samples met before the first higher one.  4 VALU + 1 LDS read per step, no
// masks, two independent chains (left / right).  A lane whose window ends without a decision
// (no low sample, no higher sample within PK_PAD) is settled by the exact wave-cooperative test.
// Returns false, having written nothing, when the frame may contain a plateau peak of three or more samples:
// the caller then runs the generic routine.
// ------------------------------------------------------------------------------------------------
constexpr int PK_PAD = 16;

// bin i (value xv) is the reported position of a peak of one sample (a rise before, a fall after) or of a two-sample
// plateau (i is its second sample: middle_position of the half-open range [i-1, i+1) is i).  x carries +INF sentinels
// on both sides, so the frame edges never qualify.
__device__ __forceinline__ bool pk_is_top(const float* x, int i, float xv) {
    const float l = x[i - 1];
    return (x[i + 1] < xv) && ((l < xv) || (l == xv && x[i - 2] < xv));
}

// LDS scratch of the lean routine besides the frame and its peak list: candidate list (u16), 64 mask words, distance-rule states (u8)
__host__ __device__ inline size_t peaks_lean_scratch_bytes(int n_bins, int dist) {
    const size_t n = (size_t)((n_bins + 63) / 64 * 64);
    return n + 256 + (dist > 4 ? n /*survivors of the distance rule, where it runs over LDS (pk_distance_rounds)*/ : 0);
}

// plist: where the frame's peak bins go (ascending, room for npad / 2 u16; the last slot is a dump slot); n_peaks receives
// their number.  The continuous outputs are NOT produced here: the caller refines the peaks of several frames side by side.
// Written without per-lane branches: a predicated LDS store goes to a dump slot instead (peak positions are never adjacent and
// never the first or last sample, so a frame holds at most npad / 2 - 1 of them and slot npad / 2 - 1 of either list is free),
// range tests that the +INF sentinels around the frame already decide are left out, and the peak mask is gathered with one
// LDS atomic per pass — every `if (lane-dependent) store` costs an exec-mask save / branch / restore on the scalar unit, which
// was as busy as the vector unit here.
// number of set bits of a ballot below this lane, plus `add`: the two v_mbcnt instructions made for it
__device__ __forceinline__ int pk_rank(unsigned long long bm, int add) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, (unsigned)add));
}

// wave-wide minimum by DPP, returned uniform (readlane of the last lane): quads, half rows, rows, then row_bcast across rows.
// (__shfl_xor costs an address computation and an LDS crossbar round trip per step: 50 vector instructions per frame)
__device__ __forceinline__ float pk_wave_min(float v) {
#define PK_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (x)), __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xf, false))
    v = fminf(v, PK_DPP(v, 0xB1, 0xf));    // quad_perm(1,0,3,2)
    v = fminf(v, PK_DPP(v, 0x4E, 0xf));    // quad_perm(2,3,0,1)
    v = fminf(v, PK_DPP(v, 0x141, 0xf));   // row_half_mirror
    v = fminf(v, PK_DPP(v, 0x140, 0xf));   // row_mirror: every lane holds its row's minimum
    v = fminf(v, PK_DPP(v, 0x142, 0xa));   // row_bcast:15 into rows 1 and 3
    v = fminf(v, PK_DPP(v, 0x143, 0xc));   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's minimum
#undef PK_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Per-bin thresholds of the candidate test, lane-constant, kept in LDS (thrH[npad], thrP[npad], filled once per workgroup by
// peaks_lean_thresholds): in registers they are either 2 NK values per lane or — what the compiler makes of the loop-invariant
// comparisons — 64-bit lane masks per chunk and condition that spill from the scalar file; an LDS read costs the busy vector unit nothing.
__device__ __forceinline__ void peaks_lean_thresholds(float* thrH, float* thrP, const PeakParamsDev& a, int tid, int n_threads) {
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    for (int i = tid; i < npad; i += n_threads) {
        const bool bass = i <= a.highest_bassnote;  // analysis.rs:338,346
        const float H = bass ? a.bass_min_height : a.peak_min_height;
        const float P = bass ? a.bass_min_prominence : a.peak_min_prominence;
        thrH[i] = (i >= a.min_bin && i < n) ? H : __builtin_huge_valf();
        thrP[i] = (P > 0.0f) ? P : -__builtin_huge_valf();   // no prominence bound: every difference passes
    }
}

// Part 1 (one frame): peak positions, the distance rule, and the frame's candidates compacted into its clist; n_cand receives
// their number.  Returns false, with n_cand = 0, when the frame goes to the generic routine.
template <int NK, bool DISTANCE>   // DISTANCE: min_distance > 1 (84 bins per octave)
__device__ __forceinline__ bool peaks_lean_scan(const float* x, unsigned char* scratch, const float* thrH, const float* thrP, uint32_t& n_cand_out,
                                                const PeakParamsDev& a, int lane) {
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    const int dump = npad / 2 - 1;
    const float INF = __builtin_huge_valf();
    uint16_t* clist = reinterpret_cast<uint16_t*>(scratch);          // candidates (strict local maxima past the cheap tests), ascending
    uint32_t* maskw = reinterpret_cast<uint32_t*>(scratch + npad);   // the frame's peak mask, one word per lane
    uint8_t* keep0 = scratch + npad + 256;                           // dist > 1 only: survivors of the distance rule (evaluated at the lower height threshold)
    float vt[NK];   // the bin's value where it is a peak position (strict local maximum or second sample of a two-sample plateau), -INF elsewhere
    float fmin_ = INF;
    bool plateau = false;
    const float hmin = fminf(a.bass_min_height, a.peak_min_height);
    int n_dl = 0;
    maskw[lane] = 0;
    // Without two equal neighbours anywhere in the frame (white noise: all but a few frames in 10^5; frames floored by
    // power_to_db do have them) a peak position is simply a rise before and a fall after: the first pass assumes that and
    // notes whether any bin equals its right neighbour; if one does the loop runs again with the full plateau logic.
    bool tie = false;
    constexpr bool SHORT = NK <= 12;   // beyond 768 bins a second copy of the unrolled loop and the LDS thresholds cost more than they save (measured: +18 % at 840 bins)
    auto scan = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        n_dl = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            vt[k] = -INF;
            if ((k << 6) >= n) continue;
            const int i = (k << 6) + lane;
            const float xv = x[i], l = x[i - 1], r = x[i + 1];   // +INF outside the frame
            bool top = (r < xv) & (l < xv);
            if (FULL && !SHORT) fmin_ = fminf(fmin_, xv);
            if (!FULL) {
                fmin_ = fminf(fmin_, xv);
                if ((k << 6) + 64 > n) tie |= (r == xv) & (i < n - 1);   // (uniform; past the frame's end the sentinels equal each other)
                else tie |= r == xv;
            } else {
                const float l2 = x[i - 2], r2 = x[i + 2];
                // a rise followed by two equal samples may start a plateau peak of three or more: leave those frames to the
                // generic code (two-sample plateaus — exact ties of neighbouring bins — are taken here: their middle_position
                // is the second sample).  (i < n - 2: the sentinels equal each other)
                plateau |= (i < n - 2) & (l < xv) & (r == xv) & (r2 == xv);
                // peak position (pk_is_top): a fall after, and a rise before or a two-sample plateau ending here; the sentinels
                // make the frame's first and last sample, and everything past them, fail by themselves
                top = (r < xv) & ((l < xv) | ((l == xv) & (l2 < xv)));
            }
            vt[k] = top ? xv : -INF;
            if (DISTANCE) {   // the distance rule's candidates (peak positions at or above the lower height threshold), compacted on the way
                const bool c = vt[k] >= hmin;
                if (a.dist > 4) keep0[i] = c ? 2 : 0;   // (uniform: the rounds over LDS)
                const unsigned long long bm = __ballot(c);
                clist[c ? pk_rank(bm, n_dl) : dump] = (uint16_t)i;
                n_dl += __popcll(bm);
            }
        }
    };
    if (SHORT) {
        scan(std::false_type{});
        if (__ballot(tie)) scan(std::true_type{});
    } else {
        scan(std::true_type{});
    }
    n_cand_out = 0;
    if (__ballot(plateau)) return false;
    if (DISTANCE)   // find_peaks' distance rule runs before its prominence test, once per height threshold: one evaluation at the
                    // lower threshold serves both (see peaks_wave_nk); clist is free again from step 1 on
        if (a.dist > 4) pk_distance_rounds(x, n, a.dist, keep0, clist, n_dl, lane);
    fmin_ = pk_wave_min(fmin_);
    auto thr = [&](int i, float& H, float& P) {   // analysis.rs:332-349 picks the bass or the general pair by bin
        if (SHORT && thrH) {   // (thrH null: no table in LDS — the list-domain test of the register distance rule reads a handful of thresholds per frame)
            H = thrH[i];
            P = thrP[i];
        } else {
            const bool bass = i <= a.highest_bassnote;
            H = (i >= a.min_bin) ? (bass ? a.bass_min_height : a.peak_min_height) : INF;
            const float Pq = bass ? a.bass_min_prominence : a.peak_min_prominence;
            P = (Pq > 0.0f) ? Pq : -INF;
        }
    };
    if (DISTANCE && a.dist <= 4) {   // (uniform) rule and candidate test on the compacted list, in registers
        n_cand_out = pk_distance_regs<(NK + 1) / 2>(x, a.dist, clist, n_dl, lane, thr, fmin_, clist, dump);
        return true;
    }

    // 1. candidates of the whole frame, compacted: the window walk then runs once per 64 candidates instead of once per 64
    //    bins (a third of the bins are local maxima, far fewer pass height / range): peak position, min_bin, height, and the
    //    prominence the frame's minimum allows at all (analysis.rs:332-349 picks the bass or the general pair by bin)
    uint32_t n_cand = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if ((k << 6) >= n) break;
        const int i = (k << 6) + lane;
        float H, P;
        thr(i, H, P);
        bool pre = (vt[k] >= H) & (vt[k] - fmin_ >= P);
        if (DISTANCE) pre &= keep0[i] != 0;
        const unsigned long long bm = __ballot(pre);
        clist[pre ? pk_rank(bm, (int)n_cand) : dump] = (uint16_t)i;
        n_cand += __popcll(bm);
    }
    n_cand_out = n_cand;
    return true;
}

// Part 2 (the FPW frames of a wave together): prominence of every candidate over a PK_PAD-sample window on both sides.  The
// candidates of the wave's frames are walked side by side — a frame has ~30, so one pass of 64 lanes serves two frames.
// x0 / scratch0 / plist0: frame 0's row, scratch and peak list; the next frame's lie row / scratch_stride / pl_cap further on.
// total[g] receives frame g's number of peaks; mask and count are written for the frames with write[g] set.
template <int NK, int FPW>
__device__ __forceinline__ void peaks_lean_walk(const float* x0, int row, unsigned char* scratch0, int scratch_stride, uint16_t* plist0, int pl_cap,
                                                const uint32_t (&n_cand)[FPW], uint32_t (&total)[FPW], const bool (&write)[FPW],
                                                const size_t (&frame)[FPW], const PeakParamsDev& a, int lane) {
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    const int words = (n + 31) / 32;
    const int dump = npad / 2 - 1;
    const float INF = __builtin_huge_valf();
    uint32_t pre[FPW + 1];
    pre[0] = 0;
#pragma unroll
    for (int g = 0; g < FPW; ++g) {
        pre[g + 1] = pre[g] + n_cand[g];
        total[g] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < pre[FPW]; base += 64) {
        const uint32_t p = base + lane;
        const bool have = p < pre[FPW];
        int g = 0;
#pragma unroll
        for (int q = 1; q < FPW; ++q) g += (p >= pre[q]) ? 1 : 0;
        uint32_t first = 0;
#pragma unroll
        for (int q = 1; q < FPW; ++q) first = (g == q) ? pre[q] : first;
        const float* x = x0 + g * row;
        const uint16_t* clist = reinterpret_cast<const uint16_t*>(scratch0 + g * scratch_stride);
        const int i = have ? (int)clist[p - first] : 0;   // idle lanes sit on the first sample, which is never a peak
        const float xv = x[i];
        const float P = (i <= a.highest_bassnote) ? a.bass_min_prominence : a.peak_min_prominence;
        // pL / pR: a sample higher than the peak has been met on that side (later samples no longer count)
        bool pL = false, pR = false;
        float mL = INF, mR = INF;
        // in blocks of four steps; the walk stops once every lane of the pass has both sides settled (a higher
        // sample met, or the prominence already reached — the minimum only falls, so that verdict is final)
#pragma unroll
        for (int s0 = 1; s0 <= PK_PAD; s0 += 4) {
#pragma unroll
            for (int s = s0; s < s0 + 4; ++s) {
                const float vl = x[i - s], vr = x[i + s];
                pL |= vl > xv;
                pR |= vr > xv;
                mL = fminf(mL, pL ? INF : vl);
                mR = fminf(mR, pR ? INF : vr);
            }
            const bool open = have && (!(pL || (xv - mL >= P)) || !(pR || (xv - mR >= P)));
            if (!__ballot(open)) break;
        }
        const bool noP = !(P > 0.0f);
        const bool okL = noP || (xv - mL >= P), okR = noP || (xv - mR >= P);
        const bool failed = (!okL && pL) || (!okR && pR);
        bool peak = have && okL && okR;
        unsigned long long cm = __ballot(have && !peak && !failed);  // window too short: exact test
        while (cm) {
            const int b = __builtin_ctzll(cm);
            cm &= cm - 1;
            const int ci = __builtin_amdgcn_readlane(i, b);
            const int cg = __builtin_amdgcn_readlane(g, b);
            const float h = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b));
            const float Pc = (ci <= a.highest_bassnote) ? a.bass_min_prominence : a.peak_min_prominence;
            float v[NK];   // that frame, one bin per lane and chunk (+INF past its end: the sentinels)
#pragma unroll
            for (int k = 0; k < NK; ++k) v[k] = ((k << 6) < npad) ? x0[cg * row + (k << 6) + lane] : INF;
            const bool ok = pk_prom_ok_wave<NK>(v, n, ci, h, Pc, lane);
            if (lane == b) peak = ok;
        }
        // peak lists and masks, frame by frame
        uint32_t slot = (uint32_t)dump;
#pragma unroll
        for (int q = 0; q < FPW; ++q) {
            const unsigned long long pm = __ballot(peak && g == q);
            if (peak && g == q) slot = (uint32_t)pk_rank(pm, (int)total[q]);
            total[q] += __popcll(pm);
        }
        plist0[g * pl_cap + (int)slot] = (uint16_t)i;
        uint32_t* maskw = reinterpret_cast<uint32_t*>(scratch0 + g * scratch_stride + npad);
        __hip_atomic_fetch_or(&maskw[i >> 5], peak ? (1u << (i & 31)) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < FPW; ++g) {
        if (!write[g]) continue;   // uniform
        const uint32_t* maskw = reinterpret_cast<const uint32_t*>(scratch0 + g * scratch_stride + npad);
        if (a.mask && lane < words) a.mask[frame[g] * words + lane] = maskw[lane];
        if (a.count && lane == 0) a.count[frame[g]] = total[g];
    }
}

}  // namespace pvq
