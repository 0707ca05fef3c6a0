// vqt_engine.hip — MI355X (gfx950 / CDNA4) kernels and the pvq::Vqt engine.
//
// Hot path being replaced: Vqt::calculate_vqt_instant_in_db (reference
// pitchvis_analysis/src/vqt.rs:866-916), power_to_db (:922-954) and the stateless peak pipeline
// (analysis.rs:332-361, analysis_modules/peak_detection.rs:26-241), batched over hops.
//
// Kernel inventory (wave64, LDS-staged, no compatibility layers):
//   vqt_fft_frames<BLOCK>   one workgroup per frame.  Per window group: gather the window from the
//                           hop stream, an in-place register-staged Stockham FFT (radix 16/8/4/2)
//                           of the packed real window in padded LDS, real-split of only the
//                           columns the sparse kernel reads, banded complex row dots reduced with
//                           16-lane shuffles; then the frame-relative dB epilogue.
//   peaks_frames            one wavefront per frame: plateau-aware local maxima, prominence
//                           walks, bass/general split, sub-bin refinement, bass promotion.
//   (block-DFT path: see vqt_blockdft.hip)
#include "vqt_engine.hpp"

#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "device_tables.hpp"
#include "peaks_device.hpp"

// Every floating-point operation of this file rounds where it is written: no implicit FMA contraction.  The FFT path exists in several forms —
// the walk (vqt_fft_frames, any window at run time, for four thread counts), its group-split launch for few frames, and one kernel per window
// size for batches (vqt_fft_group) — and a frame must have the same bits whichever form computed it (a stream's values do not depend on the size
// of the batch it is analysed in).  Left to contract on its own, the compiler fuses a product into a neighbouring addition or not depending on the
// code around it: the same source, inlined into two kernels, rounded differently in 94 % of the coefficients (1.4e-7 of the frame peak;
// profiles/r05_fft_path.txt).  Where a fused multiply-add is wanted it is written (fmaf): the same instructions in every form, by construction.
#pragma clang fp contract(off)

namespace pvq {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
void set_last_error_noexcept(const char* s) noexcept {
    try {
        g_last_error = s ? s : "";
    } catch (...) {
        g_last_error.clear();
    }
}
const char* get_last_error() { return g_last_error.c_str(); }

#define PVQ_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_last_error(std::string(#call) + " failed: " + hipGetErrorString(e_));              \
            return PVQ_ERR_DEVICE;                                                                 \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {   // two products, two fused multiply-adds
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// LDS padding: one float2 of padding after every 16, so that radix-16 strided writes
// (stride 16 elements = 128 B) land on distinct banks within each 16-lane ds_write_b64 group.
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }

// cos/sin(2*pi*k/16), k = 0..7
__device__ constexpr float kC16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                      0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float kS16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                      1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// In-register forward DFT of R points (natural order in and out), decimation in time.
template <int R>
struct RegFft;

template <int R, int K>
struct RegBfly {
    static __device__ __forceinline__ void run(float2* u, const float2* e, const float2* o) {
        constexpr int idx = K * (16 / R);  // twiddle exp(-2*pi*i*idx/16)
        float2 t;
        if constexpr (idx == 0) {
            t = o[K];
        } else if constexpr (idx == 4) {
            t = make_float2(o[K].y, -o[K].x);  // * (-i)
        } else {
            t = cmul(o[K], make_float2(kC16[idx], -kS16[idx]));
        }
        u[K] = cadd(e[K], t);
        u[K + R / 2] = csub(e[K], t);
        if constexpr (K + 1 < R / 2) RegBfly<R, K + 1>::run(u, e, o);
    }
};

template <>
struct RegFft<1> {
    static __device__ __forceinline__ void run(float2*) {}
};
template <>
struct RegFft<2> {
    static __device__ __forceinline__ void run(float2* u) {
        float2 a = u[0], b = u[1];
        u[0] = cadd(a, b);
        u[1] = csub(a, b);
    }
};
template <int R>
struct RegFft {
    static __device__ __forceinline__ void run(float2* u) {
        float2 e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = u[2 * k];
            o[k] = u[2 * k + 1];
        }
        RegFft<R / 2>::run(e);
        RegFft<R / 2>::run(o);
        RegBfly<R, 0>::run(u, e, o);
    }
};

// One in-place Stockham autosort pass of radix R over N points living in padded LDS.
//   item i in [0, N/R): k = i mod p, u[r] = Z[i + r*N/R] * w_{pR}^{k r}, DFT_R, Z[(i-k)R + k + r p] = u[r]
// Every thread first pulls all of its inputs into registers, the workgroup synchronises, then the
// outputs go back into the same buffer: one LDS buffer serves any N <= E*BLOCK.
template <int R, int BLOCK, int E>
__device__ __forceinline__ void stockham_pass(float2* __restrict__ Z, int N, int p, const float2* __restrict__ tw,
                                              int tw_stride, int tid) {
    constexpr int NB = E / R;  // butterflies a thread may own
    const int T = N / R;
    float2 u[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * BLOCK;
        if (i < T) {
#pragma unroll
            for (int r = 0; r < R; ++r) u[b][r] = Z[lpad(i + r * T)];
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * BLOCK;
        if (i < T) {
            if (p > 1) {
                const int k = i & (p - 1);
                const int base = k * tw_stride;
#pragma unroll
                for (int r = 1; r < R; ++r) u[b][r] = cmul(u[b][r], tw[base * r]);
            }
            RegFft<R>::run(u[b]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * BLOCK;
        if (i < T) {
            const int k = i & (p - 1);
            const int j = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) Z[lpad(j + r * p)] = u[b][r];
        }
    }
    __syncthreads();
}

// Output-pruned pass.  The kernel reads only the spectrum columns 0 .. M of a window group (vqt.rs:725-735 bounds them by the
// decimated Nyquist), i.e. FFT bins 0 .. M and N - M .. N - 1 (the real split pairs c with N - c).  An output o of the pass with
// stride p lands in residue k + r p of o mod p R (k = i mod p), and a later pass only moves it within that residue class of ITS
// p' = p R — so the bins the kernel needs descend from the outputs with k + r p <= M or k + r p >= p R - M.  Once M < p that leaves
// r = 0 (for k <= M) and r = R - 1 (for k >= p - M) of an item, and nothing at all of the items in between: two of R outputs
// computed and written (out[0] = sum u_r, out[R-1] = sum u_r e^{+2 pi i r / R}), the other items not even loaded.  At 48 kHz /
// 252 bins that is the last two of the 16 384-sample window's four passes (an LDS pass moves all N points whatever its radix).
template <int R, int BLOCK, int E>
__device__ __forceinline__ void stockham_pass_ends(float2* __restrict__ Z, int N, int p, const float2* __restrict__ tw, int tw_stride, int tid, int M) {
    constexpr int NB = E / R;
    const int T = N / R;
    float2 lo[NB], hi[NB];
    bool nl[NB], nh[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * BLOCK;
        const int k = i & (p - 1);
        nl[b] = i < T && k <= M;
        nh[b] = i < T && k >= p - M;
        lo[b] = make_float2(0.0f, 0.0f);
        hi[b] = lo[b];
        if (nl[b] || nh[b]) {
            float2 u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = Z[lpad(i + r * T)];
            const int base = k * tw_stride;   // (p > M >= 0: never the first pass, the twiddles apply)
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[base * r]);
            float2 s0 = u[0], s1 = u[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                constexpr int step = 16 / R;
                const int idx = r * step;   // e^{+2 pi i idx / 16}
                const float c = idx < 8 ? kC16[idx] : -kC16[idx - 8], sn = idx < 8 ? kS16[idx] : -kS16[idx - 8];
                s0 = cadd(s0, u[r]);
                s1 = cadd(s1, cmul(u[r], make_float2(c, sn)));
            }
            lo[b] = s0;
            hi[b] = s1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * BLOCK;
        const int k = i & (p - 1);
        const int j = (i - k) * R + k;
        if (nl[b]) Z[lpad(j)] = lo[b];
        if (nh[b]) Z[lpad(j + (R - 1) * p)] = hi[b];
    }
    __syncthreads();
}

// Forward FFT of N = 2^logN complex points in padded LDS (natural order in and out); only the bins 0 .. M and N - M .. N - 1 are
// guaranteed on return (M >= N / 2: all of them).
template <int BLOCK, int E>
__device__ __forceinline__ void lds_fft(float2* Z, int N, const float2* tw, int n_tw, int tid, int M) {
    int p = 1;
    int rem = N;
    while (rem >= 16) {
        if (M < p) stockham_pass_ends<16, BLOCK, E>(Z, N, p, tw, n_tw / (p * 16), tid, M);
        else stockham_pass<16, BLOCK, E>(Z, N, p, tw, n_tw / (p * 16), tid);
        p *= 16;
        rem >>= 4;
    }
    if (rem == 8) {
        if (M < p) stockham_pass_ends<8, BLOCK, E>(Z, N, p, tw, n_tw / (p * 8), tid, M);
        else stockham_pass<8, BLOCK, E>(Z, N, p, tw, n_tw / (p * 8), tid);
    } else if (rem == 4) {
        if (M < p) stockham_pass_ends<4, BLOCK, E>(Z, N, p, tw, n_tw / (p * 4), tid, M);
        else stockham_pass<4, BLOCK, E>(Z, N, p, tw, n_tw / (p * 4), tid);
    } else if (rem == 2) {
        if (M < p) stockham_pass_ends<2, BLOCK, E>(Z, N, p, tw, n_tw / (p * 2), tid, M);
        else stockham_pass<2, BLOCK, E>(Z, N, p, tw, n_tw / (p * 2), tid);
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// vqt.rs:922-954 constants
#define PVQ_REF_POWER (0.3f * 0.3f)
#define PVQ_A_MIN (1e-6f * 1e-6f)
#define PVQ_TOP_DB 60.0f

// Frame-relative dB epilogue shared by both algorithm paths.  xv: n_bins complex coefficients in
// LDS; red: 2*(BLOCK/64) floats of LDS scratch.  Writes n_bins floats to out (global).
// status: the handle's sticky flag word; bit 0 is raised when a coefficient of the frame is not finite (a NaN / Inf sample
// in the frame's windows: the reference's callers drop such input, audio_desktop.rs:102-105, and peak_detection.rs:145
// would panic on it).  T threads (T / 64 waves) share one frame; `red` holds that frame's 2 * T / 64 partial results; the
// barrier is the workgroup's (every frame of a workgroup runs the epilogue at the same time); live = false: a padding
// frame, nothing is stored.
template <int T>
__device__ __forceinline__ void db_epilogue(const float2* xv, float* red, int n_bins, float* __restrict__ out,
                                            float* lds_out, int tid, unsigned* status, bool live = true) {
    const float ref_db = 10.0f * log10f(PVQ_REF_POWER);
    constexpr int NW = T / 64;
    constexpr int PER = 1024 / T < 4 ? 4 : 1024 / T;  // supports n_bins <= 1024 (and <= 4 T)
    float d[PER];
    float mx = -3.40282347e+38f, mn = 3.40282347e+38f;
    bool bad = false;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = tid + t * T;
        d[t] = 0.0f;
        if (k < n_bins) {
            const float2 z = xv[k];
            const float ns = fmaf(z.x, z.x, z.y * z.y);
            bad |= !(ns <= 3.40282347e+38f);
            d[t] = 10.0f * log10f(fmaxf(ns, PVQ_A_MIN)) - ref_db;
            mx = fmaxf(mx, d[t]);
            mn = fminf(mn, d[t]);
        }
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    if (status && live && __builtin_amdgcn_ballot_w64(bad) != 0 && (tid & 63) == 0) atomicOr(status, 1u);
    if ((tid & 63) == 0) {
        red[tid >> 6] = mx;
        red[NW + (tid >> 6)] = mn;
    }
    __syncthreads();
    mx = red[0];
    mn = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        mx = fmaxf(mx, red[w]);
        mn = fminf(mn, red[NW + w]);
    }
    const float floor_db = mx - PVQ_TOP_DB;
    mn = fmaxf(mn, floor_db);
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = tid + t * T;
        if (k < n_bins && live) {
            const float c = fmaxf(d[t], floor_db);
            const float r = (mn > 0.0f) ? (c - mn) : fmaxf(c, 0.0f);
            out[k] = r;
            if (lds_out) lds_out[k] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1-K3 fused: FFT path, one workgroup per frame
// ------------------------------------------------------------------------------------------------
// many streams in one launch of the FFT path: frames are numbered through all streams; stream i holds the frames from frame0
struct FftStream {
    const float* pcm;
    long long n_lead, n_samples;
    long long row0;     // output row of its frame 0
    long long frame0;   // global index of its frame 0 (ascending over the table)
};
struct FftArgs {
    const FftStream* streams;   // nullptr: one stream, described by the fields below
    int n_streams;
    const float* pcm;
    long long n_lead;
    long long hop;
    long long n_samples;  // total samples in pcm (for upper-bound safety)
    int n_fft;
    int n_frames;
    int n_groups;
    int n_bins;
    int n_tw;       // twiddle table length (= largest complex FFT size)
    int max_cols;   // LDS spec capacity
    const GroupDev* groups;
    const float2* tw;
    const float2* split_tw;
    const uint32_t* row_ptr;
    const float4* ent;   // (re, im, column | conj flag, -) per kernel entry
    float* out_db;
    float2* out_cplx;
    unsigned* status;
    int dev_skip;   // developer knob PVQ_FFT_SKIP (timing experiments, wrong results): 1 skips the row dots, 2 the FFT passes
    // Few frames (the streaming front end's single frame, short clips): one workgroup per (frames of a workgroup, WINDOW GROUP) instead
    // of one walking the groups in turn — a frame's latency is then its longest group's (the 16 384-sample window: ~55 % of the
    // walk), not their sum; x_vqt goes through xv_split ([frame][bin], complex) and db_rows finishes the frames.  Same arithmetic
    // per group, same dB routine: same bits as the walk.
    float2* xv_split;   // nullptr: every workgroup walks all groups
};

// T threads per frame, F = BLOCK / T frames per workgroup side by side: a 4096-sample window is a 2048-point complex FFT =
// 128 radix-16 butterflies per pass, so a whole 512-thread workgroup on one frame leaves three quarters of its threads idle
// through every pass (the trainer's geometry, pitchvis_train/src/train.rs:30-43: 13 M frames/s that way).  All frames of a
// workgroup walk the same window groups in step, so the barriers inside the FFT passes stay the workgroup's.  The real split
// writes the spectrum columns the kernel reads in place of the FFT output (column c and N - c come from the same pair of
// FFT bins, so one thread computes both and nothing else touches that pair): no separate spectrum buffer, 19 KB of LDS per
// frame at that geometry instead of 27, eight frames in flight per CU.
template <int BLOCK, int E, int T>
__global__ __launch_bounds__(BLOCK, 4) void vqt_fft_frames(FftArgs a) {   // four waves per SIMD: two 512-thread workgroups per CU
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int F = BLOCK / T;
    const int tid = threadIdx.x, tl = tid % T, fl = tid / T;
    const int zlen = lpad(a.n_tw) + 1;          // FFT buffer of the largest window, + the slot of its Nyquist column
    const int per_frame = zlen + a.n_bins;      // complex slots per frame
    float2* Z = reinterpret_cast<float2*>(smem) + (size_t)fl * per_frame;
    float2* xv = Z + zlen;
    float* red = reinterpret_cast<float*>(reinterpret_cast<float2*>(smem) + (size_t)F * per_frame) + fl * 2 * (T / 64);

    // (split: every frame's largest window first, then the next — workgroups start in index order, the long ones must not come last)
    const int fg_step = a.xv_split ? (int)(gridDim.x / a.n_groups) : (int)gridDim.x;
    const int g_only = a.xv_split ? (int)(blockIdx.x / fg_step) : -1;
    const int fg0 = a.xv_split ? (int)(blockIdx.x % fg_step) : (int)blockIdx.x;
    for (int fg = fg0; fg * F < a.n_frames; fg += fg_step) {
        const bool live = fg * F + fl < a.n_frames;
        const int gframe = live ? fg * F + fl : a.n_frames - 1;   // a padding frame repeats the last one and stores nothing
        // the frame's stream (wave-uniform: T >= 64 threads share a frame)
        const float* pcm = a.pcm;
        long long n_lead = a.n_lead, n_samples = a.n_samples, frame = gframe, out_row = gframe;
        if (a.streams) {
            int lo = 0, n = a.n_streams;   // the last stream whose first frame is at or before gframe
            while (n > 1) {
                const int h = n >> 1;
                if (a.streams[lo + h].frame0 <= gframe) { lo += h; n -= h; } else n = h;
            }
            const FftStream st = a.streams[lo];
            pcm = st.pcm;
            n_lead = st.n_lead;
            n_samples = st.n_samples;
            frame = gframe - st.frame0;
            out_row = st.row0 + frame;
        }
        // x[j] of the reference's n_fft buffer is pcm[buf0 + j]; zeros before the stream start
        const long long buf0 = n_lead + (frame + 1) * a.hop - a.n_fft;

        for (int g = 0; g < a.n_groups; ++g) {
            if (g_only >= 0 && g != g_only) continue;   // (uniform)
            const GroupDev G = a.groups[g];
            const int N = G.n_cplx;
            // gather the window: Z[n] = (x[w0 + 2n], x[w0 + 2n + 1]): one 8-byte load (the stream's samples are 4-byte aligned, whatever the
            // hop) and one 8-byte LDS store per complex point, eight loads in flight per thread; zeros before the stream start
            {
                struct __attribute__((packed, aligned(4))) F2U { float x, y; };
                const long long s0 = buf0 + G.w0;
                for (int n0 = tl; n0 < N; n0 += 8 * T) {
                    float2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int n = n0 + u * T;
                        const long long sx = s0 + 2 * n;
                        v[u] = make_float2(0.0f, 0.0f);
                        if (n < N) {
                            if (sx >= 0 && sx + 1 < n_samples) {
                                const F2U t = *reinterpret_cast<const F2U*>(pcm + sx);
                                v[u] = make_float2(t.x, t.y);
                            } else {
                                if (sx >= 0 && sx < n_samples) v[u].x = pcm[sx];
                                if (sx + 1 >= 0 && sx + 1 < n_samples) v[u].y = pcm[sx + 1];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int n = n0 + u * T;
                        if (n < N) Z[lpad(n)] = v[u];
                    }
                }
            }
            __syncthreads();
            if (!(a.dev_skip & 2)) lds_fft<T, E>(Z, N, a.tw, a.n_tw, tl, (a.dev_skip & 4) ? N : G.n_cols - 1);   // (PVQ_FFT_SKIP=4: unpruned, for A/B)
            // real split, in place, only for the columns the kernel reads (c <= n_cols - 1 <= N): columns c and d = N - c
            // are both made of FFT bins c and N - c
            for (int c = tl; c <= N / 2; c += T) {
                const int d = N - c;
                const bool need_c = c < G.n_cols, need_d = d < G.n_cols && d != c;
                if (!need_c && !need_d) continue;
                const float2 zc = Z[lpad(c & (N - 1))], zd = Z[lpad(d & (N - 1))];
                auto column = [&](float2 za, float2 zb, int col) {   // za = Z[col], zb = conj(Z[N - col])
                    zb.y = -zb.y;
                    const float2 w = a.split_tw[G.split_off + col];
                    const float2 ev = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y + zb.y));
                    const float2 dv = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y - zb.y));
                    const float2 t = cmul(w, dv);
                    return make_float2(ev.x + t.y, ev.y - t.x);
                };
                float2 sc = make_float2(0.0f, 0.0f), sd = sc;
                if (need_c) sc = column(zc, zd, c);
                if (need_d) sd = column(zd, zc, d);
                if (need_c) Z[lpad(c)] = sc;
                if (need_d) Z[lpad(d)] = sd;   // d = N (the Nyquist column, from FFT bin 0) lands in the slot behind the buffer
            }
            __syncthreads();
            // banded complex row dots (vqt.rs:889-910), the whole workgroup over all its frames: a team of G.tpr lanes
            // walks one kernel row, every entry is fetched once (one 16-byte load) and applied to the F frames' spectra
            {
                const int tpr = G.tpr;
                const int team = tid / tpr, tm = tid & (tpr - 1), n_teams = BLOCK / tpr;
                const uint32_t* rp = a.row_ptr + G.row_ptr_off;
                const float2* Z0 = reinterpret_cast<const float2*>(smem);
                float2* xv0 = reinterpret_cast<float2*>(smem) + zlen;
                for (int row = team; row < G.n_rows && !(a.dev_skip & 1); row += n_teams) {
                    const int s = G.ent_off + rp[row], e = G.ent_off + rp[row + 1];
                    float2 acc[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) acc[f] = make_float2(0.0f, 0.0f);
                    for (int i = s + tm; i < e; i += tpr) {
                        const float4 en = a.ent[i];
                        const uint32_t c = __builtin_bit_cast(uint32_t, en.z);
                        const int idx = lpad(c & 0x7fffu);
                        const float sg = (c & 0x8000u) ? -1.0f : 1.0f;
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            float2 x = Z0[(size_t)f * per_frame + idx];
                            x.y *= sg;
                            acc[f].x = fmaf(-en.y, x.y, fmaf(en.x, x.x, acc[f].x));
                            acc[f].y = fmaf(en.y, x.x, fmaf(en.x, x.y, acc[f].y));
                        }
                    }
                    for (int o = tpr >> 1; o > 0; o >>= 1) {
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            acc[f].x += __shfl_xor(acc[f].x, o);
                            acc[f].y += __shfl_xor(acc[f].y, o);
                        }
                    }
                    if (tm == 0) {
#pragma unroll
                        for (int f = 0; f < F; ++f) xv0[(size_t)f * per_frame + G.first_bin + row] = acc[f];
                    }
                }
            }
            __syncthreads();   // the next group's gather overwrites the spectrum columns
        }
        if (g_only >= 0) {   // this group's rows of x_vqt; db_rows takes the frame from there
            const GroupDev G = a.groups[g_only];
            if (live)
                for (int k = tl; k < G.n_rows; k += T) a.xv_split[(size_t)out_row * a.n_bins + G.first_bin + k] = xv[G.first_bin + k];
            __syncthreads();
            continue;
        }
        if (a.out_cplx && live) {
            for (int k = tl; k < a.n_bins; k += T) a.out_cplx[(size_t)out_row * a.n_bins + k] = xv[k];
        }
        db_epilogue<T>(xv, red, a.n_bins, a.out_db + (size_t)out_row * a.n_bins, nullptr, tl, a.status, live);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The FFT path for BATCHES (round 5): one kernel instantiation per window size.  vqt_fft_frames above takes any window at run time:
// every LDS address of its Stockham passes is computed from N, p and the thread's item (lpad, masks, multiplies), and of the ~330
// vector instructions it spends per radix-16 item and pass two thirds are that index arithmetic.  Here N, the pass stride P and the
// threads per frame T = N / 16 are template parameters: an item's 16 reads and 16 writes are ONE base address plus immediates, the
// pass chain is unrolled, and a workgroup serves one window group (the launch is per group, as the few-frames form of the kernel
// above already is: rows of x_vqt through `xv_split`, db_rows finishes the frames).  The arithmetic is the walk's — the same twiddle
// table entries, the same cmul and RegFft<R>, the same pass radices and pruning, the same real split and row dots with the same
// lanes-per-row — so a frame's bits are what vqt_fft_frames gives (tests/test_parity_gpu.py::test_fft_batch_kernels_equal_the_walk).
// ------------------------------------------------------------------------------------------------
template <int R, int N, int P, int T>
__device__ __forceinline__ void stockham_pass_ct(float2* __restrict__ Z, const float2* __restrict__ tw, int n_tw, int tl) {
    constexpr int NB = 16 / R;            // items per thread: NB * T = N / R exactly
    constexpr int TI = N / R;             // items of the pass
    static_assert(TI % 16 == 0 && NB * T == TI, "pass geometry");
    constexpr int LTI = TI + TI / 16;     // lpad(i + r TI) = lpad(i) + r LTI
    float2 u[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float2* zi = Z + lpad(tl + b * T);
#pragma unroll
        for (int r = 0; r < R; ++r) u[b][r] = zi[r * LTI];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (P > 1) {
            const int k = (tl + b * T) & (P - 1);
            const int base = k * (n_tw / (P * R));
#pragma unroll
            for (int r = 1; r < R; ++r) u[b][r] = cmul(u[b][r], tw[base * r]);
        }
        RegFft<R>::run(u[b]);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tl + b * T;
        const int k = i & (P - 1);
        const int j = (i - k) * R + k;
        if (P >= 16) {   // lpad(j + r P) = lpad(j) + r (P + P / 16)
            float2* zj = Z + lpad(j);
#pragma unroll
            for (int r = 0; r < R; ++r) zj[r * (P + P / 16)] = u[b][r];
        } else {         // the first pass (P = 1, R = 16): j = 16 i, lpad(j + r) = 17 i + r
            static_assert(P >= 16 || (P == 1 && R == 16), "first pass");
            float2* zj = Z + 17 * i;
#pragma unroll
            for (int r = 0; r < R; ++r) zj[r] = u[b][r];
        }
    }
    __syncthreads();
}
template <int R, int N, int P, int T>
__device__ __forceinline__ void stockham_pass_ends_ct(float2* __restrict__ Z, const float2* __restrict__ tw, int n_tw, int tl, int M) {
    constexpr int NB = 16 / R;
    constexpr int TI = N / R;
    constexpr int LTI = TI + TI / 16;
    static_assert(P >= 16, "a pruned pass is never the first");
    float2 lo[NB], hi[NB];
    bool nl[NB], nh[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tl + b * T;
        const int k = i & (P - 1);
        nl[b] = k <= M;
        nh[b] = k >= P - M;
        lo[b] = make_float2(0.0f, 0.0f);
        hi[b] = lo[b];
        if (nl[b] || nh[b]) {
            float2 u[R];
            const float2* zi = Z + lpad(i);
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = zi[r * LTI];
            const int base = k * (n_tw / (P * R));
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[base * r]);
            float2 s0 = u[0], s1 = u[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                constexpr int step = 16 / R;
                const int idx = r * step;   // e^{+2 pi i idx / 16}
                const float c = idx < 8 ? kC16[idx] : -kC16[idx - 8], sn = idx < 8 ? kS16[idx] : -kS16[idx - 8];
                s0 = cadd(s0, u[r]);
                s1 = cadd(s1, cmul(u[r], make_float2(c, sn)));
            }
            lo[b] = s0;
            hi[b] = s1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tl + b * T;
        const int k = i & (P - 1);
        float2* zj = Z + lpad((i - k) * R + k);
        if (nl[b]) zj[0] = lo[b];
        if (nh[b]) zj[(R - 1) * (P + P / 16)] = hi[b];
    }
    __syncthreads();
}
// the pass chain of lds_fft for a compile-time N: radix 16 while 16 or more points remain, then one pass of radix 8, 4 or 2
template <int N, int P, int T>
__device__ __forceinline__ void lds_fft_ct(float2* Z, const float2* tw, int n_tw, int tl, int M) {
    constexpr int REM = N / P;
    if constexpr (REM >= 16) {
        if (M < P) {
            if constexpr (P >= 16) stockham_pass_ends_ct<16, N, P, T>(Z, tw, n_tw, tl, M);
        } else {
            stockham_pass_ct<16, N, P, T>(Z, tw, n_tw, tl);
        }
        lds_fft_ct<N, P * 16, T>(Z, tw, n_tw, tl, M);
    } else if constexpr (REM > 1) {
        if (M < P) stockham_pass_ends_ct<REM, N, P, T>(Z, tw, n_tw, tl, M);
        else stockham_pass_ct<REM, N, P, T>(Z, tw, n_tw, tl);
    }
}

template <int N, int BLOCK>   // N complex points (a window of 2 N samples), T = N / 16 threads per frame, F = BLOCK / T frames per workgroup side by side
__global__ __launch_bounds__(BLOCK, 4) void vqt_fft_group(FftArgs a, int g, int frame0, int n_frames_here) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = N / 16, F = BLOCK / T;
    constexpr int ZLEN = N + N / 16 + 1;   // lpad(N) + the slot of the Nyquist column
    const int tid = threadIdx.x, tl = tid % T, fl = tid / T;
    float2* Z = reinterpret_cast<float2*>(smem) + (size_t)fl * ZLEN;
    const GroupDev G = a.groups[g];
    {   // one group of F frames per workgroup, no loop over groups: a loop's hoisted invariants (addresses of every pass) cost registers and scratch
        const int fg = blockIdx.x;
        const bool live = fg * F + fl < n_frames_here;
        const int gframe = frame0 + (live ? fg * F + fl : n_frames_here - 1);   // a padding frame repeats the last one and stores nothing
        const float* pcm = a.pcm;
        long long n_lead = a.n_lead, n_samples = a.n_samples, frame = gframe, out_row = gframe;
        if (a.streams) {
            int lo = 0, n = a.n_streams;   // the last stream whose first frame is at or before gframe
            while (n > 1) {
                const int h = n >> 1;
                if (a.streams[lo + h].frame0 <= gframe) { lo += h; n -= h; } else n = h;
            }
            const FftStream st = a.streams[lo];
            pcm = st.pcm;
            n_lead = st.n_lead;
            n_samples = st.n_samples;
            frame = gframe - st.frame0;
            out_row = st.row0 + frame;
        }
        const long long s0 = n_lead + (frame + 1) * a.hop - a.n_fft + G.w0;
        // gather the window: Z[n] = (x[w0 + 2n], x[w0 + 2n + 1]), 16 points per thread, all loads in flight before the first LDS store
        {
            struct __attribute__((packed, aligned(4))) F2U { float x, y; };
            float2 v[16];
            if (s0 >= 0 && s0 + 2 * N <= n_samples) {   // the window lies inside the stream (all but a stream's first frames)
                const F2U* src = reinterpret_cast<const F2U*>(pcm + s0);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const F2U t = src[tl + u * T];
                    v[u] = make_float2(t.x, t.y);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const long long sx = s0 + 2 * (tl + u * T);
                    v[u] = make_float2(0.0f, 0.0f);
                    if (sx >= 0 && sx < n_samples) v[u].x = pcm[sx];
                    if (sx + 1 >= 0 && sx + 1 < n_samples) v[u].y = pcm[sx + 1];
                }
            }
            // (the thread's index goes through an empty asm once per frame: left to itself the compiler hoists the 15 twiddle ADDRESSES of every
            // pass — loop-invariant 64-bit values — out of the frame loop and spills them: 380 bytes of scratch per lane, reloaded in every pass)
            int tl_ = tl;
            asm volatile("" : "+v"(tl_));
            // The first pass (stride 1, radix 16, no twiddles) takes item i = tl's inputs Z[tl + r N / 16] — exactly the 16 points this thread
            // has just gathered (T = N / 16): they go through the butterfly in registers and only its outputs reach LDS.  The walk writes the
            // gathered window to LDS, meets at a barrier and reads it back: same values, 16 LDS writes, a barrier and 16 LDS reads more.
            // (n_cols > 1, checked by the host: with a single column the walk prunes even this pass)
            RegFft<16>::run(v);
            float2* zj = Z + 17 * tl_;   // lpad(16 i + r) = 17 i + r
#pragma unroll
            for (int r = 0; r < 16; ++r) zj[r] = v[r];
            __syncthreads();
            lds_fft_ct<N, 16, T>(Z, a.tw, a.n_tw, tl_, G.n_cols - 1);
        }
        // real split, in place, only for the columns the kernel reads (as in vqt_fft_frames)
        for (int c = tl; c <= N / 2; c += T) {
            const int d = N - c;
            const bool need_c = c < G.n_cols, need_d = d < G.n_cols && d != c;
            if (!need_c && !need_d) continue;
            const float2 zc = Z[lpad(c & (N - 1))], zd = Z[lpad(d & (N - 1))];
            auto column = [&](float2 za, float2 zb, int col) {   // za = Z[col], zb = conj(Z[N - col])
                zb.y = -zb.y;
                const float2 w = a.split_tw[G.split_off + col];
                const float2 ev = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y + zb.y));
                const float2 dv = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y - zb.y));
                const float2 t = cmul(w, dv);
                return make_float2(ev.x + t.y, ev.y - t.x);
            };
            float2 sc = make_float2(0.0f, 0.0f), sd = sc;
            if (need_c) sc = column(zc, zd, c);
            if (need_d) sd = column(zd, zc, d);
            if (need_c) Z[lpad(c)] = sc;
            if (need_d) Z[lpad(d)] = sd;
        }
        __syncthreads();
        // banded complex row dots: a team of G.tpr lanes walks one kernel row, every entry fetched once and applied to the F frames' spectra;
        // the sums go straight to the frames' rows of x_vqt (xv_split)
        {
            const int tpr = G.tpr;
            const int team = tid / tpr, tm = tid & (tpr - 1), n_teams = BLOCK / tpr;
            const uint32_t* rp = a.row_ptr + G.row_ptr_off;
            const float2* Z0 = reinterpret_cast<const float2*>(smem);
            for (int row = team; row < G.n_rows; row += n_teams) {
                const int s = G.ent_off + rp[row], e = G.ent_off + rp[row + 1];
                float2 acc[F];
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] = make_float2(0.0f, 0.0f);
                for (int i = s + tm; i < e; i += tpr) {
                    const float4 en = a.ent[i];
                    const uint32_t c = __builtin_bit_cast(uint32_t, en.z);
                    const int idx = lpad(c & 0x7fffu);
                    const float sg = (c & 0x8000u) ? -1.0f : 1.0f;
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        float2 x = Z0[(size_t)f * ZLEN + idx];
                        x.y *= sg;
                        acc[f].x = fmaf(-en.y, x.y, fmaf(en.x, x.x, acc[f].x));
                        acc[f].y = fmaf(en.y, x.x, fmaf(en.x, x.y, acc[f].y));
                    }
                }
                for (int o = tpr >> 1; o > 0; o >>= 1) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        acc[f].x += __shfl_xor(acc[f].x, o);
                        acc[f].y += __shfl_xor(acc[f].y, o);
                    }
                }
                if (tm == 0) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        const int fr = fg * F + f;
                        if (fr < n_frames_here) a.xv_split[(size_t)fr * a.n_bins + G.first_bin + row] = acc[f];   // (rows of the launch's frames, from frame0 on)
                    }
                }
            }
        }
    }
}

// the frames of a group-split launch: x_vqt rows -> (optional complex output), frame-relative dB; T threads per row as in the walk
template <int T>
__global__ __launch_bounds__(T) void db_rows(const float2* __restrict__ xv_rows, int n_rows, int n_bins, float* __restrict__ out_db, float2* __restrict__ out_cplx,
                                             unsigned* status) {
    __shared__ float red[2 * (T / 64)];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row >= n_rows) return;
    const float2* xv = xv_rows + (size_t)row * n_bins;
    if (out_cplx)
        for (int k = tid; k < n_bins; k += T) out_cplx[(size_t)row * n_bins + k] = xv[k];
    db_epilogue<T>(xv, red, n_bins, out_db + (size_t)row * n_bins, nullptr, tid, status, true);
}

// ... and of a vqt_fft_group launch: row r holds frame frame0 + r of the call, whose output row is the frame's own (one stream) or its stream's
template <int T>
__global__ __launch_bounds__(T) void db_rows_batch(const float2* __restrict__ xv_rows, int n_rows, int frame0, int n_bins, const FftStream* __restrict__ streams, int n_streams,
                                                   float* __restrict__ out_db, float2* __restrict__ out_cplx, unsigned* status) {
    __shared__ float red[2 * (T / 64)];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= n_rows) return;
    long long out_row = frame0 + r;
    if (streams) {
        const int gframe = frame0 + r;
        int lo = 0, n = n_streams;
        while (n > 1) {
            const int h = n >> 1;
            if (streams[lo + h].frame0 <= gframe) { lo += h; n -= h; } else n = h;
        }
        out_row = streams[lo].row0 + (gframe - streams[lo].frame0);
    }
    const float2* xv = xv_rows + (size_t)r * n_bins;
    if (out_cplx)
        for (int k = tid; k < n_bins; k += T) out_cplx[(size_t)out_row * n_bins + k] = xv[k];
    db_epilogue<T>(xv, red, n_bins, out_db + (size_t)out_row * n_bins, nullptr, tid, status, true);
}

// ------------------------------------------------------------------------------------------------
// K4 standalone: peaks of dB frames that already sit in global memory, one wavefront per frame
// (the fused forms run peaks_wave() inside the frame kernels instead)
// ------------------------------------------------------------------------------------------------
constexpr int PK_WAVES = 4;

// Common case (no plateau peak of three or more samples in the frame): the lean routine.  A frame it cannot take is
// flagged for peaks_frames_generic, which is launched right behind it.
// A workgroup (4 waves) takes PK_FPW frames per wave and pass: each wave finds the peaks of its frames (peak lists in LDS),
// then the continuous outputs of all 4 x PK_FPW frames are refined side by side, one lane per peak over the pooled list:
// a frame has ~17 peaks, so refining per frame left three quarters of the lanes idle through ~400 vector instructions;
// pooled, enhance_peaks_continuous runs on full waves and promote_bass_peaks_with_harmonics (only the few peaks at or
// below highest_bassnote need it, but any one of them made the whole wave walk through it) runs once over the pooled
// bass peaks.
// PK_FPW frames per wave and pass: 2 up to 384 bins; 1 beyond, where a second frame per wave would cost more occupancy
// (LDS) than the denser refinement gives back
// bass peaks one frame can hold: their centre is at most highest_bassnote, so their bin at most highest_bassnote + 1,
// and peaks are never adjacent
__host__ __device__ inline int peaks_bass_cap(int n_bins, int highest_bassnote) {
    const int npad = (n_bins + 63) / 64 * 64;
    const int by_bins = (highest_bassnote < n_bins ? highest_bassnote : n_bins) / 2 + 3;
    return by_bins < npad / 2 ? by_bins : npad / 2;
}
__host__ __device__ inline size_t peaks_lean_lds_bytes(int n_bins, int dist, int PK_FPW, int highest_bassnote) {
    const size_t PK_FPG = (size_t)PK_WAVES * PK_FPW;
    const size_t npad = (size_t)((n_bins + 63) / 64 * 64);
    return PK_FPG * sizeof(float) * (npad + 2 * PK_PAD)          // frames, +INF sentinels on both sides
           + PK_FPG * peaks_lean_scratch_bytes(n_bins, dist)     // per-frame scratch of the peak search
           + 2 * PK_FPG * npad                                   // peak lists: npad / 2 u16 per frame
           + ((2 * PK_FPG * (size_t)peaks_bass_cap(n_bins, highest_bassnote) + 3) & ~(size_t)3)   // pooled bass list: u16 (frame << 10 | slot)
           + 2 * PK_FPG * sizeof(uint32_t) + 16                  // per-frame peak counts, bass counter
           + (n_bins <= 768 && !(dist > 1 && dist <= 4) ? 2 * npad * sizeof(float) : 0);   // thresholds of the candidate test (NK <= 12; not with the register distance rule)
}

template <int NK, bool DISTANCE, int PK_FPW, bool DENSE>   // DENSE: 64 (NK - 1) < n_bins <= 64 NK (the host's promise): every chunk test but the last one's folds away
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(NK <= 8 ? 7 : 4, 8))) void peaks_frames_lean(const float* __restrict__ db, int n_frames, PeakParamsDev a,
                                                                       uint8_t* __restrict__ redo, int frame0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    constexpr int PK_FPG = PK_WAVES * PK_FPW;     // frames per workgroup and pass
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if constexpr (DENSE) __builtin_assume(a.n_bins > 64 * (NK - 1) && a.n_bins <= 64 * NK);
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    const int row = npad + 2 * PK_PAD;
    float* rows = reinterpret_cast<float*>(pk_smem);                                   // [PK_FPG][row]
    const int sb = (int)peaks_lean_scratch_bytes(n, a.dist);                           // scratch of one frame
    unsigned char* scratch0 = pk_smem + PK_FPG * sizeof(float) * row;                  // [PK_FPG][sb]
    uint16_t* plists = reinterpret_cast<uint16_t*>(scratch0 + PK_FPG * sb);
    const int pl_cap = npad / 2;                                                       // peaks are never adjacent
    uint16_t* bass = plists + PK_FPG * pl_cap;                                         // [PK_FPG * bass_cap]
    const int bass_cap = peaks_bass_cap(n, a.highest_bassnote);
    uint32_t* counts = reinterpret_cast<uint32_t*>(pk_smem + ((reinterpret_cast<unsigned char*>(bass + PK_FPG * bass_cap) - pk_smem + 3) & ~(size_t)3));   // [PK_FPG]
    uint32_t* n_bass = counts + PK_FPG;
    float* thrH = reinterpret_cast<float*>(n_bass + 1);                                // [npad] per-bin height / prominence thresholds of the candidate test
    float* thrP = thrH + npad;
    // the wave's first frames are asked for before the workgroup fills its threshold table and meets at the barrier: the loads fly meanwhile
    float pre[PK_FPW][NK <= 12 ? NK : 1];
    if (NK <= 12) {
#pragma unroll
        for (int g = 0; g < PK_FPW; ++g) {
            const int frame = frame0 + blockIdx.x * PK_FPG + wv * PK_FPW + g;
            const float* src = db + (size_t)(frame < n_frames ? frame : 0) * n;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = (k << 6) + lane;
                pre[g][k] = ((k << 6) < n) ? src[i < n ? i : n - 1] : 0.0f;
            }
        }
    }
    const bool thr_lds = NK <= 12 && !(DISTANCE && a.dist <= 4);   // (beyond 768 bins, and on the list of the register distance rule, peaks_lean_scan computes them on the fly)
    if (thr_lds) {
        peaks_lean_thresholds(thrH, thrP, a, tid, PK_WAVES * 64);
        __syncthreads();
    }
    // sentinels of this wave's rows
    for (int g = 0; g < PK_FPW; ++g) {
        float* xs = rows + (wv * PK_FPW + g) * row;
        for (int i = lane; i < PK_PAD; i += 64) xs[i] = __builtin_huge_valf();
        for (int i = n + lane; i < npad + PK_PAD; i += 64) xs[PK_PAD + i] = __builtin_huge_valf();
    }
    // ONE pass per workgroup (the host launches a workgroup per PK_FPG frames; more than 2^20 workgroups' worth of frames go in further
    // launches): written as a loop over passes, the compiler hoisted ~150 instructions of loop-invariant lane masks and addresses into
    // a preamble and spilled them to lanes — for a loop that ran once.
    {
        const int base = frame0 + blockIdx.x * PK_FPG;
        if (tid == 0) *n_bass = 0;
        // 1. peak search: each wave scans its PK_FPW frames one after the other, then walks their candidates side by side
        uint32_t n_cand[PK_FPW], np[PK_FPW];
        bool done[PK_FPW];
        size_t frames[PK_FPW];
#pragma unroll
        for (int g = 0; g < PK_FPW; ++g) {
            const int fi = wv * PK_FPW + g;
            const int frame = base + fi;
            n_cand[g] = 0;
            done[g] = false;
            frames[g] = (size_t)frame;
            if (frame < n_frames) {
                float* x = rows + fi * row + PK_PAD;
                const float* src = db + (size_t)frame * n;
                if (NK <= 12) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) {   // the frame into its row (the tail of its last 64 stays +INF)
                        if ((k << 6) >= n) break;
                        const int i = (k << 6) + lane;
                        x[i] = i < n ? pre[g][k] : __builtin_huge_valf();
                    }
                } else {
                    for (int i = lane; i < n; i += 64) x[i] = src[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                done[g] = peaks_lean_scan<NK, DISTANCE>(x, scratch0 + fi * sb, thr_lds ? thrH : nullptr, thrP, n_cand[g], a, lane);
                if (lane == 0) redo[frame] = done[g] ? 0 : 1;   // the generic kernel produces that frame's outputs, the continuous ones included
            }
        }
        peaks_lean_walk<NK, PK_FPW>(rows + wv * PK_FPW * row + PK_PAD, row, scratch0 + wv * PK_FPW * sb, sb, plists + wv * PK_FPW * pl_cap, pl_cap,
                                    n_cand, np, done, frames, a, lane);
#pragma unroll
        for (int g = 0; g < PK_FPW; ++g)
            if (lane == 0) counts[wv * PK_FPW + g] = np[g] < a.max_peaks ? np[g] : a.max_peaks;
        if (!a.center) return;   // uniform: mask / count only
        __syncthreads();
        // 2. enhance_peaks_continuous over the pooled peaks, one lane each; bass peaks are handed to step 3
        uint32_t pre[PK_FPG + 1];
        pre[0] = 0;
#pragma unroll
        for (int f = 0; f < PK_FPG; ++f) pre[f + 1] = pre[f] + counts[f];
        const uint32_t total = pre[PK_FPG];
        for (uint32_t p0 = 0; p0 < total; p0 += 256) {
            const uint32_t p = p0 + tid;
            if (p0 + (wv << 6) >= total) break;   // nothing for this wave
            bool is_bass = false;
            int fi = 0;
            uint32_t slot = 0;
            if (p < total) {
#pragma unroll
                for (int f = 1; f < PK_FPG; ++f) fi += (p >= pre[f]) ? 1 : 0;
            }
            uint32_t start = 0;
#pragma unroll
            for (int f = 0; f < PK_FPG; ++f) start = (fi == f) ? pre[f] : start;
            slot = p - start;
            if (p < total) {
                const float* x = rows + fi * row + PK_PAD;
                float ctr, sz;
                pk_enhance(x, (int)plists[fi * pl_cap + slot], a, ctr, sz);
                const size_t o = (size_t)(base + fi) * a.max_peaks + slot;
                a.center[o] = ctr;
                is_bass = !(ctr > (float)a.highest_bassnote);
                if (!is_bass) a.size[o] = sz;
            }
            const unsigned long long bm = __ballot(is_bass);
            if (bm) {
                uint32_t off = 0;
                if (lane == 0) off = atomicAdd(n_bass, (uint32_t)__popcll(bm));
                off = __builtin_amdgcn_readfirstlane(off);
                const uint32_t at = off + __popcll(bm & ((1ull << lane) - 1ull));
                if (is_bass && at < (uint32_t)(PK_FPG * bass_cap)) bass[at] = (uint16_t)((fi << 10) | slot);
            }
        }
        __syncthreads();
        // 3. promote_bass_peaks_with_harmonics over the pooled bass peaks (the centre is recomputed: cheaper than carrying it)
        const uint32_t nb_ = min(*n_bass, (uint32_t)(PK_FPG * bass_cap));
        for (uint32_t q = tid; q < nb_; q += 256) {
            const uint32_t e = bass[q];
            const int fi = (int)(e >> 10);
            const uint32_t slot = e & 1023u;
            const float* x = rows + fi * row + PK_PAD;
            float ctr, sz;
            pk_enhance(x, (int)plists[fi * pl_cap + slot], a, ctr, sz);
            pk_promote(x, a, ctr, sz);
            a.size[(size_t)(base + fi) * a.max_peaks + slot] = sz;
        }
    }
}

// Generic routine (plateau peaks, min_distance > 1).  redo == nullptr: every frame.
template <int NK>
__global__ __launch_bounds__(PK_WAVES * 64) void peaks_frames_generic(const float* __restrict__ db, int n_frames, PeakParamsDev a,
                                                                     const uint8_t* __restrict__ redo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = a.n_bins;
    const int npad = (n + 63) / 64 * 64;
    const size_t per_wave = sizeof(float) * npad + peaks_scratch_bytes(n, a.dist);
    float* x = reinterpret_cast<float*>(pk_smem + wv * per_wave);
    unsigned char* scratch = reinterpret_cast<unsigned char*>(x + npad);
    auto one = [&](int frame) {
        const float* src = db + (size_t)frame * n;
        for (int i = lane; i < n; i += 64) x[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        peaks_wave_nk<NK>(x, scratch, (size_t)frame, a, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (!redo) {
        for (int frame = blockIdx.x * PK_WAVES + wv; frame < n_frames; frame += gridDim.x * PK_WAVES) one(frame);
        return;
    }
    // sweep of the (mostly clear) flags: 64 per wave and step, one ballot, then only the flagged frames
    for (int base = (blockIdx.x * PK_WAVES + wv) * 64; base < n_frames; base += gridDim.x * PK_WAVES * 64) {
        unsigned long long m = __ballot(base + lane < n_frames && redo[base + lane] != 0);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            one(base + b);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Many SHORT streams, staged one behind the other (Vqt::batch_streams_device): piece p copies `count` samples from its stream to
// position dst_off of the staging buffer (zeroed before: the gaps between the streams are the zeros / the history each stream's
// first frames see).  grid: (chunks, pieces).
// ------------------------------------------------------------------------------------------------
struct StagePiece {
    const float* src;     // first sample to copy (the stream's pointer + what of its lead does not fit the gap)
    long long count;
    long long dst_off;    // samples from the start of the staging buffer
    long long zero_from;  // the gap before the piece, [zero_from, dst_off), is zeroed by it (the previous piece's end; 0 for the first)
    long long zero_to;    // ... and [dst_off + count, zero_to) behind it (the buffer's end for the last piece, nothing otherwise)
};
__global__ __launch_bounds__(256) void stage_streams(float* __restrict__ dst, const StagePiece* __restrict__ pieces) {
    const StagePiece p = pieces[blockIdx.y];
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    float* d = dst + p.dst_off;
    for (long long i = t0; i < p.count; i += step) d[i] = p.src[i];
    for (long long i = p.zero_from + t0; i < p.dst_off; i += step) dst[i] = 0.0f;
    for (long long i = p.dst_off + p.count + t0; i < p.zero_to; i += step) dst[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
const char* Vqt::slot_name(uint32_t s) {
    switch (s) {
        case SLOT_FFT_FRAMES: return "vqt_fft_frames";
        case SLOT_BLOCKDFT_GEMM: return "blockdft_gemm";
        case SLOT_BLOCKDFT_COMBINE: return "blockdft_combine";
        case SLOT_BLOCKDFT_DOTS: return "blockdft_dots_db";
        case SLOT_PEAKS: return "peaks_frames";
        default: return "";
    }
}

pvq_status Vqt::create(const VqtParameters& p, int device_id, std::unique_ptr<Vqt>& out, VqtError& err) {
    if (p.range.octaves == 0 || p.range.buckets_per_octave == 0 || p.n_fft < 4 || (p.n_fft & (p.n_fft - 1)) != 0 ||
        !(p.sr > 0.0f) || !(p.range.min_freq > 0.0f)) {
        set_last_error("invalid VqtParameters (n_fft must be a power of two, sr/min_freq/octaves/buckets > 0)");
        return PVQ_ERR_INVALID_ARG;
    }
    std::unique_ptr<Vqt> v(new Vqt());
    err = build_plan(p, v->plan_);
    if (err.kind == VqtError::AboveNyquist) {
        set_last_error(err.to_string());
        return PVQ_ERR_ABOVE_NYQUIST;
    }
    if (err.kind == VqtError::WindowExceedsNFft) {
        set_last_error(err.to_string());
        return PVQ_ERR_WINDOW_EXCEEDS_NFFT;
    }
    v->device_id_ = device_id;
    if (device_id >= 0) {
        int n_dev = 0;
        hipError_t e = hipGetDeviceCount(&n_dev);
        if (e != hipSuccess || device_id >= n_dev) {
            set_last_error("no such HIP device (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ")");
            return PVQ_ERR_DEVICE;
        }
        PVQ_HIP(hipSetDevice(device_id));
        pvq_status st = v->upload_tables();
        if (st != PVQ_OK) return st;
    }
    out = std::move(v);
    return PVQ_OK;
}

Vqt::~Vqt() {
    if (device_id_ >= 0) {
        (void)hipSetDevice(device_id_);
        if (dev_) {
            free_device_tables(dev_);
            dev_ = nullptr;
        }
        if (ws_pcm_) (void)hipFree(ws_pcm_);
        if (ws_out_) (void)hipFree(ws_out_);
        if (ws_misc_) (void)hipFree(ws_misc_);
        if (ws_flags_) (void)hipFree(ws_flags_);
        if (ws_stage_) (void)hipFree(ws_stage_);
        if (ws_stage_tab_) (void)hipFree(ws_stage_tab_);
        if (ws_split_) (void)hipFree(ws_split_);
        for (void* b : multi_buf_)
            if (b) (void)hipFree(b);
        if (multi_stream_) (void)hipStreamDestroy(multi_stream_);
        if (host_streams_ready_)
            for (int i = 0; i < 3; ++i) (void)hipStreamDestroy(host_streams_[i]);
        for (hipEvent_t e : host_events_) (void)hipEventDestroy(e);
        if (inst_stream_) (void)hipStreamDestroy(inst_stream_);
        if (order_ev_) (void)hipEventDestroy(order_ev_);
        if (inst_pin_) (void)hipHostFree(inst_pin_);
        for (int s = 0; s < N_SLOTS; ++s)
            for (int k = 0; k < 2; ++k)
                for (hipEvent_t e : ev_[s][k]) (void)hipEventDestroy(e);
    }
}

pvq_status Vqt::set_twiddle_fp16(bool on) {
    if (on == twiddle_fp16_) return PVQ_OK;
    twiddle_fp16_ = on;
    if (device_id_ < 0) return PVQ_OK;
    PVQ_HIP(hipSetDevice(device_id_));
    PVQ_HIP(hipDeviceSynchronize());
    if (dev_) {
        free_device_tables(dev_);   // frees the block-DFT tables too; they are rebuilt on the next batch call
        dev_ = nullptr;
    }
    return upload_tables();
}

pvq_status Vqt::upload_tables() {
    std::string msg;
    dev_ = build_device_tables(plan_, twiddle_fp16_, msg);
    if (!dev_) {
        set_last_error(msg);
        return msg.rfind("unsupported", 0) == 0 ? PVQ_ERR_UNSUPPORTED : PVQ_ERR_DEVICE;
    }
    return PVQ_OK;
}

pvq_status Vqt::ensure_workspace(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return PVQ_OK;
    if (*ptr) PVQ_HIP(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    PVQ_HIP(hipMalloc(ptr, bytes));
    *cap = bytes;
    return PVQ_OK;
}

void Vqt::set_profiling(int mode) {
    profiling_ = mode != 0;
    profiling_main_only_ = mode == 2;
    for (int s = 0; s < N_SLOTS; ++s) ev_count_[s] = 0;
}

void Vqt::slot_begin(int slot, hipStream_t s) {
    if (profiling_main_only_ && slot != SLOT_BLOCKDFT_GEMM && slot != SLOT_FFT_FRAMES) return;
    if (!profiling_ || ev_count_[slot] >= kMaxTimedLaunches) return;
    const int i = ev_count_[slot];
    if ((int)ev_[slot][0].size() <= i) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        ev_[slot][0].push_back(a);
        ev_[slot][1].push_back(b);
    }
    (void)hipEventRecord(ev_[slot][0][i], s);
}
void Vqt::slot_end(int slot, hipStream_t s) {
    if (profiling_main_only_ && slot != SLOT_BLOCKDFT_GEMM && slot != SLOT_FFT_FRAMES) return;
    if (!profiling_ || ev_count_[slot] >= kMaxTimedLaunches) return;
    (void)hipEventRecord(ev_[slot][1][ev_count_[slot]], s);
    ++ev_count_[slot];
}

uint32_t Vqt::last_kernel_ms(float* out, uint32_t cap) {
    uint32_t n = 0;
    for (int s = 0; s < N_SLOTS && (uint32_t)s < cap; ++s) {
        out[s] = -1.0f;
        double sum = 0.0;
        int ok = 0;
        for (int i = 0; i < ev_count_[s]; ++i) {
            if (hipEventSynchronize(ev_[s][1][i]) != hipSuccess) continue;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev_[s][0][i], ev_[s][1][i]) == hipSuccess) {
                sum += ms;
                ++ok;
            }
        }
        if (ok) out[s] = (float)(sum / ok);
        n = s + 1;
    }
    return n;
}

bool Vqt::make_peak_params(const AnalysisParameters& ap, uint32_t* d_mask, uint32_t* d_count, float* d_center,
                           float* d_size, uint32_t max_peaks, PeakParamsDev& a) const {
    a.n_bins = (int)n_bins();
    a.bpo = (int)plan_.params.range.buckets_per_octave;
    a.min_freq = plan_.params.range.min_freq;
    a.lnf = dev_->d_lnf;
    a.peak_min_prominence = ap.peak_min_prominence;
    a.peak_min_height = ap.peak_min_height;
    a.bass_min_prominence = ap.bass_min_prominence;
    a.bass_min_height = ap.bass_min_height;
    a.highest_bassnote = (int)std::min<uint32_t>(ap.highest_bassnote, 0x7fffffffu);
    a.harmonic_threshold = ap.harmonic_threshold;
    a.dist = (int)std::lround((float)a.bpo * 0.4f / 12.0f);
    a.min_bin = ((a.bpo / 12) + 1) / 2;
    a.mask = d_mask;
    a.count = d_count;
    a.center = d_center;
    a.size = d_size;
    a.max_peaks = max_peaks;
    return a.n_bins >= 3 && a.n_bins <= 1024;
}

uint32_t Vqt::last_kernel_launches(uint32_t* out, uint32_t cap) const {
    uint32_t n = 0;
    for (int s = 0; s < N_SLOTS && (uint32_t)s < cap; ++s) {
        out[s] = (uint32_t)ev_count_[s];
        n = s + 1;
    }
    return n;
}

pvq_status Vqt::launch_fft_path(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                                float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream) {
    return launch_fft_streams(nullptr, 0, d_pcm, n_lead, hop, n_frames, n_frames, d_out_db, d_out_cplx, pk, stream);
}

// st != nullptr: n_st streams in ONE launch (their frames numbered through; rows_total output rows for the peak stage)
pvq_status Vqt::launch_fft_streams(const void* st_table, size_t n_st, const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, size_t rows_total,
                                   float* d_out_db, float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream) {
    FftArgs a;
    a.streams = nullptr;
    a.n_streams = 0;
    if (st_table) {
        pvq_status es = ensure_workspace(&ws_stage_tab_, &ws_stage_tab_cap_, n_st * sizeof(FftStream));
        if (es != PVQ_OK) return es;
        PVQ_HIP(hipMemcpyAsync(ws_stage_tab_, st_table, n_st * sizeof(FftStream), hipMemcpyHostToDevice, stream));   // (pageable source: staged before the call returns)
        a.streams = static_cast<const FftStream*>(ws_stage_tab_);
        a.n_streams = (int)n_st;
    }
    a.pcm = d_pcm;
    a.n_lead = (long long)n_lead;
    a.hop = (long long)hop;
    a.n_samples = (long long)(n_lead + n_frames * hop);
    a.n_fft = (int)plan_.params.n_fft;
    a.n_frames = (int)n_frames;
    a.n_groups = (int)plan_.kernel.window_groups.size();
    a.n_bins = (int)n_bins();
    a.n_tw = dev_->n_tw;
    a.max_cols = dev_->max_cols;
    a.groups = dev_->d_groups;
    a.tw = dev_->d_tw;
    a.split_tw = dev_->d_split_tw;
    a.row_ptr = dev_->d_row_ptr;
    a.ent = dev_->d_ent;
    a.out_db = d_out_db;
    a.out_cplx = reinterpret_cast<float2*>(d_out_cplx);
    a.status = dev_->d_status;
    static const int skip_env = dev_knob("PVQ_FFT_SKIP", 0);   // (developer build only: timing probes that skip stages)
    a.dev_skip = skip_env;

    // threads per frame: one radix-16 butterfly per thread and pass for the largest window; 512-thread workgroups hold
    // 512 / T frames side by side (a single frame, e.g. the streaming front end's, keeps the whole workgroup)
    const int n_tw = dev_->n_tw;
    const int T = n_frames < 4 ? (n_tw <= 8192 ? 512 : 1024) : n_tw <= 2048 ? 128 : n_tw <= 4096 ? 256 : n_tw <= 8192 ? 512 : 1024;
    const int BLOCK = T == 1024 ? 1024 : 512;
    const int F = BLOCK / T;
    const size_t lds = (size_t)F * (sizeof(float2) * ((size_t)(n_tw + (n_tw >> 4)) + 1 + a.n_bins) + sizeof(float) * 2 * (T / 64));
    int grid = (int)std::min<size_t>((n_frames + F - 1) / F, 1u << 20);
    // few frames of one stream: a workgroup per window group (FftArgs::xv_split).  Measured against the walk (profiles/r04_fft_split.txt):
    // 1 frame 47 -> 19 us at 48 kHz / 252 bins (77 -> 27 at 96 kHz / 360), ahead up to ~400 frames there (~200 at 96 kHz, ~750 at
    // 22 050 Hz / 588 bins): up to ~1 500 workgroups, beyond which the chip is full either way and the walk's one pass over LDS wins
    static const int split_env = dev_knob("PVQ_FFT_SPLIT", 1);   // (developer build: 0 = every workgroup walks its frames' groups)
    static const int split_max_env = dev_knob("PVQ_FFT_SPLIT_MAX", 1500);
    const bool split = !st_table && split_env && a.n_groups > 1 && (size_t)grid * (size_t)a.n_groups <= (size_t)split_max_env;
    a.xv_split = nullptr;
    // batches: one kernel instantiation per window size (vqt_fft_group), a launch per window group and 16 384-frame part, db_rows_batch
    // behind them.  Same bits as the walk.  (developer build: PVQ_FFT_CT=0 keeps the walk)
    static const int ct_env = dev_knob("PVQ_FFT_CT", 1);
    bool ct = ct_env && !split && !skip_env && a.n_groups >= 1 && n_frames >= 64;
    for (int g = 0; g < a.n_groups && ct; ++g) {
        const int N = dev_->h_groups[g].n_cplx;
        ct = N >= 256 && N <= 16384 && (N & (N - 1)) == 0 && dev_->h_groups[g].n_cols > 1;
    }
    if (ct) {
        constexpr size_t PART = 16384;
        pvq_status es = ensure_workspace(&ws_split_, &ws_split_cap_, std::min(PART, n_frames) * (size_t)a.n_bins * sizeof(float2));
        if (es != PVQ_OK) return es;
        a.xv_split = static_cast<float2*>(ws_split_);
        slot_begin(SLOT_FFT_FRAMES, stream);
        auto launch_group = [&](auto n_c, auto block_c, int g, int f0, int nf) -> pvq_status {
            constexpr int N = decltype(n_c)::value, BLK = decltype(block_c)::value;
            constexpr int Tg = N / 16, Fg = BLK / Tg;
            const size_t lds_g = (size_t)Fg * (N + N / 16 + 1) * sizeof(float2);
            auto kern = vqt_fft_group<N, BLK>;
            PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));
            const int grid_g = (int)(((size_t)nf + Fg - 1) / Fg);   // a workgroup per Fg frames (the kernel makes one pass)
            hipLaunchKernelGGL(kern, dim3(grid_g), dim3(BLK), lds_g, stream, a, g, f0, nf);
            return PVQ_OK;
        };
        using std::integral_constant;
        for (size_t f0 = 0; f0 < n_frames; f0 += PART) {
            const int nf = (int)std::min(PART, n_frames - f0);
            for (int g = 0; g < a.n_groups; ++g) {
                pvq_status ls = PVQ_OK;
                switch (dev_->h_groups[g].n_cplx) {
                    case 16384: ls = launch_group(integral_constant<int, 16384>{}, integral_constant<int, 1024>{}, g, (int)f0, nf); break;
                    case 8192: ls = launch_group(integral_constant<int, 8192>{}, integral_constant<int, 512>{}, g, (int)f0, nf); break;
                    case 4096: ls = launch_group(integral_constant<int, 4096>{}, integral_constant<int, 512>{}, g, (int)f0, nf); break;
                    case 2048: ls = launch_group(integral_constant<int, 2048>{}, integral_constant<int, 512>{}, g, (int)f0, nf); break;
                    case 1024: ls = launch_group(integral_constant<int, 1024>{}, integral_constant<int, 512>{}, g, (int)f0, nf); break;
                    case 512: ls = launch_group(integral_constant<int, 512>{}, integral_constant<int, 256>{}, g, (int)f0, nf); break;
                    default: ls = launch_group(integral_constant<int, 256>{}, integral_constant<int, 128>{}, g, (int)f0, nf); break;
                }
                if (ls != PVQ_OK) return ls;
            }
            hipLaunchKernelGGL(db_rows_batch<256>, dim3((unsigned)nf), dim3(256), 0, stream, static_cast<const float2*>(ws_split_), nf, (int)f0, a.n_bins, a.streams, a.n_streams,
                               a.out_db, a.out_cplx, a.status);
        }
        slot_end(SLOT_FFT_FRAMES, stream);
        if (pk) {
            slot_begin(SLOT_PEAKS, stream);
            pvq_status ps = launch_peaks_kernel(d_out_db, rows_total, *pk, stream);
            slot_end(SLOT_PEAKS, stream);
            if (ps != PVQ_OK) return ps;
        }
        PVQ_HIP(hipGetLastError());
        last_algo_ = PVQ_ALGO_FFT;
        last_frames_per_launch_ = (uint32_t)n_frames;
        last_gemm_flop_ = 0.0;
        return PVQ_OK;
    }
    if (split) {
        pvq_status es = ensure_workspace(&ws_split_, &ws_split_cap_, n_frames * (size_t)a.n_bins * sizeof(float2));
        if (es != PVQ_OK) return es;
        a.xv_split = static_cast<float2*>(ws_split_);
        grid *= a.n_groups;
    }
    slot_begin(SLOT_FFT_FRAMES, stream);
    auto launch = [&](auto kern) -> pvq_status {
        PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, stream, a);
        return PVQ_OK;
    };
    pvq_status lst = T == 128 ? launch(vqt_fft_frames<512, 16, 128>) : T == 256 ? launch(vqt_fft_frames<512, 16, 256>)
                     : T == 512 ? launch(vqt_fft_frames<512, 16, 512>) : launch(vqt_fft_frames<1024, 16, 1024>);
    if (lst != PVQ_OK) return lst;
    if (split) {
        const float2* rows = static_cast<const float2*>(ws_split_);
        if (T == 128) hipLaunchKernelGGL(db_rows<128>, dim3((unsigned)n_frames), dim3(128), 0, stream, rows, (int)n_frames, a.n_bins, a.out_db, a.out_cplx, a.status);
        else if (T == 256) hipLaunchKernelGGL(db_rows<256>, dim3((unsigned)n_frames), dim3(256), 0, stream, rows, (int)n_frames, a.n_bins, a.out_db, a.out_cplx, a.status);
        else if (T == 512) hipLaunchKernelGGL(db_rows<512>, dim3((unsigned)n_frames), dim3(512), 0, stream, rows, (int)n_frames, a.n_bins, a.out_db, a.out_cplx, a.status);
        else hipLaunchKernelGGL(db_rows<1024>, dim3((unsigned)n_frames), dim3(1024), 0, stream, rows, (int)n_frames, a.n_bins, a.out_db, a.out_cplx, a.status);
    }
    slot_end(SLOT_FFT_FRAMES, stream);
    if (pk) {  // peak / note detection as its own launch (one wavefront per frame)
        slot_begin(SLOT_PEAKS, stream);
        pvq_status ps = launch_peaks_kernel(d_out_db, rows_total, *pk, stream);
        slot_end(SLOT_PEAKS, stream);
        if (ps != PVQ_OK) return ps;
    }
    PVQ_HIP(hipGetLastError());
    last_algo_ = PVQ_ALGO_FFT;
    last_frames_per_launch_ = (uint32_t)n_frames;
    last_gemm_flop_ = 0.0;
    return PVQ_OK;
}

pvq_status Vqt::calculate_batch_db_device(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                          float* d_out_db, float* d_out_cplx, hipStream_t stream) {
    return run_batch(d_pcm, n_lead, hop, n_frames, d_out_db, d_out_cplx, nullptr, stream);
}

pvq_status Vqt::vqt_analyze_batch_device(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames,
                                         const AnalysisParameters& ap, float* d_out_db, uint32_t* d_peak_mask,
                                         uint32_t* d_peak_count, float* d_center, float* d_size, uint32_t max_peaks,
                                         hipStream_t stream) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if ((d_center == nullptr) != (d_size == nullptr)) {
        set_last_error("analyze: only one of center/size given");
        return PVQ_ERR_INVALID_ARG;
    }
    PeakParamsDev pk;
    if (!make_peak_params(ap, d_peak_mask, d_peak_count, d_center, d_size, max_peaks, pk)) {
        set_last_error("unsupported: peak detection handles 3..1024 bins per frame");
        return PVQ_ERR_UNSUPPORTED;
    }
    return run_batch(d_pcm, n_lead, hop, n_frames, d_out_db, nullptr, &pk, stream);
}

pvq_status Vqt::order_on(hipStream_t s) {
    if (order_valid_ && order_stream_ != s) {
        if (!order_ev_) PVQ_HIP(hipEventCreateWithFlags(&order_ev_, hipEventDisableTiming));
        // a marker behind everything the previous call queued; should its stream be gone (destroyed by the caller: an invalid
        // handle), a stream that no longer exists has nothing in flight that a device-wide wait would not cover
        if (hipEventRecord(order_ev_, order_stream_) == hipSuccess) {
            PVQ_HIP(hipStreamWaitEvent(s, order_ev_, 0));
        } else {
            (void)hipGetLastError();
            PVQ_HIP(hipDeviceSynchronize());
        }
    }
    order_stream_ = s;
    order_valid_ = true;
    return PVQ_OK;
}

pvq_status Vqt::run_batch(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                          float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (n_frames == 0) return PVQ_OK;
    if (!d_pcm || !d_out_db || hop == 0 || n_frames > (size_t)0x7fffffff) {
        set_last_error("calculate_batch: null pointer, zero hop or too many frames");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    {
        pvq_status os = order_on(stream);
        if (os != PVQ_OK) return os;
    }
    // The block-DFT path takes the hop itself (r = 1), or — a hop it cannot take but whose r-fold it can, 800 -> 1 600 — r interleaved
    // block grids of hop r * hop: grid i holds the frames i, i + r, ... (each hop' block is then transformed once per grid)
    const size_t r = blockdft_hop_factor(hop);
    bool use_block = false;
    if (algo_ == PVQ_ALGO_BLOCKDFT) {
        if (r == 0) {
            set_last_error("block-DFT path: no multiple r * hop (r = 1, 2, 4, 8, 16) is a multiple of 64 samples that the windows hold at most 16 times "
                           "(or a power of two dividing every window)");
            return PVQ_ERR_UNSUPPORTED;
        }
        use_block = true;
    } else if (algo_ == PVQ_ALGO_AUTO) {
        use_block = r != 0 && n_frames >= auto_block_min_frames(hop, r);
    }
    if (use_block && r > 1 && !blockdft_takes_streams(hop * r)) {
        if (algo_ == PVQ_ALGO_BLOCKDFT) {
            set_last_error("block-DFT path: this geometry runs the unfused stages, which take the hop only as it is");
            return PVQ_ERR_UNSUPPORTED;
        }
        use_block = false;
    }
    if (use_block && r == 1) return launch_blockdft_path(d_pcm, n_lead, hop, n_frames, d_out_db, d_out_cplx, pk, stream);
    if (use_block) {
        std::vector<StreamIn> st;
        for (size_t i = 0; i < r && i < n_frames; ++i)
            st.push_back(StreamIn{d_pcm, n_lead + (i + 1) * hop, n_lead + n_frames * hop, (n_frames - i + r - 1) / r, i, r});
        return launch_blockdft_streams(st.data(), st.size(), hop * r, d_out_db, d_out_cplx, n_frames, pk, stream);
    }
    return launch_fft_path(d_pcm, n_lead, hop, n_frames, d_out_db, d_out_cplx, pk, stream);
}

// which path run_batch takes for a batch of this shape (the multi-device driver decides once for the WHOLE stream, so that a shard of a
// few frames runs the path the unsharded stream runs: the two paths agree to the parity bars, not bit for bit)
pvq_algo Vqt::resolve_algo(size_t hop, size_t n_frames) const {
    if (algo_ == PVQ_ALGO_FFT) return PVQ_ALGO_FFT;
    const size_t r = blockdft_hop_factor(hop);
    if (r == 0 || (r > 1 && !blockdft_takes_streams(hop * r))) return algo_ == PVQ_ALGO_BLOCKDFT ? PVQ_ALGO_BLOCKDFT : PVQ_ALGO_FFT;   // (forced: run_batch reports the error)
    if (algo_ == PVQ_ALGO_BLOCKDFT) return PVQ_ALGO_BLOCKDFT;
    return n_frames >= auto_block_min_frames(hop, r) ? PVQ_ALGO_BLOCKDFT : PVQ_ALGO_FFT;
}

// From how many frames on PVQ_ALGO_AUTO takes the block-DFT path (hop * r its block length).  A power-of-two hop: from 384 frames
// (and at least one tile row per grid).  A general hop is different:
// its tiles' K loops are hop * r / 2 deep, so a launch cannot end before ~180 us at 1 600 samples and ~300 us at 3 200 however few
// frames it holds, while the FFT path — a workgroup per frame, 512 of them side by side — takes 53 us for up to ~420 frames and
// 0.125 us per frame beyond (48 kHz / 252 bins; profiles/r04_small_batches.txt: 64 frames at hop 800 took 367 us on the block path
// against 64 on the FFT path).  The estimate below — both paths' time as floor + frames x slope, the slopes scaled by the geometry's
// FFT work and column count — puts the switch where the two lines cross: ~1 700 frames at hop 800 / 1 600, ~3 500 at 3 200.
size_t Vqt::auto_block_min_frames(size_t hop, size_t r) const {
    const size_t hop_eff = hop * r;
    bool divides = (hop_eff & (hop_eff - 1)) == 0;
    double fft_work = 0.0;   // sum over the window groups of W log2 W
    size_t cols = 0;         // spectrum columns the kernel reads (upper bound: every group's highest column)
    for (const WindowGroup& g : plan_.kernel.window_groups) {
        const size_t w = g.window_size();
        divides = divides && w % hop_eff == 0;
        fft_work += (double)w * std::log2((double)w);
        uint32_t top = 0;
        for (uint32_t c : g.filter_bank.col_idx) top = c > top ? c : top;
        for (uint32_t c : g.negative_filter_bank.col_idx) top = c > top ? c : top;
        cols += top + 1;
    }
    // (a power-of-two hop: both paths' launches are short; the block path's two kernels cost 58-67 us up to ~1 000 frames at 48 kHz /
    // 252 bins, the FFT path — group-split for few frames, launch_fft_streams — 19 us for one frame, 42 for 256, 56 for 400)
    if (divides) return std::max<size_t>(64 * r, 384);
    // Both paths' time for n frames, in us, as measured on one box (profiles/r05_auto_rule.txt):
    //   FFT path     t_fft (n + 500): the per-window kernels (round 5: half the walk's time) ~ the FFT work + the row dots (bins)
    //   block path   max(floor, floor / 2 + t_block n): a launch pair cannot end before `floor` however few frames it holds; per frame the K loops'
    //                depth x columns + the kernel product (bins)
    // and the switch sits at the first n (in steps of 64 r) where the block path is the faster one.  At the reference's default geometry
    // (22 050 Hz, 588 bins) the per-window FFT kernels run level with the general-hop block path — 0.035 against 0.038 us per frame at hop 1 600 —
    // and AUTO stays on the FFT path at every size.
    double t_fft = 1.45e-7 * fft_work + 1.0e-5 * (double)n_bins();
    if (plan_.params.n_fft > 0 && fft_work > 6.0e5) t_fft *= 1.15;                        // (a 32 768-sample window: 1 024 threads per frame, one workgroup per CU)
    const double floor_block = 58.0 + 0.075 * ((double)hop_eff - 256.0) + 18.0;           // us: shortest launch pair of the general-hop kernels
    const double t_block = 1.0e-5 * (double)hop_eff * ((double)cols / 871.0) + 3.2e-5 * (double)n_bins();   // (871: the column bound at 48 kHz / 252 bins, 602 of them read)
    const size_t lo = 64 * r;
    if (t_fft <= t_block) {   // the lines never cross beyond the floor: the switch, if any, lies where the FFT path reaches the block path's floor
        const double n = floor_block / t_fft - 500.0;
        return n > 0.0 && floor_block / 2 + t_block * n <= floor_block ? std::max(lo, (size_t)n) : ~(size_t)0 >> 1;
    }
    for (size_t n = lo; n < ((size_t)1 << 22); n += lo)
        if (std::max(floor_block, floor_block / 2 + t_block * (double)n) < t_fft * ((double)n + 500.0)) return n;
    return ~(size_t)0 >> 1;
}

size_t Vqt::blockdft_hop_factor(size_t hop) const {
    for (size_t r = 1; r <= 16; r *= 2)
        if (blockdft_applicable(hop * r)) return r;
    return 0;
}

pvq_status Vqt::batch_streams_device(const float* const* d_pcm, const size_t* n_lead, const size_t* n_frames, uint32_t n_streams, size_t hop,
                                     float* d_out_db, size_t stride, const AnalysisParameters* ap, uint32_t* d_peak_mask, uint32_t* d_peak_count,
                                     float* d_center, float* d_size, uint32_t max_peaks, hipStream_t stream) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (n_streams == 0) return PVQ_OK;
    if (!d_pcm || !n_frames || !d_out_db || hop == 0) {
        set_last_error("batch_streams: null pointer or zero hop");
        return PVQ_ERR_INVALID_ARG;
    }
    if ((d_center == nullptr) != (d_size == nullptr)) {
        set_last_error("batch_streams: only one of center/size given");
        return PVQ_ERR_INVALID_ARG;
    }
    bool ragged = false;
    size_t total = 0;
    for (uint32_t s = 0; s < n_streams; ++s) {
        if (n_frames[s] > stride || n_frames[s] > (size_t)0x7fffffff) {
            set_last_error("batch_streams: a stream has more frames than out_stride_frames (or than 2^31 - 1)");
            return PVQ_ERR_INVALID_ARG;
        }
        if (n_frames[s] > 0 && !d_pcm[s]) {
            set_last_error("batch_streams: null stream pointer");
            return PVQ_ERR_INVALID_ARG;
        }
        ragged |= n_frames[s] != stride;
        total += n_frames[s];
    }
    const size_t rows_total = (size_t)n_streams * stride;
    if (rows_total > (size_t)0x7fffffff) {
        set_last_error("batch_streams: more than 2^31 - 1 output rows");
        return PVQ_ERR_INVALID_ARG;
    }
    PeakParamsDev pk;
    const bool want_peaks = ap && (d_peak_mask || d_peak_count || d_center);
    if (want_peaks && !make_peak_params(*ap, d_peak_mask, d_peak_count, d_center, d_size, max_peaks, pk)) {
        set_last_error("unsupported: peak detection handles 3..1024 bins per frame");
        return PVQ_ERR_UNSUPPORTED;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    {
        pvq_status os = order_on(stream);
        if (os != PVQ_OK) return os;
    }
    const size_t nb = n_bins();
    // rows a stream does not fill are zero frames: nothing reads uninitialised memory, their peak outputs say "no peaks"
    if (ragged) PVQ_HIP(hipMemsetAsync(d_out_db, 0, rows_total * nb * sizeof(float), stream));
    if (total == 0) return PVQ_OK;
    const size_t r = blockdft_hop_factor(hop);   // 1: the hop itself; > 1: r interleaved block grids of hop r * hop (Vqt::run_batch)
    bool use_block = false;
    if (algo_ == PVQ_ALGO_BLOCKDFT) {
        if (r == 0 || (r > 1 && !blockdft_takes_streams(hop * r))) {
            set_last_error("block-DFT path: no multiple r * hop (r = 1, 2, 4, 8, 16) is a multiple of 64 samples that the windows hold at most 16 times "
                           "(or a power of two dividing every window)");
            return PVQ_ERR_UNSUPPORTED;
        }
        use_block = true;
    } else if (algo_ == PVQ_ALGO_AUTO) {
        use_block = r != 0 && total >= auto_block_min_frames(hop, r) && (r == 1 || blockdft_takes_streams(hop * r));
    }
    if (use_block && blockdft_takes_streams(hop * r)) {
        // SHORT streams are staged one behind the other into ONE buffer — each in a slot of whole 64 r-frame tiles, the zeroed gap behind
        // it holding the next stream's history — and analysed as one long stream: a tile of the GEMM then never stops at a stream's end,
        // no stream's first tile is a range-checked one (dword loads: twice the time, no 64-column pairing), and a stream shorter than a
        // tile does not leave the rest of it empty.  The price is one copy of the PCM (1 KB per frame at hop 256 against the 9.6 KB of X
        // traffic) and the gap frames (63 per stream at hop 256, 9 at hop 1 600), which are computed and dropped.  Long streams go as
        // they are (segments of the launch, no copy).  Same bits either way: a frame's values do not depend on its place in a tile.
        static const int stage_max_env = dev_knob("PVQ_STAGE_MAX", 2048);   // streams of at most this many frames are staged (0: none; measured: 2 048 frames gain 2 %, 512 gain 42 %, 4 096 lose 9 %)
        const size_t stage_max = (size_t)stage_max_env;
        const size_t wu = plan_.window_union;
        const size_t G = wu > hop ? (wu - hop + hop - 1) / hop : 0;     // frames of history a stream's first frame needs
        const size_t A = 64 * r;                                         // slots start on whole tiles of every grid
        std::vector<uint32_t> shorts;
        std::vector<StreamIn> longs;
        for (uint32_t s = 0; s < n_streams; ++s) {
            if (n_frames[s] == 0) continue;
            if (n_streams > 1 && n_frames[s] <= stage_max)
                shorts.push_back(s);
            else {
                const size_t lead = n_lead ? n_lead[s] : 0;
                for (size_t i = 0; i < r && i < n_frames[s]; ++i) {
                    StreamIn in{d_pcm[s], lead + (i + 1) * hop, lead + n_frames[s] * hop, (n_frames[s] - i + r - 1) / r, (size_t)s * stride + i, r};
                    longs.push_back(in);
                }
            }
        }
        if (shorts.size() == 1) {   // a single short stream gains nothing from a copy
            const uint32_t s = shorts[0];
            const size_t lead = n_lead ? n_lead[s] : 0;
            for (size_t i = 0; i < r && i < n_frames[s]; ++i) {
                StreamIn in{d_pcm[s], lead + (i + 1) * hop, lead + n_frames[s] * hop, (n_frames[s] - i + r - 1) / r, (size_t)s * stride + i, r};
                longs.push_back(in);
            }
            shorts.clear();
        }
        // staged buffers of at most 144 K frames (one sub-batch of the block-DFT path) and 512 MiB each, filled and analysed one after the other on `stream`
        const size_t vcap = std::max<size_t>(A * 4, std::min<size_t>((size_t)147456, ((size_t)512 << 20) / (hop * sizeof(float))) / A * A);
        size_t at = 0;
        while (at < shorts.size()) {
            std::vector<Slot> slots;
            std::vector<StagePiece> pieces;
            size_t F = (G + A - 1) / A * A;   // the first slot leaves room for the first stream's history too
            uint64_t hash = 1469598103934665603ull;
            auto mix = [&](uint64_t x) { hash = (hash ^ x) * 1099511628211ull; };
            long long longest = 0;
            while (at < shorts.size()) {
                const uint32_t s = shorts[at];
                const size_t len = (n_frames[s] + G + A - 1) / A * A;
                if (!slots.empty() && F + len > vcap) break;
                const size_t lead = n_lead ? n_lead[s] : 0;
                const size_t h = std::min(lead, std::min(G * hop, F * hop));   // what of the stream's own history the gap before its slot holds
                slots.push_back(Slot{F, n_frames[s], (size_t)s * stride});
                const long long prev_end = pieces.empty() ? 0ll : pieces.back().dst_off + pieces.back().count;
                pieces.push_back(StagePiece{d_pcm[s] + (lead - h), (long long)(h + n_frames[s] * hop), (long long)(F * hop - h), prev_end, 0ll});
                pieces.back().zero_to = pieces.back().dst_off + pieces.back().count;
                longest = std::max(longest, pieces.back().count);
                mix(F); mix(n_frames[s]); mix((uint64_t)s * stride);
                F += len;
                ++at;
            }
            const size_t FV = F, n_samp = FV * hop;
            pvq_status es = ensure_workspace(&ws_stage_, &ws_stage_cap_, n_samp * sizeof(float));
            if (es != PVQ_OK) return es;
            es = ensure_workspace(&ws_stage_tab_, &ws_stage_tab_cap_, pieces.size() * sizeof(StagePiece));
            if (es != PVQ_OK) return es;
            pieces.back().zero_to = (long long)n_samp;   // (every sample of the buffer is written: a stream's data or a gap's zeros)
            PVQ_HIP(hipMemcpyAsync(ws_stage_tab_, pieces.data(), pieces.size() * sizeof(StagePiece), hipMemcpyHostToDevice, stream));   // (pageable source: staged before the call returns)
            // (the piece table holds at most 65 535 pieces per launch of the copy kernel: grid.y)
            for (size_t p0 = 0; p0 < pieces.size(); p0 += 65535) {
                const unsigned np = (unsigned)std::min<size_t>(65535, pieces.size() - p0);
                const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((longest + 1023) / 1024, 64));
                hipLaunchKernelGGL(stage_streams, dim3(gx, np), dim3(256), 0, stream, static_cast<float*>(ws_stage_), static_cast<const StagePiece*>(ws_stage_tab_) + p0);
            }
            std::vector<StreamIn> runs;
            for (size_t i = 0; i < r; ++i) {
                StreamIn in{static_cast<const float*>(ws_stage_), (i + 1) * hop, n_samp, (FV - i + r - 1) / r, 0, r};
                in.slots = slots.data();
                in.n_slots = slots.size();
                in.grid_i = i;
                in.slot_hash = hash;
                runs.push_back(in);
            }
            pvq_status ls = launch_blockdft_streams(runs.data(), runs.size(), hop * r, d_out_db, nullptr, rows_total, nullptr, stream);
            if (ls != PVQ_OK) return ls;
        }
        if (!longs.empty()) {
            pvq_status ls = launch_blockdft_streams(longs.data(), longs.size(), hop * r, d_out_db, nullptr, rows_total, nullptr, stream);
            if (ls != PVQ_OK) return ls;
        }
        if (want_peaks) {
            slot_begin(SLOT_PEAKS, stream);
            pvq_status ps = launch_peaks_kernel(d_out_db, rows_total, pk, stream);
            slot_end(SLOT_PEAKS, stream);
            if (ps != PVQ_OK) return ps;
            PVQ_HIP(hipGetLastError());
        }
        return PVQ_OK;
    }
    if (!use_block) {   // the FFT path (any hop): all streams in ONE launch, the peaks over all rows behind it
        std::vector<FftStream> tab;
        long long f0 = 0;
        for (uint32_t s = 0; s < n_streams; ++s) {
            if (n_frames[s] == 0) continue;
            const size_t lead = n_lead ? n_lead[s] : 0;
            tab.push_back(FftStream{d_pcm[s], (long long)lead, (long long)(lead + n_frames[s] * hop), (long long)((size_t)s * stride), f0});
            f0 += (long long)n_frames[s];
        }
        return launch_fft_streams(tab.data(), tab.size(), nullptr, 0, hop, (size_t)f0, rows_total, d_out_db, nullptr, want_peaks ? &pk : nullptr, stream);
    }
    // one stream per call (the unfused block-DFT stages), the peaks once over all rows
    for (uint32_t s = 0; s < n_streams; ++s) {
        if (n_frames[s] == 0) continue;
        float* out = d_out_db + (size_t)s * stride * nb;
        pvq_status st = use_block ? launch_blockdft_path(d_pcm[s], n_lead ? n_lead[s] : 0, hop, n_frames[s], out, nullptr, nullptr, stream)
                                  : launch_fft_path(d_pcm[s], n_lead ? n_lead[s] : 0, hop, n_frames[s], out, nullptr, nullptr, stream);
        if (st != PVQ_OK) return st;
    }
    if (want_peaks) {
        slot_begin(SLOT_PEAKS, stream);
        pvq_status ps = launch_peaks_kernel(d_out_db, rows_total, pk, stream);
        slot_end(SLOT_PEAKS, stream);
        if (ps != PVQ_OK) return ps;
        PVQ_HIP(hipGetLastError());
    }
    return PVQ_OK;
}

pvq_status Vqt::calculate_batch_db(const float* pcm, size_t n_lead, size_t hop, size_t n_frames, float* out_db) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (n_frames == 0) return PVQ_OK;
    if (!pcm || !out_db || hop == 0) {
        set_last_error("calculate_batch: null pointer or zero hop");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    // a synchronous call answers for its own input only: whatever an earlier asynchronous call left in the flag (and nobody
    // polled with input_status) is dropped here, not reported against this call's clean samples
    PVQ_HIP(hipDeviceSynchronize());
    PVQ_HIP(hipMemset(dev_->d_status, 0, sizeof(uint32_t)));
    const size_t n_samples = n_lead + n_frames * hop;
    pvq_status st = ensure_workspace(&ws_pcm_, &ws_pcm_cap_, n_samples * sizeof(float));
    if (st != PVQ_OK) return st;
    st = ensure_workspace(&ws_out_, &ws_out_cap_, n_frames * n_bins() * sizeof(float));
    if (st != PVQ_OK) return st;
    const size_t nb = n_bins();
    constexpr size_t PART = 16384;   // frames per pipeline part
    if (n_frames < 2 * PART) {
        PVQ_HIP(hipMemcpy(ws_pcm_, pcm, n_samples * sizeof(float), hipMemcpyHostToDevice));
        st = calculate_batch_db_device(static_cast<const float*>(ws_pcm_), n_lead, hop, n_frames,
                                       static_cast<float*>(ws_out_), nullptr, nullptr);
        if (st != PVQ_OK) return st;
        PVQ_HIP(hipMemcpy(out_db, ws_out_, n_frames * nb * sizeof(float), hipMemcpyDeviceToHost));
        return input_status(nullptr);
    }
    // Large batches: upload, transform and download in parts on three streams, so that with page-locked host buffers
    // (pvq_host_alloc) the two PCIe directions and the kernels overlap.  Part p's frames see the parts before it as
    // history (n_lead grows), so the results are those of one call.  Pageable buffers make the copies synchronous:
    // same results, no overlap.
    if (!host_streams_ready_) {
        for (int i = 0; i < 3; ++i) PVQ_HIP(hipStreamCreateWithFlags(&host_streams_[i], hipStreamNonBlocking));
        host_streams_ready_ = true;
    }
    hipStream_t s_in = host_streams_[0], s_run = host_streams_[1], s_out = host_streams_[2];
    const size_t n_parts = (n_frames + PART - 1) / PART;
    while (host_events_.size() < 2 * n_parts) {
        hipEvent_t e;
        PVQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        host_events_.push_back(e);
    }
    float* d_pcm = static_cast<float*>(ws_pcm_);
    float* d_out = static_cast<float*>(ws_out_);
    size_t sample_done = 0;
    // the path is decided ONCE for the whole call and pinned for its parts (as analyze_batch_multi does for its shards): left to itself
    // PVQ_ALGO_AUTO would send a tail part below its threshold to the FFT path, whose values agree with the block-DFT path's to the
    // parity bars, not bit for bit — one synchronous call would mix the two
    struct AlgoPin {
        Vqt* v; pvq_algo saved;
        AlgoPin(Vqt* vv, pvq_algo a) : v(vv), saved(vv->algo()) { v->set_algo(a); }
        ~AlgoPin() { v->set_algo(saved); }
    } pin(this, resolve_algo(hop, n_frames));
    for (size_t p = 0; p < n_parts; ++p) {
        const size_t fbeg = p * PART, nf = std::min(PART, n_frames - fbeg);
        const size_t sample_end = n_lead + (fbeg + nf) * hop;
        PVQ_HIP(hipMemcpyAsync(d_pcm + sample_done, pcm + sample_done, (sample_end - sample_done) * sizeof(float), hipMemcpyHostToDevice, s_in));
        sample_done = sample_end;
        PVQ_HIP(hipEventRecord(host_events_[2 * p], s_in));
        PVQ_HIP(hipStreamWaitEvent(s_run, host_events_[2 * p], 0));
        st = calculate_batch_db_device(d_pcm, n_lead + fbeg * hop, hop, nf, d_out + fbeg * nb, nullptr, s_run);
        if (st != PVQ_OK) return st;
        PVQ_HIP(hipEventRecord(host_events_[2 * p + 1], s_run));
        PVQ_HIP(hipStreamWaitEvent(s_out, host_events_[2 * p + 1], 0));
        PVQ_HIP(hipMemcpyAsync(out_db + fbeg * nb, d_out + fbeg * nb, nf * nb * sizeof(float), hipMemcpyDeviceToHost, s_out));
    }
    PVQ_HIP(hipStreamSynchronize(s_out));
    PVQ_HIP(hipStreamSynchronize(s_run));
    return input_status(s_run);
}

pvq_status Vqt::input_status(hipStream_t stream) {
    if (!has_device() || !dev_ || !dev_->d_status) return PVQ_OK;
    PVQ_HIP(hipSetDevice(device_id_));
    // read and clear as ONE stream-ordered step behind everything the stream holds: a flag raised by work queued on this stream
    // before the call is seen, nothing of it can fall between the read and the clear
    uint32_t flag = 0;
    PVQ_HIP(hipMemcpyAsync(&flag, dev_->d_status, sizeof flag, hipMemcpyDeviceToHost, stream));
    PVQ_HIP(hipMemsetAsync(dev_->d_status, 0, sizeof flag, stream));
    PVQ_HIP(hipStreamSynchronize(stream));
    if (flag == 0) return PVQ_OK;
    set_last_error("non-finite sample (NaN / Inf) in the input: the affected frames are unspecified (the reference's audio "
                   "callback drops such chunks, audio_desktop.rs:102-105; peak_detection.rs:145 would panic)");
    return PVQ_ERR_NONFINITE_INPUT;
}

pvq_status Vqt::calculate_vqt_instant_in_db(const float* x, size_t len, float* out_db) {
    if (len != plan_.params.n_fft) {
        set_last_error("input must be exactly n_fft samples");
        return PVQ_ERR_BAD_LENGTH;
    }
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (!x || !out_db) {
        set_last_error("calculate_vqt_instant_in_db: null pointer");
        return PVQ_ERR_INVALID_ARG;
    }
    // The reference's call shape — one frame, host slice in, host vector out (the viewer: once per rendered frame) — is latency, not
    // throughput: the general host-buffer route (three streams, events, pageable copies of all n_fft samples, a flag read) took
    // 77-92 us of which the GPU worked ~20.  Here: only the window union travels (the first n_fft - window_union samples are never
    // read, SURVEY Appendix B: half of the buffer at 48 kHz, three quarters at the defaults), through page-locked staging on one
    // stream with ONE wait; the non-finite check the kernels would flag is made on the host while the samples are staged; the
    // frame runs the FFT path group-split (launch_fft_streams).  Same kernel, same samples: same bits as a batch of one.
    PVQ_HIP(hipSetDevice(device_id_));
    const size_t wu = plan_.window_union, nb = n_bins(), n_fft = plan_.params.n_fft;
    if (!inst_stream_) PVQ_HIP(hipStreamCreateWithFlags(&inst_stream_, hipStreamNonBlocking));
    if (!inst_pin_) PVQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&inst_pin_), (wu + nb) * sizeof(float), hipHostMallocDefault));
    {   // (an asynchronous batch still queued on the caller's stream owns ws_pcm_ / ws_out_ / the group-split rows until it is done)
        pvq_status os = order_on(inst_stream_);
        if (os != PVQ_OK) return os;
    }
    pvq_status st = ensure_workspace(&ws_pcm_, &ws_pcm_cap_, wu * sizeof(float));
    if (st != PVQ_OK) return st;
    st = ensure_workspace(&ws_out_, &ws_out_cap_, nb * sizeof(float));
    if (st != PVQ_OK) return st;
    const float* src = x + (n_fft - wu);
    float bad = 0.0f;   // x - x is 0 for a finite x, NaN for NaN / Inf
    for (size_t i = 0; i < wu; ++i) {
        inst_pin_[i] = src[i];
        bad += src[i] - src[i];
    }
    if (!(bad == 0.0f)) {
        set_last_error("non-finite sample (NaN / Inf) in the input: the affected frames are unspecified (the reference's audio "
                       "callback drops such chunks, audio_desktop.rs:102-105; peak_detection.rs:145 would panic)");
        return PVQ_ERR_NONFINITE_INPUT;
    }
    PVQ_HIP(hipMemcpyAsync(ws_pcm_, inst_pin_, wu * sizeof(float), hipMemcpyHostToDevice, inst_stream_));
    // frame 0 of a stream of `wu` samples with wu - 1 of them as history and a hop of 1: its n_fft buffer ends at the last sample
    st = launch_fft_path(static_cast<const float*>(ws_pcm_), wu - 1, 1, 1, static_cast<float*>(ws_out_), nullptr, nullptr, inst_stream_);
    if (st != PVQ_OK) return st;
    last_algo_ = PVQ_ALGO_FFT;
    PVQ_HIP(hipMemcpyAsync(inst_pin_ + wu, ws_out_, nb * sizeof(float), hipMemcpyDeviceToHost, inst_stream_));
    PVQ_HIP(hipStreamSynchronize(inst_stream_));
    std::copy(inst_pin_ + wu, inst_pin_ + wu + nb, out_db);
    return PVQ_OK;
}

pvq_status Vqt::launch_peaks_kernel(const float* d_db, size_t n_frames, const PeakParamsDev& a, hipStream_t stream) {
    pvq_status st = ensure_workspace(&ws_flags_, &ws_flags_cap_, n_frames);
    if (st != PVQ_OK) return st;
    return launch_peaks_frames(d_db, n_frames, a, static_cast<uint8_t*>(ws_flags_), stream);
}

// find_peaks over n_frames independent dB rows (the lean kernel, then the generic one over the few frames it flags in `redo`, n_frames bytes
// of device scratch): the frame kernels of the batch path, also the frame-parallel pre-pass of AnalysisBatch (analysis_batch.hip)
pvq_status launch_peaks_frames(const float* d_db, size_t n_frames, const PeakParamsDev& a, uint8_t* redo, hipStream_t stream) {
    const int npad = (a.n_bins + 63) / 64 * 64;
    const size_t lds_gen = PK_WAVES * (sizeof(float) * npad + peaks_scratch_bytes(a.n_bins, a.dist));
    // bins per lane: 4 (<= 256 bins), 8 (<= 512), 12 (<= 768) or 16 (<= 1024); the lean kernel also has 5 (<= 320), 6 (<= 384) and 10 (<= 640)
    auto launch_generic = [&](int g, const uint8_t* flags) {
        if (a.n_bins <= 256)
            hipLaunchKernelGGL(peaks_frames_generic<4>, dim3(g), dim3(PK_WAVES * 64), lds_gen, stream, d_db, (int)n_frames, a, flags);
        else if (a.n_bins <= 512)
            hipLaunchKernelGGL(peaks_frames_generic<8>, dim3(g), dim3(PK_WAVES * 64), lds_gen, stream, d_db, (int)n_frames, a, flags);
        else if (a.n_bins <= 768)
            hipLaunchKernelGGL(peaks_frames_generic<12>, dim3(g), dim3(PK_WAVES * 64), lds_gen, stream, d_db, (int)n_frames, a, flags);
        else
            hipLaunchKernelGGL(peaks_frames_generic<16>, dim3(g), dim3(PK_WAVES * 64), lds_gen, stream, d_db, (int)n_frames, a, flags);
    };
    const int sweep_grid = (int)std::min<size_t>(256, (n_frames + 64 * PK_WAVES - 1) / (64 * PK_WAVES));
    const int fpw = a.n_bins <= 384 ? 2 : 1;
    const size_t lds_lean = peaks_lean_lds_bytes(a.n_bins, a.dist, fpw, a.highest_bassnote);
    const size_t fpg = (size_t)PK_WAVES * fpw;   // frames per workgroup: one pass each
    auto launch_lean = [&](auto nk_c, auto dist_c) {
        constexpr int NK = decltype(nk_c)::value;
        constexpr bool D = decltype(dist_c)::value;
        for (size_t f0 = 0; f0 < n_frames; f0 += ((size_t)1 << 20) * fpg) {
            const int grid_lean = (int)std::min<size_t>((n_frames - f0 + fpg - 1) / fpg, (size_t)1 << 20);
            if (a.n_bins > 64 * (NK - 1))
                hipLaunchKernelGGL((peaks_frames_lean<NK, D, (NK <= 6 ? 2 : 1), true>), dim3(grid_lean), dim3(PK_WAVES * 64), lds_lean, stream, d_db, (int)n_frames, a, redo, (int)f0);
            else
                hipLaunchKernelGGL((peaks_frames_lean<NK, D, (NK <= 6 ? 2 : 1), false>), dim3(grid_lean), dim3(PK_WAVES * 64), lds_lean, stream, d_db, (int)n_frames, a, redo, (int)f0);
        }
    };
    using std::integral_constant;
    if (a.dist > 1) {
        if (a.n_bins <= 256) launch_lean(integral_constant<int, 4>{}, std::true_type{});
        else if (a.n_bins <= 320) launch_lean(integral_constant<int, 5>{}, std::true_type{});
        else if (a.n_bins <= 384) launch_lean(integral_constant<int, 6>{}, std::true_type{});
        else if (a.n_bins <= 512) launch_lean(integral_constant<int, 8>{}, std::true_type{});
        else if (a.n_bins <= 640) launch_lean(integral_constant<int, 10>{}, std::true_type{});   // (588 bins = 7 x 84: the reference's default geometry)
        else if (a.n_bins <= 768) launch_lean(integral_constant<int, 12>{}, std::true_type{});
        else launch_lean(integral_constant<int, 16>{}, std::true_type{});
    } else {
        if (a.n_bins <= 256) launch_lean(integral_constant<int, 4>{}, std::false_type{});
        else if (a.n_bins <= 320) launch_lean(integral_constant<int, 5>{}, std::false_type{});   // (288 bins = 8 x 36: BASELINE configs[2])
        else if (a.n_bins <= 384) launch_lean(integral_constant<int, 6>{}, std::false_type{});
        else if (a.n_bins <= 512) launch_lean(integral_constant<int, 8>{}, std::false_type{});
        else if (a.n_bins <= 640) launch_lean(integral_constant<int, 10>{}, std::false_type{});
        else if (a.n_bins <= 768) launch_lean(integral_constant<int, 12>{}, std::false_type{});
        else launch_lean(integral_constant<int, 16>{}, std::false_type{});
    }
    launch_generic(sweep_grid, redo);   // small grid: it only sweeps the (mostly clear) flags
    return PVQ_OK;
}

pvq_status Vqt::analyze_batch_device(const float* d_db, size_t n_frames, const AnalysisParameters& ap,
                                     uint32_t* d_peak_mask, uint32_t* d_peak_count, float* d_center, float* d_size,
                                     uint32_t max_peaks, hipStream_t stream) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (n_frames == 0) return PVQ_OK;
    if (!d_db || ((d_center == nullptr) != (d_size == nullptr))) {
        set_last_error("analyze_batch: null dB pointer, or only one of center/size given");
        return PVQ_ERR_INVALID_ARG;
    }
    PeakParamsDev a;
    if (!make_peak_params(ap, d_peak_mask, d_peak_count, d_center, d_size, max_peaks, a)) {
        set_last_error("unsupported: peak detection handles 3..1024 bins per frame");
        return PVQ_ERR_UNSUPPORTED;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    {
        pvq_status os = order_on(stream);
        if (os != PVQ_OK) return os;
    }
    slot_begin(SLOT_PEAKS, stream);
    pvq_status ps = launch_peaks_kernel(d_db, n_frames, a, stream);
    slot_end(SLOT_PEAKS, stream);
    if (ps != PVQ_OK) return ps;
    PVQ_HIP(hipGetLastError());
    return PVQ_OK;
}

pvq_status Vqt::analyze_batch(const float* db, size_t n_frames, const AnalysisParameters& ap, uint32_t* peak_mask,
                              uint32_t* peak_count, float* center, float* size, uint32_t max_peaks) {
    if (!has_device()) {
        set_last_error("handle was created without a device; there is no CPU fallback");
        return PVQ_ERR_NO_DEVICE;
    }
    if (n_frames == 0) return PVQ_OK;
    if (!db) {
        set_last_error("analyze_batch: null dB pointer");
        return PVQ_ERR_INVALID_ARG;
    }
    PVQ_HIP(hipSetDevice(device_id_));
    const size_t nb = n_bins(), words = (nb + 31) / 32;
    const size_t b_db = n_frames * nb * sizeof(float);
    const size_t b_mask = n_frames * words * sizeof(uint32_t);
    const size_t b_cnt = n_frames * sizeof(uint32_t);
    const size_t b_pk = n_frames * (size_t)max_peaks * sizeof(float);
    pvq_status st = ensure_workspace(&ws_out_, &ws_out_cap_, b_db);
    if (st != PVQ_OK) return st;
    st = ensure_workspace(&ws_misc_, &ws_misc_cap_, b_mask + b_cnt + 2 * b_pk + 64);
    if (st != PVQ_OK) return st;
    char* base = static_cast<char*>(ws_misc_);
    uint32_t* d_mask = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base + b_mask);
    float* d_ctr = reinterpret_cast<float*>(base + b_mask + b_cnt);
    float* d_sz = reinterpret_cast<float*>(base + b_mask + b_cnt + b_pk);
    PVQ_HIP(hipMemcpy(ws_out_, db, b_db, hipMemcpyHostToDevice));
    if (max_peaks) PVQ_HIP(hipMemset(d_ctr, 0, 2 * b_pk));
    st = analyze_batch_device(static_cast<const float*>(ws_out_), n_frames, ap, d_mask, d_cnt,
                              max_peaks ? d_ctr : nullptr, max_peaks ? d_sz : nullptr, max_peaks, nullptr);
    if (st != PVQ_OK) return st;
    PVQ_HIP(hipDeviceSynchronize());
    if (peak_mask) PVQ_HIP(hipMemcpy(peak_mask, d_mask, b_mask, hipMemcpyDeviceToHost));
    if (peak_count) PVQ_HIP(hipMemcpy(peak_count, d_cnt, b_cnt, hipMemcpyDeviceToHost));
    if (center && max_peaks) PVQ_HIP(hipMemcpy(center, d_ctr, b_pk, hipMemcpyDeviceToHost));
    if (size && max_peaks) PVQ_HIP(hipMemcpy(size, d_sz, b_pk, hipMemcpyDeviceToHost));
    return PVQ_OK;
}


// ------------------------------------------------------------------------------------------------
// several handles, one host stream (multi-device driver)
// ------------------------------------------------------------------------------------------------
namespace {
struct MultiBuf {   // (a view of one of the handle's shard buffers)
    void* p = nullptr;
    template <typename T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

// A handle's shard resources for the multi-device driver: one stream and six grow-only device buffers, kept between calls (a call per
// shard used to pay a hipStreamCreate and up to six hipMalloc / hipFree: milliseconds, and every hipFree synchronises the device).
pvq_status Vqt::multi_buffers(const size_t (&bytes)[6], void* (&out)[6], hipStream_t* st) {
    if (!multi_stream_) PVQ_HIP(hipStreamCreateWithFlags(&multi_stream_, hipStreamNonBlocking));
    for (int i = 0; i < 6; ++i) {
        if (bytes[i] == 0) {
            out[i] = nullptr;
            continue;
        }
        pvq_status e = ensure_workspace(&multi_buf_[i], &multi_cap_[i], bytes[i]);
        if (e != PVQ_OK) return e;
        out[i] = multi_buf_[i];
    }
    *st = multi_stream_;
    return PVQ_OK;
}

pvq_status analyze_batch_multi(Vqt* const* handles, uint32_t n_handles, const float* pcm, size_t n_lead, size_t hop, size_t n_frames,
                               const AnalysisParameters& ap, float* out_db, uint32_t* peak_mask, uint32_t* peak_count, float* center,
                               float* size, uint32_t max_peaks) {
    if (!handles || n_handles == 0) {
        set_last_error("analyze_batch_multi: no handles");
        return PVQ_ERR_INVALID_ARG;
    }
    if (n_frames == 0) return PVQ_OK;
    if (!pcm || !out_db || hop == 0) {
        set_last_error("analyze_batch_multi: null pointer or zero hop");
        return PVQ_ERR_INVALID_ARG;
    }
    if ((center != nullptr) != (size != nullptr) || (center && max_peaks == 0)) {
        set_last_error("analyze_batch_multi: center and size go together, with max_peaks > 0");
        return PVQ_ERR_INVALID_ARG;
    }
    for (uint32_t g = 0; g < n_handles; ++g) {
        if (!handles[g]) {
            set_last_error("analyze_batch_multi: null handle");
            return PVQ_ERR_INVALID_ARG;
        }
        if (!handles[g]->has_device()) {
            set_last_error("handle was created without a device; there is no CPU fallback");
            return PVQ_ERR_NO_DEVICE;
        }
        for (uint32_t h = 0; h < g; ++h)
            if (handles[h] == handles[g]) {
                set_last_error("analyze_batch_multi: a handle is exclusive to one worker; the same handle was passed twice");
                return PVQ_ERR_INVALID_ARG;
            }
        const VqtParameters &p0 = handles[0]->params(), &pg = handles[g]->params();
        if (std::memcmp(&p0, &pg, sizeof(VqtParameters)) != 0) {
            set_last_error("analyze_batch_multi: the handles were created with different parameters");
            return PVQ_ERR_INVALID_ARG;
        }
    }
    const size_t nb = handles[0]->n_bins(), words = (nb + 31) / 32;
    const size_t wu = handles[0]->plan().window_union;
    const bool want_peaks = peak_mask || peak_count || center;
    // one decision for the whole stream (what ONE handle would do with it), applied to every shard
    if (hipSetDevice(handles[0]->device()) != hipSuccess) {
        set_last_error("hipSetDevice failed");
        return PVQ_ERR_DEVICE;
    }
    const pvq_algo whole = handles[0]->resolve_algo(hop, n_frames);
    std::vector<pvq_status> status(n_handles, PVQ_OK);
    std::vector<std::string> message(n_handles);
    auto work = [&](uint32_t g) {
        // the worker's error text is thread-local: hand it back with the status
        auto fail = [&](pvq_status st, const std::string& msg) {
            status[g] = st;
            message[g] = msg;
        };
        try {
            ShardPlan sh;
            if (!plan_shard(n_frames, hop, wu, g, n_handles, &sh)) return fail(PVQ_ERR_INTERNAL, "plan_shard failed");
            if (sh.n_frames == 0) return;
            // plan_shard counts samples from the stream's first hop; the caller's array starts n_lead samples earlier, and a shard's
            // halo may reach into that history
            const size_t hop_begin = n_lead + (size_t)sh.first_frame * hop;
            const size_t halo = wu > hop ? wu - hop : 0;
            const size_t begin = hop_begin > halo ? hop_begin - halo : 0;
            const size_t lead = hop_begin - begin, n_samp = lead + (size_t)sh.n_frames * hop, nf = (size_t)sh.n_frames;
            Vqt* v = handles[g];
            if (hipSetDevice(v->device()) != hipSuccess) return fail(PVQ_ERR_DEVICE, "hipSetDevice failed");
            // the handle's own stream and grow-only shard buffers (kept between calls); with page-locked caller arrays (pvq_host_alloc)
            // the copies below are asynchronous DMA, with pageable ones the runtime stages them
            MultiBuf d_pcm, d_db, d_mask, d_count, d_center, d_size;
            hipStream_t st = nullptr;
            {
                const size_t bytes[6] = {n_samp * 4, nf * nb * 4, want_peaks ? nf * words * 4 : 0, want_peaks ? nf * 4 : 0,
                                         center ? nf * max_peaks * 4 : 0, center ? nf * max_peaks * 4 : 0};
                void* ptrs[6];
                const pvq_status bs = v->multi_buffers(bytes, ptrs, &st);
                if (bs != PVQ_OK) return fail(bs, get_last_error());
                d_pcm.p = ptrs[0]; d_db.p = ptrs[1]; d_mask.p = ptrs[2]; d_count.p = ptrs[3]; d_center.p = ptrs[4]; d_size.p = ptrs[5];
            }
            if (hipMemcpyAsync(d_pcm.p, pcm + begin, n_samp * 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail(PVQ_ERR_DEVICE, "upload failed");
            if (center) {   // entries beyond a frame's count are left as the caller passed them: start from the caller's contents
                const size_t off = (size_t)sh.first_frame * max_peaks;
                if (hipMemcpyAsync(d_center.p, center + off, nf * max_peaks * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(d_size.p, size + off, nf * max_peaks * 4, hipMemcpyHostToDevice, st) != hipSuccess)
                    return fail(PVQ_ERR_DEVICE, "upload failed");
            }
            pvq_status rs;
            const pvq_algo own = v->algo();
            v->set_algo(whole);
            if (want_peaks)
                rs = v->vqt_analyze_batch_device(d_pcm.as<float>(), lead, hop, nf, ap, d_db.as<float>(), d_mask.as<uint32_t>(), d_count.as<uint32_t>(),
                                                 center ? d_center.as<float>() : nullptr, center ? d_size.as<float>() : nullptr, max_peaks, st);
            else
                rs = v->calculate_batch_db_device(d_pcm.as<float>(), lead, hop, nf, d_db.as<float>(), nullptr, st);
            v->set_algo(own);
            if (rs != PVQ_OK) return fail(rs, get_last_error());
            const size_t f0 = (size_t)sh.first_frame;
            bool cp = hipMemcpyAsync(out_db + f0 * nb, d_db.p, nf * nb * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
            if (cp && peak_mask) cp = hipMemcpyAsync(peak_mask + f0 * words, d_mask.p, nf * words * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
            if (cp && peak_count) cp = hipMemcpyAsync(peak_count + f0, d_count.p, nf * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
            if (cp && center)
                cp = hipMemcpyAsync(center + f0 * max_peaks, d_center.p, nf * max_peaks * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipMemcpyAsync(size + f0 * max_peaks, d_size.p, nf * max_peaks * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
            if (!cp || hipStreamSynchronize(st) != hipSuccess) return fail(PVQ_ERR_DEVICE, "download failed");
            rs = v->input_status(st);
            if (rs != PVQ_OK) return fail(rs, get_last_error());
        } catch (const std::bad_alloc&) {
            fail(PVQ_ERR_INTERNAL, "out of host memory");
        } catch (const std::exception& e) {
            fail(PVQ_ERR_INTERNAL, e.what());
        } catch (...) {
            fail(PVQ_ERR_INTERNAL, "unknown exception in a shard worker");
        }
    };
    std::vector<std::thread> threads;
    threads.reserve(n_handles);
    for (uint32_t g = 1; g < n_handles; ++g) threads.emplace_back(work, g);
    work(0);
    for (auto& t : threads) t.join();
    for (uint32_t g = 0; g < n_handles; ++g)
        if (status[g] != PVQ_OK) {
            set_last_error("shard " + std::to_string(g) + ": " + message[g]);
            return status[g];
        }
    return PVQ_OK;
}

}  // namespace pvq
