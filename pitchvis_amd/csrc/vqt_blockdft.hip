// vqt_blockdft.hip — hop-block DFT path of the batched VQT (fp32 MFMA GEMM + phase-combine tree).
//
// What it replaces: the per-frame real FFTs of Vqt::calculate_vqt_instant_in_db (reference
// pitchvis_analysis/src/vqt.rs:876-887) when many frames are analysed at a small hop.  The
// reference (one call per frame) cannot share work between frames; a batch can.  With hop h
// dividing every analysis window W_g, frame f's window of group g is the union of the Nb_g = W_g/h
// hop blocks j = f .. f+Nb_g-1 (blocks of group g start at s_g + j*h), and the low spectrum
// columns c the sparse kernel reads (vqt.rs:725-735: c <= W_g/(2M)) are
//
//     X_f[c] = sum_{b<Nb} e^{-2 pi i c b / Nb} * P_g[f+b][c],   P_g[j][c] = sum_{m<h} x[s_g+jh+m] e^{-2 pi i c m / W_g}
//
// P is a dense real GEMM  [blocks x h] . [h x 2*n_cols]  shared by all Nb frames that see the
// block (exact-f32 MFMA v_mfma_f32_32x32x2_f32, the matrix is 100 % dense), the sum over b is a
// log2(Nb)-level tree  A_{l+1}[j] = A_l[j] + phi^(2^l) A_l[j+2^l]  evaluated in LDS, then the
// banded complex row dots (vqt.rs:889-910) and power_to_db (vqt.rs:922-954) as in the FFT path.
//
// Kernels:  blockdft_gemm (MFMA, LDS-tiled 128x64x32, double-buffered)
//           blockdft_combine (one workgroup per 64 frames x 32 columns, tree in LDS)
//           blockdft_dots_db (4 frames per workgroup, lane = output bin, ELL-packed kernel)
#include <algorithm>
#include <cmath>
#include <cstring>

#include "device_tables.hpp"
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

constexpr int GM_BM = 128;  // blocks (rows of P) per workgroup
constexpr int GM_BN = 64;   // real columns per workgroup = 32 complex spectrum columns
constexpr int GM_BK = 32;
constexpr int CB_T = 64;    // frames per combine workgroup
constexpr int CB_C = GM_BN / 2;  // complex columns per combine workgroup
constexpr int CB_MAX_NB = 64;
constexpr int CB_MAX_R = CB_T + CB_MAX_NB - 1;  // rows of P a combine workgroup stages
constexpr int DT_FB = 4;    // frames per dots workgroup
constexpr size_t kChunkFrames = 8192;  // frames per sub-batch: P and X of one chunk stay in the Infinity Cache

struct BlockGroup {
    int nb;         // hop blocks per window
    int levels;     // log2(nb)
    int n_cols;     // spectrum columns used
    int tile0;      // first GEMM column tile of this group
    int n_tiles;    // column tiles (of 32 complex columns)
    int tw_off;     // into d_comb_tw: levels x (n_tiles*32) entries
    long long s_rel;  // window begin relative to the end of the n_fft buffer: w0 - n_fft
};

struct BlockDftTables {
    size_t hop = 0;
    int n_groups = 0;
    int n_tiles = 0;   // total column tiles; Ntot = n_tiles*64 floats, XC = n_tiles*32 complex
    int nb_max = 0;
    int ell_len = 0;   // max entries per output row
    int n_bins_pad = 0;
    std::vector<BlockGroup> groups;
    float* d_E = nullptr;          // [hop][Ntot]
    int* d_tile_group = nullptr;   // [n_tiles]
    long long* d_tile_s = nullptr; // [n_tiles] window begin of the tile's group relative to the buffer end
    BlockGroup* d_groups = nullptr;
    float2* d_comb_tw = nullptr;
    float2* d_ell_val = nullptr;   // [ell_len][n_bins_pad]
    uint16_t* d_ell_col = nullptr; // X column | 0x8000 (conj)
    uint16_t* d_row_len = nullptr; // [n_bins_pad]
    float* d_P = nullptr;  size_t p_cap = 0;   // workspace
    float2* d_X = nullptr; size_t x_cap = 0;
};

void free_blockdft_tables(BlockDftTables* t) {
    if (!t) return;
    if (t->d_E) (void)hipFree(t->d_E);
    if (t->d_tile_group) (void)hipFree(t->d_tile_group);
    if (t->d_tile_s) (void)hipFree(t->d_tile_s);
    if (t->d_groups) (void)hipFree(t->d_groups);
    if (t->d_comb_tw) (void)hipFree(t->d_comb_tw);
    if (t->d_ell_val) (void)hipFree(t->d_ell_val);
    if (t->d_ell_col) (void)hipFree(t->d_ell_col);
    if (t->d_row_len) (void)hipFree(t->d_row_len);
    if (t->d_P) (void)hipFree(t->d_P);
    if (t->d_X) (void)hipFree(t->d_X);
    delete t;
}

// ------------------------------------------------------------------------------------------------
// GEMM: P[j][n] = sum_m pcm[s(n) + j*K + m] * E[m][n]        (exact fp32 MFMA)
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* pcm;
    long long n_samples;
    const float* E;
    int ld;            // Ntot
    float* P;
    int n_rows;        // rows of P to produce
    int K;             // hop
    const long long* tile_s;  // per column tile: window begin relative to the n_fft buffer end (w0 - n_fft)
    long long base;           // pcm index of the end of frame 0 of this launch (n_lead + hop + chunk offset)
};

__global__ __launch_bounds__(256) void blockdft_gemm(GemmArgs a) {
    __shared__ float As[2][GM_BM][GM_BK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK][GM_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const long long s = a.base + a.tile_s[nt];
    const int j0 = mt * GM_BM;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging
    float ra[16];
    float4 rb[2];
    const int a_row = tid >> 5, a_col = tid & 31;          // + 8*i rows
    const int b_row = tid >> 4, b_col = (tid & 15) * 4;    // + 16*i rows
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long gi = s + (long long)(j0 + a_row + 8 * i) * a.K + k0 + a_col;
            ra[i] = (gi >= 0 && gi < a.n_samples) ? a.pcm[gi] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rb[i] = *reinterpret_cast<const float4*>(a.E + (size_t)(k0 + b_row + 16 * i) * a.ld + nt * GM_BN + b_col);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 16; ++i) As[buf][a_row + 8 * i][a_col] = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Bs[buf][b_row + 16 * i][b_col]) = rb[i];
    };

    f32x16 acc0 = {0}, acc1 = {0};
    const int n_iter = a.K / GM_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        const int buf = it & 1;
        if (it + 1 < n_iter) load_tile((it + 1) * GM_BK);
        const int ar = wm * 64 + (lane & 31), kh = lane >> 5, bc = wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < GM_BK / 2; ++kk) {
            const float a0 = As[buf][ar][2 * kk + kh];
            const float a1 = As[buf][ar + 32][2 * kk + kh];
            const float b = Bs[buf][2 * kk + kh][bc];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
        }
        if (it + 1 < n_iter) store_tile(buf ^ 1);
        __syncthreads();
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
    const int col = nt * GM_BN + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int g0 = j0 + wm * 64 + row;
        if (g0 < a.n_rows) a.P[(size_t)g0 * a.ld + col] = acc0[r];
        if (g0 + 32 < a.n_rows) a.P[(size_t)(g0 + 32) * a.ld + col] = acc1[r];
    }
}

// ------------------------------------------------------------------------------------------------
// combine: X_f[c] = sum_b phi_c^b P[f+b][c] by a doubling tree in LDS
// ------------------------------------------------------------------------------------------------
struct CombineArgs {
    const float* P;
    int ld;            // Ntot floats
    float2* X;         // [n_frames][xc]
    int xc;
    int n_frames;      // frames in this chunk
    int n_rows;        // rows of P present
    const int* tile_group;
    const BlockGroup* groups;
    const float2* comb_tw;
};

__global__ __launch_bounds__(256) void blockdft_combine(CombineArgs a) {
    __shared__ float2 A[CB_MAX_R][CB_C];
    const int tid = threadIdx.x;
    const int ct = blockIdx.x;
    const int f0 = blockIdx.y * CB_T;
    const BlockGroup G = a.groups[a.tile_group[ct]];
    const int R = CB_T + G.nb - 1;
    const int c = tid & (CB_C - 1);
    // stage rows f0 .. f0+R-1 of this column tile
    for (int idx = tid; idx < R * CB_C; idx += 256) {
        const int j = idx / CB_C;
        const int row = f0 + j;
        float2 v = make_float2(0.0f, 0.0f);
        if (row < a.n_rows) v = *reinterpret_cast<const float2*>(a.P + (size_t)row * a.ld + ct * GM_BN + 2 * c);
        A[j][c] = v;
    }
    __syncthreads();
    constexpr int PER = (CB_MAX_R * CB_C + 255) / 256;
    int valid = R;
    for (int l = 0; l < G.levels; ++l) {
        const int s = 1 << l;
        valid -= s;  // rows with a complete span after this level
        const float2 w = a.comb_tw[G.tw_off + l * (G.n_tiles * CB_C) + (ct - G.tile0) * CB_C + c];
        float2 v[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = tid + t * 256;
            const int j = idx / CB_C;
            if (j < valid) {
                const float2 lo = A[j][c], hi = A[j + s][c];
                v[t] = make_float2(lo.x + (w.x * hi.x - w.y * hi.y), lo.y + (w.x * hi.y + w.y * hi.x));
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = tid + t * 256;
            const int j = idx / CB_C;
            if (j < valid) A[j][c] = v[t];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < CB_T * CB_C; idx += 256) {
        const int j = idx / CB_C;
        const int f = f0 + j;
        if (f < a.n_frames) a.X[(size_t)f * a.xc + ct * CB_C + c] = A[j][c];
    }
}

// ------------------------------------------------------------------------------------------------
// dots + dB: x_vqt[k] = sum_e val[k][e] * X[col[k][e]]  (conj entries flagged), then power_to_db
// ------------------------------------------------------------------------------------------------
struct DotsArgs {
    const float2* X;
    int xc;
    int n_frames;
    int n_bins;
    int n_bins_pad;
    const float2* ell_val;
    const uint16_t* ell_col;
    const uint16_t* row_len;
    float* out_db;     // [n_frames][n_bins]
    float2* out_cplx;  // optional
    int do_peaks;
    size_t frame0;     // global index of this launch's first frame (for the peak outputs)
    PeakParamsDev pk;
};

#define PVQ_REF_POWER (0.3f * 0.3f)
#define PVQ_A_MIN (1e-6f * 1e-6f)
#define PVQ_TOP_DB 60.0f

__global__ __launch_bounds__(256) void blockdft_dots_db(DotsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* Xs = reinterpret_cast<float2*>(smem_raw);             // [DT_FB][xc]
    float* red = reinterpret_cast<float*>(Xs + DT_FB * a.xc);     // [2][DT_FB][4]
    const int npad = (a.n_bins + 63) / 64 * 64;
    float* dbs = red + 2 * DT_FB * 4;                             // [DT_FB][npad] final dB values (fused peaks)
    unsigned char* pk_scratch = reinterpret_cast<unsigned char*>(dbs + DT_FB * npad);
    const int tid = threadIdx.x;
    const int f0 = blockIdx.x * DT_FB;
    for (int idx = tid; idx < DT_FB * a.xc; idx += 256) {
        const int fb = idx / a.xc, cc = idx - fb * a.xc;
        const int f = f0 + fb;
        Xs[idx] = (f < a.n_frames) ? a.X[(size_t)f * a.xc + cc] : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    const float ref_db = 10.0f * log10f(PVQ_REF_POWER);
    constexpr int PER = 4;  // n_bins <= 1024
    float d[PER][DT_FB];
    float mx[DT_FB], mn[DT_FB];
#pragma unroll
    for (int fb = 0; fb < DT_FB; ++fb) {
        mx[fb] = -3.40282347e+38f;
        mn[fb] = 3.40282347e+38f;
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int k = tid + t * 256;
        if (k < a.n_bins) {
            float2 acc[DT_FB];
#pragma unroll
            for (int fb = 0; fb < DT_FB; ++fb) acc[fb] = make_float2(0.0f, 0.0f);
            const int len = a.row_len[k];
            for (int e = 0; e < len; ++e) {
                const float2 v = a.ell_val[(size_t)e * a.n_bins_pad + k];
                const uint32_t cc = a.ell_col[(size_t)e * a.n_bins_pad + k];
                const int col = cc & 0x7fffu;
                const float sg = (cc & 0x8000u) ? -1.0f : 1.0f;
#pragma unroll
                for (int fb = 0; fb < DT_FB; ++fb) {
                    float2 x = Xs[fb * a.xc + col];
                    x.y *= sg;
                    acc[fb].x += v.x * x.x - v.y * x.y;
                    acc[fb].y += v.x * x.y + v.y * x.x;
                }
            }
#pragma unroll
            for (int fb = 0; fb < DT_FB; ++fb) {
                if (a.out_cplx && f0 + fb < a.n_frames) a.out_cplx[(size_t)(f0 + fb) * a.n_bins + k] = acc[fb];
                const float ns = acc[fb].x * acc[fb].x + acc[fb].y * acc[fb].y;
                const float v = 10.0f * log10f(fmaxf(ns, PVQ_A_MIN)) - ref_db;
                d[t][fb] = v;
                mx[fb] = fmaxf(mx[fb], v);
                mn[fb] = fminf(mn[fb], v);
            }
        }
    }
#pragma unroll
    for (int fb = 0; fb < DT_FB; ++fb) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx[fb] = fmaxf(mx[fb], __shfl_xor(mx[fb], o));
            mn[fb] = fminf(mn[fb], __shfl_xor(mn[fb], o));
        }
        if ((tid & 63) == 0) {
            red[fb * 4 + (tid >> 6)] = mx[fb];
            red[DT_FB * 4 + fb * 4 + (tid >> 6)] = mn[fb];
        }
    }
    __syncthreads();
#pragma unroll
    for (int fb = 0; fb < DT_FB; ++fb) {
        float m1 = red[fb * 4], m2 = red[DT_FB * 4 + fb * 4];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            m1 = fmaxf(m1, red[fb * 4 + w]);
            m2 = fminf(m2, red[DT_FB * 4 + fb * 4 + w]);
        }
        const float floor_db = m1 - PVQ_TOP_DB;
        m2 = fmaxf(m2, floor_db);
        if (f0 + fb < a.n_frames) {
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int k = tid + t * 256;
                if (k < a.n_bins) {
                    const float c = fmaxf(d[t][fb], floor_db);
                    const float r = (m2 > 0.0f) ? (c - m2) : fmaxf(c, 0.0f);
                    a.out_db[(size_t)(f0 + fb) * a.n_bins + k] = r;
                    if (a.do_peaks) dbs[fb * npad + k] = r;
                }
            }
        }
    }
    if (a.do_peaks) {
        __syncthreads();
        const int wv = tid >> 6;  // DT_FB == 4 waves: one frame each
        if (f0 + wv < a.n_frames)
            peaks_wave(dbs + wv * npad, pk_scratch + wv * peaks_scratch_bytes(a.n_bins, a.pk.dist), a.frame0 + f0 + wv,
                       a.pk, tid & 63);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool Vqt::blockdft_applicable(size_t hop) const {
    if (!has_device() || hop < (size_t)GM_BK || (hop & (hop - 1)) != 0 || hop > 4096) return false;
    if (n_bins() > 1024) return false;
    for (const WindowGroup& g : plan_.kernel.window_groups) {
        const size_t ws = g.window_size();
        if (ws % hop != 0) return false;
        if (ws / hop > (size_t)CB_MAX_NB) return false;
    }
    return true;
}

template <typename T>
static bool up(T** dst, const std::vector<T>& src) {
    if (hipMalloc(reinterpret_cast<void**>(dst), sizeof(T) * std::max<size_t>(src.size(), 1)) != hipSuccess) return false;
    if (!src.empty() && hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice) != hipSuccess) return false;
    return true;
}

pvq_status Vqt::prepare_blockdft(size_t hop) {
    if (dev_->block && dev_->block->hop == hop) return PVQ_OK;
    if (dev_->block) {
        free_blockdft_tables(dev_->block);
        dev_->block = nullptr;
    }
    auto* t = new BlockDftTables();
    t->hop = hop;
    const auto& groups = plan_.kernel.window_groups;
    t->n_groups = (int)groups.size();
    const double pi = 3.14159265358979323846;
    int tile = 0, tw_off = 0;
    for (size_t g = 0; g < groups.size(); ++g) {
        const GroupDev& D = dev_->h_groups[g];
        BlockGroup B{};
        B.nb = (int)(groups[g].window_size() / hop);
        B.levels = 0;
        while ((1 << B.levels) < B.nb) ++B.levels;
        B.n_cols = D.n_cols;
        B.tile0 = tile;
        B.n_tiles = (D.n_cols + CB_C - 1) / CB_C;
        B.tw_off = tw_off;
        B.s_rel = (long long)groups[g].window_begin - (long long)plan_.params.n_fft;  // + n_lead + hop at launch
        tile += B.n_tiles;
        tw_off += B.levels * B.n_tiles * CB_C;
        t->nb_max = std::max(t->nb_max, B.nb);
        t->groups.push_back(B);
    }
    t->n_tiles = tile;
    const int ntot = tile * GM_BN;
    std::vector<float> E((size_t)hop * ntot, 0.0f);
    std::vector<int> tile_group(tile);
    std::vector<float2> comb_tw((size_t)std::max(tw_off, 1), make_float2(0.0f, 0.0f));
    for (size_t g = 0; g < groups.size(); ++g) {
        const BlockGroup& B = t->groups[g];
        const double W = (double)groups[g].window_size();
        for (int tt = 0; tt < B.n_tiles; ++tt) tile_group[B.tile0 + tt] = (int)g;
        for (int c = 0; c < B.n_cols; ++c) {
            for (size_t m = 0; m < hop; ++m) {
                // reduce the angle exactly: (c*m) mod W in integers
                const long long prod = ((long long)c * (long long)m) % (long long)W;
                const double ang = -2.0 * pi * (double)prod / W;
                E[m * ntot + (size_t)B.tile0 * GM_BN + 2 * c] = (float)std::cos(ang);
                E[m * ntot + (size_t)B.tile0 * GM_BN + 2 * c + 1] = (float)std::sin(ang);
            }
            for (int l = 0; l < B.levels; ++l) {
                const long long prod = ((long long)c * (1ll << l)) % (long long)B.nb;
                const double ang = -2.0 * pi * (double)prod / (double)B.nb;
                comb_tw[B.tw_off + l * (B.n_tiles * CB_C) + c] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            }
        }
    }
    // ELL-packed sparse kernel addressed by X column
    const int nb = (int)n_bins();
    t->n_bins_pad = (nb + 63) / 64 * 64;
    std::vector<uint16_t> row_len(t->n_bins_pad, 0);
    int ell_len = 0;
    for (size_t g = 0; g < groups.size(); ++g) {
        const CsrMatrix& A = groups[g].filter_bank;
        const CsrMatrix& Bm = groups[g].negative_filter_bank;
        for (uint32_t r = 0; r < A.rows; ++r) {
            int len = (int)(A.row_ptr[r + 1] - A.row_ptr[r]);
            if (Bm.nnz() > 0) len += (int)(Bm.row_ptr[r + 1] - Bm.row_ptr[r]);
            row_len[groups[g].first_bin + r] = (uint16_t)len;
            ell_len = std::max(ell_len, len);
        }
    }
    t->ell_len = ell_len;
    std::vector<float2> ell_val((size_t)std::max(ell_len, 1) * t->n_bins_pad, make_float2(0.0f, 0.0f));
    std::vector<uint16_t> ell_col((size_t)std::max(ell_len, 1) * t->n_bins_pad, 0);
    for (size_t g = 0; g < groups.size(); ++g) {
        const CsrMatrix& A = groups[g].filter_bank;
        const CsrMatrix& Bm = groups[g].negative_filter_bank;
        const int xoff = t->groups[g].tile0 * CB_C;
        for (uint32_t r = 0; r < A.rows; ++r) {
            const int k = (int)(groups[g].first_bin + r);
            int e = 0;
            for (uint32_t i = A.row_ptr[r]; i < A.row_ptr[r + 1]; ++i, ++e) {
                ell_val[(size_t)e * t->n_bins_pad + k] = make_float2(A.values[i].re, A.values[i].im);
                ell_col[(size_t)e * t->n_bins_pad + k] = (uint16_t)(xoff + A.col_idx[i]);
            }
            if (Bm.nnz() > 0)
                for (uint32_t i = Bm.row_ptr[r]; i < Bm.row_ptr[r + 1]; ++i, ++e) {
                    ell_val[(size_t)e * t->n_bins_pad + k] = make_float2(Bm.values[i].re, -Bm.values[i].im);
                    ell_col[(size_t)e * t->n_bins_pad + k] = (uint16_t)((xoff + Bm.col_idx[i]) | 0x8000u);
                }
        }
    }
    if (tile * CB_C >= 0x8000) {
        free_blockdft_tables(t);
        set_last_error("unsupported: too many spectrum columns for the block-DFT path");
        return PVQ_ERR_UNSUPPORTED;
    }
    std::vector<long long> tile_s(tile, 0);
    for (size_t g = 0; g < groups.size(); ++g)
        for (int tt = 0; tt < t->groups[g].n_tiles; ++tt) tile_s[t->groups[g].tile0 + tt] = t->groups[g].s_rel;
    bool ok = up(&t->d_E, E) && up(&t->d_tile_group, tile_group) && up(&t->d_tile_s, tile_s) && up(&t->d_groups, t->groups) &&
              up(&t->d_comb_tw, comb_tw) && up(&t->d_ell_val, ell_val) && up(&t->d_ell_col, ell_col) &&
              up(&t->d_row_len, row_len);
    if (!ok) {
        free_blockdft_tables(t);
        set_last_error("hipMalloc/hipMemcpy failed while building block-DFT tables");
        return PVQ_ERR_DEVICE;
    }
    dev_->block = t;
    return PVQ_OK;
}

pvq_status Vqt::launch_blockdft_path(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                                     float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream) {
    pvq_status st = prepare_blockdft(hop);
    if (st != PVQ_OK) return st;
    BlockDftTables* t = dev_->block;
    const int ntot = t->n_tiles * GM_BN, xc = t->n_tiles * CB_C;
    const size_t chunk = std::min(n_frames, kChunkFrames);
    const size_t rows_cap = chunk + t->nb_max - 1;
    const size_t p_bytes = rows_cap * ntot * sizeof(float), x_bytes = chunk * xc * sizeof(float2);
    if (t->p_cap < p_bytes) {
        if (t->d_P) PVQ_HIP(hipFree(t->d_P));
        t->d_P = nullptr; t->p_cap = 0;
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_P), p_bytes));
        t->p_cap = p_bytes;
    }
    if (t->x_cap < x_bytes) {
        if (t->d_X) PVQ_HIP(hipFree(t->d_X));
        t->d_X = nullptr; t->x_cap = 0;
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_X), x_bytes));
        t->x_cap = x_bytes;
    }
    const long long n_samples = (long long)(n_lead + n_frames * hop);
    const int nb = (int)n_bins();
    const size_t n_chunks = (n_frames + chunk - 1) / chunk;
    for (size_t c = 0; c < n_chunks; ++c) {
        const size_t fbeg = c * chunk;
        const size_t nf = std::min(chunk, n_frames - fbeg);
        const int n_rows = (int)(nf + t->nb_max - 1);
        GemmArgs ga;
        ga.pcm = d_pcm;
        ga.n_samples = n_samples;
        ga.E = t->d_E;
        ga.ld = ntot;
        ga.P = t->d_P;
        ga.n_rows = n_rows;
        ga.K = (int)hop;
        ga.tile_s = t->d_tile_s;
        ga.base = (long long)n_lead + (long long)hop + (long long)fbeg * (long long)hop;
        slot_begin(SLOT_BLOCKDFT_GEMM, stream);
        hipLaunchKernelGGL(blockdft_gemm, dim3(t->n_tiles, (n_rows + GM_BM - 1) / GM_BM), dim3(256), 0, stream, ga);
        slot_end(SLOT_BLOCKDFT_GEMM, stream);
        CombineArgs ca;
        ca.P = t->d_P;
        ca.ld = ntot;
        ca.X = t->d_X;
        ca.xc = xc;
        ca.n_frames = (int)nf;
        ca.n_rows = n_rows;
        ca.tile_group = t->d_tile_group;
        ca.groups = t->d_groups;
        ca.comb_tw = t->d_comb_tw;
        slot_begin(SLOT_BLOCKDFT_COMBINE, stream);
        hipLaunchKernelGGL(blockdft_combine, dim3(t->n_tiles, (unsigned)((nf + CB_T - 1) / CB_T)), dim3(256), 0, stream, ca);
        slot_end(SLOT_BLOCKDFT_COMBINE, stream);
        DotsArgs da;
        da.X = t->d_X;
        da.xc = xc;
        da.n_frames = (int)nf;
        da.n_bins = nb;
        da.n_bins_pad = t->n_bins_pad;
        da.ell_val = t->d_ell_val;
        da.ell_col = t->d_ell_col;
        da.row_len = t->d_row_len;
        da.out_db = d_out_db + fbeg * nb;
        da.out_cplx = d_out_cplx ? reinterpret_cast<float2*>(d_out_cplx) + fbeg * nb : nullptr;
        da.do_peaks = 0;  // peaks run as their own launch below: the latency-bound peak logic wants the
                          // high occupancy this LDS-heavy kernel cannot give it (measured: fused +77 us, separate +29 us per 8192 frames)
        da.frame0 = fbeg;
        if (pk) da.pk = *pk; else std::memset(&da.pk, 0, sizeof da.pk);
        const size_t lds = sizeof(float2) * DT_FB * xc + sizeof(float) * 2 * DT_FB * 4 +
                           sizeof(float) * DT_FB * ((nb + 63) / 64 * 64) +
                           0;
        slot_begin(SLOT_BLOCKDFT_DOTS, stream);
        hipLaunchKernelGGL(blockdft_dots_db, dim3((unsigned)((nf + DT_FB - 1) / DT_FB)), dim3(256), lds, stream, da);
        slot_end(SLOT_BLOCKDFT_DOTS, stream);
        if (pk) {
            PeakParamsDev p2 = *pk;  // outputs of this chunk
            const size_t words = (nb + 31) / 32;
            if (p2.mask) p2.mask += fbeg * words;
            if (p2.count) p2.count += fbeg;
            if (p2.center) p2.center += fbeg * p2.max_peaks;
            if (p2.size) p2.size += fbeg * p2.max_peaks;
            slot_begin(SLOT_PEAKS, stream);
            launch_peaks_kernel(d_out_db + fbeg * nb, nf, p2, stream);
            slot_end(SLOT_PEAKS, stream);
        }
    }
    PVQ_HIP(hipGetLastError());
    last_algo_ = PVQ_ALGO_BLOCKDFT;
    last_frames_per_launch_ = (uint32_t)chunk;
    return PVQ_OK;
}

}  // namespace pvq
