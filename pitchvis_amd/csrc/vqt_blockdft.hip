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
// Kernels:  blockdft_gemm_tree[_bf16x3] (MFMA GEMM + combine tree fused; windows of <= 64 hop blocks)
//           blockdft_gemm + blockdft_combine (the same two stages unfused, for longer windows)
//           blockdft_banddots_db (kernel product as a banded MFMA GEMM + power_to_db)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

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

constexpr int GM_BN = 64;   // granularity of the column tiling: 64 floats = 32 complex spectrum columns
constexpr int CB_T = 128;    // frames per combine workgroup
constexpr int CB_C = GM_BN / 2;  // complex columns per combine workgroup
constexpr int CB_MAX_NB = 256;  // hop blocks per window the combine tree supports (<= 64: 32-column tiles, else 16)
// Frames per sub-batch: the X (+ Y) workspace is sized for one.  As many as the handle's workspace limit holds (default 1 GiB:
// 131 072 frames at 48 kHz / 252 bins = 0.74 GB; wide geometries — 840 bins: 15 KB per frame — take fewer), a multiple of 64,
// at most 147 456 (BASELINE configs[2]'s 131 072 per rank is one sub-batch; against two of 65 536 the step is 4 % shorter: one ramp and
// one tail per kernel instead of two); sub-batches small enough for the Infinity Cache were measured no faster (32 768: 9 % slower).
static size_t chunk_frames(size_t limit_bytes, size_t bytes_per_frame) {
    const long knob = dev_knob("PVQ_CHUNK_FRAMES", 0);   // (developer build; read per call)
    if (knob >= 64) return (size_t)knob;
    size_t f = limit_bytes / std::max<size_t>(bytes_per_frame, 1) / 64 * 64;
    return std::min<size_t>(std::max<size_t>(f, 64), 147456);   // (131 072 + an eighth: 64 staged streams of 2 048 frames with their gaps are 135 168 frames — one launch, not one and a 4 000-frame tail)
}

struct BlockGroup {
    int nb;         // hop blocks per window
    int levels;     // log2(nb)
    int n_cols;     // spectrum columns used
    int tile0;      // first GEMM column tile of this group
    int n_tiles;    // column tiles (of 32 complex columns)
    int tw_off;     // into d_comb_tw: levels x (n_tiles*32) entries
    long long s_rel;  // window begin relative to the end of the n_fft buffer: w0 - n_fft
    int nb_f;       // blocks summed inside the fused kernel: min(nb, 64); the remaining levels run in blockdft_tree_finish
    int levels_f;   // log2(nb_f)
    // general hops (a multiple of 64 that does not divide the window, blockdft_gemm_gen): window = nq whole hop blocks + rem samples
    int nq, rem;
    int e16r_off;   // float4 index of the group's slices of E16R (the DFT matrix of the first rem samples of a block)
    int gtw_off;    // float2 index into gen_tw: phi (n_tiles * 32 columns), then tau
};

// 16 output bins (rows of one window group's kernel) and the contiguous range of X columns they read
struct BandBlock {
    int x0;      // first X column
    int kb;      // columns walked (multiple of BD_KU; coefficients beyond the true range are zero)
    int boff;    // first column of this block in d_band_B (units of 64 floats)
    int bin0;    // first output bin
    int nrows;   // 1..16
    int kg;      // 8-column groups walked by the split-bf16 form
    int boff3;   // first group of this block in d_band_B3 (units of 3 planes x 64 lanes x 8 bf16)
};
constexpr int BD_RB = 16;   // bins per block: 32 MFMA columns = 16 x (re, im)
constexpr int BD_KU = 4;    // columns per software-pipeline stage
constexpr int BD_NS = 4;    // pipeline stages
constexpr int BD8_RB = 8;   // bins per block of the 16x16x4 form
constexpr int BD8_KU = 4;   // columns per stage (two column pairs = two MFMAs per 16-frame tile)
constexpr int BD8_NS = 4;   // stages in the operand ring
constexpr int X_PAD_COLS = 32;   // zeroed columns after the last X column (the operand prefetch runs past a block's range)

struct BlockDftTables {
    size_t hop = 0;
    int n_groups = 0;
    int n_tiles = 0;   // total column tiles; Ntot = n_tiles*64 floats, XC = n_tiles*32 complex
    int nb_max = 0;
    int n_bins_pad = 0;
    std::vector<BlockGroup> groups;
    std::vector<float> h_E;        // host copy of E
    float* d_E = nullptr;          // [hop][Ntot]
    __bf16* d_Et = nullptr;        // [3][Ntot][hop] hi/mid/lo bf16 planes of E^T (split-bf16 GEMM), built on first use
    int* d_tile_group = nullptr;   // [n_tiles]
    long long* d_tile_s = nullptr; // [n_tiles] window begin of the tile's group relative to the buffer end
    BlockGroup* d_groups = nullptr;
    float2* d_comb_tw = nullptr;
    // banded kernel product: blocks of 16 output bins x their union of spectrum columns, as MFMA B operands
    struct BandBlock* d_band = nullptr;
    float* d_band_B = nullptr;     // per block and column: 64 floats in v_mfma_f32_32x32x2_f32 B-operand lane order
    __bf16* d_band_B3 = nullptr;   // per block and 8 columns: 3 planes x 64 lanes x 8 bf16 in v_mfma_f32_32x32x16_bf16 order
    // 8-bin blocks for the 16x16x4 MFMA form of the kernel product (fp32, 64-frame tiles)
    struct BandBlock* d_band8 = nullptr;
    float* d_band_B4 = nullptr;    // per block and 4 columns: 64 x (Re coefficient, Im coefficient): the no-swap form
    int* d_band_list8 = nullptr;   // [8][band_per_wave8]
    int band_per_wave8 = 0;
    int* d_band_list = nullptr;    // [band_waves][band_per_wave]: per wave of a workgroup, the count and then the blocks it walks
    int band_per_wave = 0;
    int band_waves = 4;            // waves per kernel-product workgroup (8 when the 64-frame form is used)
    float* d_P = nullptr;  size_t p_cap = 0;   // workspace
    float2* d_X = nullptr; size_t x_cap = 0;
    float2* d_Y = nullptr; size_t y_cap = 0;   // 64-block partial sums (windows of more than 64 blocks)
    // frame-stripe tile order of the fused kernels, built per (frames, tile rows) and kept for the next launch
    struct SegKey {   // one run of a launch: what decides its tiles, where its frames go
        long long pcm_off, base, out_row0;
        unsigned pcm_bytes;
        int nf, x_tile0, y_tile0, row_step;
        unsigned long long slot_hash;   // a run over a staged buffer of many streams: hash of its slots (0: none), its grid offset and first frame
        long long grid_i, fbeg;
        bool operator==(const SegKey& o) const {
            return pcm_off == o.pcm_off && base == o.base && out_row0 == o.out_row0 && pcm_bytes == o.pcm_bytes && nf == o.nf && x_tile0 == o.x_tile0 && y_tile0 == o.y_tile0 &&
                   row_step == o.row_step && slot_hash == o.slot_hash && grid_i == o.grid_i && fbeg == o.fbeg;
        }
    };
    struct TileList {
        int4* d = nullptr; size_t cap = 0;            // device: the list, then the launch's segment table and X-tile map
        const struct SegDev* d_segs = nullptr;
        const struct XTile* d_xmap = nullptr;
        std::vector<SegKey> key;                      // the runs the list was built for (stream geometry included: which tiles may pair up / take 16-byte loads)
        int bm = 0, wide = 0, blocks = 0, kind = 0;   // kind: 0 power-of-two hop (GEMM + tree); 1 / 2: R / Q tiles of a general hop
        bool multi = false;
        double eff_tiles = 0.0;                       // MFMA work of the list in whole 32-column tiles
        double eff_flop = 0.0;                        // ... in flop (general hops: the depth differs by tile kind and group)
        std::vector<size_t> slot_data;                // the staged streams' slots the X-tile map was built from (compared by content: the key's slot_hash alone is a hash)
    } tile_lists[8];   // eight slots: a batch's first, middle and last sub-batch alternate without rebuilding; a general hop takes two lists per launch shape
    int tile_list_next = 0;
    unsigned long long* d_clk = nullptr; size_t clk_cap = 0; int clk_n = 0;   // K-loop clock samples of the last profiled launch: 4 slots per sampled tile
    float4* d_E16 = nullptr;       // E in the B-operand order of the 16x16x4 GEMM: [column tile][k < hop / 2][n < 16]
    bool general = false;          // the hop does not divide the windows: blockdft_gemm_gen (whole hop blocks + the window's remainder)
    float4* d_E16R = nullptr;      // general hops: per group and column tile [k < rem / 2][n < 16]
    float2* d_gen_tw = nullptr;    // general hops: per group phi, tau (n_tiles * 32 columns each)
};

void free_blockdft_tables(BlockDftTables* t) {
    if (!t) return;
    if (t->d_E) (void)hipFree(t->d_E);
    if (t->d_Et) (void)hipFree(t->d_Et);
    if (t->d_tile_group) (void)hipFree(t->d_tile_group);
    if (t->d_tile_s) (void)hipFree(t->d_tile_s);
    if (t->d_groups) (void)hipFree(t->d_groups);
    if (t->d_comb_tw) (void)hipFree(t->d_comb_tw);
    if (t->d_band) (void)hipFree(t->d_band);
    if (t->d_band_B) (void)hipFree(t->d_band_B);
    if (t->d_band8) (void)hipFree(t->d_band8);
    if (t->d_band_B4) (void)hipFree(t->d_band_B4);
    if (t->d_band_list8) (void)hipFree(t->d_band_list8);
    if (t->d_band_B3) (void)hipFree(t->d_band_B3);
    if (t->d_band_list) (void)hipFree(t->d_band_list);
    if (t->d_P) (void)hipFree(t->d_P);
    if (t->d_X) (void)hipFree(t->d_X);
    if (t->d_Y) (void)hipFree(t->d_Y);
    for (auto& tl : t->tile_lists)
        if (tl.d) (void)hipFree(tl.d);
    if (t->d_clk) (void)hipFree(t->d_clk);
    if (t->d_E16) (void)hipFree(t->d_E16);
    if (t->d_E16R) (void)hipFree(t->d_E16R);
    if (t->d_gen_tw) (void)hipFree(t->d_gen_tw);
    delete t;
}

// ------------------------------------------------------------------------------------------------
// GEMM: P[j][n] = sum_m pcm[s(n) + j*K + m] * E[m][n]        (exact fp32 MFMA)
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 16-byte raw buffer load.  Bound to the LLVM intrinsic by name: this compiler lowers
// __builtin_amdgcn_raw_buffer_load_b64 / _b128 to a single-dword load.
__device__ f32x4 pvq_raw_buffer_load_f32x4(i32x4 srsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// LDS-DMA (global_load_lds_*): 64 lanes x 16 (4) bytes from per-lane global addresses to wave-uniform LDS base + 16 (4) * lane, no register
// destination; counted in vmcnt like any load.
typedef __attribute__((address_space(1))) const void pvq_gvoid;
typedef __attribute__((address_space(3))) void pvq_lvoid;
__device__ __forceinline__ void lds_dma16(const void* g_lane, void* lds_wave) {
    __builtin_amdgcn_global_load_lds((pvq_gvoid*)g_lane, (pvq_lvoid*)lds_wave, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const void* g_lane, void* lds_wave) {
    __builtin_amdgcn_global_load_lds((pvq_gvoid*)g_lane, (pvq_lvoid*)lds_wave, 4, 0, 0);
}

struct GemmArgs {
    const float* pcm_base;    // rebased per launch so that byte offsets fit 32 bits
    unsigned pcm_bytes;       // bytes readable from pcm_base (hardware bounds check: beyond -> 0)
    const float* E;
    int ld;                   // Ntot
    float* P;
    int n_rows;               // rows of P to produce
    int K;                    // hop
    const long long* tile_s;  // per 64-float column tile: window begin relative to the n_fft buffer end (w0 - n_fft)
    long long base;           // index, relative to pcm_base, of the end of frame 0 of this launch
    int n_col_tiles;          // column tiles of this kernel's BN
    int p_rows;               // row capacity of the tile-major P: P[(tile64 * p_rows + row) * 64 + (col & 63)]
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// Fused form (windows of <= 64 hop blocks): the 128 x 32-complex-column tile of P never leaves the
// workgroup.  After the K loop the accumulators go to LDS (aliasing the staging buffers), the doubling
// tree runs there, and only X_f for the tile's 128 - Nb + 1 complete frames is written.  Row tiles of a
// group therefore advance by S_g = 129 - Nb_g blocks (1.02x ... 1.97x recomputation of the GEMM rows,
// +20 % MFMA work at 48 kHz / hop 256) in exchange for dropping the P round trip through memory
// (172 MB written + ~200 MB read per 32 768 frames) and the separate combine launch.
// ------------------------------------------------------------------------------------------------
// MANY streams in one launch (pvq_vqt_*_streams: the trainer's shape, pitchvis_train/src/train.rs:146-163 — many files side by side).
// A launch covers a list of SEGMENTS, each a contiguous run of frames of one stream; a tile list entry names its segment
// (.x bits 16..31), the segment table gives the tile its stream (offset from the launch's base pointer, readable bytes, where
// frame 0 of the run ends) and the run's first 64-frame tile in X / Y; XTile maps an X tile back to the output rows it holds.
// segs == nullptr: one segment described by the kernel arguments themselves (the single-stream entry points: unchanged).
struct alignas(16) SegDev {
    long long pcm_off;    // samples from GemmTreeArgs::pcm_base to the segment's rebased stream pointer
    long long base;       // index, relative to that pointer, of the end of the segment's frame 0
    unsigned pcm_bytes;   // bytes readable from that pointer
    int n_frames;         // frames of the segment
    int x_tile0, y_tile0; // its first 64-frame tile in X / in Y
};
struct XTile {
    long long out_row0;   // output row (of out_db, masks, ...) of the tile's frame 0
    int live_step;        // bits 0..7: frames of the tile that exist (<= 64); bits 8..: output rows between consecutive frames (1, or r
                          // for a run that holds every r-th frame of its stream: Vqt::run_batch's interleaved block grids)
    int y_tile;           // the Y tile that holds the same frames' 64-block partial sums
};

struct GemmTreeArgs {
    const float* pcm_base;
    unsigned pcm_bytes;
    const float* E;           // [K][Ntot], (cos, sin) of e^{-i th_c u_m} interleaved per column; the fp32 form reads rows m < K/2
    int ld;                   // Ntot
    float2* X;                // frame-tile blocked: X[((frame / 64) * xcp + col) * 64 + frame % 64]
    float2* Y;                // same layout, 64-block partial sums of the groups whose windows span more than 64 blocks
    int xcp;                  // columns per frame tile (incl. the zeroed pad columns)
    int n_frames;             // frames of this launch
    int K;                    // hop
    long long base;           // index, relative to pcm_base, of the end of frame 0 of this launch
    int n_groups;
    const int4* tile_list;    // (group, column tile in the group, first frame, position) per tile — frame-stripe order in eight queues, see launch
    const BlockGroup* groups;
    BlockGroup gv[8];         // the same descriptors by value (the fused path takes at most 8 window groups): read from the kernel argument segment, not through a second dependent memory round trip
    const float2* comb_tw;
    const __bf16* Et;         // [3][Ntot][K] hi/mid/lo planes of E^T (split-bf16 form only)
    const float4* E16;        // [column tile][k < K / 2][n < 16]: (cos c_n, cos c_{n+16}, -sin c_n, -sin c_{n+16}): B operands of the 16x16x4 fp32 form
    unsigned long long* stamps;   // developer knob PVQ_STAMPS: [workgroup][8] 100 MHz clock: 0 start, 1 after K loop, 2 after tree, 3 end, 4 all waves past the K loop, 5 P tile in LDS, 6 register levels done
    unsigned long long* clk;      // profiling only (pvq_vqt_set_profiling): every 64th workgroup stores (shader clock, 100 MHz clock) before and after its K loop
    const SegDev* segs;           // many-streams launches: the segment table (nullptr: one segment = the arguments above)
    // general hops (blockdft_gemm_gen)
    const float4* E16R;           // per group and column tile: [k < rem / 2][n < 16], as E16
    const float2* gen_tw;         // per group: phi_c = e^{-2 pi i c hop / W} and tau_c (see the kernel), n_tiles * 32 columns each
    int gen_kind;                 // 1: remainder tiles (R' -> Y, or -> X for windows shorter than the hop); 2: whole-block tiles (Q', combined, + tau R' -> X)
};
#define PVQ_STAMP(i) \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)stamp_slot * 8 + (i)] = wall_clock64();   // stamp_slot: the tile's (workgroup's) row of the dump

constexpr int FT_BM = 128, FT_BN = 64;

// which (group, row tile, column tile) a workgroup of the fused kernels owns
struct FusedTile {
    BlockGroup G;
    int S;        // complete frames per row tile
    int ntl, nt;  // column tile within the group / global
    int f0;       // first frame == first block row
    int nfr;      // rows this group produces: n_frames complete frames, or n_frames + nb - 64 partial sums when nb > 64
    int xt0, yt0; // first 64-frame tile of the tile's segment in X / Y (0 in a single-stream launch)
};
// the stream a tile reads and where its frames go: the kernel arguments, or the tile's entry of the segment table
struct TileStream {
    const float* pcm_base;
    unsigned pcm_bytes;
    long long base;
    int n_frames, xt0, yt0;
};
__device__ __forceinline__ TileStream tile_stream(const GemmTreeArgs& a, int entry_x) {
    TileStream ts{a.pcm_base, a.pcm_bytes, a.base, a.n_frames, 0, 0};
    if (a.segs) {
        // The entry is the same for every lane, but the compiler cannot know that of loaded data: without the readfirstlanes the
        // buffer resource built from it counts as lane-dependent and EVERY operand load of the K loop is wrapped in a waterfall loop.
        const int4* sp = reinterpret_cast<const int4*>(a.segs + ((unsigned)entry_x >> 16));
        const int4 lo = sp[0], hi = sp[1];   // (pcm_off, base), (pcm_bytes, n_frames, x_tile0, y_tile0)
        auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        const long long pcm_off = (long long)(((unsigned long long)(unsigned)rfl(lo.y) << 32) | (unsigned)rfl(lo.x));
        ts.pcm_base = a.pcm_base + pcm_off;
        ts.base = (long long)(((unsigned long long)(unsigned)rfl(lo.w) << 32) | (unsigned)rfl(lo.z));
        ts.pcm_bytes = (unsigned)rfl(hi.x);
        ts.n_frames = rfl(hi.y);
        ts.xt0 = rfl(hi.z);
        ts.yt0 = rfl(hi.w);
    }
    return ts;
}
template <int BM = FT_BM>
__device__ __forceinline__ FusedTile fused_tile_of(const GemmTreeArgs& a, const int4& e, const TileStream& ts) {   // tile list entry: (group, column tile, first frame, slot)
    FusedTile t;
    t.G = a.gv[e.x];
    t.S = BM - t.G.nb_f + 1;
    t.nfr = ts.n_frames + t.G.nb - t.G.nb_f;
    t.xt0 = ts.xt0;
    t.yt0 = ts.yt0;
    t.ntl = e.y;
    t.f0 = e.z;
    t.nt = t.G.tile0 + t.ntl;
    return t;
}
template <int BM = FT_BM>
__device__ __forceinline__ FusedTile fused_tile(const GemmTreeArgs& a, TileStream& ts) {   // (the kernels of 32-column tiles: their lists carry no wide entries)
    int4 e = a.tile_list[blockIdx.x];
    ts = tile_stream(a, e.x);
    e.x &= 7;
    return fused_tile_of<BM>(a, e, ts);
}

// doubling tree over the [128][32 complex] P tile in LDS (rows padded to 33 so that the transposed store
// below is bank-conflict free), then the store of the S complete frames, column-major
constexpr int FT_LDP = CB_C + 1;                      // P tile row stride in complex elements
// the tile's combine twiddles (levels x 32 columns) into LDS, issued before the K loop so the tree never waits on memory
constexpr int FT_MAXL = 6;
__device__ __forceinline__ void fused_stage_twiddles(float2 (*tw)[CB_C], const FusedTile& t, const GemmTreeArgs& a, int tid) {
    const int l = tid >> 5, c = tid & (CB_C - 1);
    if (l < t.G.levels_f) tw[l][c] = a.comb_tw[t.G.tw_off + l * (t.G.n_tiles * CB_C) + t.ntl * CB_C + c];
}

// lo + w * hi with a fixed operation order — (re, im) = fma((-w.y, w.y), (hi.y, hi.x), fma((w.x, w.x), (hi.x, hi.y), lo)),
// two packed fp32 fmas (v_pk_fma_f32): every tree level, whichever code path evaluates it, rounds identically, so a
// frame's result does not depend on where it sits in a tile
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 tree_cmadd(float2 lo, float2 w, float2 hi) {
    const f32x2 t = __builtin_elementwise_fma((f32x2){w.x, w.x}, (f32x2){hi.x, hi.y}, (f32x2){lo.x, lo.y});
    const f32x2 r = __builtin_elementwise_fma((f32x2){-w.y, w.y}, (f32x2){hi.y, hi.x}, t);
    return make_float2(r.x, r.y);
}

// the first R <= 4 tree levels (strides 1 .. 8) in registers: a thread owns 16 consecutive rows of one column and
// reads them plus the 2^R - 1 rows above once; every A_{l+1}[j] = A_l[j] + w_l A_l[j + 2^l] is evaluated exactly as
// the level-by-level form would, without a pass through LDS per level.  Rows past the tile read as zero: they only
// feed outputs that are themselves incomplete.
template <int R, int BM>
__device__ __forceinline__ void fused_tree_register_levels(float2 (*A)[CB_C + 1], const float2 (*tw)[CB_C], int tid) {
    constexpr int H = (1 << R) - 1;
    const int c = tid & (CB_C - 1), j0 = (tid >> 5) * 16;
    float2 v[16 + H];
#pragma unroll
    for (int i = 0; i < 16 + H; ++i) v[i] = A[j0 + i][c];   // rows BM .. BM + 14 are spare rows of the tile (zeroed by the caller)
    int len = 16 + H;
#pragma unroll
    for (int l = 0; l < R; ++l) {
        const int st = 1 << l;
        const float2 w = tw[l][c];
        len -= st;
#pragma unroll
        for (int i = 0; i < 16 + H; ++i)
            if (i < len) v[i] = tree_cmadd(v[i], w, v[i + st]);
    }
    __syncthreads();   // every thread has read its halo
#pragma unroll
    for (int i = 0; i < 16; ++i) A[j0 + i][c] = v[i];
    __syncthreads();
}

template <int BM = FT_BM>   // BM rows, 2 * BM threads
__device__ __forceinline__ void fused_tree_levels(float* smem, const float2 (*tw)[CB_C], const FusedTile& t, const GemmTreeArgs& a, int tid, int stamp_slot) {
    float2 (*A)[FT_LDP] = reinterpret_cast<float2 (*)[FT_LDP]>(smem);  // [BM][33]
    const int c = tid & (CB_C - 1);
    constexpr int THREADS = 2 * BM;
    constexpr int PER = BM * CB_C / THREADS;  // 16
    auto cmadd = [](float2 lo, float2 w, float2 hi) { return tree_cmadd(lo, w, hi); };
    const int levels = t.G.levels_f;
    int l = levels < 4 ? levels : 4;
    switch (l) {   // wave-uniform
        case 1: fused_tree_register_levels<1, BM>(A, tw, tid); break;
        case 2: fused_tree_register_levels<2, BM>(A, tw, tid); break;
        case 3: fused_tree_register_levels<3, BM>(A, tw, tid); break;
        case 4: fused_tree_register_levels<4, BM>(A, tw, tid); break;
        default: break;
    }
    PVQ_STAMP(6);
    int valid = BM - ((1 << l) - 1);
    // remaining levels (strides >= 16) through LDS, two per pass where possible: evaluated exactly as two radix-2
    // levels (same operations in the same order), outputs in groups of four to bound the registers
    for (; l + 1 < levels; l += 2) {
        const int st = 1 << l;
        const float2 w1 = tw[l][c], w2 = tw[l + 1][c];
        valid -= 3 * st;
        float2 v[PER];
#pragma unroll
        for (int g = 0; g < PER; g += 4) {
#pragma unroll
            for (int q = g; q < g + 4; ++q) {
                const int j = (tid + q * THREADS) / CB_C;
                if (j < valid) {
                    const float2 t0 = cmadd(A[j][c], w1, A[j + st][c]);
                    const float2 t1 = cmadd(A[j + 2 * st][c], w1, A[j + 3 * st][c]);
                    v[q] = cmadd(t0, w2, t1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / CB_C;
            if (j < valid) A[j][c] = v[q];
        }
        __syncthreads();
    }
    for (; l < levels; ++l) {
        const int st = 1 << l;
        valid -= st;
        const float2 w = tw[l][c];
        float2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / CB_C;
            if (j < valid) v[q] = cmadd(A[j][c], w, A[j + st][c]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / CB_C;
            if (j < valid) A[j][c] = v[q];
        }
        __syncthreads();
    }
    PVQ_STAMP(2);
}
// the tile's S complete frames to X (or Y): lanes walk the frames of one column: 512-byte runs in memory, conflict-free LDS reads
template <int BM = FT_BM>
__device__ __forceinline__ void fused_store_x(float* smem, const FusedTile& t, const GemmTreeArgs& a, int tid) {
    float2 (*A)[FT_LDP] = reinterpret_cast<float2 (*)[FT_LDP]>(smem);  // [BM][33]
    const int j = tid % BM;
    const int f = t.f0 + j;
    if (j < t.S && f < t.nfr) {
        // windows of more than 64 blocks: 64-block partial sums go to Y, blockdft_tree_finish adds the last levels
        const bool to_y = t.G.nb > t.G.nb_f;
        float2* dst = (to_y ? a.Y : a.X) + ((size_t)((f >> 6) + (to_y ? t.yt0 : t.xt0)) * a.xcp + t.nt * CB_C) * 64 + (f & 63);
        const int ncv = t.G.n_cols - t.ntl * CB_C < CB_C ? t.G.n_cols - t.ntl * CB_C : CB_C;   // the tile's real columns: the padding of a group's last tile is never read with a non-zero coefficient and never written (X starts out zeroed)
#pragma unroll 4
        for (int cc = tid / BM; cc < ncv; cc += 2) {   // streamed out (non-temporal): the kernel-product stage that reads X back runs 5 % faster for it
            const float2 val = A[j][cc];
            __builtin_nontemporal_store((f32x2){val.x, val.y}, reinterpret_cast<f32x2*>(&dst[cc * 64]));   // one 8-byte store
        }
    }
}
template <int BM = FT_BM>
__device__ __forceinline__ void fused_tree_store(float* smem, const float2 (*tw)[CB_C], const FusedTile& t, const GemmTreeArgs& a, int tid, int stamp_slot) {
    fused_tree_levels<BM>(smem, tw, t, a, tid, stamp_slot);
    fused_store_x<BM>(smem, t, a, tid);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // stores issued and acknowledged
        __syncthreads();
        PVQ_STAMP(3);
    }
}

// fp32 MFMA form.  The hop DFT is evaluated about the centre of the hop block: with u_m = m - (K-1)/2,
//     P'[j][c] = sum_{m < K/2} (x[m] + x[K-1-m]) cos(th_c u_m)  +  i sum_{m < K/2} (x[m] - x[K-1-m]) (-sin(th_c u_m))
// (cos is even, sin odd about the centre), i.e. two real GEMMs of depth K/2 — the real parts from the mirrored sums,
// the imaginary parts from the mirrored differences — instead of one of depth K: half the MFMA work, exactly.
// P = rho_c P' with rho_c = e^{-i th_c (K-1)/2}; the tree is linear per column, so X = rho_c X' and the constant
// phase is folded into the kernel-product coefficients on the host (prepare_blockdft).
//
// The PCM matrix goes from memory straight into MFMA operand registers, no LDS staging and no barrier in the K loop:
// a wave owns 32 block rows x all 32 complex columns of the tile; lane (row = lane & 31, half = lane >> 5) fetches
// the 16 consecutive samples k0 + 16 half .. + 15 of its row and the 16 mirrored ones (64-byte runs: every cache line
// it touches is consumed by four back-to-back loads), forms the 16 sums and 16 differences in registers, and these
// ARE the A operands of 16 v_mfma_f32_32x32x2_f32 pairs (k order within a stage: k0 + 16 half + t; any order works
// as long as the B rows follow it).  acc0 += sums x cosines, acc1 += differences x (-sines).  The tile's slice of E
// (rows m < K/2, (cos, -sin) interleaved) is staged into LDS once per 128 rows of K/2 and read as one b64 per pair.
constexpr int FR_KC = 128;   // rows of E staged per pass (32 KB)
// idx_f / idx_b: sample index (relative to pcm_base) of the lane's front run and of its mirrored run
template <bool VEC>
__device__ __forceinline__ void fused_f32_stage_load(const i32x4& rsrc4, __amdgpu_buffer_rsrc_t rsrc, long long idx_f, long long idx_b,
                                                     float (&fr)[16], float (&bk)[16]) {
    if (VEC) {   // the whole tile lies inside the stream: offsets are plain non-negative byte offsets
        const unsigned off_f = (unsigned)(idx_f * 4ll), off_b = (unsigned)(idx_b * 4ll);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)(off_f + 16u * q), 0, 0);
            const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)(off_b + 16u * q), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fr[4 * q + i] = v[i];
                bk[4 * q + i] = w[i];
            }
        }
    } else {
        // tiles that touch the stream start / end: samples before the stream get an explicit out-of-range offset
        // (a wrapped negative offset plus the instruction's immediate offset would not wrap in the hardware's range
        // check), samples past the end are zeroed by the range check itself
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long long jf = idx_f + q, jb = idx_b + q;
            const unsigned of = jf >= 0 ? (unsigned)(jf * 4ll) : 0xFFFFFFFCu;
            const unsigned ob = jb >= 0 ? (unsigned)(jb * 4ll) : 0xFFFFFFFCu;
            fr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, of, 0, 0));
            bk[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ob, 0, 0));
        }
    }
}

template <bool VEC, int BM, typename Args>
__device__ __forceinline__ void fused_f32_kloop(const Args& a, float* smem, long long idx_f0, long long idx_b0, const float* e_tile,
                                                int tid, f32x16& acc0, f32x16& acc1) {
    constexpr int THREADS = 2 * BM;
    const int lane = tid & 63;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(a.pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)a.pcm_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pcm_base), 0, a.pcm_bytes, 0x00020000);
    const int K2 = a.K / 2;
    const float2* bsl = reinterpret_cast<const float2*>(smem) + (lane >> 5) * 16 * CB_C + (lane & 31);   // row 16 half, column lane & 31
    float fr[16], bk[16];
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        fused_f32_stage_load<VEC>(rsrc4, rsrc, idx_f0 + kc, idx_b0 - kc, fr, bk);
        if (kc > 0) __syncthreads();   // every wave is done with the previous slice
        for (int i = tid; i < rows * (FT_BN / 4); i += THREADS) {
            const int r = i / (FT_BN / 4), c4 = i % (FT_BN / 4);
            *reinterpret_cast<float4*>(smem + r * FT_BN + c4 * 4) = *reinterpret_cast<const float4*>(e_tile + (size_t)(kc + r) * a.ld + c4 * 4);
        }
        __syncthreads();
        for (int k0 = 0; k0 < rows; k0 += 32) {
            float sm[16], df[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                sm[t] = fr[t] + bk[15 - t];
                df[t] = fr[t] - bk[15 - t];
            }
            if (k0 + 32 < rows)
                fused_f32_stage_load<VEC>(rsrc4, rsrc, idx_f0 + (kc + k0 + 32), idx_b0 - (kc + k0 + 32), fr, bk);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 b = bsl[(k0 + t) * CB_C];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sm[t], b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(df[t], b.y, acc1, 0, 0, 0);
            }
        }
    }
}

// The same mirrored GEMM on v_mfma_f32_16x16x4_f32 (the form blockdft_gemm_tree runs).  A wave still owns 32 block rows x 32
// complex columns (two 16-row tiles x (re, im) x two 16-column halves = 8 accumulators of 4 registers), but lane
// (row = lane & 15, kq = lane >> 4) now fetches 4 consecutive samples of its row and the 4 mirrored ones with ONE 16-byte load
// each: the four lanes of a row read one contiguous 64-byte run, 16 cache lines per load instruction where the 32x32x2 form's
// lanes touch 64 — its address processing took as long as its MFMAs (16 vs 15.6 us per workgroup pair), which is what held a
// workgroup running its K loop alone (its CU partner in its tree / store phase) at 63 % of the matrix pipe.  MFMA t of a k
// group (16 mirrored sample pairs) takes sample 4 kq + t of every lane; B operand of lane (n, kq): row 16 g + 4 kq + t of the
// E slice, (cos c_n, cos c_{n+16}, -sin c_n, -sin c_{n+16}) as one 16-byte LDS read.  Operands double-buffered, one k group ahead.
typedef float f32x4a __attribute__((ext_vector_type(4)));
template <int BM, bool HALF>   // tiles that lie wholly inside the stream (all but a handful per launch); HALF: at most 16 columns (a group's last tile): the second 16-column half is not computed
// depth: samples of a row the DFT runs over (the hop; the general-hop kernel's second GEMM runs over the first `rem` samples of rows that
// still lie a.K = hop samples apart)
__device__ __forceinline__ void fused_f32_kloop16(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, float* smem, long long tile_lo, const float4* e_tile, int tid,
                                                  f32x4a (&accR)[2][2], f32x4a (&accI)[2][2], const float* tw_src, float* tw_dst, int tw_levels, int tw_stride, int stamp_slot, int depth) {
    constexpr int THREADS = 2 * BM;
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)pcm_bytes, 0x00020000};
    const int K2 = depth / 2;
    const int nG = K2 / 32;   // (K2 is a multiple of 32: the fused path takes hops that are multiples of 64)
    // A load step fetches a DOUBLE k group (32 mirrored sample pairs): lane (row, kq) takes the 8 consecutive samples 32 G + 8 kq ...
    // of its row and the 8 mirrored ones, two 16-byte loads each, issued back to back — the four lanes of a row read one whole
    // 128-byte line at a time.  Fetched 16 pairs at a time (one 64-byte half line per step, the other half a step later) every line
    // crossed the L2 -> L1 path twice: by then the CU's other waves had pushed it out of the L1 again, and the K loop ran at the
    // L2's 64-66 GB/s per CU, not at the matrix pipe's rate.  MFMA t of half h of double group G takes sample
    // 32 G + 8 kq + 4 h + t of every lane; the B rows follow that order.
    // Addresses: one loop-invariant byte offset per lane, row tile and direction; the double group moves in the instruction's
    // SCALAR offset (front runs + 128 G bytes; the mirrored runs are anchored at the LAST double group and take + 128 (nG - 1 - G)).
    unsigned vf[2], vb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long row_lo = tile_lo + (long long)(wave * 32 + mt * 16 + m16) * a.K;
        vf[mt] = (unsigned)((row_lo + 8 * kq) * 4ll);
        vb[mt] = (unsigned)((row_lo + depth - 8 - 8 * kq) * 4ll) - 128u * (unsigned)(nG - 1);
    }
    float fr[2][2][8], bk[2][2][8];
    // The prefetch is UNCONDITIONAL (the group index is clamped: past the last group the last one is fetched again into the idle
    // buffer).  Inside a uniform `if` the compiler must place the s_waitcnt for the path on which the loads were NOT issued: it waited
    // for vmcnt(4), then vmcnt(0) right after issuing the eight loads of the next group, i.e. for the loads it had just issued — the
    // K loop ran with no prefetch distance at all, hidden only while a CU's other workgroup had its own K loop to run.
    auto load_dgroup = [&](int buf, int G) {   // G: double k group of the whole depth
        const int Gc = G < nG - 1 ? G : nG - 1;
        const int sf = 128 * Gc, sb = 128 * (nG - 1 - Gc);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)vf[mt], sf + 16 * h, 0);   // (the half's 16 bytes ride in the scalar offset too: one address register per run)
                const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)vb[mt], sb + 16 * h, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fr[buf][mt][4 * h + t] = v[t];
                    bk[buf][mt][4 * h + t] = w[t];
                }
            }
        }
    };
    float4* El = reinterpret_cast<float4*>(smem);   // [rows][16]
    // one k row = 8 MFMAs; its B operand (one 16-byte LDS read per lane) is fetched one row ahead, so that no MFMA waits on the LDS
    const float4* erow = El + (8 * kq) * 16 + m16;   // row 32 Gl + 8 kq + 4 h + t of the staged slice
    auto b_at = [&](int Gl, int h, int t) { return erow[(32 * Gl + 4 * h + t) * 16]; };
    auto mfma_row = [&](int buf, int h, int t, const float4& b) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float sm = fr[buf][mt][4 * h + t] + bk[buf][mt][7 - 4 * h - t];
            const float df = fr[buf][mt][4 * h + t] - bk[buf][mt][7 - 4 * h - t];
            accR[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, b.x, accR[mt][0], 0, 0, 0);
            if (!HALF) accR[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, b.y, accR[mt][1], 0, 0, 0);
            accI[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, b.z, accI[mt][0], 0, 0, 0);
            if (!HALF) accI[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, b.w, accI[mt][1], 0, 0, 0);
        }
    };
    // the 8 k rows of double group Gl from operand buffer `buf`; bc: the first row's B operand (already fetched), returns the next group's
    auto mfma_dgroup = [&](int buf, int Gl, float4 bc) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int h = r >> 2, t = r & 3;
            // (past the slice's last row this reads on into the workgroup's LDS region: in bounds, never used)
            const float4 bn = r < 7 ? b_at(Gl, (r + 1) >> 2, (r + 1) & 3) : b_at(Gl + 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // the next row's read is in flight before this row's MFMAs
            mfma_row(buf, h, t, bc);
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
        }
        return bc;
    };
    // The slice of E goes straight to LDS (LDS-DMA, 1 KB per wave instruction, no registers), issued ahead of the first operand
    // loads: ONE memory round trip before the first MFMA.  (As a load -> wait -> ds_write loop it was four dependent round
    // trips, each also waiting for the operand loads issued before it, behind two more for the tile's twiddles and descriptor.)
    auto stage_e = [&](int kc, int rows) {   // 1 KB pieces of the slice (rows x 256 bytes), dealt to the waves; a full slice unrolled: all pieces in flight together
        constexpr int NWV = THREADS / 64, FULL = FR_KC / 4 / NWV;
        const float4* src = e_tile + (size_t)kc * 16 + wave * 64 + lane;
        float4* dst = El + wave * 64;
        if (rows == FR_KC && FR_KC / 4 % NWV == 0) {
#pragma unroll
            for (int q = 0; q < FULL; ++q) lds_dma16(src + q * (NWV * 64), dst + q * (NWV * 64));
        } else {
            for (int j = wave; j < rows / 4; j += NWV) lds_dma16(e_tile + (size_t)kc * 16 + j * 64 + lane, El + j * 64);
        }
    };
    stage_e(0, K2 < FR_KC ? K2 : FR_KC);
    for (int l = wave; l < tw_levels; l += THREADS / 64)   // the tile's combine twiddles: one level (32 complex = 64 floats) per wave instruction
        lds_dma4(tw_src + (size_t)l * tw_stride, tw_dst + l * (2 * CB_C));
    load_dgroup(0, 0);
    // (one wait for everything the prologue fetched.  Waiting only for the DMA pieces — s_waitcnt vmcnt(8) + a raw s_barrier — lets a
    // wave start on its own first operands, but hipcc does not credit a hand-written wait: with an LDS-DMA "possibly in flight" it
    // waits vmcnt(0) at the first use of every later load, which takes the K loop's prefetch distance away again.)
    __syncthreads();
    PVQ_STAMP(7);
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        if (kc > 0) {   // (hops of 512 and more: the next slice of E replaces the one every wave is done with)
            __syncthreads();
            stage_e(kc, rows);
            __syncthreads();
        }
        const int ng = rows / 32, G0 = kc / 32;
        float4 bc = b_at(0, 0, 0);
        for (int Gl = 0; Gl < ng; Gl += 2) {   // two double groups per pass: buffer indices stay compile-time
            load_dgroup(1, G0 + Gl + 1);
            bc = mfma_dgroup(0, Gl, bc);
            if (Gl + 1 >= ng) break;           // (a 32-row slice: hop 64)
            load_dgroup(0, G0 + Gl + 2);
            bc = mfma_dgroup(1, Gl + 1, bc);
        }
    }
}

// The same loop for the tiles that touch the stream's start or end (dword loads, each range-checked by the buffer hardware;
// samples before the stream get an explicit out-of-range offset, see fused_f32_stage_load): same MFMAs on the same samples in
// the same order — k group g = 2 G + h — but one 16-pair half of a double group per load step, which keeps its 32 dword loads
// per step inside the register budget.
template <int BM>
__device__ __forceinline__ void fused_f32_kloop16_edge(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, float* smem, long long tile_lo, const float4* e_tile, int tid,
                                                       f32x4a (&accR)[2][2], f32x4a (&accI)[2][2], const float* tw_src, float* tw_dst, int tw_levels, int tw_stride, int stamp_slot, int depth) {
    constexpr int THREADS = 2 * BM;
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pcm_base), 0, pcm_bytes, 0x00020000);
    const int K2 = depth / 2;
    int jf0[2], jb0[2];   // sample indices relative to pcm_base (|.| < 2^30: the launch's stream is at most 4 GB)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row_lo = (int)tile_lo + (wave * 32 + mt * 16 + m16) * a.K;
        jf0[mt] = row_lo + 8 * kq;
        jb0[mt] = row_lo + depth - 4 - 8 * kq;
    }
    float fr[2][2][4], bk[2][2][4];
    auto load_group = [&](int buf, int g) {
        const int so = 32 * (g >> 1) + 4 * (g & 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xf = jf0[mt] + so + t, xb = jb0[mt] - so + t;
                fr[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xf >= 0 ? (unsigned)xf * 4u : 0xFFFFFFFCu, 0, 0));
                bk[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xb >= 0 ? (unsigned)xb * 4u : 0xFFFFFFFCu, 0, 0));
            }
    };
    float4* El = reinterpret_cast<float4*>(smem);   // [rows][16]
    auto mfma_group = [&](int buf, int gl) {        // gl: k group inside the staged slice
        const float4* e = El + (32 * (gl >> 1) + 8 * kq + 4 * (gl & 1)) * 16 + m16;
        float4 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = e[t * 16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float sm = fr[buf][mt][t] + bk[buf][mt][3 - t];
                const float df = fr[buf][mt][t] - bk[buf][mt][3 - t];
                accR[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, b[t].x, accR[mt][0], 0, 0, 0);
                accR[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, b[t].y, accR[mt][1], 0, 0, 0);
                accI[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, b[t].z, accI[mt][0], 0, 0, 0);
                accI[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, b[t].w, accI[mt][1], 0, 0, 0);
            }
        }
    };
    load_group(0, 0);
    for (int l = wave; l < tw_levels; l += THREADS / 64) tw_dst[l * (2 * CB_C) + lane] = tw_src[(size_t)l * tw_stride];
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        if (kc > 0) __syncthreads();   // every wave is done with the previous slice
        for (int i = tid; i < rows * 16; i += THREADS) El[i] = e_tile[(size_t)kc * 16 + i];
        __syncthreads();
        const int ng = rows / 16, g0 = kc / 16;
        for (int gl = 0; gl < ng; gl += 2) {   // two k groups per pass: buffer indices stay compile-time (rows is a multiple of 32)
            load_group(1, g0 + gl + 1);
            mfma_group(0, gl);
            if (g0 + gl + 2 < K2 / 16) load_group(0, g0 + gl + 2);
            mfma_group(1, gl + 1);
        }
    }
}

// WIDE tiles: the same rows against TWO neighbouring column tiles (64 complex columns) in one K loop.  What bounds a workgroup is
// not a resource but the chain of latencies a tile pays once — launch gap, descriptor, the first operand burst, the waves' skew at
// the loop's end — so a wide tile pays them once for twice the MFMAs.  A wave holds 16 accumulators (64 registers);
// the A operands are single-buffered per 16-row tile and fetched one STAGE ahead — stage = (row tile, double k group) = 64 MFMAs,
// the same prefetch distance in MFMAs as the narrow loop's double buffer: while row tile 1 of double group G runs, row tile 0's
// loads of group G + 1 are in flight, and so on in turn.  Every accumulator adds the same products in the same order as in the
// narrow loop: results are bit-identical whichever form a tile takes.  The two tiles' slices of E lie 32 KB apart in LDS.
template <int BM>
__device__ __forceinline__ void fused_f32_kloop64(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, float* smem, long long tile_lo, const float4* e_tile, int tid,
                                                  f32x4a (&accR)[2][4], f32x4a (&accI)[2][4], const float* tw_src, float* tw_dst, int tw_levels, int tw_stride, int stamp_slot) {
    constexpr int THREADS = 2 * BM, NWV = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)pcm_bytes, 0x00020000};
    const int K2 = a.K / 2;
    const int nG = K2 / 32;
    unsigned vf[2], vb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long row_lo = tile_lo + (long long)(wave * 32 + mt * 16 + m16) * a.K;
        vf[mt] = (unsigned)((row_lo + 8 * kq) * 4ll);
        vb[mt] = (unsigned)((row_lo + a.K - 8 - 8 * kq) * 4ll) - 128u * (unsigned)(nG - 1);
    }
    float fr[2][8], bk[2][8];
    auto load_stage = [&](int mt, int G) {   // unconditional, clamped (see fused_f32_kloop16)
        const int Gc = G < nG - 1 ? G : nG - 1;
        const int sf = 128 * Gc, sb = 128 * (nG - 1 - Gc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)vf[mt], sf + 16 * h, 0);
            const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)vb[mt], sb + 16 * h, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fr[mt][4 * h + t] = v[t];
                bk[mt][4 * h + t] = w[t];
            }
        }
    };
    float4* El = reinterpret_cast<float4*>(smem);   // [column tile][FR_KC rows][16]
    const float4* erow = El + (8 * kq) * 16 + m16;
    auto b_at = [&](int Gl, int r, int ct) { return erow[(32 * Gl + r) * 16 + ct * (FR_KC * 16)]; };
    // one stage: 8 k rows x 2 column tiles = 16 steps of 4 MFMAs, the B operand fetched one step ahead
    auto mfma_stage = [&](int mt, int Gl, int Gl_next, float4 bc) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = i >> 1, ct = i & 1;
            const float4 bn = i < 15 ? b_at(Gl, (i + 1) >> 1, (i + 1) & 1) : b_at(Gl_next, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const float sm = fr[mt][r] + bk[mt][7 - r];
            const float df = fr[mt][r] - bk[mt][7 - r];
            accR[mt][2 * ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, bc.x, accR[mt][2 * ct], 0, 0, 0);
            accR[mt][2 * ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, bc.y, accR[mt][2 * ct + 1], 0, 0, 0);
            accI[mt][2 * ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, bc.z, accI[mt][2 * ct], 0, 0, 0);
            accI[mt][2 * ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, bc.w, accI[mt][2 * ct + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
        }
        return bc;
    };
    auto stage_e = [&](int kc, int rows) {   // both tiles' slices, 1 KB pieces dealt to the waves
        constexpr int FULL = FR_KC / 4 / NWV;
        if (rows == FR_KC && FR_KC / 4 % NWV == 0) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < FULL; ++q)
                    lds_dma16(e_tile + (size_t)ct * K2 * 16 + (size_t)kc * 16 + (q * NWV + wave) * 64 + lane, El + ct * (FR_KC * 16) + (q * NWV + wave) * 64);
        } else {
            for (int j = wave; j < 2 * (rows / 4); j += NWV) {
                const int ct = j >= rows / 4, jj = ct ? j - rows / 4 : j;
                lds_dma16(e_tile + (size_t)ct * K2 * 16 + (size_t)kc * 16 + jj * 64 + lane, El + ct * (FR_KC * 16) + jj * 64);
            }
        }
    };
    stage_e(0, K2 < FR_KC ? K2 : FR_KC);
    // twiddles: levels x 64 complex columns = two 64-float pieces per level, [column tile][level][32]
    for (int j = wave; j < 2 * tw_levels; j += NWV) {
        const int l = j >> 1, ct = j & 1;
        lds_dma4(tw_src + (size_t)l * tw_stride + ct * (2 * CB_C), tw_dst + (ct * FT_MAXL + l) * (2 * CB_C));
    }
    load_stage(0, 0);
    load_stage(1, 0);
    __syncthreads();
    PVQ_STAMP(7);
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        if (kc > 0) {
            __syncthreads();
            stage_e(kc, rows);
            __syncthreads();
        }
        const int ng = rows / 32, G0 = kc / 32;
        float4 bc = b_at(0, 0, 0);
        for (int Gl = 0; Gl < ng; ++Gl) {
            bc = mfma_stage(0, Gl, Gl, bc);
            load_stage(0, G0 + Gl + 1);
            bc = mfma_stage(1, Gl, Gl + 1, bc);
            load_stage(1, G0 + Gl + 1);
        }
    }
}

// P' accumulators of one 32-column tile -> LDS as [row][32 complex + pad]  (C/D layout of the 16x16 MFMA: column = lane & 15,
// rows 4 (lane >> 4) + r), and the 15 spare rows zeroed
template <int BM, int NP, int NP0>
__device__ __forceinline__ void fused_dump_p(float* smem, const f32x4a (&accR)[2][NP], const f32x4a (&accI)[2][NP], int tid) {
    float2 (*Pt)[FT_LDP] = reinterpret_cast<float2 (*)[FT_LDP]>(smem);
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Pt[wave * 32 + mt * 16 + 4 * kq + r][np * 16 + m16] = make_float2(accR[mt][NP0 + np][r], accI[mt][NP0 + np][r]);
    for (int i = tid; i < 15 * FT_LDP; i += 2 * BM) Pt[BM][i] = make_float2(0.0f, 0.0f);   // the spare rows (Pt[BM][..] runs on through them)
}

// one 32-column tile from the K loop to the store
template <int BM>
__device__ __forceinline__ void fused_f32_narrow_tile(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, float* smem, float2 (*tw_lds)[CB_C], const FusedTile& T, bool inside,
                                                      long long tile_lo, int tid, int stamp_slot) {
    const int lane = tid & 63;
    // the tile's combine twiddles (levels x 32 complex columns = 64 floats per level): a wave copies a level, a dword per lane
    const float* tw_src = reinterpret_cast<const float*>(a.comb_tw + T.G.tw_off + T.ntl * CB_C) + lane;   // level l: + l * tw_stride floats
    float* tw_dst = reinterpret_cast<float*>(&tw_lds[0][0]);
    const int tw_levels = T.G.levels_f, tw_stride = 2 * T.G.n_tiles * CB_C;
    const float4* e_tile = a.E16 + (size_t)T.nt * (a.K / 2) * 16;
    f32x4a accR[2][2], accI[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accR[mt][np][r] = 0.0f;
                accI[mt][np][r] = 0.0f;
            }
    // the clock the chip holds under this kernel's MFMA load: shader-clock ticks over 100 MHz ticks across the K loop.  The
    // start stamps go straight to memory so that nothing stays live in registers across the loop
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 6) * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    // a group's last column tile may hold 16 columns or fewer (3 of the 21 tiles at 48 kHz / 252 bins): half the MFMAs
    const bool half = T.ntl == T.G.n_tiles - 1 && T.G.n_cols - T.ntl * CB_C <= 16;
    if (!inside)
        fused_f32_kloop16_edge<BM>(a, pcm_base, pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot, a.K);
    else if (half)
        fused_f32_kloop16<BM, true>(a, pcm_base, pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot, a.K);
    else
        fused_f32_kloop16<BM, false>(a, pcm_base, pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot, a.K);
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 6) * 4 + 2] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    PVQ_STAMP(1);
    __syncthreads();   // the E slice is dead: the P' tile takes its place
    PVQ_STAMP(4);
    fused_dump_p<BM, 2, 0>(smem, accR, accI, tid);
    __syncthreads();
    PVQ_STAMP(5);
    fused_tree_store<BM>(smem, tw_lds, T, a, tid, stamp_slot);
}


// three more stamps per workgroup, behind the gridDim.x rows of eight: its first instruction, its last wave's last store issued, and that
// store acknowledged — what lies between one workgroup's last and the next one's first is the dispatcher's (scripts/dev_conc.py)
#define PVQ_END_STAMPS \
    if (a.stamps) { \
        unsigned long long* ends = a.stamps + (size_t)gridDim.x * 8 + (size_t)stamp_slot * 4; \
        if (lane == 0) atomicMax(ends + 1, wall_clock64()); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        if (lane == 0) atomicMax(ends + 2, wall_clock64()); \
        if (tid == 0) ends[0] = t_entry; \
    }
template <int BM, int KFIX = 0>   // rows of hop blocks per tile; 2 * BM threads = BM / 32 waves of 32 rows x 32 (wide tiles: 64) complex columns; KFIX: the hop, where the instantiation knows it (0: any)
__global__ __launch_bounds__(2 * BM, 4) void blockdft_gemm_tree(GemmTreeArgs a) {
    if constexpr (KFIX != 0) __builtin_assume(a.K == KFIX);   // 4 waves per SIMD = two 512-thread (four 256-thread) workgroups per CU: at most 128 registers
    constexpr bool WIDE = BM == 256;                   // (the 128-row form, a test shape, takes narrow tiles only)
    constexpr int B_FLOATS = (WIDE ? 2 : 1) * FR_KC * FT_BN;   // a wide tile's two slices of E
    constexpr int P_FLOATS = (BM + 15) * FT_LDP * 2;   // 15 spare rows: the register tree levels read their halo without a range check
    __shared__ __attribute__((aligned(16))) float smem[B_FLOATS > P_FLOATS ? B_FLOATS : P_FLOATS];  // the E slice(s), then the P tile
    __shared__ float2 tw_lds[2][FT_MAXL][CB_C];
    const unsigned long long t_entry = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int4 entry = a.tile_list[blockIdx.x];   // .x: group | wide << 8 | segment << 16
    const TileStream ts = tile_stream(a, entry.x);
    FusedTile T = fused_tile_of<BM>(a, make_int4(entry.x & 7, entry.y, entry.z, entry.w), ts);   // (gv[8])
    const bool wide = ((entry.x >> 8) & 1) != 0;
    if (T.f0 >= T.nfr) return;
    const int stamp_slot = blockIdx.x;
    PVQ_STAMP(0);
    const long long s = ts.base + T.G.s_rel;
    const long long tile_lo = s + (long long)T.f0 * a.K, tile_hi = tile_lo + (long long)BM * a.K;  // sample range of the tile
    const bool inside = tile_lo >= 0 && tile_hi * 4ll <= (long long)ts.pcm_bytes;
    if (WIDE && wide && inside) {
        const float* tw_src = reinterpret_cast<const float*>(a.comb_tw + T.G.tw_off + T.ntl * CB_C) + lane;
        float* tw_dst = reinterpret_cast<float*>(&tw_lds[0][0][0]);
        const float4* e_tile = a.E16 + (size_t)T.nt * (a.K / 2) * 16;
        f32x4a accR[2][4], accI[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int np = 0; np < 4; ++np)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accR[mt][np][r] = 0.0f;
                    accI[mt][np][r] = 0.0f;
                }
        if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
            a.clk[(blockIdx.x >> 6) * 4 + 0] = __builtin_amdgcn_s_memtime();
            a.clk[(blockIdx.x >> 6) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        fused_f32_kloop64<BM>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, T.G.levels_f, 2 * T.G.n_tiles * CB_C, stamp_slot);
        if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
            a.clk[(blockIdx.x >> 6) * 4 + 2] = __builtin_amdgcn_s_memtime();
            a.clk[(blockIdx.x >> 6) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        }
        PVQ_STAMP(1);
        // the two tiles' P' one after the other through the same buffer (the second waits in its accumulators)
        __syncthreads();   // the E slices are dead
        PVQ_STAMP(4);
        fused_dump_p<BM, 4, 0>(smem, accR, accI, tid);
        __syncthreads();
        PVQ_STAMP(5);
        fused_tree_store<BM>(smem, tw_lds[0], T, a, tid, stamp_slot);
        __syncthreads();
        fused_dump_p<BM, 4, 2>(smem, accR, accI, tid);
        __syncthreads();
        T.ntl += 1;
        T.nt += 1;
        fused_tree_store<BM>(smem, tw_lds[1], T, a, tid, stamp_slot);
        PVQ_END_STAMPS
        return;
    }
    // (the host pairs only tiles that lie inside the stream — same test, same numbers, launch(): the range-checked loop takes one tile)
    fused_f32_narrow_tile<BM>(a, ts.pcm_base, ts.pcm_bytes, smem, tw_lds[0], T, inside, tile_lo, tid, stamp_slot);
    PVQ_END_STAMPS
}

// ------------------------------------------------------------------------------------------------
// THREE workgroups per CU (round 5; DESIGN.md 5.2 "(i)").  The 256 x 32 tile's life is a serial chain of latencies of which the K loop
// is about a third; two such lives per CU leave the matrix pipe idle whenever both are outside their K loops.  This form buys a third
// life with registers and LDS instead of a longer K share per life: at most 80 registers (6 waves per SIMD) and 38 KB of LDS, so
//   * the K loop keeps ONE operand buffer per 16-row tile, fetched one stage (row tile x double k group = 32 MFMAs) ahead — the wide
//     loop's scheme on one column tile: 32 accumulator + 32 operand registers;
//   * the P' tile goes through LDS in two 16-column QUARTERS ([256 + 15][17] complex = 36.9 KB), one after the other through the same
//     buffer (the second waits in its 16 accumulator registers), each with the doubling tree and the store of its own.
// Every accumulator adds the same products in the same order as in the other K loops, every tree level is the same tree_cmadd: a
// frame's bits do not depend on which form computed it (tests/test_configs_gpu.py::test_tile_shapes_bit_identical_in_subprocesses).
// ------------------------------------------------------------------------------------------------
constexpr int Q_C = 16;            // complex columns of a P quarter
constexpr int Q_LDP = Q_C + 1;     // its row stride in LDS
template <int BM, bool HALF>
__device__ __forceinline__ void fused_f32_kloop16s(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, float* smem, long long tile_lo, const float4* e_tile, int tid,
                                                   f32x4a (&accR)[2][2], f32x4a (&accI)[2][2], const float* tw_src, float* tw_dst, int tw_levels, int tw_stride, int stamp_slot) {
    constexpr int THREADS = 2 * BM, NWV = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)pcm_bytes, 0x00020000};
    const int K2 = a.K / 2;
    const int nG = K2 / 32;
    // one byte offset per lane and direction (row tile 0); row tile 1 lies 16 rows = 16 K samples further on: that, the double group and
    // the half ride in the instruction's SCALAR offset
    const long long row_lo = tile_lo + (long long)(wave * 32 + m16) * a.K;
    const unsigned vf = (unsigned)((row_lo + 8 * kq) * 4ll);
    const unsigned vb = (unsigned)((row_lo + a.K - 8 - 8 * kq) * 4ll) - 128u * (unsigned)(nG - 1);
    const int mt_step = 64 * a.K;   // bytes between the two row tiles of a wave
    float fr[2][8], bk[2][8];
    auto load_stage = [&](int mt, int G) {   // unconditional, clamped (see fused_f32_kloop16)
        const int Gc = G < nG - 1 ? G : nG - 1;
        const int sf = 128 * Gc + mt * mt_step, sb = 128 * (nG - 1 - Gc) + mt * mt_step;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)vf, sf + 16 * h, 0);
            const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)vb, sb + 16 * h, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fr[mt][4 * h + t] = v[t];
                bk[mt][4 * h + t] = w[t];
            }
        }
    };
    float4* El = reinterpret_cast<float4*>(smem);   // [rows][16]
    const float4* erow = El + (8 * kq) * 16 + m16;
    auto b_at = [&](int Gl, int r) { return erow[(32 * Gl + r) * 16]; };
    // one stage: the 8 k rows of a double group for one 16-row tile, 4 MFMAs each, the B operand fetched one row ahead
    auto mfma_stage = [&](int mt, int Gl, int Gl_next, float4 bc) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float4 bn = r < 7 ? b_at(Gl, r + 1) : b_at(Gl_next, 0);
            __builtin_amdgcn_sched_barrier(0);
            const float sm = fr[mt][r] + bk[mt][7 - r];
            const float df = fr[mt][r] - bk[mt][7 - r];
            accR[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, bc.x, accR[mt][0], 0, 0, 0);
            if (!HALF) accR[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, bc.y, accR[mt][1], 0, 0, 0);
            accI[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, bc.z, accI[mt][0], 0, 0, 0);
            if (!HALF) accI[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, bc.w, accI[mt][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
        }
        return bc;
    };
    auto stage_e = [&](int kc, int rows) {
        constexpr int FULL = FR_KC / 4 / NWV;
        const float4* src = e_tile + (size_t)kc * 16 + wave * 64 + lane;
        float4* dst = El + wave * 64;
        if (rows == FR_KC && FR_KC / 4 % NWV == 0) {
#pragma unroll
            for (int q = 0; q < FULL; ++q) lds_dma16(src + q * (NWV * 64), dst + q * (NWV * 64));
        } else {
            for (int j = wave; j < rows / 4; j += NWV) lds_dma16(e_tile + (size_t)kc * 16 + j * 64 + lane, El + j * 64);
        }
    };
    stage_e(0, K2 < FR_KC ? K2 : FR_KC);
    for (int l = wave; l < tw_levels; l += NWV) lds_dma4(tw_src + (size_t)l * tw_stride, tw_dst + l * (2 * CB_C));
    load_stage(0, 0);
    load_stage(1, 0);
    __syncthreads();
    PVQ_STAMP(7);
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        if (kc > 0) {
            __syncthreads();
            stage_e(kc, rows);
            __syncthreads();
        }
        const int ng = rows / 32, G0 = kc / 32;
        float4 bc = b_at(0, 0);
        for (int Gl = 0; Gl < ng; ++Gl) {
            bc = mfma_stage(0, Gl, Gl, bc);
            load_stage(0, G0 + Gl + 1);
            bc = mfma_stage(1, Gl, Gl + 1, bc);
            load_stage(1, G0 + Gl + 1);
        }
    }
}

// the 16 columns NP of the accumulators -> LDS as [row][16 complex + pad], the 15 spare rows zeroed
template <int BM>
__device__ __forceinline__ void q_dump_p(float* smem, const f32x4a (&accR)[2][2], const f32x4a (&accI)[2][2], int np, int tid) {
    float2 (*Pt)[Q_LDP] = reinterpret_cast<float2 (*)[Q_LDP]>(smem);
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Pt[wave * 32 + mt * 16 + 4 * kq + r][m16] = np == 0 ? make_float2(accR[mt][0][r], accI[mt][0][r]) : make_float2(accR[mt][1][r], accI[mt][1][r]);
    for (int i = tid; i < 15 * Q_LDP; i += 2 * BM) Pt[BM][i] = make_float2(0.0f, 0.0f);
}
// the first R <= 4 tree levels in registers: a thread owns 8 consecutive rows of one of the quarter's 16 columns (+ the 2^R - 1 rows above)
template <int R, int BM>
__device__ __forceinline__ void q_tree_register_levels(float2 (*A)[Q_LDP], const float2 (*tw)[CB_C], int cq, int tid) {
    constexpr int H = (1 << R) - 1;
    constexpr int RPT = BM * Q_C / (2 * BM);   // 8 rows per thread
    const int c = tid & (Q_C - 1), j0 = (tid >> 4) * RPT;
    float2 v[RPT + H];
#pragma unroll
    for (int i = 0; i < RPT + H; ++i) v[i] = A[j0 + i][c];
    int len = RPT + H;
#pragma unroll
    for (int l = 0; l < R; ++l) {
        const int st = 1 << l;
        const float2 w = tw[l][cq + c];
        len -= st;
#pragma unroll
        for (int i = 0; i < RPT + H; ++i)
            if (i < len) v[i] = tree_cmadd(v[i], w, v[i + st]);
    }
    __syncthreads();   // every thread has read its halo
#pragma unroll
    for (int i = 0; i < RPT; ++i) A[j0 + i][c] = v[i];
    __syncthreads();
}
// one quarter from the P' tile in LDS to X: the tree (same operations, level by level, as fused_tree_levels), then the store
template <int BM>
__device__ __forceinline__ void q_tree_store(float* smem, const float2 (*tw)[CB_C], const FusedTile& t, const GemmTreeArgs& a, int cq, int tid, int stamp_slot) {
    float2 (*A)[Q_LDP] = reinterpret_cast<float2 (*)[Q_LDP]>(smem);
    constexpr int THREADS = 2 * BM;
    constexpr int PER = BM * Q_C / THREADS;  // 8
    const int c = tid & (Q_C - 1);
    const int levels = t.G.levels_f;
    int l = levels < 4 ? levels : 4;
    switch (l) {   // wave-uniform
        case 1: q_tree_register_levels<1, BM>(A, tw, cq, tid); break;
        case 2: q_tree_register_levels<2, BM>(A, tw, cq, tid); break;
        case 3: q_tree_register_levels<3, BM>(A, tw, cq, tid); break;
        case 4: q_tree_register_levels<4, BM>(A, tw, cq, tid); break;
        default: break;
    }
    PVQ_STAMP(6);
    int valid = BM - ((1 << l) - 1);
    for (; l + 1 < levels; l += 2) {
        const int st = 1 << l;
        const float2 w1 = tw[l][cq + c], w2 = tw[l + 1][cq + c];
        valid -= 3 * st;
        float2 v[PER];
#pragma unroll
        for (int g = 0; g < PER; g += 4) {
#pragma unroll
            for (int q = g; q < g + 4; ++q) {
                const int j = (tid + q * THREADS) / Q_C;
                if (j < valid) {
                    const float2 t0 = tree_cmadd(A[j][c], w1, A[j + st][c]);
                    const float2 t1 = tree_cmadd(A[j + 2 * st][c], w1, A[j + 3 * st][c]);
                    v[q] = tree_cmadd(t0, w2, t1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / Q_C;
            if (j < valid) A[j][c] = v[q];
        }
        __syncthreads();
    }
    for (; l < levels; ++l) {
        const int st = 1 << l;
        valid -= st;
        const float2 w = tw[l][cq + c];
        float2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / Q_C;
            if (j < valid) v[q] = tree_cmadd(A[j][c], w, A[j + st][c]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / Q_C;
            if (j < valid) A[j][c] = v[q];
        }
        __syncthreads();
    }
    PVQ_STAMP(2);
    // store: lanes walk the frames of one column (512-byte runs)
    const int j = tid % BM;
    const int f = t.f0 + j;
    if (j < t.S && f < t.nfr) {
        const bool to_y = t.G.nb > t.G.nb_f;
        float2* dst = (to_y ? a.Y : a.X) + ((size_t)((f >> 6) + (to_y ? t.yt0 : t.xt0)) * a.xcp + t.nt * CB_C + cq) * 64 + (f & 63);
        int ncv = t.G.n_cols - t.ntl * CB_C - cq;   // the quarter's real columns
        ncv = ncv < Q_C ? ncv : Q_C;
#pragma unroll 4
        for (int cc = tid / BM; cc < ncv; cc += 2) {
            const float2 val = A[j][cc];
            __builtin_nontemporal_store((f32x2){val.x, val.y}, reinterpret_cast<f32x2*>(&dst[cc * 64]));
        }
    }
}

template <int BM>
__global__ __launch_bounds__(2 * BM, 6) void blockdft_gemm_tree3(GemmTreeArgs a) {   // 6 waves per SIMD = three 512-thread workgroups per CU: at most 80 registers
    constexpr int B_FLOATS = FR_KC * FT_BN;
    constexpr int P_FLOATS = (BM + 15) * Q_LDP * 2;
    __shared__ __attribute__((aligned(16))) float smem[B_FLOATS > P_FLOATS ? B_FLOATS : P_FLOATS];  // the E slice, then a P quarter
    __shared__ float2 tw_lds[FT_MAXL][CB_C];
    const unsigned long long t_entry = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int4 entry = a.tile_list[blockIdx.x];   // .x: group | segment << 16 (no wide entries in this kernel's lists)
    const TileStream ts = tile_stream(a, entry.x);
    const FusedTile T = fused_tile_of<BM>(a, make_int4(entry.x & 7, entry.y, entry.z, entry.w), ts);
    if (T.f0 >= T.nfr) return;
    const int stamp_slot = blockIdx.x;
    PVQ_STAMP(0);
    const long long s = ts.base + T.G.s_rel;
    const long long tile_lo = s + (long long)T.f0 * a.K, tile_hi = tile_lo + (long long)BM * a.K;
    const bool inside = tile_lo >= 0 && tile_hi * 4ll <= (long long)ts.pcm_bytes;
    const float* tw_src = reinterpret_cast<const float*>(a.comb_tw + T.G.tw_off + T.ntl * CB_C) + lane;
    float* tw_dst = reinterpret_cast<float*>(&tw_lds[0][0]);
    const int tw_levels = T.G.levels_f, tw_stride = 2 * T.G.n_tiles * CB_C;
    const float4* e_tile = a.E16 + (size_t)T.nt * (a.K / 2) * 16;
    f32x4a accR[2][2], accI[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accR[mt][np][r] = 0.0f;
                accI[mt][np][r] = 0.0f;
            }
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 6) * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    const bool half = T.ntl == T.G.n_tiles - 1 && T.G.n_cols - T.ntl * CB_C <= 16;
    if (!inside)
        fused_f32_kloop16_edge<BM>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot, a.K);
    else if (half)
        fused_f32_kloop16s<BM, true>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot);
    else
        fused_f32_kloop16s<BM, false>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, tw_src, tw_dst, tw_levels, tw_stride, stamp_slot);
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 6) * 4 + 2] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    PVQ_STAMP(1);
    __syncthreads();   // the E slice is dead: the first P' quarter takes its place
    PVQ_STAMP(4);
    q_dump_p<BM>(smem, accR, accI, 0, tid);
    __syncthreads();
    PVQ_STAMP(5);
    q_tree_store<BM>(smem, tw_lds, T, a, 0, tid, stamp_slot);
    if (T.G.n_cols - T.ntl * CB_C > Q_C) {   // (a last tile of at most 16 columns has no second quarter)
        __syncthreads();
        q_dump_p<BM>(smem, accR, accI, 1, tid);
        __syncthreads();
        q_tree_store<BM>(smem, tw_lds, T, a, Q_C, tid, stamp_slot);
    }
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PVQ_STAMP(3);
    }
    PVQ_END_STAMPS
}

// ------------------------------------------------------------------------------------------------
// General hops: a multiple of 64 samples that does NOT divide the windows (1 600 samples = 30 analyses per second at 48 kHz, the
// cadence of pitchvis_serial/src/main.rs:41; the trainer's 3 x chunk, pitchvis_train/src/train.rs:43).  With W = nq hop + rem,
//
//     X_f[c] = sum_{q < nq} phi_c^q Q[f + q][c]  +  phi_c^nq R[f + nq][c],     phi_c = e^{-2 pi i c hop / W},
//     Q[j][c] = sum_{m < hop} x[s + j hop + m] e^{-2 pi i c m / W}   (the whole hop block, as in the power-of-two case),
//     R[j][c] = sum_{m < rem} x[s + j hop + m] e^{-2 pi i c m / W}   (the block's first rem samples),
//
// so every hop block is still transformed once (twice: whole and head) for all the frames that share it — the cost is per SAMPLE,
// not per frame — and only the combine changes: nq <= 16 terms by Horner's rule instead of a power-of-two tree.  Both GEMMs are the
// mirrored half-depth form about their block's centre (Q' = Q / rho_Q, R' = R / rho_R); rho_Q goes into the kernel-product
// coefficients as before and tau_c = phi_c^nq rho_R / rho_Q multiplies R'.  Two launches of this kernel: the R tiles first (256 rows
// = 256 frames, rows taken nq blocks further on, results to Y — or straight to X for a window shorter than the hop, nq = 0), then
// the Q tiles (257 - nq frames per tile), which add tau Y on their way out.  Same K loop, same P tile, same X layout as
// blockdft_gemm_tree: the kernel-product and peak stages do not know the difference.
// ------------------------------------------------------------------------------------------------
constexpr int GEN_MAX_NQ = 16;   // whole blocks per window the combine takes (the P tile's 15 spare rows are its halo)
template <int BM>
__device__ __forceinline__ void gen_horner(float* smem, const float2* phi, int nq, int tid) {
    float2 (*A)[FT_LDP] = reinterpret_cast<float2 (*)[FT_LDP]>(smem);  // [BM + 15][33]
    const int c = tid & (CB_C - 1), j0 = (tid >> 5) * 16;
    const float2 w = phi[c];
    float2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = make_float2(0.0f, 0.0f);
    // out[i] = v[i] + w (v[i + 1] + w (... + w v[i + nq - 1])): rows taken from the far end; row r feeds output i as term q = r - i
    for (int r = 15 + nq - 1; r >= 0; --r) {
        const float2 x = A[j0 + r][c];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = r - i;
            if (q >= 0 && q < nq) acc[i] = tree_cmadd(x, w, acc[i]);   // (uniform; the first term: x + w * 0 = x exactly)
        }
    }
    __syncthreads();   // every thread has read its halo
#pragma unroll
    for (int i = 0; i < 16; ++i) A[j0 + i][c] = acc[i];
    __syncthreads();
}

template <int BM>
__global__ __launch_bounds__(2 * BM, 4) void blockdft_gemm_gen(GemmTreeArgs a) {
    constexpr int B_FLOATS = FR_KC * FT_BN;
    constexpr int P_FLOATS = (BM + 15) * FT_LDP * 2;
    __shared__ __attribute__((aligned(16))) float smem[B_FLOATS > P_FLOATS ? B_FLOATS : P_FLOATS];  // the E slice, then the P tile
    __shared__ float2 gtw[2][CB_C];   // phi, tau of the tile's columns
    const int tid = threadIdx.x;
    const int4 entry = a.tile_list[blockIdx.x];   // .x: group | segment << 16
    const TileStream ts = tile_stream(a, entry.x);
    const BlockGroup G = a.gv[entry.x & 7];
    const int ntl = entry.y, f0 = entry.z, nt = G.tile0 + ntl;
    if (f0 >= ts.n_frames) return;
    const int stamp_slot = blockIdx.x;
    const bool is_r = a.gen_kind == 1;
    const int depth = is_r ? G.rem : a.K;
    // rows of the tile: hop blocks f0 .. f0 + BM - 1 of the group's block grid (R tiles: nq blocks further on)
    const long long tile_lo = ts.base + G.s_rel + (long long)(f0 + (is_r ? G.nq : 0)) * a.K;
    const long long tile_hi = tile_lo + (long long)(BM - 1) * a.K + depth;
    const bool inside = tile_lo >= 0 && tile_hi * 4ll <= (long long)ts.pcm_bytes;
    const float4* e_tile = is_r ? a.E16R + G.e16r_off + (size_t)ntl * (G.rem / 2) * 16 : a.E16 + (size_t)nt * (a.K / 2) * 16;
    if (tid < 2 * CB_C) gtw[tid >> 5][tid & (CB_C - 1)] = a.gen_tw[G.gtw_off + (tid >> 5) * (G.n_tiles * CB_C) + ntl * CB_C + (tid & (CB_C - 1))];
    f32x4a accR[2][2], accI[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accR[mt][np][r] = 0.0f;
                accI[mt][np][r] = 0.0f;
            }
    const bool half = ntl == G.n_tiles - 1 && G.n_cols - ntl * CB_C <= 16;
    float* no_tw = reinterpret_cast<float*>(&gtw[0][0]);   // (no tree twiddles to stage: tw_levels = 0)
    if (!inside)
        fused_f32_kloop16_edge<BM>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, no_tw, no_tw, 0, 0, stamp_slot, depth);
    else if (half)
        fused_f32_kloop16<BM, true>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, no_tw, no_tw, 0, 0, stamp_slot, depth);
    else
        fused_f32_kloop16<BM, false>(a, ts.pcm_base, ts.pcm_bytes, smem, tile_lo, e_tile, tid, accR, accI, no_tw, no_tw, 0, 0, stamp_slot, depth);
    __syncthreads();   // the E slice is dead: the P' tile takes its place
    fused_dump_p<BM, 2, 0>(smem, accR, accI, tid);
    __syncthreads();
    if (!is_r && G.nq > 1) gen_horner<BM>(smem, gtw[0], G.nq, tid);
    // store: lanes walk the frames of one column (512-byte runs); Q tiles add tau * R'[f + nq] (the R launch left it in Y)
    float2 (*A)[FT_LDP] = reinterpret_cast<float2 (*)[FT_LDP]>(smem);
    const int j = tid % BM;
    const int f = f0 + j;
    const int S = is_r ? BM : BM - (G.nq > 1 ? G.nq - 1 : 0);
    if (j < S && f < ts.n_frames) {
        const size_t at = ((size_t)((f >> 6) + ts.xt0) * a.xcp + nt * CB_C) * 64 + (f & 63);
        float2* dst = (is_r && G.nq > 0 ? a.Y : a.X) + at;
        const float2* ysrc = a.Y + at;
        const bool add_y = !is_r && G.rem > 0;
        const int ncv = G.n_cols - ntl * CB_C < CB_C ? G.n_cols - ntl * CB_C : CB_C;
#pragma unroll 4
        for (int cc = tid / BM; cc < ncv; cc += 2) {
            float2 val = A[j][cc];
            if (add_y) val = tree_cmadd(val, gtw[1][cc], ysrc[cc * 64]);
            if (is_r && G.nq > 0)
                dst[cc * 64] = val;   // read back by the Q launch: through the L2
            else
                __builtin_nontemporal_store((f32x2){val.x, val.y}, reinterpret_cast<f32x2*>(&dst[cc * 64]));
        }
    }
}

// Unfused form of the same GEMM (windows of more than 64 hop blocks: the tree runs as its own kernel over P' in memory):
// BM rows x 32 complex columns per workgroup, the K loop of the fused kernel, P' written tile-major as (re, im) pairs.
template <int BM>
__global__ __launch_bounds__(2 * BM, 2) void blockdft_gemm_rows(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[FR_KC * FT_BN];   // the E slice
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups are dealt round robin over the 8 XCDs (b and b + 8 share an L2), so an XCD owns
    // whole row panels: all column tiles that re-read one panel of the stream hit one L2
    const int b = blockIdx.x;
    const int xcd = b & 7, bi = b >> 3;
    const int nt = bi % a.n_col_tiles;
    const int mt = (bi / a.n_col_tiles) * 8 + xcd;
    if (mt * BM >= a.n_rows) return;
    const int j0 = mt * BM;
    const long long s = a.base + a.tile_s[nt];
    const long long tile_lo = s + (long long)j0 * a.K, tile_hi = tile_lo + (long long)BM * a.K;
    const long long row0 = tile_lo + (long long)(wave * 32 + (lane & 31)) * a.K;
    const int half = lane >> 5;
    const long long off_f0 = row0 + 16 * half;
    const long long off_b0 = row0 + a.K - 16 - 16 * half;
    const float* e_tile = a.E + (size_t)nt * FT_BN;
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        acc0[q] = 0.0f;
        acc1[q] = 0.0f;
    }
    if (tile_lo >= 0 && tile_hi * 4ll <= (long long)a.pcm_bytes)
        fused_f32_kloop<true, BM>(a, smem, off_f0, off_b0, e_tile, tid, acc0, acc1);
    else
        fused_f32_kloop<false, BM>(a, smem, off_f0, off_b0, e_tile, tid, acc0, acc1);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5); a row of the tile is 256 contiguous bytes
    float2* Pt = reinterpret_cast<float2*>(a.P) + (size_t)nt * a.p_rows * CB_C;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = j0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < a.n_rows) Pt[(size_t)row * CB_C + (lane & 31)] = make_float2(acc0[q], acc1[q]);
    }
}

// Split-bf16 form of the fused kernel ("bf16x3", the default; pvq_vqt_set_gemm_precision): same tile, same
// epilogue.  Each fp32 operand is written exactly as hi + mid + lo with three bf16 values (8+8+8 mantissa bits)
// and the product is accumulated in fp32 from the six partial products whose weight is >= 2^-16 (hh, hm, mh, hl,
// lh, mm); the dropped terms are below 2^-24 of the product, i.e. at fp32 rounding level, and every partial
// product of two bf16 numbers is exact in fp32.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA,
// so six of them replace eight fp32 MFMAs at 6/16 of the matrix-pipe time.  E is split once on the host (planes
// stored [plane][n][k], k contiguous = the B-operand fragment order); the PCM tile is split while it is staged
// (16 consecutive samples per thread: 16-byte loads, 16-byte LDS writes).  LDS: 3 planes x (128 + 64) rows x 32
// bf16 = 36 KB of staging, swizzled instead of padded (4 workgroups per CU), aliased by the P tile.  Measured
// accuracy equals the fp32 MFMA form (6e-7 of the frame peak); the K loop runs at ~35 % of the bf16 matrix peak,
// bounded by the LDS staging and barrier structure of a 128 x 64 tile, not by the matrix pipe.
constexpr int FB_BK = 32;
template <int BM> struct FbGeom {
    static constexpr int PLANE = (BM + FT_BN) * FB_BK;            // bf16 elements of one plane: BM PCM rows, then 64 E^T rows
    static constexpr int STAGE_BYTES = 3 * PLANE * 2;             // 36 864 B at BM = 128
    static constexpr int P_BYTES = (BM + 15) * FT_LDP * 8;        // the P tile that aliases the staging area (+ 15 spare rows: the tree's halo reads need no range check)
    static constexpr int LDS_BYTES = STAGE_BYTES > P_BYTES ? STAGE_BYTES : P_BYTES;
};
// element offset of the 8-sample chunk `ch` (0..3) of row `row` inside a plane.  Rows are 64 bytes, unpadded; the
// chunk index is XORed with (row / 4) % 4, which makes the b128 fragment reads (16 consecutive rows, one chunk),
// the PCM staging writes (8 rows x 2 chunks) and the E^T staging writes (4 rows x 4 chunks) bank-conflict free.
__device__ __forceinline__ int fb_off(int row, int ch) { return row * FB_BK + ((ch ^ ((row >> 2) & 3)) << 3); }

// K loop of the split-bf16 fused kernel.  VEC: the tile's samples all lie inside the stream, so each
// thread's 16 consecutive samples come as four 16-byte loads; otherwise (tiles that touch the stream start
// or end) as 16 dword loads, each range-checked by the buffer hardware.
template <bool VEC, int BM>
__device__ __forceinline__ void fused_bf16x3_kloop(const GemmTreeArgs& a, const float* pcm_base, unsigned pcm_bytes, unsigned char* smem_raw, long long a_idx0, const __bf16* e_ptr,
                                                   int tid, f32x16& acc0, f32x16& acc1) {
    constexpr int FB_PLANE = FbGeom<BM>::PLANE;
    __bf16* lds = reinterpret_cast<__bf16*>(smem_raw);
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)pcm_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pcm_base), 0, pcm_bytes, 0x00020000);
    // A staging: thread -> (row = tid / 2, 16 consecutive k); B staging: thread -> (n = tid / 4, 8 consecutive k) x 3 planes
    const int a_row = tid >> 1, b_n = tid >> 2;
    const size_t plane = (size_t)a.ld * a.K;
    float ra[16];
    bf16x8 rb[3];
    auto load = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)((unsigned)(a_idx0 * 4ll) + (unsigned)k0 * 4u + 16u * q), 0, 0);
                ra[4 * q + 0] = v[0];
                ra[4 * q + 1] = v[1];
                ra[4 * q + 2] = v[2];
                ra[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {   // samples before the stream: an explicit out-of-range offset (see fused_f32_stage_load)
                const long long j = a_idx0 + k0 + q;
                ra[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, j >= 0 ? (unsigned)(j * 4ll) : 0xFFFFFFFCu, 0, 0));
            }
        }
        if (BM == 128 || tid < 256) {
#pragma unroll
            for (int p = 0; p < 3; ++p) rb[p] = *reinterpret_cast<const bf16x8*>(e_ptr + p * plane + k0);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 vh, vm, vl;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = ra[8 * h + q];
                const __bf16 hi = (__bf16)x;
                const float r1 = x - (float)hi;
                const __bf16 mid = (__bf16)r1;
                vh[q] = hi;
                vm[q] = mid;
                vl[q] = (__bf16)(r1 - (float)mid);
            }
            const int o = fb_off(a_row, (tid & 1) * 2 + h);
            *reinterpret_cast<bf16x8*>(lds + o) = vh;
            *reinterpret_cast<bf16x8*>(lds + FB_PLANE + o) = vm;
            *reinterpret_cast<bf16x8*>(lds + 2 * FB_PLANE + o) = vl;
        }
        if (BM == 128 || tid < 256) {   // 64 E^T rows x 4 chunks: 256 threads
            const int ob = fb_off(BM + b_n, tid & 3);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(lds + p * FB_PLANE + ob) = rb[p];
        }
    };
    const int n_iter = a.K / FB_BK;
    const int ar = wm * 64 + (lane & 31), kh = lane >> 5, bc = BM + wn * 32 + (lane & 31);
    load(0);
    for (int it = 0; it < n_iter; ++it) {
        store();
        __syncthreads();
        if (it + 1 < n_iter) load((it + 1) * FB_BK);
#pragma unroll
        for (int kk = 0; kk < FB_BK / 16; ++kk) {
            bf16x8 a0[3], a1[3], bv[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a0[p] = *reinterpret_cast<const bf16x8*>(lds + p * FB_PLANE + fb_off(ar, kk * 2 + kh));
                a1[p] = *reinterpret_cast<const bf16x8*>(lds + p * FB_PLANE + fb_off(ar + 32, kk * 2 + kh));
                bv[p] = *reinterpret_cast<const bf16x8*>(lds + p * FB_PLANE + fb_off(bc, kk * 2 + kh));
            }
            // smallest terms first
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], bv[1], acc0, 0, 0, 0);  // mid*mid
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bv[2], acc0, 0, 0, 0);  // hi*lo
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bv[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], bv[0], acc0, 0, 0, 0);  // lo*hi
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], bv[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bv[1], acc0, 0, 0, 0);  // hi*mid
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bv[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], bv[0], acc0, 0, 0, 0);  // mid*hi
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], bv[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bv[0], acc0, 0, 0, 0);  // hi*hi
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], bv[0], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
}

template <int BM>   // rows of hop blocks per tile; 2 * BM threads (wave tile 64 x 32)
__global__ __launch_bounds__(2 * BM, BM == 128 ? 4 : 2) void blockdft_gemm_tree_bf16x3(GemmTreeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[FbGeom<BM>::LDS_BYTES];
    __shared__ float2 tw_lds[FT_MAXL][CB_C];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileStream ts;
    const FusedTile T = fused_tile<BM>(a, ts);
    const int f0 = T.f0, nt = T.nt;
    if (f0 >= T.nfr) return;
    if (tid < 256) fused_stage_twiddles(tw_lds, T, a, tid);
    const int wm = wave >> 1, wn = wave & 1;
    const long long s = ts.base + T.G.s_rel;
    const long long tile_lo = s + (long long)f0 * a.K, tile_hi = tile_lo + (long long)BM * a.K;  // sample range of the tile
    const long long a_off0 = tile_lo + (long long)(tid >> 1) * a.K + (tid & 1) * 16;   // sample index of the thread's 16-sample run
    const __bf16* e_ptr = a.Et + (size_t)(nt * FT_BN + ((tid & 255) >> 2)) * a.K + (tid & 3) * 8;
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        acc0[q] = 0.0f;
        acc1[q] = 0.0f;
    }
    if (tile_lo >= 0 && tile_hi * 4ll <= (long long)ts.pcm_bytes)
        fused_bf16x3_kloop<true, BM>(a, ts.pcm_base, ts.pcm_bytes, smem_raw, a_off0, e_ptr, tid, acc0, acc1);
    else
        fused_bf16x3_kloop<false, BM>(a, ts.pcm_base, ts.pcm_bytes, smem_raw, a_off0, e_ptr, tid, acc0, acc1);
    const int bc = wn * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = wm * 64 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        smem[row * (2 * FT_LDP) + bc] = acc0[q];
        smem[(row + 32) * (2 * FT_LDP) + bc] = acc1[q];
    }
    if (tid < 15 * FT_LDP) reinterpret_cast<float2*>(smem)[BM * FT_LDP + tid] = make_float2(0.0f, 0.0f);   // the spare rows
    __syncthreads();
    fused_tree_store<BM>(smem, tw_lds, T, a, tid, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Last tree levels for windows of more than 64 hop blocks: the fused kernel leaves Y_f = sum_{b<64} phi^b P'[f+b];
// X'_f = sum_q phi^{64 q} Y_{f+64q} (2 or 4 terms) with the same level-by-level multiply-adds as the rest of the tree.
// Y_{f+64q} is the same lane of the same column q frame tiles further on: 512-byte runs in, 512-byte runs out.
// ------------------------------------------------------------------------------------------------
struct FinishArgs {
    const float2* Y;
    float2* X;
    int xcp;
    int n_frames;
    int col0;          // first X column of the group
    int n_cols;        // columns of the group (padded to its tiles): the twiddle table's row stride
    int n_real;        // columns of the group that exist (the padding is neither written by the fused kernel nor combined here)
    int levels_f, levels;
    const float2* tw;  // the group's combine twiddles: [levels][n_cols]
    const XTile* xmap; // many-streams launches: per X tile, the Y tile of the same frames and the number of frames that exist
};
__global__ __launch_bounds__(256) void blockdft_tree_finish(FinishArgs a) {
    const int fr = threadIdx.x & 63, cc = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int tile = blockIdx.x;
    int ytile = tile, n_live = a.n_frames - tile * 64;
    if (a.xmap) {
        ytile = a.xmap[tile].y_tile;
        n_live = a.xmap[tile].live_step & 255;
    }
    if (cc >= a.n_real || fr >= n_live) return;
    const int col = a.col0 + cc;
    const int nq = 1 << (a.levels - a.levels_f);   // 2 or 4
    float2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q < nq) v[q] = a.Y[((size_t)(ytile + q) * a.xcp + col) * 64 + fr];
    const float2 w0 = a.tw[(size_t)a.levels_f * a.n_cols + cc];
    if (nq == 2) {
        v[0] = tree_cmadd(v[0], w0, v[1]);
    } else {
        const float2 w1 = a.tw[(size_t)(a.levels_f + 1) * a.n_cols + cc];
        v[0] = tree_cmadd(tree_cmadd(v[0], w0, v[1]), w1, tree_cmadd(v[2], w0, v[3]));
    }
    a.X[((size_t)tile * a.xcp + col) * 64 + fr] = v[0];
}

// ------------------------------------------------------------------------------------------------
// combine: X_f[c] = sum_b phi_c^b P[f+b][c] by a doubling tree in LDS
// ------------------------------------------------------------------------------------------------
struct CombineArgs {
    const float* P;
    int p_rows;        // row capacity of the tile-major P
    float2* X;         // frame-tile blocked: X[((frame / 64) * xcp + col) * 64 + frame % 64]
    int xcp;
    int n_frames;      // frames in this chunk
    int n_rows;        // rows of P present
    const int* tile_group;
    const BlockGroup* groups;
    const float2* comb_tw;
};

// CT frames x CW complex columns per workgroup; windows of up to MAXNB hop blocks.
template <int CT, int CW, int MAXNB>
__global__ __launch_bounds__(256) void blockdft_combine(CombineArgs a) {
    constexpr int MAXR = CT + MAXNB - 1;
    __shared__ float2 A[MAXR][CW + 1];   // +1: the transposed store below reads a column per wave
    const int tid = threadIdx.x;
    const int col0 = blockIdx.x * CW;          // first complex column of this tile (global X column)
    const int f0 = blockIdx.y * CT;
    const BlockGroup G = a.groups[a.tile_group[col0 / CB_C]];
    const int R = CT + G.nb - 1;
    const int c = tid & (CW - 1);
    // stage rows f0 .. f0+R-1 of this column tile
    for (int idx = tid; idx < R * CW; idx += 256) {
        const int j = idx / CW;
        const int row = f0 + j;
        float2 v = make_float2(0.0f, 0.0f);
        if (row < a.n_rows)
            v = *reinterpret_cast<const float2*>(a.P + ((size_t)(col0 / CB_C) * a.p_rows + row) * 64 + 2 * ((col0 % CB_C) + c));
        A[j][c] = v;
    }
    __syncthreads();
    constexpr int PER = (MAXR * CW + 255) / 256;
    int valid = R;
    for (int l = 0; l < G.levels; ++l) {
        const int s = 1 << l;
        valid -= s;  // rows with a complete span after this level
        const float2 w = a.comb_tw[G.tw_off + l * (G.n_tiles * CB_C) + (col0 - G.tile0 * CB_C) + c];
        float2 v[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = tid + t * 256;
            const int j = idx / CW;
            if (j < valid) {
                const float2 lo = A[j][c], hi = A[j + s][c];
                v[t] = make_float2(lo.x + (w.x * hi.x - w.y * hi.y), lo.y + (w.x * hi.y + w.y * hi.x));
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int idx = tid + t * 256;
            const int j = idx / CW;
            if (j < valid) A[j][c] = v[t];
        }
        __syncthreads();
    }
    {
        const int j = tid & (CT - 1);
        const int f = f0 + j;
        if (f < a.n_frames)
            for (int cc = tid / CT; cc < CW; cc += 256 / CT) a.X[((size_t)(f >> 6) * a.xcp + col0 + cc) * 64 + (f & 63)] = A[j][cc];
    }
}

// ------------------------------------------------------------------------------------------------
// kernel product + dB:  x_vqt[k] = sum_c K[k][c] X[c] (+ conjugate part), then power_to_db (vqt.rs:889-954)
//
// The rows of a window group's spectral kernel are band-limited wavelets: consecutive bins read
// overlapping, nearly contiguous column ranges (27 % of a 16-bin x 70-column block is non-zero).  A block
// of 16 bins is therefore a small dense real GEMM per frame tile,
//     [32 frames x 2 kb] . [2 kb x 32]   (columns: Xr, Xi of kb spectrum columns; outputs: 16 x re, 16 x im),
// one v_mfma_f32_32x32x2_f32 per spectrum column, with both operands read straight from memory in lane
// order (X is column-major, so the 32 frames of a column are 256 contiguous bytes; the coefficient table
// is stored as B operands and stays in L2).  The conjugate part (negative_filter_bank) lands in the same
// B matrix with the signs of the Xi row flipped.  A workgroup owns FT frames and all bins: its four waves
// walk disjoint lists of blocks, drop the dB values into LDS, and the frame-wide max / floor / shift of
// power_to_db is applied from there.
// ------------------------------------------------------------------------------------------------
struct BandArgs {
    const float* X;            // complex spectrum columns, as floats: [frame / 64][column][frame % 64][re, im]
    int xcp;                   // columns per 64-frame tile of X (incl. pad)
    int n_frames;
    int n_bins;
    int ldb;                   // LDS row stride of the dB tile
    const BandBlock* blocks;
    const float* B;
    const __bf16* B3;          // split-bf16 coefficient planes
    const int* list;           // [waves][per_wave]: count, then the blocks of that wave
    int per_wave;
    float* out_db;             // [n_frames][n_bins]
    float2* out_cplx;          // optional
    unsigned* status;          // the handle's sticky flag word: bit 0 <- a live frame holds a non-finite power value
    const XTile* xmap;         // many-streams launches: per X tile, its output rows and how many of its frames exist (nullptr: tile t holds rows 64 t ...)
    unsigned long long* stamps;   // developer knob PVQ_STAMPS_DOTS: [workgroup][8] 100 MHz clock: 0 start, 1 wave 0 done with its blocks, 2 all waves done, 3 end
};

#define PVQ_REF_POWER (0.3f * 0.3f)
#define PVQ_A_MIN (1e-6f * 1e-6f)
#define PVQ_TOP_DB 60.0f

// gfx950 lane-row swaps.  Inline asm: this compiler's two-result builtins (__builtin_amdgcn_permlane16_swap /
// permlane32_swap) were seen to hand the same register to both results once inlined into a larger kernel.
// x' = [x.rows 0, y.rows 0, x.rows 2, y.rows 2],  y' = [x.rows 1, y.rows 1, x.rows 3, y.rows 3]   (rows of 16 lanes)
__device__ __forceinline__ void permlane16_swap(float& x, float& y) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
// x' = [x.lo, y.lo],  y' = [x.hi, y.hi]   (halves of 32 lanes)
__device__ __forceinline__ void permlane32_swap(float& x, float& y) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

// wave-wide max / min by DPP (within rows of 16 lanes) and the gfx950 row / half swaps (across rows)
#define PVQ_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, PVQ_DPP(v, 0xB1));    // quad_perm(1,0,3,2)
    v = fmaxf(v, PVQ_DPP(v, 0x4E));    // quad_perm(2,3,0,1)
    v = fmaxf(v, PVQ_DPP(v, 0x141));   // row_half_mirror
    v = fmaxf(v, PVQ_DPP(v, 0x140));   // row_mirror
    float x = v, y = v;
    permlane16_swap(x, y);
    v = fmaxf(x, y);
    x = v;
    y = v;
    permlane32_swap(x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

// results of one block (C layout: column n = lane & 31: bin row = n & 15, re / im = n >> 4; frame =
// (q&3) + 8(q>>2) + 4(lane>>5)) -> |x_vqt|^2 into the LDS tile (+ the optional complex output).
// v_permlane16_swap brings the im column's value into the re column's lane.
// LDB: compile-time row stride of the tile (0: a.ldb) — with it every LDS address below is one base plus an immediate
// offset; the optional complex output recomputes its addresses per block (the row stride is laundered through an
// asm so that 32 loop-invariant 64-bit addresses are not kept live across the whole block loop).
constexpr int BAND_LDB2 = 260;   // the 64-frame form: up to 256 bins + 4 (rows 4 apart land 16 banks apart)
constexpr int BAND_LDB3 = 308;   // the 64-frame 8-bin form up to 304 bins (78.8 KB: still two workgroups per CU)
template <int MT, int LDB>
__device__ __forceinline__ void band_writeout(const f32x16 (&acc)[MT], float* dbs, const BandArgs& a, long long row0, int n_live, int rstep, int bin0, int nrows,
                                              int lane) {
    const int ldb = LDB ? LDB : a.ldb;
    const int n = lane & 31, kx = lane >> 5;
    const int row = n & 15, part = n >> 4;
    const bool mine = part == 0 && row < nrows;
    const int bin = bin0 + row;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float im[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float re = acc[mt][q], o = 0.0f;
            permlane16_swap(re, o);   // o: rows 0 / 2 now hold the im columns' values
            im[q] = o;
        }
        if (mine) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int fr = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * kx;
                dbs[fr * ldb + bin] = acc[mt][q] * acc[mt][q] + im[q] * im[q];
            }
            if (a.out_cplx) {
                int row_stride = a.n_bins;
                asm volatile("" : "+s"(row_stride));
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int fr = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * kx;
                    if (fr < n_live) a.out_cplx[(size_t)(row0 + (long long)fr * rstep) * row_stride + bin] = make_float2(acc[mt][q], im[q]);
                }
            }
        }
    }
}

// power_to_db per frame (vqt.rs:922-954): a wave per frame, lanes over bins.  Up to 512 bins: four (two) frames at a
// time with their dB values in registers, so the read -> log -> reduce -> rescale chains of the frames overlap
// (the phase is latency-bound at two waves per SIMD); more bins: one frame at a time through LDS.
template <int MT, int NW, int LDB = (MT == 2 ? BAND_LDB2 : 0)>
__device__ __forceinline__ void band_finish(float* dbs, const BandArgs& a, long long row0, int n_live, int rstep, int wave, int lane) {
    const int ldb = LDB ? LDB : a.ldb;
    const float ref_db = 10.0f * log10f(PVQ_REF_POWER);
    // 10 log10(p) = 10 log10(2) * log2(p) on the hardware log2 (1 ulp): within 2e-5 dB of the libm route
    auto to_db = [&](float p) { return 3.01029995663981f * __log2f(fmaxf(p, PVQ_A_MIN)) - ref_db; };
    // non-finite input (a NaN / Inf sample inside one of the frame's windows) reaches every bin of the frame as a NaN or Inf
    // power; fmaxf would silently turn it into the A_MIN floor, so it is flagged instead (Vqt::input_status)
    bool bad = false;
    auto flag = [&]() {
        if (a.status && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.status, 1u);
    };
    auto in_registers = [&](auto fu_c, auto nkb_c) {
        constexpr int FU = decltype(fu_c)::value, NKB = decltype(nkb_c)::value;
        for (int fr0 = wave; fr0 < MT * 32; fr0 += NW * FU) {
            float d[FU][NKB], mx[FU], mn[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                mx[u] = -3.40282347e+38f;
                mn[u] = 3.40282347e+38f;
                const bool live = fr0 + NW * u < n_live;
#pragma unroll
                for (int kk = 0; kk < NKB; ++kk) {
                    const int k = lane + 64 * kk;
                    const bool in = k < a.n_bins;
                    const float p = in ? dbs[(fr0 + NW * u) * ldb + k] : 1.0f;
                    bad |= live && !(p <= 3.40282347e+38f);
                    d[u][kk] = to_db(p);
                    mx[u] = fmaxf(mx[u], in ? d[u][kk] : -3.40282347e+38f);
                    mn[u] = fminf(mn[u], in ? d[u][kk] : 3.40282347e+38f);
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                mx[u] = wave_max(mx[u]);
                mn[u] = wave_min(mn[u]);
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int fr = fr0 + NW * u;
                if (fr >= n_live) continue;
                const float floor_db = mx[u] - PVQ_TOP_DB;
                const float m2 = fmaxf(mn[u], floor_db);
                float* dst = a.out_db + (size_t)(row0 + (long long)fr * rstep) * a.n_bins;
#pragma unroll
                for (int kk = 0; kk < NKB; ++kk) {
                    const int k = lane + 64 * kk;
                    const float c = fmaxf(d[u][kk], floor_db);
                    if (k < a.n_bins) dst[k] = (m2 > 0.0f) ? (c - m2) : fmaxf(c, 0.0f);
                }
            }
        }
    };
    if (a.n_bins <= 256) {
        in_registers(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
        flag();
        return;
    }
    if (a.n_bins <= 512) {
        in_registers(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
        flag();
        return;
    }
    for (int fr = wave; fr < MT * 32; fr += NW) {
        if (fr >= n_live) break;
        float* rowp = dbs + fr * ldb;
        float mx = -3.40282347e+38f, mn = 3.40282347e+38f;
        for (int k = lane; k < a.n_bins; k += 64) {
            bad |= !(rowp[k] <= 3.40282347e+38f);
            const float d = to_db(rowp[k]);
            rowp[k] = d;
            mx = fmaxf(mx, d);
            mn = fminf(mn, d);
        }
        mx = wave_max(mx);
        mn = wave_min(mn);
        const float floor_db = mx - PVQ_TOP_DB;
        const float m2 = fmaxf(mn, floor_db);
        float* dst = a.out_db + (size_t)(row0 + (long long)fr * rstep) * a.n_bins;
        for (int k = lane; k < a.n_bins; k += 64) {
            const float c = fmaxf(rowp[k], floor_db);
            dst[k] = (m2 > 0.0f) ? (c - m2) : fmaxf(c, 0.0f);
        }
    }
    flag();
}

template <int MT, int NW>   // 32-frame MFMA row tiles per workgroup, waves per workgroup
__global__ __launch_bounds__(64 * NW, NW / 2) void blockdft_banddots_db(BandArgs a) {
    const int stamp_slot = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) float dbs[];   // [MT * 32][ldb]: |x_vqt|^2, then dB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * (MT * 32);
    const int n = lane & 31, kx = lane >> 5;
    constexpr int col_stride = 128;   // floats between consecutive X columns of a 64-frame tile
    const float* xtile = a.X + ((size_t)(f0 >> 6) * a.xcp) * col_stride + (f0 & 63) * 2;   // this workgroup's frames
    // the output rows of this workgroup's frames: rows f0 ... of a single stream, or what the X tile's entry of the map says
    long long row0 = f0;
    int n_live = a.n_frames - f0, rstep = 1;
    if (a.xmap) {   // (uniform)
        const XTile xt = a.xmap[f0 >> 6];
        rstep = xt.live_step >> 8;
        row0 = xt.out_row0 + (long long)(f0 & 63) * rstep;
        n_live = (xt.live_step & 255) - (f0 & 63);
        if (n_live <= 0) return;   // (uniform) a staged buffer's gap frames: nothing of this tile is wanted (Vqt::batch_streams_device)
    }
    PVQ_STAMP(0);
    const int* my_list = a.list + wave * a.per_wave;
    const int n_blocks = __builtin_amdgcn_readfirstlane(my_list[0]);
    // MT == 2: a lane loads (Re, Im) of one of 64 frames; a half swap then leaves Re of frames 0..31 / Im of frames
    // 0..31 in the two lane halves of one register (the A operand of row tile 0) and frames 32..63 in the other.
    // MT == 1: a lane loads the one float it feeds to the MFMA.
    const float* xa = nullptr;
    const float2* bp = nullptr;   // column pairs
    float2 av[BD_NS][BD_KU], bv[BD_NS][BD_KU / 2];
    auto fetch = [&](int s, int c) {
#pragma unroll
        for (int u = 0; u < BD_KU / 2; ++u) bv[s][u] = bp[(size_t)(c / 2 + u) * 64];
#pragma unroll
        for (int u = 0; u < BD_KU; ++u) {
            if (MT == 2)
                av[s][u] = *reinterpret_cast<const float2*>(xa + (size_t)(c + u) * col_stride);
            else
                av[s][u].x = xa[(size_t)(c + u) * col_stride];
        }
    };
    // point the operand streams at a block and put its first BD_NS - 1 stages in flight
    auto open_block = [&](const BandBlock& blk) {
        xa = MT == 2 ? xtile + (size_t)blk.x0 * col_stride + lane * 2 : xtile + (size_t)blk.x0 * col_stride + n * 2 + kx;
        bp = reinterpret_cast<const float2*>(a.B) + (size_t)blk.boff * 32 + lane;
#pragma unroll
        for (int s = 0; s < BD_NS - 1; ++s) fetch(s, s * BD_KU);
    };
    BandBlock blk{};
    if (n_blocks > 0) {
        blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[1])];
        open_block(blk);
    }
    for (int bi = 0; bi < n_blocks; ++bi) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
        auto mul = [&](int s) {
#pragma unroll
            for (int u = 0; u < BD_KU; ++u) {
                const float b = (u & 1) ? bv[s][u / 2].y : bv[s][u / 2].x;
                if (MT == 2) {
                    float t0 = av[s][u].x, t1 = av[s][u].y;
                    permlane32_swap(t0, t1);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t0, b, acc[0], 0, 0, 0);
                    acc[MT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(t1, b, acc[MT - 1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][u].x, b, acc[0], 0, 0, 0);
                }
            }
        };
        // ring of BD_NS stages of BD_KU columns: BD_NS - 1 stages of operands in flight while one is multiplied.
        // kb is a multiple of BD_KU; the fetches run up to (BD_NS - 1) * BD_KU columns past the block (in bounds
        // by construction, never multiplied).
        const int kb = __builtin_amdgcn_readfirstlane(blk.kb);
        const int kb_full = kb - kb % (BD_NS * BD_KU);
        int c = 0;
        for (; c < kb_full; c += BD_NS * BD_KU) {   // steady state: no branches, exact load counting
#pragma unroll
            for (int s = 0; s < BD_NS; ++s) {
                fetch((s + BD_NS - 1) % BD_NS, c + (s + BD_NS - 1) * BD_KU);
                mul(s);
                // keep a stage's lane swaps (and the waits on its operands) inside the stage: the scheduler would
                // otherwise hoist the swaps of later stages to the top and wait for the whole ring
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < BD_NS - 1; ++s)         // remainder: the operands are already in flight
            if (c + s * BD_KU < kb) mul(s);
        // the next block's first operands fly while this block's results are written out
        const int bin0 = blk.bin0, nrows = blk.nrows;
        if (bi + 1 < n_blocks) {
            blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[bi + 2])];
            open_block(blk);
        }
        band_writeout<MT, MT == 2 ? BAND_LDB2 : 0>(acc, dbs, a, row0, n_live, rstep, bin0, nrows, lane);
    }
    PVQ_STAMP(1);
    __syncthreads();
    PVQ_STAMP(2);
    band_finish<MT, NW>(dbs, a, row0, n_live, rstep, wave, lane);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PVQ_STAMP(3);
    }
}

// 16x16x4 form of the fp32 kernel product (64-frame tiles, up to 304 bins: the default): blocks of 8 bins, so a block walks the union
// of only 8 rows' columns (about 35 instead of 57) — the same products in 39 % fewer matrix-pipe cycles than the 32x32x2 form above.
// Its first version (round 2) fetched 8 bytes per lane — one complex value of one column — and turned four such registers into the
// MFMA operands of the four 16-frame tiles with two v_permlane32_swap + two v_permlane16_swap per column pair: 133-150 us per
// 65 536 frames, bound by its vector-memory instructions.  This form needs NO lane swaps and issues half the loads (16 bytes each):
// 113-125 us on the same boxes.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// Lane (i = lane & 15, kq = lane >> 4) loads 16 bytes of column c + kq: (Re, Im) of the frame PAIR 16 u + i of the tile (frames 32 u + 2 i and + 1), and each
// of its four registers IS an A operand of v_mfma_f32_16x16x4_f32 as it stands: the k slots of an MFMA are the four columns
// c .. c + 3 (Re parts for register 0 / 2, Im parts for 1 / 3), its rows the 16 even (registers 0, 1) or odd (2, 3) frames of the
// half tile u; the B operand of lane (n, kq) is the coefficient of column c + kq for output n (n < 8: re of bin row n, else im),
// one register for the Re parts and one for the Im parts.  Per four columns: two 16-byte X loads and one 8-byte B load per lane
// instead of four 8-byte loads + one, eight MFMAs as before, no swaps.  C layout: output n = lane & 15, frame 32 u + 2 (4 (lane >> 4) + r) + p.
template <int NW, int NS, int LDB, int NU>   // LDB: row stride of the LDS tile (4 mod 16, >= bins); NU: half tiles of 32 frames per workgroup (2: a whole X tile; 1 — half a tile, 4 waves, four workgroups per CU — was measured slower: 141-150 against 121 us)
__global__ __launch_bounds__(64 * NW, NU == 2 || NW == 8 ? NW / 2 : NW) void blockdft_banddots4c_db(BandArgs a) {   // (four waves per SIMD)
    // the bin counts an LDS row stride serves (the host's choice of the instantiation): the finish's other size classes fold away
    if constexpr (LDB == 260) __builtin_assume(a.n_bins <= 256);
    else if constexpr (LDB == 308) __builtin_assume(a.n_bins > 256 && a.n_bins <= 304);
    else if constexpr (LDB == 372) __builtin_assume(a.n_bins > 256 && a.n_bins <= 368);   // (257 ... 304 bins come here with the developer knob PVQ_DOTS_F32 behind the split-bf16 GEMM)
    else if constexpr (LDB == 596) __builtin_assume(a.n_bins > 368 && a.n_bins <= 592);
    else if constexpr (LDB == 852) __builtin_assume(a.n_bins > 592 && a.n_bins <= 848);
    else if constexpr (LDB == 1028) __builtin_assume(a.n_bins > 848 && a.n_bins <= 1024);
    const int stamp_slot = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) float dbs[];   // [64][LDB]: |x_vqt|^2, then dB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * (32 * NU);
    constexpr int col_stride = 128;   // floats between consecutive X columns of a 64-frame tile
    const float* xtile = a.X + ((size_t)(f0 >> 6) * a.xcp) * col_stride + (f0 & 63) * 2;   // (a half-tile workgroup starts at frame pair 16 of its tile)
    // the output rows of this workgroup's frames: rows f0 ... of a single stream, or what the X tile's entry of the map says
    long long row0 = f0;
    int n_live = a.n_frames - f0, rstep = 1;
    if (a.xmap) {   // (uniform)
        const XTile xt = a.xmap[f0 >> 6];
        rstep = xt.live_step >> 8;
        row0 = xt.out_row0 + (long long)(f0 & 63) * rstep;
        n_live = (xt.live_step & 255) - (f0 & 63);
        if (n_live <= 0) return;   // (uniform) a staged buffer's gap frames: nothing of this tile is wanted (Vqt::batch_streams_device)
    }
    PVQ_STAMP(0);
    const int* my_list = a.list + wave * a.per_wave;
    const int n_blocks = __builtin_amdgcn_readfirstlane(my_list[0]);
    const int n = lane & 15, kq = lane >> 4;
    const float* xa = nullptr;
    const float2* bp = nullptr;
    f32x4 av[NS][NU];
    float2 bv[NS];
    auto fetch = [&](int s, int c) {   // stage: columns c .. c + 3 of the block
        bv[s] = bp[(size_t)(c / 4) * 64];
#pragma unroll
        for (int u = 0; u < NU; ++u) av[s][u] = *reinterpret_cast<const f32x4*>(xa + (size_t)c * col_stride + u * 64);
    };
    auto open_block = [&](const BandBlock& blk) {
        xa = xtile + (size_t)(blk.x0 + kq) * col_stride + n * 4;
        bp = reinterpret_cast<const float2*>(a.B) + (size_t)blk.boff3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) fetch(s, s * 4);
    };
    BandBlock blk{};
    if (n_blocks > 0) {
        blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[1])];
        open_block(blk);
    }
    for (int bi = 0; bi < n_blocks; ++bi) {
        f32x4v acc[NU][2];   // [half tile u][p: even / odd frames]
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u][p][q] = 0.0f;
        auto mul = [&](int s) {   // independent accumulators between two uses of one
#pragma unroll
            for (int part = 0; part < 2; ++part) {   // Re parts of the four columns, then Im parts
                const float b = part ? bv[s].y : bv[s].x;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][u][part], b, acc[u][0], 0, 0, 0);
                    acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][u][2 + part], b, acc[u][1], 0, 0, 0);
                }
            }
        };
        const int kb = __builtin_amdgcn_readfirstlane(blk.kb);
        const int kb_full = kb - kb % (NS * 4);
        int c = 0;
        for (; c < kb_full; c += NS * 4) {   // steady state: no branches, exact load counting
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fetch((s + NS - 1) % NS, c + (s + NS - 1) * 4);
                mul(s);
                __builtin_amdgcn_sched_barrier(0);   // keep the waits on a stage's operands inside the stage
            }
        }
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)          // remainder: the operands are already in flight
            if (c + s * 4 < kb) mul(s);
        const int bin0 = blk.bin0, nrows = blk.nrows;
        if (bi + 1 < n_blocks) {                      // the next block's first operands fly during the write-out
            blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[bi + 2])];
            open_block(blk);
        }
        const bool mine = n < nrows;                  // re columns of live rows
        const int bin = bin0 + n;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float im[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // row_ror:8: lane n <- lane n ^ 8 (through a scalar copy: the DPP of a vector element was seen merged across q)
                    const float re_q = acc[u][p][q];
                    im[q] = PVQ_DPP(re_q, 0x128);
                }
                if (mine) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dbs[(32 * u + 8 * kq + 2 * q + p) * LDB + bin] = acc[u][p][q] * acc[u][p][q] + im[q] * im[q];
                    if (a.out_cplx) {
                        int row_stride = a.n_bins;
                        asm volatile("" : "+s"(row_stride));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int fr = 32 * u + 8 * kq + 2 * q + p;
                            if (fr < n_live) a.out_cplx[(size_t)(row0 + (long long)fr * rstep) * row_stride + bin] = make_float2(acc[u][p][q], im[q]);
                        }
                    }
                }
            }
    }
    PVQ_STAMP(1);
    __syncthreads();
    PVQ_STAMP(2);
    band_finish<NU, NW, LDB>(dbs, a, row0, n_live, rstep, wave, lane);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PVQ_STAMP(3);
    }
}

// Split-bf16 form of the kernel product (the default, with the split-bf16 GEMM): the fp32 MFMA above runs at 1/16
// of the bf16 matrix rate, and a 16-bin block is 73 % zeros, so the stage is matrix-bound.  Here X and the
// coefficients are written as hi + mid + lo bf16 (exact 3-way split, see blockdft_gemm_tree_bf16x3) and eight
// spectrum columns (16 real k) go through six v_mfma_f32_32x32x16_bf16: 6 x 32 cycles instead of 8 x 64.  A lane
// (frame m, half kh) loads (Re, Im) of columns 4 kh .. 4 kh + 3 of its frame — the same bytes per lane as the fp32
// form — and splits them in registers; the coefficient planes come pre-split in B-operand order.
constexpr int B3_NS = 3;   // 8-column stages in flight

template <int MT, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void blockdft_banddots_db_bf16x3(BandArgs a) {
    const int stamp_slot = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) float dbs[];   // [MT * 32][ldb]: |x_vqt|^2, then dB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * (MT * 32);
    const int n = lane & 31, kx = lane >> 5;
    constexpr int col_stride = 128;   // floats between consecutive X columns of a 64-frame tile
    const float* xtile = a.X + ((size_t)(f0 >> 6) * a.xcp) * col_stride + (f0 & 63) * 2;   // this workgroup's frames
    // the output rows of this workgroup's frames: rows f0 ... of a single stream, or what the X tile's entry of the map says
    long long row0 = f0;
    int n_live = a.n_frames - f0, rstep = 1;
    if (a.xmap) {   // (uniform)
        const XTile xt = a.xmap[f0 >> 6];
        rstep = xt.live_step >> 8;
        row0 = xt.out_row0 + (long long)(f0 & 63) * rstep;
        n_live = (xt.live_step & 255) - (f0 & 63);
        if (n_live <= 0) return;   // (uniform) a staged buffer's gap frames: nothing of this tile is wanted (Vqt::batch_streams_device)
    }
    PVQ_STAMP(0);
    const int* my_list = a.list + wave * a.per_wave;
    const int n_blocks = __builtin_amdgcn_readfirstlane(my_list[0]);
    const float* xa = nullptr;
    const bf16x8* bp = nullptr;
    float2 av[B3_NS][MT][4];
    bf16x8 bv[B3_NS][3];
    auto fetch = [&](int s, int g) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bv[s][p] = bp[((size_t)g * 3 + p) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) av[s][mt][q] = *reinterpret_cast<const float2*>(xa + (size_t)(8 * g + q) * col_stride + mt * 64);
    };
    auto open_block = [&](const BandBlock& blk) {
        xa = xtile + (size_t)(blk.x0 + 4 * kx) * col_stride + n * 2;
        bp = reinterpret_cast<const bf16x8*>(a.B3) + (size_t)blk.boff3 * 3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < B3_NS - 1; ++s) fetch(s, s);
    };
    BandBlock blk{};
    if (n_blocks > 0) {
        blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[1])];
        open_block(blk);
    }
    for (int bi = 0; bi < n_blocks; ++bi) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
        auto mul = [&](int s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x8 vh, vm, vl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        const float x = part ? av[s][mt][q].y : av[s][mt][q].x;
                        const __bf16 hi = (__bf16)x;
                        const float r1 = x - (float)hi;
                        const __bf16 mid = (__bf16)r1;
                        vh[2 * q + part] = hi;
                        vm[2 * q + part] = mid;
                        vl[2 * q + part] = (__bf16)(r1 - (float)mid);
                    }
                }
                // smallest terms first
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, bv[s][1], acc[mt], 0, 0, 0);  // mid*mid
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, bv[s][2], acc[mt], 0, 0, 0);  // hi*lo
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, bv[s][0], acc[mt], 0, 0, 0);  // lo*hi
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, bv[s][1], acc[mt], 0, 0, 0);  // hi*mid
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm, bv[s][0], acc[mt], 0, 0, 0);  // mid*hi
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, bv[s][0], acc[mt], 0, 0, 0);  // hi*hi
            }
        };
        const int kg = __builtin_amdgcn_readfirstlane(blk.kg);
        const int kg_full = kg - kg % B3_NS;
        int g = 0;
        for (; g < kg_full; g += B3_NS) {   // steady state: no branches, exact load counting
#pragma unroll
            for (int s = 0; s < B3_NS; ++s) {
                fetch((s + B3_NS - 1) % B3_NS, g + s + B3_NS - 1);
                mul(s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < B3_NS - 1; ++s)         // remainder: the operands are already in flight
            if (g + s < kg) mul(s);
        // the next block's first operands fly while this block's results are written out
        const int bin0 = blk.bin0, nrows = blk.nrows;
        if (bi + 1 < n_blocks) {
            blk = a.blocks[__builtin_amdgcn_readfirstlane(my_list[bi + 2])];
            open_block(blk);
        }
        band_writeout<MT, MT == 2 ? BAND_LDB2 : 0>(acc, dbs, a, row0, n_live, rstep, bin0, nrows, lane);
    }
    PVQ_STAMP(1);
    __syncthreads();
    PVQ_STAMP(2);
    band_finish<MT, NW>(dbs, a, row0, n_live, rstep, wave, lane);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PVQ_STAMP(3);
    }
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// round-to-nearest-even bf16 of a finite float (host)
static inline uint16_t host_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float host_from_bf16(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// median over the sampled workgroups of the last profiled fused-GEMM launch: shader clock (MHz) held inside the K loop
float Vqt::last_sclk_mhz() {
    if (!dev_ || !dev_->block || !dev_->block->d_clk || dev_->block->clk_n <= 0) return 0.0f;
    BlockDftTables* t = dev_->block;
    const int per = 4;
    std::vector<unsigned long long> h((size_t)t->clk_n * per);
    if (hipSetDevice(device_id_) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0.0f;
    if (hipMemcpy(h.data(), t->d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0.0f;
    std::vector<double> r;
    for (int i = 0; i < t->clk_n; ++i)
        if (h[per * i + 3] > h[per * i + 1]) r.push_back(100.0 * (double)(h[per * i + 2] - h[per * i]) / (double)(h[per * i + 3] - h[per * i + 1]));
    if (r.empty()) return 0.0f;
    std::sort(r.begin(), r.end());
    return (float)r[r.size() / 2];
}

uint32_t Vqt::blockdft_columns() const { return (dev_ && dev_->block) ? (uint32_t)(dev_->block->n_tiles * CB_C) : 0u; }

// whether the block-DFT path can take several streams in one launch (the fused GEMM + tree kernels; the unfused fallback stages —
// more than 8 window groups — run one stream per call).  A pure predicate on the plan: it builds no tables and evicts none (a query
// such as pvq_vqt_resolve_algo must not move the handle's tables of the hop in use, let alone allocate on whatever device is current).
bool Vqt::blockdft_takes_streams(size_t hop) const {
    if (!blockdft_applicable(hop)) return false;
    const auto& groups = plan_.kernel.window_groups;
    bool divides = (hop & (hop - 1)) == 0;
    size_t nb_max = 0;
    for (const WindowGroup& g : groups) {
        divides = divides && g.window_size() % hop == 0;
        nb_max = std::max(nb_max, g.window_size() / hop);
    }
    if (!divides) return true;   // a general hop: blockdft_gemm_gen (blockdft_applicable has checked its conditions)
    const bool use_bf = gemm_split_bf16_ && hop % FB_BK == 0;
    return !dev_knob("PVQ_NO_FUSE", 0) && nb_max <= (size_t)CB_MAX_NB && groups.size() <= 8 && hop % (use_bf ? FB_BK : 64) == 0;
}

bool Vqt::blockdft_applicable(size_t hop) const {
    if (!has_device() || hop < 64 || hop % 64 != 0 || hop > 4096) return false;   // the mirrored K loop walks hop / 2 in stages of 32
    if (n_bins() > 1024) return false;
    bool divides = (hop & (hop - 1)) == 0;
    for (const WindowGroup& g : plan_.kernel.window_groups) divides = divides && g.window_size() % hop == 0;
    if (divides) {   // power-of-two hop dividing every window: hop-block GEMM + doubling tree
        for (const WindowGroup& g : plan_.kernel.window_groups)
            if (g.window_size() / hop > (size_t)CB_MAX_NB) return false;
        return true;
    }
    // general hop (a multiple of 64): whole hop blocks + the window's remainder, combined by Horner's rule over at most GEN_MAX_NQ blocks;
    // the fused kernel only (at most 8 window groups), windows a multiple of 64 samples
    if (plan_.kernel.window_groups.size() > 8) return false;
    for (const WindowGroup& g : plan_.kernel.window_groups) {
        const size_t ws = g.window_size();
        if (ws % 64 != 0 || ws / hop > (size_t)GEN_MAX_NQ) return false;
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
    PVQ_HIP(hipSetDevice(device_id_));   // the tables live on the handle's device, whatever device the calling thread has current
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
    // Only the spectrum columns some kernel row actually reads are computed (e.g. 602 of 871 at
    // 48 kHz / 7x36): col_of[g][i] is the i-th used column of group g, idx_of[g][c] its compressed index.
    std::vector<std::vector<uint32_t>> col_of(groups.size());
    std::vector<std::vector<int>> idx_of(groups.size());
    for (size_t g = 0; g < groups.size(); ++g) {
        std::vector<char> used(groups[g].filter_bank.cols + 1, 0);
        for (uint32_t c : groups[g].filter_bank.col_idx) used[c] = 1;
        for (uint32_t c : groups[g].negative_filter_bank.col_idx) used[c] = 1;
        idx_of[g].assign(used.size(), -1);
        for (uint32_t c = 0; c < used.size(); ++c)
            if (used[c]) {
                idx_of[g][c] = (int)col_of[g].size();
                col_of[g].push_back(c);
            }
    }
    {
        bool divides = (hop & (hop - 1)) == 0;
        for (const WindowGroup& g : groups) divides = divides && g.window_size() % hop == 0;
        t->general = !divides;
    }
    int e16r_off = 0, gtw_off = 0;
    for (size_t g = 0; g < groups.size(); ++g) {
        BlockGroup B{};
        if (t->general) {   // window = nq whole hop blocks + rem samples; no tree
            B.nq = (int)(groups[g].window_size() / hop);
            B.rem = (int)(groups[g].window_size() % hop);
            B.nb = B.nb_f = 1;
            B.levels = B.levels_f = 0;
        } else {
            B.nb = (int)(groups[g].window_size() / hop);
            B.levels = 0;
            while ((1 << B.levels) < B.nb) ++B.levels;
            B.nb_f = std::min(B.nb, 64);
            B.levels_f = std::min(B.levels, 6);
        }
        B.n_cols = (int)col_of[g].size();
        B.tile0 = tile;
        B.n_tiles = (B.n_cols + CB_C - 1) / CB_C;
        B.tw_off = tw_off;
        B.s_rel = (long long)groups[g].window_begin - (long long)plan_.params.n_fft;  // + n_lead + hop at launch
        B.e16r_off = e16r_off;
        B.gtw_off = gtw_off;
        e16r_off += B.n_tiles * (B.rem / 2) * 16;
        gtw_off += 2 * B.n_tiles * CB_C;
        tile += B.n_tiles;
        tw_off += B.levels * B.n_tiles * CB_C;
        t->nb_max = std::max(t->nb_max, B.nb);
        t->groups.push_back(B);
    }
    t->n_tiles = tile;
    const int ntot = tile * GM_BN;
    auto tq = [&](double v) { const float f = (float)v; return twiddle_fp16_ ? round_to_half(f) : f; };   // config 4: fp16 twiddles
    // The hop DFT is taken about the centre of the hop block (see blockdft_gemm_tree): E[m][c] = e^{-i th_c u_m},
    // u_m = m - (hop-1)/2, th_c = 2 pi c / W.  Every GEMM form therefore yields P' = P / rho_c, rho_c = e^{-i th_c (hop-1)/2},
    // and the tree X' = X / rho_c; rho_c goes into the kernel-product coefficients below.
    //   E [hop][Ntot], (cos, sin) interleaved per column; the mirrored fp32 form reads its first hop/2 rows
    std::vector<float> E((size_t)hop * ntot, 0.0f);
    std::vector<std::vector<std::pair<double, double>>> rho(groups.size());
    std::vector<int> tile_group(tile);
    std::vector<float2> comb_tw((size_t)std::max(tw_off, 1), make_float2(0.0f, 0.0f));
    for (size_t g = 0; g < groups.size(); ++g) {
        const BlockGroup& B = t->groups[g];
        const double W = (double)groups[g].window_size();
        const long long W2 = 2ll * (long long)groups[g].window_size();
        for (int tt = 0; tt < B.n_tiles; ++tt) tile_group[B.tile0 + tt] = (int)g;
        rho[g].resize(B.n_cols);
        for (int ci = 0; ci < B.n_cols; ++ci) {
            const long long c = (long long)col_of[g][ci];  // actual spectrum column
            {   // the phase of the centred block DFT whose results the kernel product sees: the hop block's, or — a window shorter than a
                // general hop, where the transform is the remainder GEMM alone — the window's
                const long long D = t->general && B.nq == 0 ? (long long)B.rem : (long long)hop;
                const long long prod = (c * (D - 1)) % W2;
                const double ang = -2.0 * pi * (double)prod / (double)W2;
                rho[g][ci] = {std::cos(ang), std::sin(ang)};
            }
            for (size_t m = 0; m < hop; ++m) {
                // reduce the angle exactly: c * (2m - hop + 1) mod 2W in integers
                long long prod = (c * (2ll * (long long)m - (long long)hop + 1ll)) % W2;
                if (prod < 0) prod += W2;
                const double ang = -2.0 * pi * (double)prod / (double)W2;
                const float er = tq(std::cos(ang)), ei = tq(std::sin(ang));
                E[m * ntot + (size_t)B.tile0 * GM_BN + 2 * ci] = er;
                E[m * ntot + (size_t)B.tile0 * GM_BN + 2 * ci + 1] = ei;
            }
            for (int l = 0; l < B.levels; ++l) {
                const long long prod = (c * (1ll << l)) % (long long)B.nb;
                const double ang = -2.0 * pi * (double)prod / (double)B.nb;
                comb_tw[B.tw_off + l * (B.n_tiles * CB_C) + ci] = make_float2(tq(std::cos(ang)), tq(std::sin(ang)));
            }
        }
        (void)W;
    }
    // Banded kernel product tables: per window group, blocks of BD_RB consecutive bins; a block walks the
    // union of its rows' (compressed) columns.  B operand of column c, lane l (n = l & 31, k = l >> 5):
    // output n = part * 16 + row, k = 0 multiplies Re X, k = 1 multiplies Im X:
    //     y += v X        : re += vr Xr - vi Xi,  im += vi Xr + vr Xi      (filter_bank, vqt.rs:889-895)
    //     y += conj(w X)  : re += wr Xr - wi Xi,  im += -wi Xr - wr Xi     (negative_filter_bank, vqt.rs:896-910)
    // with v, w the reference's coefficients times rho_c (the GEMM stages deliver X' = X / rho_c, see above).
    const int nb = (int)n_bins();
    t->n_bins_pad = (nb + 63) / 64 * 64;
    std::vector<BandBlock> band;
    std::vector<float> band_B;
    // columns are stored in pairs: element (column c, lane l) lives at [(c / 2) * 64 + l] * 2 + (c & 1)
    auto at = [](float* Bp, int c, int l) -> float& { return Bp[((size_t)(c >> 1) * 64 + l) * 2 + (c & 1)]; };
    for (size_t g = 0; g < groups.size(); ++g) {
        const CsrMatrix& A = groups[g].filter_bank;
        const CsrMatrix& Bm = groups[g].negative_filter_bank;
        const int xoff = t->groups[g].tile0 * CB_C;
        for (uint32_t r0 = 0; r0 < A.rows; r0 += BD_RB) {
            const uint32_t r1 = std::min<uint32_t>(A.rows, r0 + BD_RB);
            int lo = 1 << 30, hi = -1;
            for (uint32_t r = r0; r < r1; ++r) {
                for (uint32_t q = A.row_ptr[r]; q < A.row_ptr[r + 1]; ++q) {
                    lo = std::min(lo, idx_of[g][A.col_idx[q]]);
                    hi = std::max(hi, idx_of[g][A.col_idx[q]]);
                }
                if (Bm.nnz() > 0)
                    for (uint32_t q = Bm.row_ptr[r]; q < Bm.row_ptr[r + 1]; ++q) {
                        lo = std::min(lo, idx_of[g][Bm.col_idx[q]]);
                        hi = std::max(hi, idx_of[g][Bm.col_idx[q]]);
                    }
            }
            BandBlock bb{};
            bb.bin0 = (int)(groups[g].first_bin + r0);
            bb.nrows = (int)(r1 - r0);
            bb.boff = (int)(band_B.size() / 64);
            if (hi < 0) {   // rows without coefficients: one all-zero stage
                lo = 0;
                hi = 0;
            }
            bb.x0 = xoff + lo;
            bb.kb = ((hi - lo + 1) + BD_KU - 1) / BD_KU * BD_KU;
            band_B.resize(band_B.size() + (size_t)bb.kb * 64, 0.0f);
            float* Bp = band_B.data() + (size_t)bb.boff * 64;
            for (uint32_t r = r0; r < r1; ++r) {
                const int row = (int)(r - r0);
                for (uint32_t q = A.row_ptr[r]; q < A.row_ptr[r + 1]; ++q) {
                    const int ci = idx_of[g][A.col_idx[q]];
                    const int cc = ci - lo;
                    // v * rho_c, in double, rounded once
                    const double ar_ = A.values[q].re, ai_ = A.values[q].im;
                    const float vr = (float)(ar_ * rho[g][ci].first - ai_ * rho[g][ci].second);
                    const float vi = (float)(ar_ * rho[g][ci].second + ai_ * rho[g][ci].first);
                    at(Bp, cc, 0 * 32 + row) += vr;        // k = 0 (Re X) -> re
                    at(Bp, cc, 1 * 32 + row) += -vi;       // k = 1 (Im X) -> re
                    at(Bp, cc, 0 * 32 + 16 + row) += vi;   // k = 0 -> im
                    at(Bp, cc, 1 * 32 + 16 + row) += vr;   // k = 1 -> im
                }
                if (Bm.nnz() > 0)
                    for (uint32_t q = Bm.row_ptr[r]; q < Bm.row_ptr[r + 1]; ++q) {
                        const int ci = idx_of[g][Bm.col_idx[q]];
                        const int cc = ci - lo;
                        const double br_ = Bm.values[q].re, bi_ = Bm.values[q].im;
                        const float wr = (float)(br_ * rho[g][ci].first - bi_ * rho[g][ci].second);
                        const float wi = (float)(br_ * rho[g][ci].second + bi_ * rho[g][ci].first);
                        at(Bp, cc, 0 * 32 + row) += wr;
                        at(Bp, cc, 1 * 32 + row) += -wi;
                        at(Bp, cc, 0 * 32 + 16 + row) += -wi;
                        at(Bp, cc, 1 * 32 + 16 + row) += -wr;
                    }
            }
            band.push_back(bb);
        }
    }
    // The same coefficients for the 16x16x4 MFMA form: blocks of BD8_RB = 8 bins (16 output columns = 8 x (re, im)) walk the
    // union of 8 rows' columns — about 35 instead of 57 columns per block, so 39 % fewer matrix operations for the same
    // products.  B operand of a column pair, lane l (n = l & 15: part = n >> 3, row = n & 7; k = l >> 4: column k >> 1 of
    // the pair, k & 1 = 0 multiplies Re X, 1 multiplies Im X).
    std::vector<BandBlock> band8;
    std::vector<float> band_B8, band_B4;
    for (size_t g = 0; g < groups.size(); ++g) {
        const CsrMatrix& A = groups[g].filter_bank;
        const CsrMatrix& Bm = groups[g].negative_filter_bank;
        const int xoff = t->groups[g].tile0 * CB_C;
        for (uint32_t r0 = 0; r0 < A.rows; r0 += BD8_RB) {
            const uint32_t r1 = std::min<uint32_t>(A.rows, r0 + BD8_RB);
            int lo = 1 << 30, hi = -1;
            for (uint32_t r = r0; r < r1; ++r) {
                for (uint32_t q = A.row_ptr[r]; q < A.row_ptr[r + 1]; ++q) {
                    lo = std::min(lo, idx_of[g][A.col_idx[q]]);
                    hi = std::max(hi, idx_of[g][A.col_idx[q]]);
                }
                if (Bm.nnz() > 0)
                    for (uint32_t q = Bm.row_ptr[r]; q < Bm.row_ptr[r + 1]; ++q) {
                        lo = std::min(lo, idx_of[g][Bm.col_idx[q]]);
                        hi = std::max(hi, idx_of[g][Bm.col_idx[q]]);
                    }
            }
            if (hi < 0) {
                lo = 0;
                hi = 0;
            }
            BandBlock bb{};
            bb.bin0 = (int)(groups[g].first_bin + r0);
            bb.nrows = (int)(r1 - r0);
            bb.boff = (int)(band_B8.size() / 64);   // in column pairs
            bb.x0 = xoff + lo;
            bb.kb = ((hi - lo + 1) + BD8_KU - 1) / BD8_KU * BD8_KU;
            band_B8.resize(band_B8.size() + (size_t)(bb.kb / 2) * 64, 0.0f);
            float* Bp = band_B8.data() + (size_t)bb.boff * 64;
            // element (column cc, kk = Re / Im input, part = re / im output, row)
            // pairs are stored two by two (one 8-byte load per lane and stage): pair p, lane l at ((p >> 1) * 64 + l) * 2 + (p & 1)
            auto at8 = [&](int cc, int kk, int part, int row) -> float& {
                const int pr = cc >> 1, l = ((cc & 1) * 2 + kk) * 16 + part * 8 + row;
                return Bp[((size_t)(pr >> 1) * 64 + l) * 2 + (pr & 1)];
            };
            for (uint32_t r = r0; r < r1; ++r) {
                const int row = (int)(r - r0);
                for (uint32_t q = A.row_ptr[r]; q < A.row_ptr[r + 1]; ++q) {
                    const int ci = idx_of[g][A.col_idx[q]];
                    const int cc = ci - lo;
                    const double ar_ = A.values[q].re, ai_ = A.values[q].im;
                    const float vr = (float)(ar_ * rho[g][ci].first - ai_ * rho[g][ci].second);
                    const float vi = (float)(ar_ * rho[g][ci].second + ai_ * rho[g][ci].first);
                    at8(cc, 0, 0, row) += vr;
                    at8(cc, 1, 0, row) += -vi;
                    at8(cc, 0, 1, row) += vi;
                    at8(cc, 1, 1, row) += vr;
                }
                if (Bm.nnz() > 0)
                    for (uint32_t q = Bm.row_ptr[r]; q < Bm.row_ptr[r + 1]; ++q) {
                        const int ci = idx_of[g][Bm.col_idx[q]];
                        const int cc = ci - lo;
                        const double br_ = Bm.values[q].re, bi_ = Bm.values[q].im;
                        const float wr = (float)(br_ * rho[g][ci].first - bi_ * rho[g][ci].second);
                        const float wi = (float)(br_ * rho[g][ci].second + bi_ * rho[g][ci].first);
                        at8(cc, 0, 0, row) += wr;
                        at8(cc, 1, 0, row) += -wi;
                        at8(cc, 0, 1, row) += -wi;
                        at8(cc, 1, 1, row) += -wr;
                    }
            }
            // the same block in the order of blockdft_banddots4c_db: per 4 columns (group gq) and lane (kq = column of the group, n = part * 8 + row)
            // a float2 (coefficient of Re X, coefficient of Im X)
            bb.boff3 = (int)(band_B4.size() / 128);   // in groups of 4 columns
            band_B4.resize(band_B4.size() + (size_t)(bb.kb / 4) * 128, 0.0f);
            {
                float* B4 = band_B4.data() + (size_t)bb.boff3 * 128;
                for (int cc = 0; cc < bb.kb; ++cc)
                    for (int kk = 0; kk < 2; ++kk)
                        for (int part = 0; part < 2; ++part)
                            for (int row = 0; row < 8; ++row)
                                B4[((size_t)(cc >> 2) * 64 + (cc & 3) * 16 + part * 8 + row) * 2 + kk] = at8(cc, kk, part, row);
            }
            band8.push_back(bb);
        }
    }
    band_B4.resize(band_B4.size() + (size_t)8 * 128, 0.0f);   // the prefetch of the last block runs on past it
    band_B8.resize(band_B8.size() + (size_t)8 * (BD8_KU / 2) * 64, 0.0f);   // the prefetch of the last block runs on past it
    t->band_per_wave8 = (int)band8.size() + 2;
    std::vector<int> band_list8((size_t)8 * t->band_per_wave8, 0);
    {
        std::vector<std::vector<int>> per_wave(8);
        for (size_t i = 0; i < band8.size(); ++i) per_wave[i % 8].push_back((int)i);
        for (int w = 0; w < 8; ++w) {
            int* row = band_list8.data() + (size_t)w * t->band_per_wave8;
            row[0] = (int)per_wave[w].size();
            for (size_t i = 0; i < per_wave[w].size(); ++i) row[1 + i] = per_wave[w][i];
        }
    }
    band_B.resize(band_B.size() + (size_t)BD_NS * BD_KU * 64, 0.0f);   // the prefetch of the last block runs on past it
    // split-bf16 planes of the same coefficients, 8 columns (16 real k) per MFMA: lane (n = l & 31, kh = l >> 5)
    // holds k = 8 kh + t, t = 0..7  <->  column 4 kh + t / 2, Re / Im row t & 1
    std::vector<uint16_t> band_B3;
    for (BandBlock& bb : band) {
        bb.kg = (bb.kb + 7) / 8;
        bb.boff3 = (int)(band_B3.size() / (3 * 64 * 8));
        band_B3.resize(band_B3.size() + (size_t)bb.kg * 3 * 64 * 8, 0);
        float* Bp = band_B.data() + (size_t)bb.boff * 64;
        for (int g = 0; g < bb.kg; ++g)
            for (int l = 0; l < 64; ++l)
                for (int tt = 0; tt < 8; ++tt) {
                    const int col = 8 * g + 4 * (l >> 5) + (tt >> 1);
                    const float x = col < bb.kb ? at(Bp, col, (tt & 1) * 32 + (l & 31)) : 0.0f;
                    const uint16_t h = host_to_bf16(x);
                    const float r1 = x - host_from_bf16(h);
                    const uint16_t m = host_to_bf16(r1);
                    const uint16_t lo = host_to_bf16(r1 - host_from_bf16(m));
                    const size_t base = ((size_t)(bb.boff3 + g) * 3) * 64 * 8 + (size_t)l * 8 + tt;
                    band_B3[base] = h;
                    band_B3[base + 64 * 8] = m;
                    band_B3[base + 2 * 64 * 8] = lo;
                }
    }
    band_B3.resize(band_B3.size() + (size_t)B3_NS * 3 * 64 * 8, 0);
    // blocks to waves: round robin in bin order, so that the four waves of a workgroup walk neighbouring blocks
    // (whose column ranges overlap) at the same time and share the X columns through L1 / L2
    // two sets of lists: for `band_waves` waves per workgroup (fp32 form) and for 4 (split-bf16 form, whose register
    // budget does not fit four waves per SIMD)
    t->band_waves = 8;   // fp32 forms: 8 waves per workgroup (4 waves per SIMD when two workgroups fit a CU)
    t->band_per_wave = (int)band.size() + 2;
    std::vector<int> band_list((size_t)(t->band_waves + 4) * t->band_per_wave, 0);
    auto deal = [&](int first_row, int waves) {
        // dealt round robin in bin order: the waves of a workgroup walk neighbouring blocks (overlapping column ranges)
        // at the same time and share the X columns through L1 / L2 (balancing by cost instead was measured no faster)
        std::vector<std::vector<int>> per_wave(waves);
        for (size_t i = 0; i < band.size(); ++i) per_wave[i % waves].push_back((int)i);
        for (int w = 0; w < waves; ++w) {
            int* row = band_list.data() + (size_t)(first_row + w) * t->band_per_wave;
            row[0] = (int)per_wave[w].size();
            for (size_t i = 0; i < per_wave[w].size(); ++i) row[1 + i] = per_wave[w][i];
        }
    };
    deal(0, t->band_waves);
    deal(t->band_waves, 4);
    if (tile * CB_C >= 0x8000) {
        free_blockdft_tables(t);
        set_last_error("unsupported: too many spectrum columns for the block-DFT path");
        return PVQ_ERR_UNSUPPORTED;
    }
    std::vector<long long> tile_s(tile, 0);
    for (size_t g = 0; g < groups.size(); ++g)
        for (int tt = 0; tt < t->groups[g].n_tiles; ++tt) tile_s[t->groups[g].tile0 + tt] = t->groups[g].s_rel;
    std::vector<float4> E16;
    {   // E in the B-operand order of the 16x16x4 GEMM: [column tile][k < hop / 2][n < 16]: (cos c_n, cos c_{n+16}, -sin c_n, -sin c_{n+16})
        const size_t K2 = hop / 2;
        E16.resize((size_t)tile * K2 * 16);
        for (int tt = 0; tt < tile; ++tt)
            for (size_t m = 0; m < K2; ++m)
                for (int n = 0; n < 16; ++n) {
                    const float* e = E.data() + m * ntot + (size_t)tt * GM_BN;
                    E16[((size_t)tt * K2 + m) * 16 + n] = make_float4(e[2 * n], e[2 * (n + 16)], e[2 * n + 1], e[2 * (n + 16) + 1]);
                }
    }
    std::vector<float4> E16R((size_t)std::max(e16r_off, 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<float2> gen_tw((size_t)std::max(gtw_off, 1), make_float2(1.0f, 0.0f));
    if (t->general) {
        for (size_t g = 0; g < groups.size(); ++g) {
            const BlockGroup& B = t->groups[g];
            const long long W = (long long)groups[g].window_size(), W2 = 2 * W;
            auto cs = [&](long long num, long long den) {   // e^{-2 pi i num / den}, the angle reduced exactly
                long long r = num % den;
                if (r < 0) r += den;
                const double ang = -2.0 * pi * (double)r / (double)den;
                return make_float2(tq(std::cos(ang)), tq(std::sin(ang)));
            };
            for (int ci = 0; ci < B.n_cols; ++ci) {
                const long long c = (long long)col_of[g][ci];
                const int tt = ci / CB_C, n32 = ci % CB_C;
                // E_R[m][c] = e^{-i th_c (m - (rem - 1) / 2)}, m < rem / 2 (the mirrored form reads the first half), B-operand order of E16
                for (int m = 0; m < B.rem / 2; ++m) {
                    const float2 e = cs(c * (2ll * m - (long long)B.rem + 1ll), W2);
                    float4& dst = E16R[(size_t)B.e16r_off + ((size_t)tt * (B.rem / 2) + m) * 16 + (n32 & 15)];
                    if (n32 < 16) { dst.x = e.x; dst.z = e.y; } else { dst.y = e.x; dst.w = e.y; }
                }
                // phi_c = e^{-2 pi i c hop / W};  tau_c = phi_c^nq rho_R / rho_Q = e^{-i th_c (nq hop + (rem - hop) / 2)}
                gen_tw[(size_t)B.gtw_off + ci] = cs(c * (long long)hop, W);
                gen_tw[(size_t)B.gtw_off + B.n_tiles * CB_C + ci] = B.nq > 0 ? cs(c * (2ll * B.nq * (long long)hop + (long long)B.rem - (long long)hop), W2) : make_float2(1.0f, 0.0f);
            }
        }
    }
    t->h_E = E;  // kept for the lazily built bf16 planes
    bool ok = up(&t->d_E16R, E16R) && up(&t->d_gen_tw, gen_tw) && up(&t->d_E, E) && up(&t->d_tile_group, tile_group) && up(&t->d_tile_s, tile_s) && up(&t->d_groups, t->groups) &&
              up(&t->d_comb_tw, comb_tw) && up(&t->d_band, band) && up(&t->d_band_B, band_B) && up(&t->d_band_list, band_list) && up(&t->d_band8, band8) &&
              up(&t->d_band_B4, band_B4) && up(&t->d_band_list8, band_list8) &&
              up(reinterpret_cast<uint16_t**>(&t->d_band_B3), band_B3) &&
              up(&t->d_E16, E16);
    if (!ok) {
        free_blockdft_tables(t);
        set_last_error("hipMalloc/hipMemcpy failed while building block-DFT tables");
        return PVQ_ERR_DEVICE;
    }
    dev_->block = t;
    return PVQ_OK;
}

// Every kernel of a sub-batch runs on the caller's stream; peaks once over the whole batch.  (A two-stream
// variant that overlapped the GEMM of sub-batch c+1 with the memory-bound stages of c was measured slower:
// the streams contend for the same CUs.)
pvq_status Vqt::launch_blockdft_path(const float* d_pcm, size_t n_lead, size_t hop, size_t n_frames, float* d_out_db,
                                     float* d_out_cplx, const PeakParamsDev* pk, hipStream_t stream) {
    const StreamIn one{d_pcm, n_lead + hop, n_lead + n_frames * hop, n_frames, 0, 1};
    return launch_blockdft_streams(&one, 1, hop, d_out_db, d_out_cplx, n_frames, pk, stream);
}

// The same path over MANY streams (pvq_vqt_*_streams).  Every stream is cut into runs of at most one sub-batch; runs are packed
// into launches up to the sub-batch size (the workspace limit), so 64 streams of 2 048 frames are ONE launch per stage where a
// loop over single-stream calls pays 64 ramps and tails per stage; a long stream still runs sub-batch by sub-batch, alone in
// its launches, exactly as through the single-stream entry point.  Stream s writes its rows to out rows st[s].out_row0 ...
pvq_status Vqt::launch_blockdft_streams(const StreamIn* st, size_t n_st, size_t hop, float* d_out_db, float* d_out_cplx, size_t rows_total,
                                        const PeakParamsDev* pk, hipStream_t stream) {
    pvq_status pst = prepare_blockdft(hop);
    if (pst != PVQ_OK) return pst;
    BlockDftTables* t = dev_->block;
    const int ntot = t->n_tiles * GM_BN, xc = t->n_tiles * CB_C;
    // X is blocked by 64-frame tiles: [tile][column][64 frames], so the kernel-product workgroup of a tile streams one
    // contiguous region (and a column step is a constant 512 bytes); X_PAD_COLS zeroed columns close every tile
    const int xcp = xc + X_PAD_COLS;
    size_t longest = 0, total_frames = 0;
    for (size_t i = 0; i < n_st; ++i) {
        longest = std::max(longest, st[i].n_frames);
        total_frames += st[i].n_frames;
    }
    const size_t chunk = std::min(n_st == 1 ? longest : std::max<size_t>((total_frames + 63) / 64 * 64, 64),
                                  chunk_frames(workspace_limit_, (size_t)xcp * sizeof(float2) * (t->nb_max > 64 ? 2 : 1)));
    const bool use_bf = gemm_split_bf16_ && hop % FB_BK == 0 && !t->general;   // (a general hop runs the fp32 GEMM whatever the setting; the kernel product follows the setting)
    static const bool fuse_env = !dev_knob("PVQ_NO_FUSE", 0);
    const bool fused = t->general || (fuse_env && t->nb_max <= CB_MAX_NB && t->n_groups <= 8 && hop % (use_bf ? FB_BK : 64) == 0);
    if (!fused && n_st > 1) {
        set_last_error("internal: the unfused block-DFT stages take one stream per call");
        return PVQ_ERR_INTERNAL;
    }
    // runs -> launches: a launch holds runs of whole 64-frame tiles up to the sub-batch size
    struct Run { size_t stream, fbeg, nf; };
    std::vector<std::vector<Run>> launches;
    {
        const size_t budget = (chunk + 63) / 64;   // tiles per launch
        size_t used = 0;
        for (size_t i = 0; i < n_st; ++i) {
            size_t fbeg = 0, left = st[i].n_frames;
            while (left > 0) {
                const size_t nf = std::min(left, chunk);
                const size_t tiles = (nf + 63) / 64;
                if (launches.empty() || used + tiles > budget || launches.back().size() >= 0xFFFFu) {   // (a tile-list entry names its run in 16 bits)
                    launches.emplace_back();
                    used = 0;
                }
                launches.back().push_back(Run{i, fbeg, nf});
                used += tiles;
                fbeg += nf;
                left -= nf;
            }
        }
    }
    size_t x_tiles = 1, y_tiles = 1;
    for (const auto& L : launches) {
        size_t xt = 0, yt = 0;
        for (const Run& r : L) {
            xt += (r.nf + 63) / 64;
            yt += (r.nf + (size_t)std::max(t->nb_max - 64, 0) + 63) / 64;
        }
        x_tiles = std::max(x_tiles, xt);
        y_tiles = std::max(y_tiles, yt);
    }
    const size_t rows_cap = chunk + t->nb_max - 1;
    const size_t x_bytes = x_tiles * (size_t)xcp * 64 * sizeof(float2);
    if (t->x_cap < x_bytes) {
        if (t->d_X) PVQ_HIP(hipFree(t->d_X));
        t->d_X = nullptr; t->x_cap = 0;
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_X), x_bytes));
        PVQ_HIP(hipMemsetAsync(t->d_X, 0, x_bytes, stream));
        t->x_cap = x_bytes;
    }
    if (fused && (t->nb_max > 64 || t->general)) {   // (general hops: the remainder GEMM's results, laid out like X)
        const size_t y_bytes = y_tiles * (size_t)xcp * 64 * sizeof(float2);
        if (t->y_cap < y_bytes) {
            if (t->d_Y) PVQ_HIP(hipFree(t->d_Y));
            t->d_Y = nullptr; t->y_cap = 0;
            PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_Y), y_bytes));
            t->y_cap = y_bytes;
        }
    }
    if (!fused) {
        const size_t p_bytes = rows_cap * ntot * sizeof(float);
        if (t->p_cap < p_bytes) {
            if (t->d_P) PVQ_HIP(hipFree(t->d_P));
            t->d_P = nullptr; t->p_cap = 0;
            PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_P), p_bytes));
            t->p_cap = p_bytes;
        }
    }
    if (use_bf && fused && !t->d_Et) {
        // bf16 split of E^T (round-to-nearest-even on the bit patterns), built on first use
        std::vector<uint16_t> Et((size_t)3 * ntot * hop);
        for (int n = 0; n < ntot; ++n)
            for (size_t m = 0; m < hop; ++m) {
                const float x = t->h_E[m * ntot + n];
                const uint16_t h = host_to_bf16(x);
                const float r1 = x - host_from_bf16(h);
                const uint16_t mid = host_to_bf16(r1);
                Et[((size_t)0 * ntot + n) * hop + m] = h;
                Et[((size_t)1 * ntot + n) * hop + m] = mid;
                Et[((size_t)2 * ntot + n) * hop + m] = host_to_bf16(r1 - host_from_bf16(mid));
            }
        PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_Et), Et.size() * 2));
        PVQ_HIP(hipMemcpy(t->d_Et, Et.data(), Et.size() * 2, hipMemcpyHostToDevice));
    }
    const int nb = (int)n_bins();
    float2* X = t->d_X;
    // the launch's base pointer: the lowest stream pointer (segment offsets are counted from it, in samples)
    const float* pcm_min = st[0].d_pcm;
    for (size_t i = 1; i < n_st; ++i)
        if (st[i].d_pcm < pcm_min) pcm_min = st[i].d_pcm;
    for (const auto& L : launches) {
        // per run: rebase its stream so that every byte offset of the launch fits 32 bits
        bool strided = false;   // a run that holds every r-th frame of its stream: its rows go through the X-tile map too
        std::vector<BlockDftTables::SegKey> segs(L.size());
        size_t xt_n = 0, yt_n = 0, nf_launch = 0;
        for (size_t u = 0; u < L.size(); ++u) {
            const Run& r = L[u];
            const StreamIn& S = st[r.stream];
            const long long n_samples = (long long)S.n_samples;
            const long long first_needed = (long long)S.first_end + (long long)r.fbeg * (long long)hop - (long long)plan_.params.n_fft;
            const long long rebase = std::max<long long>(0, std::min<long long>(first_needed, n_samples));
            const long long extent = std::min<long long>(n_samples - rebase, (long long)(r.nf + 2) * (long long)hop + (long long)plan_.params.n_fft + 4096);
            BlockDftTables::SegKey& k = segs[u];
            k.pcm_off = (long long)(S.d_pcm - pcm_min) + rebase;
            k.pcm_bytes = (unsigned)std::min<long long>(extent * 4, 0xFFFFF000ll);
            k.base = (long long)S.first_end + (long long)r.fbeg * (long long)hop - rebase;
            k.nf = (int)r.nf;
            k.x_tile0 = (int)xt_n;
            k.y_tile0 = (int)yt_n;
            k.out_row0 = (long long)(S.out_row0 + r.fbeg * S.row_step);
            k.row_step = (int)S.row_step;
            k.slot_hash = S.slots ? (S.slot_hash | 1ull) : 0ull;
            k.grid_i = (long long)S.grid_i;
            k.fbeg = (long long)r.fbeg;
            strided |= S.row_step != 1 || S.slots != nullptr;
            xt_n += (r.nf + 63) / 64;
            yt_n += (r.nf + (size_t)std::max(t->nb_max - 64, 0) + 63) / 64;
            nf_launch += r.nf;
        }
        // a launch of one run goes through the kernel arguments (segs == nullptr), as the single-stream entry point always did
        const bool multi = L.size() > 1 || strided;
        const float* pcm_base = multi ? pcm_min : pcm_min + segs[0].pcm_off;
        const unsigned pcm_bytes = segs[0].pcm_bytes;
        const long long base = segs[0].base;
        const size_t nf = multi ? xt_n * 64 : (size_t)segs[0].nf;   // frames the per-frame stages of the launch cover
        const SegDev* d_segs = nullptr;
        const XTile* d_xmap = nullptr;
        if (fused) {
            GemmTreeArgs fa;
            fa.pcm_base = pcm_base;
            fa.pcm_bytes = pcm_bytes;
            fa.E = t->d_E;
            fa.ld = ntot;
            fa.X = X;
            fa.Y = t->d_Y;
            fa.xcp = xcp;
            fa.n_frames = (int)segs[0].nf;
            fa.K = (int)hop;
            fa.base = base;
            fa.n_groups = t->n_groups;
            static const int dyn_lds_env = dev_knob("PVQ_DYN_LDS", 0);      // extra LDS -> one workgroup per CU
            static const int bm_env = dev_knob("PVQ_FUSED_BM", 0);          // 128: 128-row tiles (the tile-shape bit-identity test)
            // matrix work of the launch in whole-tile units, for tiles of bm rows
            auto count_tiles = [&](int bm) {
                double n = 0.0;
                for (const auto& k : segs)
                    for (int g = 0; g < t->n_groups; ++g) {
                        const int S = bm - t->groups[g].nb_f + 1;
                        const int rows_g = k.nf + t->groups[g].nb - t->groups[g].nb_f;
                        // a last tile of at most 16 columns runs half the MFMAs (fp32 kernel; the few tiles at the stream's ends run the full
                        // loop: counted as half all the same)
                        const bool half_last = !use_bf && t->groups[g].n_cols - (t->groups[g].n_tiles - 1) * CB_C <= 16;
                        n += (t->groups[g].n_tiles - (half_last ? 0.5 : 0.0)) * ((rows_g + S - 1) / S);
                    }
                return n;
            };
            // 256-row tiles: 257 - Nb complete frames per tile (1.08x row recomputation instead of 1.2x with 128 rows) — for a launch that
            // fills the chip's 512 workgroup slots a few times over.  A smaller one (fewer than FUSED_SMALL 256-row tiles: up to ~12 000
            // frames at 48 kHz / 252 bins) takes 128-row tiles: twice the workgroups, each half as long — what such a launch lacks is
            // parallelism, not efficiency (hop 1 600: 4 096 frames 294 -> 216 us, 8 192: 324 -> 292; hop 256: 2 048 frames 61 -> 54;
            // profiles/r04_small_tiles.txt).  Same bits either way (a frame's values do not depend on its tile: tests/test_tile_shapes).
            constexpr double FUSED_SMALL = 1400.0;
            int fused_bm = 256;
            if (!use_bf && (bm_env == 128 || (bm_env == 0 && count_tiles(256) < FUSED_SMALL))) fused_bm = 128;
            double eff_tiles = count_tiles(fused_bm);
            // Frame-stripe tile order: the launch's frames are cut into stripes of 2 048, stripe s belongs to XCD queue s & 7 (workgroup b
            // runs on the XCD of all b' = b mod 8), and a queue takes its stripes in order, within a stripe every group's row tiles with
            // all their column tiles.  All window groups then read a stripe's PCM rows from that XCD's L2 while they are resident (once
            // per stripe, not once per group).  Entry: (group | wide << 8 | segment << 16, column tile, first frame, position i * 8 + queue).
            static const int FS = dev_knob("PVQ_TILE_FS", 2048);
            static const int balance_env = dev_knob("PVQ_BALANCE", 1);   // 0: queues as the stripes fall
            static const int order_env = dev_knob("PVQ_ORDER", 1);       // 0: wide and narrow tiles of a stripe interleaved
            static const int pair_half_env = dev_knob("PVQ_PAIR_HALF", 0); // 1: a group's last tile of at most 16 columns pairs up too (measured: same time, more MFMAs)
            static const int tail_env = dev_knob("PVQ_TAIL", 128);       // narrow entries at the end of every queue
            static const int wide_env = dev_knob("PVQ_WIDE", 1);         // 0: narrow tiles only; 2: wide tiles to the very end of every queue; 3: none in a queue's last stripe
            static const int tree3_env = dev_knob("PVQ_TREE3", 0);       // 1: blockdft_gemm_tree3 (three workgroups per CU, 32-column tiles, P' in 16-column quarters)
            const bool tree3 = tree3_env && !use_bf && fused_bm == 256 && !t->general;
            const int wide_mode = !use_bf && fused_bm == 256 && !t->general && !tree3 ? wide_env : 0;   // (the split-bf16 kernel, the 128-row form and the general-hop kernel take 32-column tiles only)
            if (segs.size() > 0xFFFFu) {
                set_last_error("too many streams in one launch");
                return PVQ_ERR_INTERNAL;
            }
            // The list is cached per (tile rows, kernel family, the runs' stream geometry): which tiles lie wholly inside their stream —
            // 16-byte loads, wide entries — is decided here with the kernel's own test, so a list built for one geometry must never be
            // used for another (round 3: a list keyed on the frame count alone let a launch read past a shorter stream's end)
            // (a launch of one run hands its stream pointer and output rows over in the kernel arguments: they are not part of its key,
            // so the middle sub-batches of a long stream share one list, and so do different buffers of one geometry)
            std::vector<BlockDftTables::SegKey> key = segs;
            if (!multi) key[0].pcm_off = key[0].out_row0 = key[0].fbeg = 0;
            std::vector<size_t> slot_data;   // runs over a staged buffer: their slots by content (runs of one buffer share one slot list: taken once)
            {
                const Slot* seen = nullptr;
                for (const Run& r : L) {
                    const StreamIn& S = st[r.stream];
                    if (!S.slots || S.slots == seen) continue;
                    seen = S.slots;
                    slot_data.push_back(S.n_slots);
                    for (size_t i = 0; i < S.n_slots; ++i) {
                        slot_data.push_back(S.slots[i].vframe0);
                        slot_data.push_back(S.slots[i].n_frames);
                        slot_data.push_back(S.slots[i].out_row0);
                    }
                }
            }
            // kind 0: the tiles of a power-of-two hop (GEMM + tree); 1 / 2: the remainder / whole-block tiles of a general hop
            auto get_list = [&](int kind, BlockDftTables::TileList*& tl) -> pvq_status {
            tl = nullptr;
            for (auto& c : t->tile_lists)
                if (c.bm == fused_bm && c.wide == wide_mode && c.multi == multi && c.kind == kind && c.key == key && c.slot_data == slot_data) tl = &c;
            if (!tl) {
                tl = &t->tile_lists[t->tile_list_next];
                t->tile_list_next = (t->tile_list_next + 1) & 7;
                auto grp = [](const int4& e) { return e.x & 255; };
                auto is_wide = [](const int4& e) { return ((e.x >> 8) & 1) != 0; };
                auto seg_of = [](const int4& e) { return (int)((unsigned)e.x >> 16); };
                auto inside_of = [&](int g, int seg, int f0) {   // the kernel's own test
                    const BlockGroup& G = t->groups[g];
                    if (kind == 0) {
                        const long long tile_lo = segs[seg].base + G.s_rel + (long long)f0 * (long long)hop, tile_hi = tile_lo + (long long)fused_bm * (long long)hop;
                        return tile_lo >= 0 && tile_hi * 4ll <= (long long)segs[seg].pcm_bytes;
                    }
                    const long long depth = kind == 1 ? G.rem : (long long)hop;
                    const long long tile_lo = segs[seg].base + G.s_rel + (long long)(f0 + (kind == 1 ? G.nq : 0)) * (long long)hop;
                    const long long tile_hi = tile_lo + (long long)(fused_bm - 1) * (long long)hop + depth;
                    return tile_lo >= 0 && tile_hi * 4ll <= (long long)segs[seg].pcm_bytes;
                };
                std::vector<std::vector<int4>> q(8);
                for (size_t u = 0; u < segs.size(); ++u)
                    for (int g = 0; g < t->n_groups; ++g) {
                        const BlockGroup& G = t->groups[g];
                        if ((kind == 1 && G.rem == 0) || (kind == 2 && G.nq == 0)) continue;
                        const int S = kind == 0 ? fused_bm - G.nb_f + 1 : kind == 1 ? fused_bm : fused_bm - (G.nq > 1 ? G.nq - 1 : 0);
                        const int rows_g = kind == 0 ? segs[u].nf + G.nb - G.nb_f : segs[u].nf;
                        const bool half_last = G.n_cols - (G.n_tiles - 1) * CB_C <= 16;
                        for (int f0 = 0; f0 < rows_g; f0 += S)
                            for (int ntl = 0; ntl < G.n_tiles; ++ntl) {
                                // two neighbouring column tiles as one WIDE entry (.x bit 8; fp32 kernel, 256-row tiles)
                                // (a last tile of at most 16 columns keeps its own entry and its half-depth loop)
                                const bool pair = wide_mode && inside_of(g, (int)u, f0) && ntl + 1 < G.n_tiles && (pair_half_env || !(half_last && ntl + 1 == G.n_tiles - 1));
                                const int stripe = (segs[u].x_tile0 * 64 + f0) / FS;   // position in the launch's frame order
                                q[stripe & 7].push_back(make_int4(g | (pair ? 256 : 0) | (int)((unsigned)u << 16), ntl, f0, stripe));
                                if (pair) ++ntl;
                            }
                    }
                size_t Lq = 0;
                // what a tile costs its workgroup, roughly in us: K loop (half for a last tile of at most 16 columns) + tree levels
                auto tile_cost = [&](const int4& e) {
                    const BlockGroup& G = t->groups[grp(e)];
                    const bool half = e.y == G.n_tiles - 1 && G.n_cols - e.y * CB_C <= 16;
                    if (kind != 0) {   // a general hop's tiles: the K loop's depth decides
                        const int depth = kind == 1 ? G.rem : (int)hop;
                        return (inside_of(grp(e), seg_of(e), e.z) ? (half ? 1 : 2) : 4) * (depth / 32) + 8 + (kind == 2 ? G.nq : 0);
                    }
                    if (!inside_of(grp(e), seg_of(e), e.z)) return 2 * 16 + G.levels_f;   // the range-checked loop: dword loads
                    return ((half ? 8 : 16) + G.levels_f) * (is_wide(e) ? 2 : 1);
                };
                for (auto& v : q)   // by stripe; (segment, group, row tile, column tile) order kept
                    std::stable_sort(v.begin(), v.end(), [&](const int4& x, const int4& y) {
                        if (x.w != y.w) return x.w < y.w;
                        return order_env ? is_wide(x) > is_wide(y) : false;   // wide tiles of a stripe before its narrow ones (see below)
                    });
                // Even queues: stripes are dealt round robin, but the stream's first and last stripe carry extra tiles (the range-checked
                // ones at the ends, which do not pair up, and the long windows' partial-sum rows past the last frame) — queue 0 ran 28 us
                // longer than the rest of a 290 us launch.  The heaviest queue hands entries of its last stripe to the lightest until they
                // differ by less than a tile (those read their PCM rows through another XCD's L2: a few dozen tiles per launch).
                if (balance_env) {
                    long long cost[8];
                    for (int x = 0; x < 8; ++x) {
                        cost[x] = 0;
                        for (const int4& e : q[x]) cost[x] += tile_cost(e);
                    }
                    for (int it = 0; it < 4096; ++it) {
                        int h = 0, l = 0;
                        for (int x = 1; x < 8; ++x) {
                            if (cost[x] > cost[h]) h = x;
                            if (cost[x] < cost[l]) l = x;
                        }
                        if (q[h].empty()) break;
                        int4 e = q[h].back();
                        const int c = tile_cost(e);
                        if (cost[h] - cost[l] <= c) break;
                        q[h].pop_back();
                        if (!q[l].empty()) e.w = q[l].back().w;   // it joins the receiving queue's last stripe
                        q[l].push_back(e);
                        cost[h] -= c;
                        cost[l] += c;
                    }
                }
                for (auto& v : q) {
                    // the queue's last stripe: long tiles first, so that what is still running when the queues run dry is short
                    if (!v.empty()) {
                        const int last = v.back().w;
                        auto first_of_last = std::find_if(v.begin(), v.end(), [&](const int4& e) { return e.w == last; });
                        std::stable_sort(first_of_last, v.end(), [&](const int4& x, const int4& y) { return tile_cost(x) > tile_cost(y); });
                        if (wide_mode == 1 || wide_mode == 3) {
                            // ... and the queue's last entries narrow again (two per workgroup slot of the XCD; 3: the whole stripe): what is
                            // still running when the queues run dry sets the launch's tail
                            const size_t n_tail = wide_mode == 3 ? v.size() : (size_t)tail_env;
                            const size_t lo = first_of_last - v.begin();
                            std::vector<int4> tail;
                            while (v.size() > lo && tail.size() < n_tail) {
                                const int4 e = v.back();
                                v.pop_back();
                                if (is_wide(e)) {
                                    tail.push_back(make_int4(e.x & ~256, e.y, e.z, e.w));
                                    tail.push_back(make_int4(e.x & ~256, e.y + 1, e.z, e.w));
                                } else
                                    tail.push_back(e);
                            }
                            std::stable_sort(tail.begin(), tail.end(), [&](const int4& x, const int4& y) { return tile_cost(x) > tile_cost(y); });
                            v.insert(v.end(), tail.begin(), tail.end());
                        }
                    }
                    Lq = std::max(Lq, v.size());
                }
                tl->eff_tiles = 0.0;
                tl->eff_flop = 0.0;
                for (auto& v : q)
                    for (const int4& e : v) {
                        const BlockGroup& G = t->groups[grp(e)];
                        const bool half = e.y == G.n_tiles - 1 && G.n_cols - e.y * CB_C <= 16;
                        const double et = is_wide(e) ? 2.0 : (half ? 0.5 : 1.0);   // (the few range-checked tiles run the full loop: counted as half all the same)
                        tl->eff_tiles += et;
                        tl->eff_flop += et * fused_bm * (2 * CB_C) * ((kind == 1 ? (double)G.rem : (double)hop) / 2) * 2.0;   // (mirrored fp32 form: half depth)
                    }
                std::vector<int4> list(8 * std::max<size_t>(Lq, 1), make_int4(0, 0, 0x3FFFFFFF, 0));   // padding entries: past every group's rows
                for (int x = 0; x < 8; ++x) {
                    for (size_t i = 0; i < q[x].size(); ++i) {
                        list[i * 8 + x] = q[x][i];
                        list[i * 8 + x].w = (int)(i * 8 + x);
                    }
                }
                // the segment table and the X-tile map travel with the list (one allocation: list | segments | map)
                std::vector<SegDev> hsegs(segs.size());
                std::vector<XTile> hmap(xt_n);
                for (size_t u = 0; u < segs.size(); ++u) {
                    hsegs[u] = SegDev{segs[u].pcm_off, segs[u].base, segs[u].pcm_bytes, segs[u].nf, segs[u].x_tile0, segs[u].y_tile0};
                    const int tiles = (segs[u].nf + 63) / 64;
                    const StreamIn& S = st[L[u].stream];
                    for (int i = 0; i < tiles; ++i) {
                        XTile xt{segs[u].out_row0 + 64ll * i * segs[u].row_step, std::min(64, segs[u].nf - 64 * i) | (segs[u].row_step << 8), segs[u].y_tile0 + i};
                        if (S.slots) {   // frame t of the run = frame grid_i + row_step * t of the staged buffer: the slot it falls into names its rows
                            const size_t rs = S.row_step, t0 = L[u].fbeg + 64 * (size_t)i, v0 = S.grid_i + rs * t0;
                            const Slot* lo = S.slots;   // the last slot that starts at or before v0 (slots ascend; they start on multiples of 64 row_step frames)
                            size_t n = S.n_slots;
                            while (n > 1) {
                                const size_t h = n / 2;
                                if (lo[h].vframe0 <= v0) { lo += h; n -= h; } else n = h;
                            }
                            long long live = 0;
                            if (S.n_slots && lo->vframe0 <= v0 && v0 < lo->vframe0 + lo->n_frames)
                                live = std::min<long long>((long long)((lo->vframe0 + lo->n_frames - v0 + rs - 1) / rs), std::min(64, segs[u].nf - 64 * i));
                            xt.out_row0 = S.n_slots ? (long long)(lo->out_row0 + (v0 - std::min(v0, lo->vframe0))) : 0;
                            xt.live_step = (int)live | ((int)rs << 8);
                        }
                        hmap[segs[u].x_tile0 + i] = xt;
                    }
                }
                const size_t b_list = list.size() * sizeof(int4), b_segs = (hsegs.size() * sizeof(SegDev) + 15) / 16 * 16, b_map = hmap.size() * sizeof(XTile);
                if (tl->cap < b_list + b_segs + b_map) {
                    if (tl->d) PVQ_HIP(hipFree(tl->d));
                    tl->d = nullptr; tl->cap = 0;
                    PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&tl->d), b_list + b_segs + b_map));
                    tl->cap = b_list + b_segs + b_map;
                }
                PVQ_HIP(hipStreamSynchronize(stream));   // an earlier launch may still read this slot
                char* dbase = reinterpret_cast<char*>(tl->d);
                PVQ_HIP(hipMemcpy(dbase, list.data(), b_list, hipMemcpyHostToDevice));
                PVQ_HIP(hipMemcpy(dbase + b_list, hsegs.data(), hsegs.size() * sizeof(SegDev), hipMemcpyHostToDevice));
                PVQ_HIP(hipMemcpy(dbase + b_list + b_segs, hmap.data(), b_map, hipMemcpyHostToDevice));
                tl->d_segs = reinterpret_cast<const SegDev*>(dbase + b_list);
                tl->d_xmap = reinterpret_cast<const XTile*>(dbase + b_list + b_segs);
                tl->key = key;
                tl->slot_data = slot_data;
                tl->bm = fused_bm;
                tl->wide = wide_mode;
                tl->multi = multi;
                tl->kind = kind;
                tl->blocks = (int)list.size();
            }
            return PVQ_OK;
            };   // get_list
            BlockDftTables::TileList* tl = nullptr;
            BlockDftTables::TileList* tl_r = nullptr;
            if (t->general) {
                pvq_status ls = get_list(1, tl_r);
                if (ls != PVQ_OK) return ls;
                ls = get_list(2, tl);
                if (ls != PVQ_OK) return ls;
                // (the second lookup may have evicted the first — the slots are handed out round robin — look it up again)
                ls = get_list(1, tl_r);
                if (ls != PVQ_OK) return ls;
            } else {
                pvq_status ls = get_list(0, tl);
                if (ls != PVQ_OK) return ls;
            }
            fa.tile_list = tl->d;
            d_segs = multi ? tl->d_segs : nullptr;
            d_xmap = multi ? tl->d_xmap : nullptr;
            fa.segs = d_segs;
            const int off = tl->blocks;   // list entries = workgroups of the non-persistent forms = rows of the stamp dump
            fa.groups = t->d_groups;
            for (int g = 0; g < 8; ++g) fa.gv[g] = t->groups[std::min(g, t->n_groups - 1)];
            fa.comb_tw = t->d_comb_tw;
            fa.Et = t->d_Et;
            fa.E16 = t->d_E16;
            static const char* stamps_env = dev_knob_str("PVQ_STAMPS");   // dump per-tile phase stamps of the first launch
            static bool stamps_done = false;
            static int stamps_skip = dev_knob("PVQ_STAMPS_SKIP", 0);         // ... of launch n + 1 (a warm one)
            const bool do_stamps = stamps_env && !stamps_done && stamps_skip-- <= 0;
            fa.stamps = nullptr;
            if (do_stamps) PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&fa.stamps), (size_t)off * 12 * 8 + 8));   // rows of 8, then rows of 4 (PVQ_END_STAMPS)
            if (do_stamps) PVQ_HIP(hipMemset(fa.stamps, 0, (size_t)off * 12 * 8 + 8));
            // flop the GEMM's matrix instructions issue in this launch: tiles x rows x 64 real columns x depth x 2
            // (depth hop / 2 in the mirrored fp32 form, hop in the split-bf16 form, where it counts fp32-equivalent products)
            if (!use_bf) eff_tiles = tl->eff_tiles;   // (what the list's entries issue: a wide entry two whole tiles, a lone half tile half a tile)
            last_gemm_flop_ = eff_tiles * fused_bm * (2 * CB_C) * (use_bf ? (double)hop : (double)hop / 2) * 2.0;
            fa.E16R = t->d_E16R;
            fa.gen_tw = t->d_gen_tw;
            fa.gen_kind = 0;
            fa.clk = nullptr;
            if (profiling_ && !use_bf) {
                const size_t need = ((size_t)off / 64 + 1) * 4 * sizeof(unsigned long long);
                if (t->clk_cap < need) {
                    if (t->d_clk) PVQ_HIP(hipFree(t->d_clk));
                    t->d_clk = nullptr; t->clk_cap = 0;
                    PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_clk), need));
                    PVQ_HIP(hipMemset(t->d_clk, 0, need));   // once: every sampled workgroup rewrites its slot at every launch (a fill per launch cost the stream 3 us)
                    t->clk_cap = need;
                }
                t->clk_n = off / 64 + 1;
                fa.clk = t->d_clk;
            }
            slot_begin(SLOT_BLOCKDFT_GEMM, stream);
            if (t->general) {
                // the remainder tiles first (their results wait in Y), then the whole-block tiles; flop: both launches' K loops
                double flop = 0.0;
                for (int kind = 1; kind <= 2; ++kind) {
                    const BlockDftTables::TileList* L = kind == 1 ? tl_r : tl;
                    if (L->blocks == 0 || L->eff_tiles == 0.0) continue;
                    GemmTreeArgs ga = fa;
                    ga.tile_list = L->d;
                    ga.segs = multi ? L->d_segs : nullptr;
                    ga.gen_kind = kind;
                    ga.stamps = nullptr;
                    ga.clk = nullptr;
                    if (fused_bm == 256)
                        hipLaunchKernelGGL(blockdft_gemm_gen<256>, dim3(L->blocks), dim3(512), 0, stream, ga);
                    else
                        hipLaunchKernelGGL(blockdft_gemm_gen<128>, dim3(L->blocks), dim3(256), 0, stream, ga);
                    flop += L->eff_flop;
                }
                last_gemm_flop_ = flop;
                d_xmap = multi ? tl->d_xmap : nullptr;
            } else if (use_bf)
                hipLaunchKernelGGL(blockdft_gemm_tree_bf16x3<256>, dim3(off), dim3(512), 0, stream, fa);
            else if (tree3)
                hipLaunchKernelGGL(blockdft_gemm_tree3<256>, dim3(off), dim3(512), 0, stream, fa);
            else if (fused_bm == 256 && fa.K == 256 && dev_knob("PVQ_KFIX", 1))   // the instantiations that know the hop: 2 % fewer cycles (its strides and trip counts fold)
                hipLaunchKernelGGL((blockdft_gemm_tree<256, 256>), dim3(off), dim3(512), dyn_lds_env, stream, fa);
            else if (fused_bm == 256)
                hipLaunchKernelGGL(blockdft_gemm_tree<256>, dim3(off), dim3(512), dyn_lds_env, stream, fa);
            else
                hipLaunchKernelGGL(blockdft_gemm_tree<128>, dim3(off), dim3(256), 0, stream, fa);
            slot_end(SLOT_BLOCKDFT_GEMM, stream);
            if (do_stamps) {
                stamps_done = true;
                std::vector<unsigned long long> h((size_t)off * 12);
                PVQ_HIP(hipStreamSynchronize(stream));
                PVQ_HIP(hipMemcpy(h.data(), fa.stamps, h.size() * 8, hipMemcpyDeviceToHost));
                PVQ_HIP(hipFree(fa.stamps));
                if (FILE* fp = fopen(stamps_env, "wb")) {
                    fwrite(h.data(), 8, h.size(), fp);
                    fclose(fp);
                }
            }
            if (t->nb_max > 64) {   // the last one or two tree levels of the long windows
                slot_begin(SLOT_BLOCKDFT_COMBINE, stream);
                for (int g = 0; g < t->n_groups; ++g) {
                    const BlockGroup& G = t->groups[g];
                    if (G.nb <= G.nb_f) continue;
                    FinishArgs fin;
                    fin.Y = t->d_Y;
                    fin.X = X;
                    fin.xcp = xcp;
                    fin.n_frames = (int)nf;
                    fin.col0 = G.tile0 * CB_C;
                    fin.n_cols = G.n_tiles * CB_C;
                    fin.n_real = G.n_cols;
                    fin.levels_f = G.levels_f;
                    fin.levels = G.levels;
                    fin.tw = t->d_comb_tw + G.tw_off;
                    fin.xmap = d_xmap;
                    hipLaunchKernelGGL(blockdft_tree_finish, dim3((unsigned)((nf + 63) / 64), (unsigned)((fin.n_real + 3) / 4)), dim3(256), 0,
                                       stream, fin);
                }
                slot_end(SLOT_BLOCKDFT_COMBINE, stream);
            }
        } else {
            const int n_rows = (int)(nf + t->nb_max - 1);
            GemmArgs ga;
            ga.pcm_base = pcm_base;
            ga.pcm_bytes = pcm_bytes;
            ga.E = t->d_E;
            ga.ld = ntot;
            ga.P = t->d_P;
            ga.n_rows = n_rows;
            ga.K = (int)hop;
            ga.tile_s = t->d_tile_s;
            ga.base = base;
            ga.n_col_tiles = t->n_tiles;
            ga.p_rows = (int)rows_cap;
            const int m_tiles8 = (((n_rows + 255) / 256) + 7) / 8 * 8;
            last_gemm_flop_ = (double)ga.n_col_tiles * ((n_rows + 255) / 256) * 256.0 * FT_BN * ((double)hop / 2) * 2.0;
            slot_begin(SLOT_BLOCKDFT_GEMM, stream);
            hipLaunchKernelGGL(blockdft_gemm_rows<256>, dim3(ga.n_col_tiles * m_tiles8), dim3(512), 0, stream, ga);
            slot_end(SLOT_BLOCKDFT_GEMM, stream);
            CombineArgs ca;
            ca.P = t->d_P;
            ca.p_rows = (int)rows_cap;
            ca.X = X;
            ca.xcp = xcp;
            ca.n_frames = (int)nf;
            ca.n_rows = n_rows;
            ca.tile_group = t->d_tile_group;
            ca.groups = t->d_groups;
            ca.comb_tw = t->d_comb_tw;
            slot_begin(SLOT_BLOCKDFT_COMBINE, stream);
            if (t->nb_max <= 64)
                hipLaunchKernelGGL((blockdft_combine<128, 16, 64>), dim3(t->n_tiles * 2, (unsigned)((nf + 127) / 128)), dim3(256), 0,
                                   stream, ca);
            else
                hipLaunchKernelGGL((blockdft_combine<CB_T, 16, 256>), dim3(t->n_tiles * 2, (unsigned)((nf + CB_T - 1) / CB_T)), dim3(256),
                                   0, stream, ca);
            slot_end(SLOT_BLOCKDFT_COMBINE, stream);
        }
        BandArgs da;
        da.X = reinterpret_cast<const float*>(X);
        da.xcp = xcp;
        da.n_frames = (int)nf;
        da.n_bins = nb;
        // 4 rows apart (the two lane halves of a C tile) land 16 banks apart; the 64-frame form has the stride compiled in
        // more than 256 bins (32-frame tiles): the smallest stride >= n_bins that is 4 mod 16, so that up to 596 bins still fit two workgroups per CU
        // 64-frame tiles while two 64-row tiles fit a CU: up to 256 bins (stride 260) or up to 304 (stride 308; fp32 8-bin form only)
        const bool wide308 = !gemm_split_bf16_ && t->n_bins_pad > 256 && nb <= BAND_LDB3 - 4;
        const bool wide = t->n_bins_pad <= 256 || wide308;
        da.ldb = wide308 ? BAND_LDB3 : wide ? BAND_LDB2 : ((nb + 11) / 16 * 16 + 4);
        da.blocks = t->d_band;
        da.B = t->d_band_B;
        da.B3 = t->d_band_B3;
        da.list = t->d_band_list;
        da.per_wave = t->band_per_wave;
        // one run: its rows follow each other from its first output row; several: the X-tile map names every tile's rows
        da.xmap = d_xmap;
        const size_t row_first = multi ? 0 : (size_t)segs[0].out_row0;
        da.out_db = d_out_db + row_first * nb;
        da.out_cplx = d_out_cplx ? reinterpret_cast<float2*>(d_out_cplx) + row_first * nb : nullptr;
        da.status = dev_->d_status;
        static const char* dstamps_env = dev_knob_str("PVQ_STAMPS_DOTS");   // dump per-workgroup phase stamps of the first launch
        static bool dstamps_done = false;
        const bool do_dstamps = dstamps_env && !dstamps_done;
        da.stamps = nullptr;
        const size_t n_wg = (nf + 63) / 32;   // upper bound of the grid
        if (do_dstamps) PVQ_HIP(hipMalloc(reinterpret_cast<void**>(&da.stamps), n_wg * 8 * 8));
        if (do_dstamps) PVQ_HIP(hipMemset(da.stamps, 0, n_wg * 8 * 8));
        slot_begin(SLOT_BLOCKDFT_DOTS, stream);
        const int mt = wide ? 2 : 1;
        static const int dots16_env = dev_knob("PVQ_DOTS_16BIN", 0);   // the 32x32x2 form
        static const int dots_f32_env = dev_knob("PVQ_DOTS_F32", 0);
        const bool dots_split = gemm_split_bf16_ && !dots_f32_env;   // the kernel product follows the GEMM arithmetic
        const size_t lds = sizeof(float) * 32 * mt * da.ldb;
        const dim3 grid((unsigned)((nf + 32 * mt - 1) / (32 * mt)));
        if (mt == 2) {
            if (dots_split) {
                da.list = t->d_band_list + (size_t)t->band_waves * t->band_per_wave;   // the 4-wave lists
                hipLaunchKernelGGL((blockdft_banddots_db_bf16x3<2, 4>), grid, dim3(256), lds, stream, da);
            } else if (dots16_env && !wide308) {
                hipLaunchKernelGGL((blockdft_banddots_db<2, 8>), grid, dim3(512), lds, stream, da);
            } else {   // 8-bin blocks, 16x16x4 MFMAs
                da.blocks = t->d_band8;
                da.list = t->d_band_list8;
                da.per_wave = t->band_per_wave8;
                da.B = t->d_band_B4;
                if (wide308)
                    hipLaunchKernelGGL((blockdft_banddots4c_db<8, BD8_NS, BAND_LDB3, 2>), grid, dim3(512), lds, stream, da);
                else
                    hipLaunchKernelGGL((blockdft_banddots4c_db<8, BD8_NS, BAND_LDB2, 2>), grid, dim3(512), lds, stream, da);
            }
        } else {
            if (dots_split) {
                da.list = t->d_band_list + (size_t)t->band_waves * t->band_per_wave;   // the 4-wave lists
                hipLaunchKernelGGL((blockdft_banddots_db_bf16x3<1, 4>), grid, dim3(256), lds, stream, da);
            } else if (dots16_env || nb > 1024 - 4) {
                hipLaunchKernelGGL((blockdft_banddots_db<1, 8>), grid, dim3(512), lds, stream, da);
            } else {
                // more than 304 bins (the reference's default 588, 360, 840): the 8-bin / 16x16x4 / no-swap form on HALF tiles (32 frames x all bins
                // per workgroup, 8 waves), its LDS row stride compiled in per class of bin counts — round 5; before, these geometries ran the
                // 16-bin 32x32x2 form (PVQ_DOTS_16BIN=1 in the developer library)
                da.blocks = t->d_band8;
                da.list = t->d_band_list8;
                da.per_wave = t->band_per_wave8;
                da.B = t->d_band_B4;
                const int ldb_c = nb <= 368 ? 372 : nb <= 592 ? 596 : nb <= 848 ? 852 : 1028;
                da.ldb = ldb_c;
                const size_t lds_c = sizeof(float) * 32 * ldb_c;
                auto launch_c = [&](auto kern) -> pvq_status {
                    if (lds_c > 64 * 1024) PVQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds_c, stream, da);
                    return PVQ_OK;
                };
                pvq_status lcs = ldb_c == 372 ? launch_c(blockdft_banddots4c_db<8, BD8_NS, 372, 1>) : ldb_c == 596 ? launch_c(blockdft_banddots4c_db<8, BD8_NS, 596, 1>)
                                 : ldb_c == 852 ? launch_c(blockdft_banddots4c_db<8, BD8_NS, 852, 1>) : launch_c(blockdft_banddots4c_db<8, BD8_NS, 1028, 1>);
                if (lcs != PVQ_OK) return lcs;
            }
        }
        slot_end(SLOT_BLOCKDFT_DOTS, stream);
        if (do_dstamps) {
            dstamps_done = true;
            std::vector<unsigned long long> h(n_wg * 8);
            PVQ_HIP(hipStreamSynchronize(stream));
            PVQ_HIP(hipMemcpy(h.data(), da.stamps, h.size() * 8, hipMemcpyDeviceToHost));
            PVQ_HIP(hipFree(da.stamps));
            if (FILE* fp = fopen(dstamps_env, "wb")) {
                fwrite(h.data(), 8, h.size(), fp);
                fclose(fp);
            }
        }
        last_frames_per_launch_ = (uint32_t)nf_launch;
    }
    if (pk) {
        slot_begin(SLOT_PEAKS, stream);
        pvq_status ps = launch_peaks_kernel(d_out_db, rows_total, *pk, stream);
        slot_end(SLOT_PEAKS, stream);
        if (ps != PVQ_OK) return ps;
    }
    PVQ_HIP(hipGetLastError());
    last_algo_ = PVQ_ALGO_BLOCKDFT;
    if (n_st == 1) last_frames_per_launch_ = (uint32_t)chunk;
    return PVQ_OK;
}

}  // namespace pvq
