// vqt_fused2.hpp — the block-DFT path as ONE software-pipelined kernel per sub-batch (included by vqt_blockdft.hip).
//
// blockdft_gemm_tree + blockdft_banddots8_db write and re-read the intermediate spectrum X (5.4 KB per frame, 770 MB per
// 65 536 frames) and leave the matrix pipe idle while a workgroup is in its tree / store phase.  Here a workgroup owns a
// (window group, 256-block row tile) UNIT and walks ALL column tiles of it:
//
//     iteration i:   hop-DFT GEMM of column tile i            (v_mfma_f32_16x16x4_f32, accumulators in registers)
//                 || combine tree + kernel product of tile i-1 (P / X tile in LDS, never in memory)
//
// * One workgroup per CU (136 KB of LDS, up to 256 registers per lane), 8 waves = 2 per SIMD.  Inside every slot of an
//   iteration waves 0-3 run their slice of the tree / kernel-product work first and their 32 GEMM MFMAs second, waves 4-7
//   the other way round: at any time one wave of a SIMD feeds the matrix pipe while its partner is in vector / LDS code.
// * GEMM on 16x16x4 MFMAs: lane (row = lane & 15, kq = lane >> 4) fetches 4 consecutive samples of its row and the 4
//   mirrored ones with one 16-byte load each; the four lanes of a row read one contiguous 64-byte run (16 cache lines per
//   load instruction instead of 64 with the 32x32x2 form, whose address processing took as long as its MFMAs).
// * Kernel product (vqt.rs:889-910) C-stationary in registers: the 8-bin blocks of a window group are dealt to the 8 waves so
//   that a wave never has two blocks open at once (interval colouring over column tiles, done on the host); a wave keeps its
//   block's 16 x 16-frame accumulator tiles (64 registers) across column tiles, adds the columns of every tile as they come
//   out of the tree, and writes |x_vqt|^2 when its block closes.  Output: power rows pw[bin][frame] (64-byte runs), turned
//   into dB frames by power_rows_to_db (frame-wide max / floor / shift of power_to_db, vqt.rs:922-954).
// Applies when hop == 256 (8 MFMA slots per tile), every window has <= 64 hop blocks and every group colours with 8 waves;
// anything else takes the two-kernel path above.
#pragma once

namespace pvq {

constexpr int F2_LDP = 34;                       // P tile row stride (complex): 4 f + 2 kq banks apart -> conflict-free ds_read_b64 of the kernel-product A operands
constexpr int F2_P_BYTES = 256 * F2_LDP * 8;     // 69 632: the GEMM's P' tile, later the tree's last levels
constexpr int F2_Q_ROWS = 256;                   // (a window of 4 hop blocks leaves 253 complete rows after its two tree levels)
constexpr int F2_Q_BYTES = F2_Q_ROWS * F2_LDP * 8;   // 69 632: the tree's first levels go out of place, P -> Q (no registers held across a barrier)
constexpr int F2_EQ_BYTES = 32 * 16 * 16;        // a quarter of an E slice: [k 32][n 16][cos lo, cos hi, -sin lo, -sin hi] = 8 192
constexpr int F2_LDS_BYTES = F2_P_BYTES + F2_Q_BYTES + 2 * F2_EQ_BYTES + FT_MAXL * CB_C * 8;   // 157 184 of the CU's 163 840

struct F2Args {
    const float* pcm_base;
    unsigned pcm_bytes;
    const float4* E16;        // [column tile][k < hop / 2][n < 16]: (cos c_n, cos c_{n+16}, -sin c_n, -sin c_{n+16}) of the centred hop DFT
    int K;                    // hop (256)
    int n_frames;
    long long base;           // index, relative to pcm_base, of the end of frame 0 of this launch
    const int4* units;        // (group, first frame, -, -) per workgroup, frame-stripe order
    const BlockGroup* groups;
    const float2* comb_tw;
    const int4* segs;         // [column tile][wave]: x = first quad in B2, y = q0 | nq << 8 | flags << 16 | nrows << 24, z = bin0
    const float2* B2;         // [quad][lane]: (coefficient of Re X, coefficient of Im X) for lane (n = lane & 15, kq = lane >> 4)
    float* pw;                // [n_bins][pw_stride] |x_vqt|^2
    int pw_stride;
    float2* out_cplx;         // optional [n_frames][n_bins]
    int n_bins;
    unsigned long long* clk;  // profiling only: every 16th workgroup stores (shader clock, 100 MHz clock) around one iteration
    int exp;                  // developer knob PVQ_F2_EXP (timing experiments, wrong results): 1 no operand loads in the loop, 2 no side work, 4 no per-pass barrier, 8 no E refill
};

typedef float f32x4m __attribute__((ext_vector_type(4)));

// rows j0 .. j0 + NO - 1 of column c after the first R tree levels, from rows j0 .. j0 + NO - 1 + (2^R - 1) of A, written to B
// (rows below F2_Q_ROWS only); same operations in the same order as fused_tree_register_levels (a frame's bits must not
// depend on the kernel that produced them)
template <int R, int NO>
__device__ __forceinline__ void f2_tree_reg_levels(const float2 (*A)[F2_LDP], float2 (*B)[F2_LDP], const float2 (*tw)[CB_C], int c, int j0) {
    constexpr int H = (1 << R) - 1;
    float2 v[NO + H];
#pragma unroll
    for (int i = 0; i < NO + H; ++i) v[i] = (j0 + i < 256) ? A[j0 + i][c] : make_float2(0.0f, 0.0f);
    int len = NO + H;
#pragma unroll
    for (int l = 0; l < R; ++l) {
        const int st = 1 << l;
        const float2 w = tw[l][c];
        len -= st;
#pragma unroll
        for (int i = 0; i < NO + H; ++i)
            if (i < len) v[i] = tree_cmadd(v[i], w, v[i + st]);
    }
#pragma unroll
    for (int i = 0; i < NO; ++i)
        if (j0 + i < F2_Q_ROWS) B[j0 + i][c] = v[i];
}

template <bool VEC, bool CPLX>   // VEC: the unit's samples all lie inside the stream (16-byte loads, plain byte offsets); CPLX: complex output too
__device__ __forceinline__ void f2_unit(const F2Args& a, unsigned char* f2_smem, const BlockGroup& G, int f0) {
    float2 (*Pt)[F2_LDP] = reinterpret_cast<float2 (*)[F2_LDP]>(f2_smem);
    float2 (*Qt)[F2_LDP] = reinterpret_cast<float2 (*)[F2_LDP]>(f2_smem + F2_P_BYTES);
    float4* El = reinterpret_cast<float4*>(f2_smem + F2_P_BYTES + F2_Q_BYTES);          // ring of two E quarters: [2][32 * 16]
    float2 (*tw)[CB_C] = reinterpret_cast<float2 (*)[CB_C]>(f2_smem + F2_P_BYTES + F2_Q_BYTES + 2 * F2_EQ_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.clk != nullptr && (blockIdx.x & 15) == 0 && tid == 0) a.clk[(blockIdx.x >> 4) * 16 + 11] = __builtin_amdgcn_s_memrealtime();
    const bool first_half = wave < 4;
    const int S = 257 - G.nb_f;                  // complete frames of the row tile
    const int levels = G.levels_f;
    const int n_ct = G.n_tiles;
    const int m16 = lane & 15, kq = lane >> 4;

    // ---- GEMM operand addressing: rows wave * 32 + mt * 16 + m16, k group gq covers m = 16 gq + 4 kq + t
    const long long s = a.base + G.s_rel;
    const long long tile_lo = s + (long long)f0 * a.K;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(a.pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)a.pcm_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pcm_base), 0, a.pcm_bytes, 0x00020000);
    // sample index (relative to pcm_base) of the lane's first front / first mirrored sample of k group 0, per row tile; in the
    // VEC form they are non-negative and fit 30 bits, so byte offsets are plain unsigned arithmetic
    long long jf0[2], jb0[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long row_lo = tile_lo + (long long)(wave * 32 + mt * 16 + m16) * a.K;
        jf0[mt] = row_lo + 4 * kq;
        jb0[mt] = row_lo + a.K - 4 - 4 * kq;
    }
    float fr[2][2][4], bk[2][2][4];              // [buffer][row tile][sample]
    auto load_group = [&](int buf, int gq) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (VEC) {
                const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)((unsigned)jf0[mt] * 4u + 64u * (unsigned)gq), 0, 0);
                const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)((unsigned)jb0[mt] * 4u - 64u * (unsigned)gq), 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fr[buf][mt][t] = v[t];
                    bk[buf][mt][t] = w[t];
                }
            } else {   // units that touch the stream start / end: samples before the stream get an explicit out-of-range offset
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long long xf = jf0[mt] + 16 * gq + t, xb = jb0[mt] - 16 * gq + t;
                    fr[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xf >= 0 ? (unsigned)(xf * 4ll) : 0xFFFFFFFCu, 0, 0));
                    bk[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xb >= 0 ? (unsigned)(xb * 4ll) : 0xFFFFFFFCu, 0, 0));
                }
            }
        }
    };
    f32x4m accR[2][2], accI[2][2];               // [row tile][column half]: real / imaginary parts of 16 rows x 16 columns
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int np = 0; np < 2; ++np)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accR[mt][np][r] = 0.0f;
                    accI[mt][np][r] = 0.0f;
                }
    };
    // 32 MFMAs: k group gq (the E quarter gq / 2 sits in ring slot (gq / 2) & 1)
    auto gemm_group = [&](int buf, int gq) {
        const float4* e = El + ((gq >> 1) & 1) * (32 * 16) + (16 * (gq & 1) + 4 * kq) * 16 + m16;
        float4 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = e[t * 16];   // all four operand reads in flight before the first MFMA
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
    // E quarter `piece` of column tile ct: fetched into a register while the ring slot it will take is still in use, stored
    // behind the barrier that ends that use
    float4 e_stage;
    auto e_load = [&](int ct, int piece) {
        e_stage = a.E16[((size_t)(G.tile0 + ct) * (a.K / 2)) * 16 + piece * 512 + tid];
    };
    auto e_store = [&](int piece) { El[(piece & 1) * (32 * 16) + tid] = e_stage; };

    // ---- side work on the tile that sits in LDS
    const int tc = tid & 31, tj0 = (tid >> 5) * 16;
    const int R = levels < 4 ? levels : 4;
    // first R <= 4 levels, P -> Q, eight of the thread's sixteen rows per call (four at a time: fewer live registers)
    auto tree_a = [&](int half) {
#pragma unroll 1
        for (int h4 = 0; h4 < 2; ++h4) {
            const int j0 = tj0 + 8 * half + 4 * h4;
            switch (R) {   // workgroup-uniform
                case 0: {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (j0 + i < F2_Q_ROWS) Qt[j0 + i][tc] = Pt[j0 + i][tc];
                    break;
                }
                case 1: f2_tree_reg_levels<1, 4>(Pt, Qt, tw, tc, j0); break;
                case 2: f2_tree_reg_levels<2, 4>(Pt, Qt, tw, tc, j0); break;
                case 3: f2_tree_reg_levels<3, 4>(Pt, Qt, tw, tc, j0); break;
                default: f2_tree_reg_levels<4, 4>(Pt, Qt, tw, tc, j0); break;
            }
        }
    };
    // levels 4 (and 5), Q -> P, evaluated exactly as fused_tree_store does (two radix-2 levels where there are two); eight of
    // the thread's sixteen outputs per call
    auto tree_c = [&](int half) {
        if (levels == 6) {
            const float2 w1 = tw[4][tc], w2 = tw[5][tc];
#pragma unroll
            for (int q = 8 * half; q < 8 * half + 8; ++q) {
                const int j = (tid >> 5) + 16 * q;
                if (j < 256 - 15 - 48) {
                    const float2 t0 = tree_cmadd(Qt[j][tc], w1, Qt[j + 16][tc]);
                    const float2 t1 = tree_cmadd(Qt[j + 32][tc], w1, Qt[j + 48][tc]);
                    Pt[j][tc] = tree_cmadd(t0, w2, t1);
                }
            }
        } else if (levels == 5) {
            const float2 w1 = tw[4][tc];
#pragma unroll
            for (int q = 8 * half; q < 8 * half + 8; ++q) {
                const int j = (tid >> 5) + 16 * q;
                if (j < 256 - 15 - 16) Pt[j][tc] = tree_cmadd(Qt[j][tc], w1, Qt[j + 16][tc]);
            }
        }
    };
    const float2 (*Xt)[F2_LDP] = levels >= 5 ? Pt : Qt;   // where the tile's spectrum ends up
    // kernel product of this wave's block with the tile's columns
    f32x4m dacc[16];                             // 16 frame tiles x (8 bins x (re, im)): the block's partial sums over the column tiles
    int4 seg = make_int4(0, 0, 0, 0);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto seg_open = [&](int ct) {                // scalar loads: no vector-memory traffic
        const int4 sg = a.segs[(size_t)(G.tile0 + ct) * 8 + wave_s];
        seg.x = sg.x;
        seg.y = sg.y;
        seg.z = sg.z;
    };
    auto seg_clear = [&]() {
        if (((seg.y >> 8) & 0xff) != 0 && ((seg.y >> 16) & 1)) {
#pragma unroll
            for (int ft = 0; ft < 16; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) dacc[ft][r] = 0.0f;
        }
    };
    // the coefficients of the four quads a pass may walk are fetched at the top of the pass, unconditionally (clamped), so that
    // every pass issues the same vector-memory instructions and the compiler's count of loads in flight stays exact
    float2 bq[4];
    auto b_load = [&](int pr) {
        const int nq = (seg.y >> 8) & 0xff;
        const int qmax = nq > 0 ? nq - 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = min(4 * (pr & 1) + i, qmax);
            bq[i] = a.B2[(size_t)(seg.x + q) * 64 + lane];
        }
    };
    auto dots_quads = [&](int qa, int i0) {      // quads qa, qa + 1 of the segment (coefficients bq[i0], bq[i0 + 1])
        const int q0 = seg.y & 0xff, nq = (seg.y >> 8) & 0xff;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (qa + i >= nq) break;
            const float2 b = bq[i0 + i];
            const float2* xcol = &Xt[m16][4 * (q0 + qa + i) + kq];
#pragma unroll
            for (int f4 = 0; f4 < 16; f4 += 4) {
                float2 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = xcol[(f4 + u) * 16 * F2_LDP];
#pragma unroll
                for (int u = 0; u < 4; ++u) dacc[f4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].x, b.x, dacc[f4 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) dacc[f4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u].y, b.y, dacc[f4 + u], 0, 0, 0);
            }
        }
    };
    // The block's last column has been added: |x_vqt|^2 out.  Stores go through a buffer resource over the power rows:
    // a lane that has nothing to write (a padding row of the block, a frame past the tile's complete frames or past the
    // batch) gets an offset beyond the buffer and the hardware drops its store — no branch per store.
    auto dots_close = [&]() {
        const int nq = (seg.y >> 8) & 0xff;
        if (nq == 0 || !((seg.y >> 17) & 1)) return;
        const int nrows = (seg.y >> 24) & 0xff, bin0 = seg.z;
        const unsigned OOR = 0x80000000u;        // the rows hold far less than 2 GB
        int jmax = min(S, a.n_frames - f0) - 4 * kq;   // the lane's frames j = 16 ft + 4 kq + r are live while 16 ft + r < jmax
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.pw, 0, (int)((unsigned)a.n_bins * (unsigned)a.pw_stride * 4u), 0x00020000);
        const unsigned pbase = m16 < nrows ? ((unsigned)(bin0 + m16) * (unsigned)a.pw_stride + (unsigned)(f0 + 4 * kq)) * 4u : OOR;
#pragma unroll
        for (int ft = 0; ft < 16; ++ft) {
            asm volatile("" : "+v"(jmax));   // keeps the 64 frame-limit compares where they are used (hoisted, their masks cost 128 scalar registers)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float re = dacc[ft][r];
                const float im = PVQ_DPP(re, 0x128);   // row_ror:8: lane n <- lane n ^ 8 (the im column of the same bin)
                const unsigned off = (ft * 16 + r < jmax ? pbase : OOR) + (unsigned)(ft * 16 + r) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, re * re + im * im), prs, off, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (CPLX) {                              // optional complex coefficients (tests): same walk, 8-byte stores
            const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(a.out_cplx, 0, 0x7FFFFFF0, 0x00020000);
            const bool fits = (unsigned long long)a.n_frames * (unsigned long long)a.n_bins * 8ull < 0x7FFFFFF0ull;
            const unsigned row_b = (unsigned)a.n_bins * 8u;
            const unsigned cbase = (m16 < nrows && fits) ? ((unsigned)(f0 + 4 * kq) * (unsigned)a.n_bins + (unsigned)(bin0 + m16)) * 8u : OOR;
#pragma unroll
            for (int ft = 0; ft < 16; ++ft) {
                asm volatile("" : "+v"(jmax));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float re = dacc[ft][r];
                    const float im = PVQ_DPP(re, 0x128);
                    const unsigned off = ft * 16 + r < jmax ? cbase + (unsigned)(ft * 16 + r) * row_b : OOR;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, re), crs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, im), crs, off + 4u, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // the side work on the tile in LDS in eight steps, two per pair of GEMM slots
    auto side_even = [&](int pr) {
        switch (pr) {   // workgroup-uniform
            case 0: tree_a(0); break;
            case 1: tree_c(0); break;
            case 2: seg_clear(); dots_quads(0, 0); break;
            default: dots_quads(4, 0); break;
        }
    };
    auto side_odd = [&](int pr) {
        switch (pr) {
            case 0: tree_a(1); break;
            case 1: tree_c(1); break;
            case 2: dots_quads(2, 2); break;
            default: dots_quads(6, 2); break;
        }
    };

    // ---- prologue: combine twiddles of the first tile, its first two E quarters, k group 0
    if (tid < 192) {
        const int l = tid >> 5, c = tid & 31;
        if (l < levels) tw[l][c] = a.comb_tw[G.tw_off + l * (n_ct * CB_C) + 0 * CB_C + c];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        e_load(0, p);
        e_store(p);
    }
    load_group(0, 0);   // k group 0: the same samples for every column tile of the unit
    __syncthreads();

    for (int ct = 0; ct <= n_ct; ++ct) {         // iteration ct: GEMM of tile ct beside the side work on tile ct - 1
        const bool do_gemm = ct < n_ct, do_side = ct > 0;
        if (a.clk != nullptr && ct == 1 && (blockIdx.x & 15) == 0 && tid == 0) {
            a.clk[(blockIdx.x >> 4) * 16 + 0] = __builtin_amdgcn_s_memtime();
            a.clk[(blockIdx.x >> 4) * 16 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        if (do_gemm) zero_acc();
        if (do_side) seg_open(ct - 1);
        const int ct_e = ct + 1 < n_ct ? ct + 1 : n_ct - 1;   // the tile whose E quarters passes 2 and 3 fetch (clamped: fetched, not used, at the end)
        const int ct_c = ct < n_ct ? ct : n_ct - 1;
        for (int pr = 0; pr < 4; ++pr) {          // two GEMM slots (k groups 2 pr, 2 pr + 1) per pass: buffer indices stay static
            if (a.clk != nullptr && ct == 1 && (blockIdx.x & 15) == 0 && tid == 0) a.clk[(blockIdx.x >> 4) * 16 + 4 + pr] = __builtin_amdgcn_s_memrealtime();
            // Every pass issues the same vector-memory instructions whether or not their data is used (the drain iteration,
            // the last tile): with no branch around them the compiler's count of loads in flight is exact and its waits fall
            // on loads issued a slot earlier, not on the ones just issued.
            // the E quarter that takes this pass's ring slot once the pass is over: quarter pr + 2 of this tile, or quarter
            // pr - 2 of the next one
            if (!(a.exp & 8)) e_load(pr < 2 ? ct_c : ct_e, (pr + 2) & 3);
            if (!(a.exp & 1)) load_group(1, 2 * pr + 1);
            b_load(pr);
            // waves w and w + 4 share a SIMD: the first four run their step of the side work before their 32 MFMAs, the other
            // four behind them, so that one wave of a SIMD feeds the matrix pipe while its partner is in vector / LDS code
            const bool side_on = do_side && !(a.exp & 2);
            if (side_on && first_half) side_even(pr);
            __builtin_amdgcn_sched_barrier(0);
            if (do_gemm) gemm_group(0, 2 * pr);
            __builtin_amdgcn_sched_barrier(0);
            if (side_on && !first_half) side_even(pr);
            if (!(a.exp & 1)) load_group(0, pr < 3 ? 2 * pr + 2 : 0);   // after the last pass: k group 0 again, for the next tile
            if (side_on && first_half) side_odd(pr);
            __builtin_amdgcn_sched_barrier(0);
            if (do_gemm) gemm_group(1, 2 * pr + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (side_on && !first_half) side_odd(pr);
            if (!(a.exp & 4)) __syncthreads();   // the pass's E quarter is free; tree steps hand rows between threads
            if (!(a.exp & 8)) e_store(pr);
        }
        if (do_side) dots_close();   // stores: behind the passes, so that they do not sit between a pass's loads and their use
        if (a.clk != nullptr && ct == 1 && (blockIdx.x & 15) == 0 && tid == 0) {
            a.clk[(blockIdx.x >> 4) * 16 + 2] = __builtin_amdgcn_s_memtime();
            a.clk[(blockIdx.x >> 4) * 16 + 3] = __builtin_amdgcn_s_memrealtime();
        }
        // (the barrier that ends the last pass: tile ct - 1 is done with, its place is taken by tile ct)
        if (do_gemm) {
            // C layout of the 16x16 MFMA: column = lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int np = 0; np < 2; ++np)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Pt[wave * 32 + mt * 16 + 4 * kq + r][np * 16 + m16] = make_float2(accR[mt][np][r], accI[mt][np][r]);
            if (tid < 192) {                      // the tile's combine twiddles
                const int l = tid >> 5, c = tid & 31;
                if (l < levels) tw[l][c] = a.comb_tw[G.tw_off + l * (n_ct * CB_C) + ct * CB_C + c];
            }
        }
        __syncthreads();
        if (a.clk != nullptr && ct == 1 && (blockIdx.x & 15) == 0 && tid == 0) a.clk[(blockIdx.x >> 4) * 16 + 8] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.clk != nullptr && (blockIdx.x & 15) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 4) * 16 + 9] = __builtin_amdgcn_s_memrealtime();
        a.clk[(blockIdx.x >> 4) * 16 + 10] = (unsigned long long)n_ct;
    }
}

template <bool CPLX>
__global__ __launch_bounds__(512, 2) void blockdft_fused2(F2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f2_smem[];
    const int4 unit = a.units[blockIdx.x];
    const int f0 = unit.y;
    if (f0 >= a.n_frames) return;
    const BlockGroup G = a.groups[unit.x];
    const long long tile_lo = a.base + G.s_rel + (long long)f0 * a.K, tile_hi = tile_lo + 256ll * a.K;
    if (tile_lo >= 0 && tile_hi * 4ll <= (long long)a.pcm_bytes)
        f2_unit<true, CPLX>(a, f2_smem, G, f0);
    else
        f2_unit<false, CPLX>(a, f2_smem, G, f0);
}

// ------------------------------------------------------------------------------------------------
// power rows -> dB frames: pw[bin][frame] (what blockdft_fused2 leaves) -> out_db[frame][bin] with the frame-wide
// max / floor / shift of power_to_db (vqt.rs:922-954).  A workgroup transposes 32 frames x all bins through LDS
// (128-byte runs in, whole rows out) and runs the dB phase of the two-kernel path on the tile.
// ------------------------------------------------------------------------------------------------
struct PowArgs {
    const float* pw;
    int pw_stride;
    BandArgs b;   // n_frames, n_bins, ldb, out_db, status (the fields band_finish reads)
};
__global__ __launch_bounds__(256) void power_rows_to_db(PowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pdb[];   // [32][ldb]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * 32;
    const int fr = lane & 31, half = lane >> 5;
    const bool live = f0 + fr < a.b.n_frames;
    for (int bin = wave * 2 + half; bin < a.b.n_bins; bin += 8)
        pdb[fr * a.b.ldb + bin] = live ? a.pw[(size_t)bin * a.pw_stride + f0 + fr] : 1.0f;
    __syncthreads();
    band_finish<1, 4, 0>(pdb, a.b, f0, wave, lane);
}

}  // namespace pvq
