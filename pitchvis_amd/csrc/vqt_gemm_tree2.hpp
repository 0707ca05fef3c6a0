// vqt_gemm_tree2.hpp — hop-DFT GEMM + combine tree, software-pipelined over the column tiles of a unit
// (included by vqt_blockdft.hip; the kernel-product stage blockdft_banddots8_db consumes its X unchanged).
//
// blockdft_gemm_tree gives every (group, column tile, row tile) its own workgroup: K loop, then tree, then store, two
// workgroups per CU so that one's tree / store can sit beside the other's K loop — measured 0.43 of the matrix peak, the pipe
// idle whenever both are outside their K loops.  Here a workgroup owns a (window group, 256-block row tile) UNIT and walks all
// its column tiles:
//
//     iteration i:   GEMM of column tile i (v_mfma_f32_16x16x4_f32, accumulators in registers)
//                 || tree + store of tile i - 1 (P tile in LDS; first tree levels P -> Q, last Q -> P, both out of place)
//
// One workgroup per CU (150 KB of LDS), 8 waves = 2 per SIMD.  An iteration is 8 slots of 32 MFMAs per wave (one k group of
// 16 mirrored sample pairs each); in every slot waves 0-3 run their step of the side work before their MFMAs and waves 4-7
// behind them, so one wave of a SIMD feeds the matrix pipe while its partner is in vector / LDS / store code.
// GEMM operands: lane (row = lane & 15, kq = lane >> 4) fetches 4 consecutive samples of its row and the 4 mirrored ones with
// one 16-byte load each (the four lanes of a row read one contiguous 64-byte run: 16 cache lines per instruction where the
// 32x32x2 form touches 64); the samples are the same for every column tile of the unit; four buffers, fetched three slots
// ahead.  Every slot issues the same vector-memory instructions (operand loads, a quarter of the next E slice every other
// slot, X stores whose dead lanes point past the buffer): no branch around them, so the compiler's count of operations in
// flight is exact and its waits fall on loads issued three slots earlier.
// Applies when hop == 256 and every window has <= 64 hop blocks; anything else takes blockdft_gemm_tree.
#pragma once

namespace pvq {

constexpr int T2_LDP = CB_C + 1;                 // 33: conflict-free both for the tree (lanes across columns) and the transposed store (lanes across rows)
constexpr int T2_P_BYTES = (256 + 15) * T2_LDP * 8;   // 71 544: 15 spare rows, so that the first tree levels read their halo without a range check (what they add only reaches incomplete rows)
constexpr int T2_Q_BYTES = 256 * T2_LDP * 8;     // 67 584
constexpr int T2_EQ_BYTES = 32 * 16 * 16;        // a quarter of an E slice: [k 32][n 16][cos lo, cos hi, -sin lo, -sin hi] = 8 192
constexpr int T2_LDS_BYTES = T2_P_BYTES + T2_Q_BYTES + 2 * T2_EQ_BYTES + FT_MAXL * CB_C * 8;   // 157 048 of the CU's 163 840

struct T2Args {
    const float* pcm_base;
    unsigned pcm_bytes;
    const float4* E16;        // [column tile][k < hop / 2][n < 16]: (cos c_n, cos c_{n+16}, -sin c_n, -sin c_{n+16}) of the centred hop DFT
    int K;                    // hop (256)
    int n_frames;
    long long base;           // index, relative to pcm_base, of the end of frame 0 of this launch
    const int4* units;        // (group, first frame, -, -) per workgroup, frame-stripe order
    const BlockGroup* groups;
    const float2* comb_tw;
    float2* X;                // frame-tile blocked: X[((frame / 64) * xcp + col) * 64 + frame % 64]
    unsigned x_bytes;         // bytes of X (buffer stores: a dead lane's offset lies beyond)
    int xcp;
    unsigned long long* clk;  // profiling only: every 16th workgroup stores (shader clock, 100 MHz clock) around one iteration, + per-pass stamps
};

typedef float f32x4t __attribute__((ext_vector_type(4)));

// rows j0 .. j0 + NO - 1 of column c after the first R tree levels, from rows j0 .. j0 + NO - 1 + (2^R - 1) of A, written to B;
// same operations in the same order as fused_tree_register_levels (a frame's bits must not depend on the kernel that produced them)
template <int R, int NO>
__device__ __forceinline__ void t2_tree_reg_levels(const float2 (*A)[T2_LDP], float2 (*B)[T2_LDP], const float2 (*tw)[CB_C], int c, int j0) {
    constexpr int H = (1 << R) - 1;
    float2 v[NO + H];
#pragma unroll
    for (int i = 0; i < NO + H; ++i) v[i] = A[j0 + i][c];   // rows 256 .. 270 exist (spare rows of the P tile)
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
    for (int i = 0; i < NO; ++i) B[j0 + i][c] = v[i];
}

template <bool VEC>   // VEC: the unit's samples all lie inside the stream (16-byte loads, plain byte offsets)
__device__ __forceinline__ void t2_unit(const T2Args& a, unsigned char* smem, const BlockGroup& G, int f0) {
    float2 (*Pt)[T2_LDP] = reinterpret_cast<float2 (*)[T2_LDP]>(smem);
    float2 (*Qt)[T2_LDP] = reinterpret_cast<float2 (*)[T2_LDP]>(smem + T2_P_BYTES);
    float4* El = reinterpret_cast<float4*>(smem + T2_P_BYTES + T2_Q_BYTES);              // ring of two E quarters: [2][32 * 16]
    float2 (*tw)[CB_C] = reinterpret_cast<float2 (*)[CB_C]>(smem + T2_P_BYTES + T2_Q_BYTES + 2 * T2_EQ_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first_half = wave < 4;            // waves w and w + 4 share a SIMD
    const bool stamp = a.clk != nullptr && (blockIdx.x & 15) == 0 && tid == 0;
    if (stamp) a.clk[(blockIdx.x >> 4) * 16 + 11] = __builtin_amdgcn_s_memrealtime();
    const int S = 257 - G.nb_f;                  // complete frames of the row tile
    const int levels = G.levels_f;
    const int n_ct = G.n_tiles;
    const int m16 = lane & 15, kq = lane >> 4;

    // ---- GEMM operands: rows wave * 32 + mt * 16 + m16, k group gq covers m = 16 gq + 4 kq + t
    const long long tile_lo = a.base + G.s_rel + (long long)f0 * a.K;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(a.pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)a.pcm_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pcm_base), 0, a.pcm_bytes, 0x00020000);
    long long jf0[2], jb0[2];                    // the lane's first front / first mirrored sample of k group 0, per row tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long row_lo = tile_lo + (long long)(wave * 32 + mt * 16 + m16) * a.K;
        jf0[mt] = row_lo + 4 * kq;
        jb0[mt] = row_lo + a.K - 4 - 4 * kq;
    }
    float fr[4][2][4], bk[4][2][4];              // [buffer][row tile][sample]
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
    f32x4t accR[2][2], accI[2][2];               // [row tile][column half]: real / imaginary parts of 16 rows x 16 columns
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
    auto e_load = [&](int ct, int piece) { e_stage = a.E16[((size_t)(G.tile0 + ct) * (a.K / 2)) * 16 + piece * 512 + tid]; };
    auto e_store = [&](int piece) { El[(piece & 1) * (32 * 16) + tid] = e_stage; };

    // ---- side work on the tile that sits in LDS
    const int tc = tid & 31, tj0 = (tid >> 5) * 16;
    const int R = levels < 4 ? levels : 4;
    auto tree_a = [&](int half) {                // first R <= 4 levels, P -> Q, eight of the thread's sixteen rows per call
        const int j0 = tj0 + 8 * half;
        switch (R) {   // workgroup-uniform
            case 0: {
#pragma unroll
                for (int i = 0; i < 8; ++i) Qt[j0 + i][tc] = Pt[j0 + i][tc];
                break;
            }
            case 1: t2_tree_reg_levels<1, 8>(Pt, Qt, tw, tc, j0); break;
            case 2: t2_tree_reg_levels<2, 8>(Pt, Qt, tw, tc, j0); break;
            case 3: t2_tree_reg_levels<3, 8>(Pt, Qt, tw, tc, j0); break;
            default: t2_tree_reg_levels<4, 8>(Pt, Qt, tw, tc, j0); break;
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
    const float2 (*Xt)[T2_LDP] = levels >= 5 ? Pt : Qt;   // where the tile's spectrum ends up
    // X store: lanes walk the frames of a column (512-byte runs in memory, conflict-free LDS reads); four of the thread's
    // sixteen columns per call.  Buffer stores: a lane without a live frame (or a call without a tile) points past the buffer.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(a.X, 0, (int)a.x_bytes, 0x00020000);
    const int sj = tid & 255, sc0 = tid >> 8;
    const bool s_live = sj < S && f0 + sj < a.n_frames;
    const unsigned s_row = ((unsigned)((f0 + sj) >> 6) * (unsigned)a.xcp * 64u + (unsigned)((f0 + sj) & 63)) * 8u;
    auto x_store = [&](int part, int nt, bool on) {
        const unsigned base = (on && s_live) ? s_row + (unsigned)(nt * CB_C) * 512u : 0x80000000u;
#pragma unroll
        for (int k = 4 * part; k < 4 * part + 4; ++k) {
            const int cc = sc0 + 2 * k;
            const float2 val = Xt[sj][cc];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val.x), xrs, base + (unsigned)cc * 512u, 0, 2);   // aux 2: streamed (slc), as the non-temporal stores of blockdft_gemm_tree
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val.y), xrs, base + (unsigned)cc * 512u + 4u, 0, 2);
        }
    };
    // the eight steps of the side work on column tile nt_side (global tile index; `on`: there is such a tile)
    auto side = [&](int sl, int nt_side, bool on) {
        switch (sl) {   // compile-time after unrolling
            case 0: if (on) tree_a(0); break;
            case 1: if (on) tree_a(1); break;
            case 2: if (on) tree_c(0); break;
            case 3: if (on) tree_c(1); break;
            default: x_store(sl - 4, nt_side, on); break;
        }
    };

    // ---- prologue: combine twiddles of the first tile, its first two E quarters, k groups 0 .. 2
    if (tid < 15 * T2_LDP) reinterpret_cast<float2*>(smem)[256 * T2_LDP + tid] = make_float2(0.0f, 0.0f);   // the spare rows
    if (tid < 192) {
        const int l = tid >> 5, c = tid & 31;
        if (l < levels) tw[l][c] = a.comb_tw[G.tw_off + l * (n_ct * CB_C) + 0 * CB_C + c];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        e_load(0, p);
        e_store(p);
    }
    load_group(0, 0);   // the same samples for every column tile of the unit
    load_group(1, 1);
    load_group(2, 2);
    __syncthreads();

    for (int ct = 0; ct <= n_ct; ++ct) {         // iteration ct: GEMM of tile ct beside the side work on tile ct - 1
        const bool do_gemm = ct < n_ct, do_side = ct > 0;
        if (stamp && ct == 1) {
            a.clk[(blockIdx.x >> 4) * 16 + 0] = __builtin_amdgcn_s_memtime();
            a.clk[(blockIdx.x >> 4) * 16 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        if (do_gemm) zero_acc();
        const int ct_e = ct + 1 < n_ct ? ct + 1 : n_ct - 1;   // the tile whose E quarters the second half fetches (clamped at the end: fetched, not used)
        const int ct_c = ct < n_ct ? ct : n_ct - 1;
        const int nt_side = G.tile0 + (do_side ? ct - 1 : 0);
        for (int hp = 0; hp < 2; ++hp) {          // four slots per pass of this loop: buffer indices stay compile-time
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int pr = 2 * hp + (s4 >> 1);   // E quarter / pass index 0..3
                if (stamp && ct == 1 && (s4 & 1) == 0) a.clk[(blockIdx.x >> 4) * 16 + 4 + pr] = __builtin_amdgcn_s_memrealtime();
                // the same vector-memory instructions in every slot, whether or not their data is used
                if ((s4 & 1) == 0) e_load(pr < 2 ? ct_c : ct_e, (pr + 2) & 3);   // the E quarter that takes this pass's ring slot once the pass is over
                load_group((s4 + 3) & 3, (4 * hp + s4 + 3) & 7);                  // three slots ahead (wraps into the next tile: same samples)
                if (hp == 0) {
                    if (first_half) side(s4, nt_side, do_side);
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_gemm) gemm_group(s4, s4);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!first_half) side(s4, nt_side, do_side);
                } else {
                    if (first_half) side(4 + s4, nt_side, do_side);
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_gemm) gemm_group(s4, 4 + s4);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!first_half) side(4 + s4, nt_side, do_side);
                }
                if (s4 & 1) {
                    __syncthreads();   // the pass's E quarter is free; tree steps hand rows between threads
                    e_store(pr);
                }
            }
        }
        if (stamp && ct == 1) {
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
        if (stamp && ct == 1) a.clk[(blockIdx.x >> 4) * 16 + 8] = __builtin_amdgcn_s_memrealtime();
    }
    if (stamp) {
        a.clk[(blockIdx.x >> 4) * 16 + 9] = __builtin_amdgcn_s_memrealtime();
        a.clk[(blockIdx.x >> 4) * 16 + 10] = (unsigned long long)n_ct;
    }
}

__global__ __launch_bounds__(512, 2) void blockdft_gemm_tree2(T2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t2_smem[];
    const int4 unit = a.units[blockIdx.x];
    const int f0 = unit.y;
    if (f0 >= a.n_frames) return;
    const BlockGroup G = a.groups[unit.x];
    const long long tile_lo = a.base + G.s_rel + (long long)f0 * a.K, tile_hi = tile_lo + 256ll * a.K;
    if (tile_lo >= 0 && tile_hi * 4ll <= (long long)a.pcm_bytes)
        t2_unit<true>(a, t2_smem, G, f0);
    else
        t2_unit<false>(a, t2_smem, G, f0);
}

}  // namespace pvq
