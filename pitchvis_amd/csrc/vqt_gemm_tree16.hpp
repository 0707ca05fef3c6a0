// vqt_gemm_tree16.hpp — hop-DFT GEMM + combine tree on 16-column tiles (included by vqt_blockdft.hip).
//
// blockdft_gemm_tree works on 256-block x 32-column tiles: a 70 KB P tile, two workgroups per CU, and the matrix pipe idles
// whenever both are outside their K loops (0.46-0.50 busy).  The same kernel on 256 x 16 tiles has a 37 KB P tile: THREE
// workgroups = 24 waves per CU at <= 80 registers, each with half the matrix work per tile (a K loop of 3.9 us of MFMAs against
// ~6 us of skew / tree / store), so two of the three are normally inside their K loops.  Side effects: a group's last tile
// wastes at most 15 padded columns instead of 31 (624 columns computed instead of 672 at 48 kHz / 252 bins), the PCM rows
// are fetched by twice as many workgroups (L2 -> L1 traffic doubles; HBM traffic does not: the stripe's rows stay L2-resident).
// Everything a frame's bits depend on is unchanged: same mirrored sums / differences, same k order, same MFMA, same tree_cmadd
// levels in the same order — a column's values do not depend on which tile width computed it (tests compare the two forms
// bit for bit).
//
// K loop: a wave owns 32 block rows x 16 complex columns = two 16-row tiles x (Re, Im) = 4 accumulators of
// v_mfma_f32_16x16x4_f32; operands exactly as fused_f32_kloop16 (lane (row = lane & 15, kq = lane >> 4): 4 consecutive samples
// and the 4 mirrored ones per 16-byte load, double-buffered one k group ahead); B operand of lane (n, kq) for MFMA t of k group
// g = 2 G + h: (cos, -sin) of row 32 G + 8 kq + 4 h + t, column n — the four t of a lane stored side by side (E16h), two ds_read_b128 per group.
// Tree: a thread owns 8 consecutive rows of a column (+ 7 halo rows): the first three levels in registers, the rest through LDS
// two at a time, as in fused_tree_store.
#pragma once

namespace pvq {

constexpr int H16_C = 16;                 // complex columns per tile
constexpr int H16_LDP = H16_C + 1;        // P tile row stride (complex): odd, so that the transposed store (lanes across rows) is conflict-free
constexpr int H16_SPARE = 7;              // spare zeroed rows: the register levels read their halo without a range check

template <int BM> struct H16Geom {
    static constexpr int P_FLOATS = (BM + H16_SPARE) * H16_LDP * 2;
    static constexpr int E_FLOATS = FR_KC * H16_C * 2;   // the E slice: 128 rows x 16 columns x (cos, -sin) = 16 KB
    static constexpr int SMEM_FLOATS = P_FLOATS > E_FLOATS ? P_FLOATS : E_FLOATS;
};

// which (group, 16-column tile, row tile) a workgroup owns
struct H16Tile {
    BlockGroup G;
    int S;        // complete frames per row tile
    int t16;      // 16-column tile within the group
    int col0;     // first X column of the tile
    int f0;       // first frame == first block row
    int nfr;      // rows this group produces
};

template <int BM>
__device__ __forceinline__ H16Tile h16_tile(const GemmTreeArgs& a) {
    H16Tile t;
    const int4 e = a.tile_list[blockIdx.x];
    t.G = a.groups[e.x];
    t.S = BM - t.G.nb_f + 1;
    t.nfr = a.n_frames + t.G.nb - t.G.nb_f;
    t.t16 = e.y;
    t.f0 = e.z;
    t.col0 = t.G.tile0 * CB_C + t.t16 * H16_C;
    return t;
}

// the first R <= 3 tree levels (strides 1, 2, 4) in registers: a thread owns 8 consecutive rows of one column and reads them
// plus the 2^R - 1 rows above once (same operations in the same order as fused_tree_register_levels).  Thread -> (column,
// chunk): the two 16-lane groups of a half wave take chunks two apart (rows 16 apart: 16 * 17 = 272 = 16 mod 32 slots), so
// that their reads fall into disjoint LDS banks.
template <int R, int BM>
__device__ __forceinline__ void h16_tree_register_levels(float2 (*A)[H16_LDP], const float2 (*tw)[H16_C], int tid) {
    constexpr int H = (1 << R) - 1;
    const int c = tid & 15;
    const int q = (tid >> 4) & 3;                                   // 16-lane group of the wave: 0, 1, 2, 3 -> chunks 0, 2, 1, 3
    const int j0 = (((tid >> 6) << 2) + ((q & 1) << 1) + (q >> 1)) * 8;
    float2 v[8 + H];
#pragma unroll
    for (int i = 0; i < 8 + H; ++i) v[i] = A[j0 + i][c];            // rows BM .. BM + 6 are spare rows of the tile (zeroed by the caller)
    int len = 8 + H;
#pragma unroll
    for (int l = 0; l < R; ++l) {
        const int st = 1 << l;
        const float2 w = tw[l][c];
        len -= st;
#pragma unroll
        for (int i = 0; i < 8 + H; ++i)
            if (i < len) v[i] = tree_cmadd(v[i], w, v[i + st]);
    }
    __syncthreads();   // every thread has read its halo
#pragma unroll
    for (int i = 0; i < 8; ++i) A[j0 + i][c] = v[i];
    __syncthreads();
}

template <int BM>   // BM rows, 2 * BM threads
__device__ __forceinline__ void h16_tree_store(float* smem, const float2 (*tw)[H16_C], const H16Tile& t, const GemmTreeArgs& a, int tid) {
    float2 (*A)[H16_LDP] = reinterpret_cast<float2 (*)[H16_LDP]>(smem);
    const int c = tid & (H16_C - 1);
    constexpr int THREADS = 2 * BM;
    constexpr int PER = BM * H16_C / THREADS;  // 8
    const int levels = t.G.levels_f;
    int l = levels < 3 ? levels : 3;
    switch (l) {   // wave-uniform
        case 1: h16_tree_register_levels<1, BM>(A, tw, tid); break;
        case 2: h16_tree_register_levels<2, BM>(A, tw, tid); break;
        case 3: h16_tree_register_levels<3, BM>(A, tw, tid); break;
        default: break;
    }
    PVQ_STAMP(6);
    int valid = BM - ((1 << l) - 1);
    // remaining levels (strides >= 8) through LDS, two per pass where possible: evaluated exactly as two radix-2 levels
    for (; l + 1 < levels; l += 2) {
        const int st = 1 << l;
        const float2 w1 = tw[l][c], w2 = tw[l + 1][c];
        valid -= 3 * st;
        float2 v[PER];
#pragma unroll
        for (int g = 0; g < PER; g += 4) {
#pragma unroll
            for (int q = g; q < g + 4; ++q) {
                const int j = (tid + q * THREADS) / H16_C;
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
            const int j = (tid + q * THREADS) / H16_C;
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
            const int j = (tid + q * THREADS) / H16_C;
            if (j < valid) v[q] = tree_cmadd(A[j][c], w, A[j + st][c]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = (tid + q * THREADS) / H16_C;
            if (j < valid) A[j][c] = v[q];
        }
        __syncthreads();
    }
    PVQ_STAMP(2);
    // lanes walk the frames of one column: 512-byte runs in memory, conflict-free LDS reads
    const int j = tid & (BM - 1);
    const int f = t.f0 + j;
    if (j < t.S && f < t.nfr) {
        // windows of more than 64 blocks: 64-block partial sums go to Y, blockdft_tree_finish adds the last levels
        float2* dst = (t.G.nb > t.G.nb_f ? a.Y : a.X) + ((size_t)(f >> 6) * a.xcp + t.col0) * 64 + (f & 63);
#pragma unroll 4
        for (int cc = tid / BM; cc < H16_C; cc += 2) {   // streamed out (non-temporal), as in fused_tree_store
            const float2 val = A[j][cc];
            __builtin_nontemporal_store((f32x2){val.x, val.y}, reinterpret_cast<f32x2*>(&dst[cc * 64]));   // one 8-byte store
        }
    }
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // stores issued and acknowledged
        __syncthreads();
        PVQ_STAMP(3);
    }
}

typedef float f32x4h __attribute__((ext_vector_type(4)));

// the mirrored half-depth GEMM of a wave's 32 rows x 16 columns (see fused_f32_kloop16 for the operand scheme)
template <bool VEC, int BM>
__device__ __forceinline__ void h16_kloop(const GemmTreeArgs& a, float* smem, long long tile_lo, const float4* e_tile, int tid,
                                          f32x4h (&accR)[2], f32x4h (&accI)[2]) {
    constexpr int THREADS = 2 * BM;
    const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
    const unsigned long long pcm_addr = reinterpret_cast<unsigned long long>(a.pcm_base);
    const i32x4 rsrc4 = {(int)(unsigned)pcm_addr, (int)(unsigned)(pcm_addr >> 32), (int)a.pcm_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pcm_base), 0, a.pcm_bytes, 0x00020000);
    const int K2 = a.K / 2;
    long long jf0[2], jb0[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long row_lo = tile_lo + (long long)(wave * 32 + mt * 16 + m16) * a.K;
        jf0[mt] = row_lo + 8 * kq;             // k group g = 2 G + h takes the samples 32 G + 8 kq + 4 h + t: the k order of fused_f32_kloop16
        jb0[mt] = row_lo + a.K - 4 - 8 * kq;   // (here one 16-sample half of its double group per load step: 80 registers do not hold both)
    }
    float fr[2][2][4], bk[2][2][4];
    auto load_group = [&](int buf, int g) {   // g: k group of the whole depth (16 sample pairs each)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (VEC) {
                const unsigned so = 128u * (unsigned)(g >> 1) + 16u * (unsigned)(g & 1);
                const f32x4 v = pvq_raw_buffer_load_f32x4(rsrc4, (int)((unsigned)jf0[mt] * 4u + so), 0, 0);
                const f32x4 w = pvq_raw_buffer_load_f32x4(rsrc4, (int)((unsigned)jb0[mt] * 4u - so), 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fr[buf][mt][t] = v[t];
                    bk[buf][mt][t] = w[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {   // samples before the stream: an explicit out-of-range offset (see fused_f32_stage_load)
                    const int so = 32 * (g >> 1) + 4 * (g & 1);
                    const long long xf = jf0[mt] + so + t, xb = jb0[mt] - so + t;
                    fr[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xf >= 0 ? (unsigned)(xf * 4ll) : 0xFFFFFFFCu, 0, 0));
                    bk[buf][mt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, xb >= 0 ? (unsigned)(xb * 4ll) : 0xFFFFFFFCu, 0, 0));
                }
            }
        }
    };
    float4* El = reinterpret_cast<float4*>(smem);   // [k group][lane 64][2]: (cos t0, -sin t0, cos t1, -sin t1), (t2, t3)
    auto mfma_group = [&](int buf, int gl) {        // gl: k group inside the staged slice
        const float4 b01 = El[(gl * 64 + lane) * 2], b23 = El[(gl * 64 + lane) * 2 + 1];
        const float bc[4] = {b01.x, b01.z, b23.x, b23.z}, bs[4] = {b01.y, b01.w, b23.y, b23.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float sm = fr[buf][mt][t] + bk[buf][mt][3 - t];
                const float df = fr[buf][mt][t] - bk[buf][mt][3 - t];
                accR[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, bc[t], accR[mt], 0, 0, 0);
                accI[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(df, bs[t], accI[mt], 0, 0, 0);
            }
        }
    };
    load_group(0, 0);
    for (int kc = 0; kc < K2; kc += FR_KC) {
        const int rows = K2 - kc < FR_KC ? K2 - kc : FR_KC;
        if (kc > 0) __syncthreads();   // every wave is done with the previous slice
        for (int i = tid; i < rows * 8; i += THREADS) El[i] = e_tile[(size_t)kc * 8 + i];   // rows / 16 groups x 128 float4
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

template <int BM>   // rows of hop blocks per tile; 2 * BM threads = BM / 32 waves of 32 rows x 16 complex columns
__global__ __launch_bounds__(2 * BM, 6) void blockdft_gemm_tree16(GemmTreeArgs a) {   // 6 waves per SIMD = three 512-thread workgroups per CU: at most 80 registers
    __shared__ __attribute__((aligned(16))) float smem[H16Geom<BM>::SMEM_FLOATS];  // the E slice, then the P tile
    __shared__ float2 tw_lds[FT_MAXL][H16_C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const H16Tile T = h16_tile<BM>(a);
    if (T.f0 >= T.nfr) return;
    PVQ_STAMP(0);
    if (tid < FT_MAXL * H16_C) {   // the tile's combine twiddles (levels x 16 columns), before the K loop so that the tree never waits on memory
        const int l = tid >> 4, c = tid & 15;
        if (l < T.G.levels_f) tw_lds[l][c] = a.comb_tw[T.G.tw_off + l * (T.G.n_tiles * CB_C) + T.t16 * H16_C + c];
    }
    const long long s = a.base + T.G.s_rel;
    const long long tile_lo = s + (long long)T.f0 * a.K, tile_hi = tile_lo + (long long)BM * a.K;  // sample range of the tile
    const float4* e_tile = a.E16h + (size_t)(T.col0 / H16_C) * (a.K / 2) * 8;
    f32x4h accR[2], accI[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accR[mt][r] = 0.0f;
            accI[mt][r] = 0.0f;
        }
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {   // shader clock under this kernel's load (see blockdft_gemm_tree)
        a.clk[(blockIdx.x >> 6) * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (tile_lo >= 0 && tile_hi * 4ll <= (long long)a.pcm_bytes)
        h16_kloop<true, BM>(a, smem, tile_lo, e_tile, tid, accR, accI);
    else
        h16_kloop<false, BM>(a, smem, tile_lo, e_tile, tid, accR, accI);
    if (a.clk != nullptr && (blockIdx.x & 63) == 0 && tid == 0) {
        a.clk[(blockIdx.x >> 6) * 4 + 2] = __builtin_amdgcn_s_memtime();
        a.clk[(blockIdx.x >> 6) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    PVQ_STAMP(1);
    __syncthreads();   // the E slice is dead: the P' tile takes its place
    PVQ_STAMP(4);
    // P' tile -> LDS as [row][16 complex + pad]  (C/D layout of the 16x16 MFMA: column = lane & 15, rows 4 (lane >> 4) + r)
    float2 (*Pt)[H16_LDP] = reinterpret_cast<float2 (*)[H16_LDP]>(smem);
    {
        const int m16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pt[wave * 32 + mt * 16 + 4 * kq + r][m16] = make_float2(accR[mt][r], accI[mt][r]);
        if (tid < H16_SPARE * H16_LDP) Pt[BM][tid] = make_float2(0.0f, 0.0f);   // the spare rows (Pt[BM][..] runs on through them)
    }
    __syncthreads();
    PVQ_STAMP(5);
    h16_tree_store<BM>(smem, tw_lds, T, a, tid);
}

}  // namespace pvq
