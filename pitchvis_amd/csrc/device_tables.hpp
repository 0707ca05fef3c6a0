// device_tables.hpp — device-resident, read-only tables derived from a HostPlan: per-group
// metadata, FFT twiddles, real-split twiddles and the sparse spectral kernel in a merged
// (filter_bank + negative_filter_bank) row-major entry list.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "vqt_host.hpp"

namespace pvq {

struct GroupDev {
    int w0;           // window begin inside the n_fft buffer (vqt.rs:391)
    int n_cplx;       // N = window_size/2: length of the packed complex FFT
    int n_rows;       // filters in this group
    int first_bin;    // output row offset
    int n_cols;       // spectrum columns [0, n_cols) that any entry of this group reads
    int row_ptr_off;  // into d_row_ptr (n_rows+1 entries, relative to ent_off)
    int ent_off;      // into d_ent_val / d_ent_col
    int split_off;    // into d_split_tw (n_cols entries: exp(-2 pi i c / window_size))
    int tpr;          // lanes that share one kernel row in the FFT path's row dots: the largest power of two <= 512 / n_rows, at most 16
};

// block-DFT path tables (vqt_blockdft.hip), built lazily per hop
struct BlockDftTables;

struct DeviceTables {
    int n_groups = 0;
    int n_tw = 0;      // largest n_cplx; d_tw[m] = exp(-2 pi i m / n_tw)
    int max_cols = 0;  // max n_cols
    int total_entries = 0;
    std::vector<GroupDev> h_groups;
    GroupDev* d_groups = nullptr;
    float2* d_tw = nullptr;
    float2* d_split_tw = nullptr;
    uint32_t* d_row_ptr = nullptr;
    // one 16-byte record per kernel entry: (re, im, column | 0x8000 when the entry multiplies conj(X[col]) [as bits], 0);
    // filter_bank values as they are, negative_filter_bank values stored conjugated
    float4* d_ent = nullptr;
    float* d_lnf = nullptr;        // ln(f_k) per bin (host libm, peak_detection.rs:81-86)
    uint32_t* d_status = nullptr;  // [0]: sticky "a frame's spectrum was not finite" flag set by the dB stages (see Vqt::input_status)
    BlockDftTables* block = nullptr;
};

// returns nullptr and fills msg on failure ("unsupported: ..." for geometry limits)
// twiddle_fp16: every twiddle factor is rounded to the nearest IEEE half before use (BASELINE config 4's "fp16 FFT
// twiddles"); the tables stay fp32 arrays and all accumulation stays fp32
DeviceTables* build_device_tables(const HostPlan& plan, bool twiddle_fp16, std::string& msg);
// value of the nearest fp16 (round to nearest even), for |x| <= 1
float round_to_half(float x);
void free_device_tables(DeviceTables* t);
void free_blockdft_tables(BlockDftTables* t);

}  // namespace pvq
