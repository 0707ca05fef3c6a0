// device_tables.hip — builds and uploads the read-only tables (see device_tables.hpp).
#include "device_tables.hpp"

#include <cmath>

namespace pvq {

namespace {
template <typename T>
bool upload(T** dst, const std::vector<T>& src, std::string& msg) {
    const size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
    if (e != hipSuccess) {
        msg = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return false;
    }
    if (!src.empty()) {
        e = hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            msg = std::string("hipMemcpy failed: ") + hipGetErrorString(e);
            return false;
        }
    }
    return true;
}
}  // namespace

float round_to_half(float x) { return static_cast<float>(static_cast<_Float16>(x)); }

DeviceTables* build_device_tables(const HostPlan& plan, bool twiddle_fp16, std::string& msg) {
    auto q = [&](double v) { const float f = static_cast<float>(v); return twiddle_fp16 ? round_to_half(f) : f; };
    const auto& groups = plan.kernel.window_groups;
    auto* t = new DeviceTables();
    t->n_groups = static_cast<int>(groups.size());

    std::vector<float2> split_tw;
    std::vector<uint32_t> row_ptr;
    std::vector<float4> ent;
    const double pi = 3.14159265358979323846;

    for (const WindowGroup& g : groups) {
        const uint32_t ws = g.window_size();
        if (ws < 32 || (ws & (ws - 1)) != 0) {
            msg = "unsupported: analysis window of " + std::to_string(ws) + " samples (need a power of two >= 32)";
            delete t;
            return nullptr;
        }
        if (ws / 2 > 16384) {
            msg = "unsupported: analysis window of " + std::to_string(ws) + " samples exceeds 32768";
            delete t;
            return nullptr;
        }
        GroupDev d{};
        d.w0 = static_cast<int>(g.window_begin);
        d.n_cplx = static_cast<int>(ws / 2);
        d.n_rows = static_cast<int>(g.filter_bank.rows);
        d.first_bin = static_cast<int>(g.first_bin);
        d.row_ptr_off = static_cast<int>(row_ptr.size());
        d.ent_off = static_cast<int>(ent.size());
        d.split_off = static_cast<int>(split_tw.size());
        uint32_t max_col = 0;
        uint32_t rel = 0;
        const CsrMatrix& A = g.filter_bank;
        const CsrMatrix& B = g.negative_filter_bank;
        for (uint32_t r = 0; r < A.rows; ++r) {
            row_ptr.push_back(rel);
            for (uint32_t i = A.row_ptr[r]; i < A.row_ptr[r + 1]; ++i) {
                ent.push_back(make_float4(A.values[i].re, A.values[i].im, __builtin_bit_cast(float, static_cast<uint32_t>(A.col_idx[i])), 0.0f));
                max_col = std::max(max_col, A.col_idx[i]);
                ++rel;
            }
            if (B.nnz() > 0) {
                // x_vqt += conj(Kneg . X)  ==  sum conj(Kneg) * conj(X)   (vqt.rs:896-910)
                for (uint32_t i = B.row_ptr[r]; i < B.row_ptr[r + 1]; ++i) {
                    ent.push_back(make_float4(B.values[i].re, -B.values[i].im, __builtin_bit_cast(float, static_cast<uint32_t>(B.col_idx[i] | 0x8000u)), 0.0f));
                    max_col = std::max(max_col, B.col_idx[i]);
                    ++rel;
                }
            }
        }
        row_ptr.push_back(rel);
        d.n_cols = static_cast<int>(max_col + 1);
        d.tpr = 1;
        while (d.tpr < 16 && d.tpr * 2 * std::max(d.n_rows, 1) <= 512) d.tpr *= 2;
        for (int c = 0; c < d.n_cols; ++c) {
            const double ang = -2.0 * pi * static_cast<double>(c) / static_cast<double>(ws);
            split_tw.push_back(make_float2(q(std::cos(ang)), q(std::sin(ang))));
        }
        t->n_tw = std::max(t->n_tw, d.n_cplx);
        t->max_cols = std::max(t->max_cols, d.n_cols);
        t->h_groups.push_back(d);
    }
    t->total_entries = static_cast<int>(ent.size());

    std::vector<float2> tw(static_cast<size_t>(t->n_tw));
    for (int m = 0; m < t->n_tw; ++m) {
        const double ang = -2.0 * pi * static_cast<double>(m) / static_cast<double>(t->n_tw);
        tw[m] = make_float2(q(std::cos(ang)), q(std::sin(ang)));
    }

    std::vector<float> lnf;
    bin_log_frequencies(plan.params, lnf);
    const std::vector<uint32_t> status(4, 0u);

    if (!upload(&t->d_status, status, msg) || !upload(&t->d_lnf, lnf, msg) || !upload(&t->d_groups, t->h_groups, msg) || !upload(&t->d_tw, tw, msg) || !upload(&t->d_split_tw, split_tw, msg) ||
        !upload(&t->d_row_ptr, row_ptr, msg) || !upload(&t->d_ent, ent, msg)) {
        free_device_tables(t);
        return nullptr;
    }
    return t;
}

void free_device_tables(DeviceTables* t) {
    if (!t) return;
    if (t->block) free_blockdft_tables(t->block);
    if (t->d_groups) (void)hipFree(t->d_groups);
    if (t->d_tw) (void)hipFree(t->d_tw);
    if (t->d_split_tw) (void)hipFree(t->d_split_tw);
    if (t->d_row_ptr) (void)hipFree(t->d_row_ptr);
    if (t->d_ent) (void)hipFree(t->d_ent);
    if (t->d_lnf) (void)hipFree(t->d_lnf);
    if (t->d_status) (void)hipFree(t->d_status);
    delete t;
}

}  // namespace pvq
