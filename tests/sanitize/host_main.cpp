// ASan / UBSan driver for the host side of libpvq (test infrastructure; built and run by tests/test_sanitize_cpu.py, CPU only — GPU
// AddressSanitizer is not available on this pool).  Replays, through the instrumented objects, what tests/test_host_plan.py,
// tests/test_analysis_state.py, tests/test_consumers.py and tests/test_multi_device.py feed them: kernel construction for every test
// geometry (including the two constructor errors and a reference panic), the AnalysisState recurrence in three smoothing modes, the
// peak helpers on crafted frames, the AGC / dataset / LED / .npy consumers and the shard planner.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "analysis_host.hpp"
#include "consumers_host.hpp"
#include "multi_host.hpp"
#include "vqt_host.hpp"

using namespace pvq;

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {   // xorshift64*, [0, 1)
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (float)((rng_state * 2685821657736338717ull) >> 40) / 16777216.0f;
}
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); std::exit(2); } } while (0)

int main() {
    // ---- kernel construction: the six test geometries ----------------------------------------------------------------------
    struct G { float sr; float f0; unsigned oct, bpo; float q; } geoms[] = {
        {22050.0f, 55.0f, 7, 84, 1.6f}, {48000.0f, 55.0f, 7, 36, 1.6f}, {48000.0f, 55.0f, 8, 36, 1.6f},
        {96000.0f, 27.5f, 10, 36, 1.6f}, {96000.0f, 27.5f, 10, 84, 1.6f}, {22050.0f, 55.0f, 5, 36, 1.8f}};
    size_t nnz_default = 0, neg_default = 0;
    for (const G& g : geoms) {
        VqtParameters p;
        p.sr = g.sr; p.range.min_freq = g.f0; p.range.octaves = g.oct; p.range.buckets_per_octave = g.bpo; p.quality = g.q;
        HostPlan plan;
        const VqtError e = build_plan(p, plan);
        CHECK(e.kind == VqtError::None);
        size_t nnz = 0, neg = 0, rows = 0;
        for (const WindowGroup& w : plan.kernel.window_groups) {
            nnz += w.filter_bank.nnz(); neg += w.negative_filter_bank.nnz(); rows += w.filter_bank.rows;
            CHECK(w.filter_bank.row_ptr.size() == w.filter_bank.rows + 1);
            for (uint32_t c : w.filter_bank.col_idx) CHECK(c < w.filter_bank.cols);
        }
        CHECK(rows == p.range.n_buckets());
        CHECK(plan.bandwidth_lo_hz.size() == rows);
        if (g.bpo == 84 && g.sr < 30000.0f) { nnz_default = nnz; neg_default = neg; }
        std::vector<float> lnf;
        bin_log_frequencies(p, lnf);
        CHECK(lnf.size() == rows);
    }
    CHECK(neg_default == 379);                       // VQT_REVIEW.md:369
    CHECK(nnz_default > 15000 && nnz_default < 20000);
    {   // constructor errors (vqt.rs:518-528, :567-573) and a reference panic turned exception
        VqtParameters p; p.sr = 96000.0f; p.range.octaves = 10; p.range.buckets_per_octave = 36;
        HostPlan plan;
        CHECK(build_plan(p, plan).kind == VqtError::AboveNyquist);
        VqtParameters q; q.quality = 40.0f;
        CHECK(build_plan(q, plan).kind == VqtError::WindowExceedsNFft);
        CHECK(!build_plan(q, plan).to_string().empty());
    }
    // ---- AnalysisState: three smoothing modes over noise + tones -------------------------------------------------------------
    for (int mode = 0; mode < 3; ++mode) {
        VqtRange r; r.buckets_per_octave = mode == 2 ? 36 : 84; r.octaves = mode == 2 ? 5 : 7;
        FullAnalysisParameters ap;
        AnalysisState st(r, ap);
        if (mode == 1) st.update_vqt_smoothing_duration(false, Duration{});
        const uint32_t n = r.n_buckets();
        std::vector<float> x(n);
        for (int f = 0; f < 120; ++f) {
            for (uint32_t k = 0; k < n; ++k) x[k] = 8.0f * frand();
            x[(37 + f) % n] = 45.0f; x[(200 + 3 * f) % n] = 38.0f; x[5] = 30.0f;
            if (f == 60) std::fill(x.begin(), x.end(), 0.0f);
            CHECK(st.preprocess(x.data(), n, Duration{(uint64_t)(f % 7 == 0 ? 1100 : 16) * 1000000ull}));
            CHECK(st.x_vqt_peakfiltered.size() == n && st.calmness.size() == n);
            for (const ContinuousPeak& c : st.peaks_continuous) CHECK(std::isfinite(c.center) && std::isfinite(c.size));
            (void)st.bin_to_frequency((uint32_t)f % n);
        }
        CHECK(!st.preprocess(x.data(), n - 1, Duration{16000000ull}));   // analysis.rs:289: wrong length
    }
    {   // peak helpers on crafted frames: plateaus, edges, empty
        VqtRange r; r.buckets_per_octave = 84;
        std::vector<float> x(r.n_buckets(), 0.0f);
        PeakDetectionParameters cfg{5.0f, 3.5f};
        CHECK(find_peaks(cfg, x.data(), (uint32_t)x.size(), 84).empty());
        x[0] = 50.0f; x[x.size() - 1] = 50.0f; x[100] = x[101] = x[102] = 20.0f; x[300] = 40.0f; x[302] = 39.0f;
        std::vector<uint32_t> pk = find_peaks(cfg, x.data(), (uint32_t)x.size(), 84);
        std::vector<ContinuousPeak> cp = enhance_peaks_continuous(pk, x.data(), r);
        promote_bass_peaks_with_harmonics(cp, x.data(), r, 28, 0.3f);
        CHECK(cp.size() == pk.size());
    }
    // ---- consumers -------------------------------------------------------------------------------------------------------------
    {
        std::string why;
        CHECK(MonoAgc::valid(0.07f, 0.0001f, &why));
        CHECK(!MonoAgc::valid(-1.0f, 0.5f, &why));
        MonoAgc agc(0.07f, 0.0001f);
        const size_t chunk = train_chunk_samples(0.0915, 22050.0f), n_chunks = 9;
        CHECK(chunk % 64 == 0 && chunk > 0);
        std::vector<float> L(n_chunks * chunk), R(n_chunks * chunk), mono(n_chunks * chunk), gains(n_chunks);
        for (size_t i = 0; i < L.size(); ++i) { L[i] = 0.3f * (frand() - 0.5f); R[i] = 0.3f * (frand() - 0.5f); }
        std::fill(L.begin() + 2 * chunk, L.begin() + 3 * chunk, 0.0f);   // a silent chunk: gain frozen
        std::fill(R.begin() + 2 * chunk, R.begin() + 3 * chunk, 0.0f);
        train_condition_stream(agc, L.data(), R.data(), n_chunks, chunk, mono.data(), gains.data());
        train_condition_stream(agc, L.data(), nullptr, n_chunks, chunk, mono.data(), gains.data());
        const uint32_t nb = 252, nf = 3;
        std::vector<float> db(nf * nb, 1.0f), rows(nf * (nb + 128));
        std::vector<uint32_t> vptr = {0, 2, 2, 3};
        std::vector<int32_t> vkey = {60, 64, 127};
        std::vector<float> gl = {0.9f, 0.2f, 1.0f}, gr = {0.9f, 0.4f, 1.0f}, ag = {1.0f, 1.0f, 1.0f};
        CHECK(train_rows(db.data(), nf, nb, vptr.data(), vkey.data(), gl.data(), gr.data(), ag.data(), rows.data(), &why));
        vkey[0] = 128;   // (frame 0's voices label frame 1's row)
        CHECK(!train_rows(db.data(), nf, nb, vptr.data(), vkey.data(), gl.data(), gr.data(), ag.data(), rows.data(), &why));
        const char* path = "/tmp/pvq_sanitize_rows.npy";
        CHECK(npy_write_f32(path, rows.data(), rows.size(), &why));
        std::remove(path);
        CHECK(!npy_write_f32("/nonexistent-dir/x.npy", rows.data(), 4, &why));
        float colors[12][3];
        for (int i = 0; i < 12; ++i) for (int c = 0; c < 3; ++c) colors[i][c] = frand();
        float rgb[3];
        for (float b = 0.0f; b < 72.0f; b += 0.37f) calculate_color(36, b, colors, 0.6f, 0.8f, rgb);
        const uint32_t n_b = 180;
        std::vector<float> ctr = {0.2f, 17.5f, 100.9f, 179.4f}, sz = {3.0f, 20.0f, 45.0f, 9.0f};
        std::vector<uint8_t> out(3 + 3 * n_b);
        CHECK(led_frame(n_b, 36, ctr.data(), sz.data(), 4, colors, 0.6f, 0.8f, out.data()) == out.size());
        CHECK(led_frame(n_b, 36, nullptr, nullptr, 0, colors, 0.6f, 0.8f, out.data()) == out.size());
    }
    // ---- shard planner -------------------------------------------------------------------------------------------------------------
    for (uint32_t world = 1; world <= 9; ++world) {
        uint64_t total = 0;
        for (uint32_t r = 0; r < world; ++r) {
            ShardPlan s;
            CHECK(plan_shard(1000003, 256, 16384, r, world, &s));
            CHECK(s.first_frame == total);
            total += s.n_frames;
        }
        CHECK(total == 1000003);
        ShardPlan s;
        CHECK(!plan_shard(10, 256, 16384, world, world, &s));
    }
    std::puts("SANITIZE_HOST_OK");
    return 0;
}
