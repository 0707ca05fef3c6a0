/* ASan / UBSan driver for the CPU oracle (test infrastructure; built and run by tests/test_sanitize_cpu.py): constructs the oracle at
 * the test geometries, runs batches that start inside the zero-filled ring buffer (ragged first frames), the three power_to_db
 * regimes, the peak pipeline on crafted frames and the FFT contract checks of vqt.rs:1087-1128. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pvq_oracle.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); exit(2); } } while (0)

static unsigned long long st = 88172645463325252ull;
static float frand(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)(st >> 40) / 16777216.0f; }

int main(void)
{
    struct { float sr, f0; unsigned oct, bpo; } g[] = {{22050.0f, 55.0f, 7, 84}, {48000.0f, 55.0f, 7, 36}, {96000.0f, 27.5f, 10, 36}};
    for (unsigned gi = 0; gi < 3; gi++) {
        orc_params p;
        orc_default_params(&p);
        p.sr = g[gi].sr; p.min_freq = g[gi].f0; p.octaves = g[gi].oct; p.buckets_per_octave = g[gi].bpo;
        orc_vqt *v = NULL;
        float err[2];
        CHECK(orc_vqt_new(&p, &v, err) == 0 && v);
        uint32_t nb = orc_n_bins(v);
        CHECK(nb == g[gi].oct * g[gi].bpo);
        size_t hop = gi == 2 ? 128 : 256, nf = 40, n_lead = 17;
        size_t ns = n_lead + nf * hop;
        float *pcm = (float *)malloc(sizeof(float) * ns);
        for (size_t i = 0; i < ns; i++) pcm[i] = 0.5f * (frand() - 0.5f) * (i > ns / 2 ? 30.0f : 1.0f);   /* clip, then shift branch */
        memset(pcm + 8 * hop, 0, sizeof(float) * hop);
        float *db = (float *)malloc(sizeof(float) * nf * nb);
        orc_calculate_batch(v, pcm, n_lead, hop, nf, db, NULL);
        for (size_t i = 0; i < nf * nb; i++) CHECK(db[i] >= 0.0f && db[i] <= 60.0f);
        orc_analysis_params ap;
        orc_default_analysis_params(&ap);
        uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * nb);
        float *ctr = (float *)malloc(sizeof(float) * nb), *sz = (float *)malloc(sizeof(float) * nb);
        for (size_t f = 0; f < nf; f++) {
            uint32_t n = orc_analyze_frame(db + f * nb, nb, p.min_freq, p.octaves, p.buckets_per_octave, &ap, idx, ctr, sz);
            CHECK(n <= nb);
        }
        float *zero = (float *)calloc(p.n_fft, sizeof(float));
        orc_calculate_vqt_instant_in_db(v, zero, db);
        for (uint32_t k = 0; k < nb; k++) CHECK(db[k] == 0.0f);   /* analysis.rs:415-428 / A_MIN floor */
        free(zero); free(idx); free(ctr); free(sz); free(db); free(pcm);
        orc_vqt_free(v);
    }
    {   /* constructor errors */
        orc_params p; orc_default_params(&p);
        p.sr = 96000.0f; p.octaves = 10; p.buckets_per_octave = 36;
        orc_vqt *v = NULL; float err[2];
        CHECK(orc_vqt_new(&p, &v, err) != 0 && v == NULL);
    }
    {   /* vqt.rs:1087-1128 */
        float a[512]; memset(a, 0, sizeof a); a[0] = 1.0f;
        orc_fft_complex(a, 256, 0); orc_fft_complex(a, 256, 1);
        CHECK(fabsf(a[0] - 256.0f) < 1e-3f);
        float x[64], out[66];
        for (int i = 0; i < 64; i++) x[i] = frand();
        orc_fft_real(x, 64, out);
        CHECK(isfinite(out[0]) && isfinite(out[64]));
    }
    puts("SANITIZE_ORACLE_OK");
    return 0;
}
