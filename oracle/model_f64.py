"""Independent float64 NumPy model of the VQT math (TEST INFRASTRUCTURE, not product code).

Purpose: cross-check the single-precision C oracle (oracle/pvq_oracle.c) with a second,
differently-written implementation of the same mathematics (vqt.rs:517-587 params, :599-759
grouping/remap, :769-852 filter, :866-954 frame + dB), using numpy.fft in float64.  It also
serves as the "truth" against which both the f32 CPU oracle's and the GPU path's rounding error
are measured (tests/test_parity_gpu.py reports err_gpu_vs_f64 next to err_oracle_vs_f64).

Integer decisions (decimation factor, window sizes, filter placement) are recomputed here from
the formulas in float64; they must agree with the oracle's f32 decisions for every geometry the
tests use (asserted by tests/test_oracle_vs_f64.py).
"""
from __future__ import annotations

import numpy as np


class ModelF64:
    def __init__(self, sr, n_fft, min_freq, octaves, bpo, sparsity_quantile, quality, gamma,
                 pattern_from=None, values_from=None):
        """pattern_from: optional OracleVqt; if given, the sparsity pattern (which coefficients
        survive) is taken from the oracle so that values can be compared coefficient by
        coefficient.
        values_from: optional OracleVqt; if given, the kernel VALUES are the oracle's f32 CSR
        values widened to f64 (the reference evaluates the wavelet phase in f32, vqt.rs:797-800,
        which moves kernel values by ~1e-4 relative to exact math; that rounding is part of the
        reference's kernel).  The frame transform (rFFT, products, sums) is then exact-in-f64:
        the "truth given the kernel" against which f32 frame errors are measured."""
        self.sr, self.n_fft, self.min_freq = float(sr), int(n_fft), float(min_freq)
        self.octaves, self.bpo = int(octaves), int(bpo)
        self.q, self.quality, self.gamma = float(sparsity_quantile), float(quality), float(gamma)
        self.n_bins = self.octaves * self.bpo
        k = np.arange(self.n_bins, dtype=np.float64)
        self.freq = self.min_freq * 2.0 ** (k / self.bpo)
        r = 2.0 ** (1.0 / self.bpo)
        alpha = (r * r - 1.0) / (r * r + 1.0)
        self.wl = self.quality * self.sr / (alpha * self.freq + self.gamma)
        msr = np.ceil(self.freq * 2.0 * 1.15)
        self.M = (1 << np.floor(np.log2(self.sr / msr)).astype(np.int64))
        self.minwin = self.n_fft >> np.floor(np.log2(self.n_fft / self.wl)).astype(np.int64)
        self.window_center = self.n_fft - self.wl[0] / 2.0
        self.delay = (self.n_fft - self.window_center) / self.sr
        # rate groups -> window groups
        rgs = []
        s = 0
        while s < self.n_bins:
            e = s + 1
            while e < self.n_bins and self.M[e] == self.M[s]:
                e += 1
            ws = int(self.minwin[s:e].max())
            if self.window_center + ws / 2.0 < self.n_fft:
                w = (int(self.window_center - ws / 2.0), int(self.window_center + ws / 2.0))
            else:
                w = (self.n_fft - ws, self.n_fft)
            rgs.append((int(self.M[s]), w, s, e))
            s = e
        self.groups = []  # dict(window, rows(first,last), K dense complex [rows x (ws/2+1)], Kneg)
        i = 0
        gain = np.sqrt(self.sr)
        g_idx = 0
        while i < len(rgs):
            j = i + 1
            while j < len(rgs) and rgs[j][1] == rgs[i][1]:
                j += 1
            w = rgs[i][1]
            ws = w[1] - w[0]
            first, last = rgs[i][2], rgs[j - 1][3]
            K = np.zeros((last - first, ws // 2 + 1), np.complex128)
            Kn = np.zeros_like(K)
            if pattern_from is not None:
                rp, ci, _ = pattern_from.group_csr(g_idx, neg=False)
                rpn, cin, _ = pattern_from.group_csr(g_idx, neg=True)
            row = 0
            for (M, _, s, e) in rgs[i:j]:
                S = ws // M
                for kb in range(s, e):
                    L = int(np.round(self.wl[kb] / M))  # numpy rounds half to even; .5 never occurs here
                    c = int(np.floor((self.window_center - w[0]) / M))
                    begin = c - L // 2
                    n = np.arange(L, dtype=np.float64)
                    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (L - 1))
                    v = np.zeros(S, np.complex128)
                    v[begin:begin + L] = hann * np.exp(2j * np.pi * n * self.freq[kb] * M / self.sr)
                    v /= np.abs(v).sum()
                    H = np.conj(np.fft.fft(v))
                    if pattern_from is None:
                        mag = np.sort(np.abs(H))
                        target = (1.0 - self.q) * mag.sum()
                        cs = np.cumsum(mag)
                        idx = int(np.searchsorted(cs, target, side="left")) + 1  # first idx with cs>=target
                        cutoff = mag[idx - 1] if idx > 0 else 0.0
                        H = np.where(np.abs(H) < cutoff, 0.0, H)
                        keep_pos = [jj for jj in range(S // 2 + 1) if H[jj] != 0]
                        keep_neg = [S - jj for jj in range(S // 2 + 1, S) if H[jj] != 0]
                    else:
                        keep_pos = list(ci[rp[row]:rp[row + 1]])
                        keep_neg = list(cin[rpn[row]:rpn[row + 1]])
                    for col in keep_pos:
                        K[row, col] = H[col] * gain / ws
                    for col in keep_neg:
                        Kn[row, col] = np.conj(H[S - col] * gain / ws)
                    row += 1
            if values_from is not None:
                K[:] = 0; Kn[:] = 0
                for (dst, neg) in ((K, False), (Kn, True)):
                    rpv, civ, vav = values_from.group_csr(g_idx, neg=neg)
                    for rr in range(last - first):
                        dst[rr, civ[rpv[rr]:rpv[rr + 1]]] = vav[rpv[rr]:rpv[rr + 1]].astype(np.complex128)
            self.groups.append(dict(window=w, rows=(first, last), K=K, Kneg=Kn))
            g_idx += 1
            i = j

    def frame_complex(self, x):
        x = np.asarray(x, np.float64)
        out = np.zeros(self.n_bins, np.complex128)
        for g in self.groups:
            w0, w1 = g["window"]
            X = np.fft.rfft(x[w0:w1])
            a, b = g["rows"]
            out[a:b] = g["K"] @ X + np.conj(g["Kneg"] @ X)
        return out

    @staticmethod
    def power_to_db(xc):
        ref_db = 10.0 * np.log10(0.09)
        d = 10.0 * np.log10(np.maximum(np.abs(xc) ** 2, 1e-12)) - ref_db
        floor = d.max() - 60.0
        mn = max(d.min(), floor)
        c = np.maximum(d, floor)
        return c - mn if mn > 0 else np.maximum(c, 0.0)

    def frame_db(self, x):
        return self.power_to_db(self.frame_complex(x))

    def batch_complex(self, pcm, hop, n_frames, n_lead=0):
        pcm = np.asarray(pcm, np.float64)
        out = np.zeros((n_frames, self.n_bins), np.complex128)
        for f in range(n_frames):
            end = n_lead + (f + 1) * hop
            beg = end - self.n_fft
            x = np.zeros(self.n_fft)
            lo = max(beg, 0)
            x[lo - beg:] = pcm[lo:end]
            out[f] = self.frame_complex(x)
        return out


def from_oracle_params(p, pattern_from=None, values_from=None) -> ModelF64:
    return ModelF64(p.sr, p.n_fft, p.min_freq, p.octaves, p.buckets_per_octave, p.sparsity_quantile,
                    p.quality, p.gamma, pattern_from=pattern_from or values_from, values_from=values_from)
