"""CPU ORACLE (test infrastructure only) for the callers either side of the path — NumPy-f32 restatements,
written independently of pitchvis_amd/csrc/consumers_host.cpp, each citing the reference lines it follows.
Parity with the Rust binaries is UNPINNED (no toolchain); the `lab` crate (0.11.0, Cargo.lock:3985) that
pitchvis_colors calls is not vendored, its sRGB/XYZ/Lab/LCh conversions are restated from the published maths."""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


class MonoAgc:
    """dagc_fork/src/lib.rs:19-87"""

    def __init__(self, desired_output_rms, distortion_factor):
        rms, d = f32(desired_output_rms), f32(distortion_factor)
        if not (rms > 0 and np.isfinite(rms)):                      # lib.rs:37-41
            raise ValueError(f"`desired_output_rms` must be a finite positive number, but got {rms}")
        if not (f32(0) <= d <= f32(1)):                              # lib.rs:42-46
            raise ValueError(f"`distortion_factor` must be a number within `0.0 ..= 1.0`, but got {d}")
        self.rms, self.d, self.gain, self.frozen = rms, d, f32(1.0), False

    def freeze_gain(self, freeze):
        self.frozen = bool(freeze)

    def process(self, samples):                                      # lib.rs:76-86
        one = f32(1.0)
        for i in range(samples.size):
            x = f32(samples[i] * self.gain)
            samples[i] = x
            if not self.frozen:
                y = f32(f32(x * x) / self.rms)
                g = f32(one + f32(self.d * f32(one - y)))
                g = f32(np.fmax(g, self.d))   # f32::max: a NaN operand is ignored
                self.gain = f32(self.gain * g)


def train_chunk_samples(delay_seconds, sr):
    """train.rs:128-129"""
    delay_ms = int(delay_seconds * 1000.0)
    return (delay_ms * int(sr) // 1000) // 64 * 64


def train_loop(ov, left, right, voices, chunk, step=3, bufsize=None):
    """pitchvis_train::synthesize_midi_to_wav (train.rs:252-351) + generate_data (train.rs:443-460), literally:
    a ring buffer of BUFSIZE samples, one calculate_vqt_instant_in_db per analysed chunk.  ov: oracle.OracleVqt."""
    n_fft = ov.params.n_fft
    bufsize = bufsize or 2 * int(ov.params.sr)                       # train.rs:31
    agc = MonoAgc(0.07, 0.001)                                       # train.rs:265
    ring = np.zeros(bufsize, f32)                                    # train.rs:268-269
    n_chunks = left.size // chunk
    rows, gains = [], []
    prev, active = {}, {}
    f = 0
    for c in range(1, n_chunks + 1):
        l = left[(c - 1) * chunk:c * chunk].astype(f32)
        if right is not None:
            l = ((l + right[(c - 1) * chunk:c * chunk].astype(f32)) / f32(2.0)).astype(f32)   # train.rs:286-289
        sq = f32(0.0)
        for v in l:
            sq = f32(sq + f32(v * v))                                # train.rs:292
        agc.freeze_gain(sq < f32(1e-6))                              # train.rs:293
        ring = np.concatenate([ring[chunk:], l])                     # train.rs:298-299
        new = ring[-chunk:].copy()
        agc.process(new)                                             # train.rs:300-301
        ring[-chunk:] = new
        gains.append(agc.gain)
        if c % step != 0:                                            # train.rs:303-305
            continue
        prev, active = active, {}                                    # train.rs:314-315
        for key, gl, gr in voices[f]:                                # train.rs:317-337
            gain = f32(f32(f32(f32(gl) + f32(gr)) / f32(2.0)) * agc.gain)
            if key in active:
                if gain > active[key]:
                    active[key] = gain
            else:
                active[key] = gain
        x_vqt = ov.calculate_vqt_instant_in_db(ring[-n_fft:])        # train.rs:341
        targets = np.zeros(128, f32)                                 # train.rs:447-458
        for key, attack in prev.items():
            targets[key] = 1.0 if attack > f32(0.5) else 0.0
        rows.append(np.concatenate([x_vqt, targets]))
        f += 1
    return np.concatenate(rows).astype(f32) if rows else np.zeros(0, f32), np.asarray(gains, f32), ring


# ---- the `lab` crate's conversions (f32) -------------------------------------------------------------
KAPPA = f32(24389.0 / 27.0)
EPSILON = f32(216.0 / 24389.0)
CBRT_EPSILON = f32(6.0 / 29.0)
S_0 = f32(0.003130668442500564)
E_0_255 = f32(f32(3294.6) * S_0)
WHITE_X = f32(0.9504492182750991)
WHITE_Z = f32(1.0889166484304715)


def _expand(c):
    c = f32(c)
    if c > E_0_255:
        return f32(np.power(f32((c + f32(0.055 * 255.0)) / f32(1.055 * 255.0)), f32(2.4)))
    return f32(c / f32(12.92 * 255.0))


def _compress(c):
    c = f32(c)
    v = f32(f32(1.055) * f32(np.power(c, f32(1.0 / 2.4))) - f32(0.055)) if c > S_0 else f32(f32(12.92) * c)
    return f32(max(min(v, f32(1.0)), f32(0.0)))


def _lab_map(c):
    c = f32(c)
    return f32(np.power(c, f32(1.0 / 3.0))) if c > EPSILON else f32(f32(KAPPA * c + f32(16.0)) / f32(116.0))


def rgb_to_lch(rgb):
    r, g, b = (_expand(v) for v in rgb)
    x = f32(r * f32(0.4124108464885388) + g * f32(0.3575845678529519) + b * f32(0.18045380393360833))
    y = f32(r * f32(0.21264934272065283) + g * f32(0.7151691357059038) + b * f32(0.07218152157344333))
    z = f32(r * f32(0.019331758429150258) + g * f32(0.11919485595098397) + b * f32(0.9503900340503373))
    fx, fy, fz = _lab_map(x / WHITE_X), _lab_map(y), _lab_map(z / WHITE_Z)
    l = f32(f32(116.0) * fy - f32(16.0))
    a, bb = f32(f32(500.0) * (fx - fy)), f32(f32(200.0) * (fy - fz))
    return l, f32(math.hypot(a, bb)), f32(math.atan2(bb, a))


def lch_to_rgb(l, c, h):
    a, bb = f32(c * f32(math.cos(h))), f32(c * f32(math.sin(h)))
    fy = f32((l + f32(16.0)) / f32(116.0))
    fx = f32(a / f32(500.0) + fy)
    fz = f32(fy - bb / f32(200.0))
    xr = f32(fx * fx * fx) if fx > CBRT_EPSILON else f32((fx * f32(116.0) - f32(16.0)) / KAPPA)
    yr = f32(fy * fy * fy) if l > f32(EPSILON * KAPPA) else f32(l / KAPPA)
    zr = f32(fz * fz * fz) if fz > CBRT_EPSILON else f32((fz * f32(116.0) - f32(16.0)) / KAPPA)
    x, y, z = f32(xr * WHITE_X), yr, f32(zr * WHITE_Z)
    r = f32(x * f32(3.240812398895283) - y * f32(1.5373084456298136) - z * f32(0.4985865229069666))
    g = f32(x * f32(-0.9692430170086407) + y * f32(1.8759663029085742) + z * f32(0.04155503085668564))
    b = f32(x * f32(0.055638398436112804) - y * f32(0.20400746093241362) + z * f32(1.0571295702861434))
    return [int(np.round(f32(_compress(v) * f32(255.0)))) for v in (r, g, b)]


def _as_u8(v):
    """Rust `as u8`: saturating, NaN -> 0, truncating"""
    v = float(v)
    if not (v > 0.0):
        return 0
    return 255 if v >= 255.0 else int(v)


def calculate_color(buckets_per_octave, bucket, colors, gray_level, easing_pow):
    """pitchvis_colors/src/lib.rs:86-117"""
    pc = f32(f32(12.0) * f32(bucket) / f32(buckets_per_octave))
    rounded = f32(math.floor(float(pc) + 0.5)) if pc >= 0 else f32(-math.floor(-float(pc) + 0.5))   # f32::round: half away from zero
    base = [_as_u8(f32(f32(c) * f32(255.0))) for c in colors[int(max(rounded, 0)) % 12]]
    inacc = f32(abs(pc - rounded))
    l, c, h = rgb_to_lch(base)
    sat = f32(f32(1.0) - f32(np.power(f32(f32(2.0) * inacc), f32(easing_pow))))
    c = f32(c * sat)
    l = f32(sat * l + f32(f32(1.0) - sat) * f32(gray_level))
    return tuple(f32(v) / f32(255.0) for v in lch_to_rgb(l, c, h))


def led_frame(n_buckets, bpo, peaks_continuous, colors, gray_level, easing_pow):
    """pitchvis_serial/src/main.rs:122-175"""
    x = np.zeros(n_buckets, f32)
    for center, size in peaks_continuous:
        center, size = f32(center), f32(size)
        lower = int(math.floor(center))
        fract = f32(center - f32(np.trunc(center)))
        x[lower] = f32(size * f32(f32(1.0) - f32(np.power(fract, f32(1.9)))))
        if lower < n_buckets - 1:
            x[lower + 1] = f32(size * f32(np.power(fract, f32(1.9))))
    k_max = int(np.argmax(x))          # util::arg_max: first maximum
    max_size = x[k_max]
    out = bytearray([0xFF, n_buckets // 256, n_buckets % 256])
    shift = bpo - 3 * (bpo // 12)
    with np.errstate(all="ignore"):
        for idx in range(n_buckets):
            rgb = calculate_color(bpo, float(f32(math.fmod(float(idx + shift), float(bpo)))), colors, gray_level, easing_pow)
            coef = f32(f32(1.0) - f32(f32(1.0) - f32(x[idx] / max_size)))
            out.extend(_as_u8(f32(f32(v * coef) * f32(254.0))) for v in rgb)
    return bytes(out)
