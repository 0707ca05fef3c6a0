"""Workgroup slots over time from a PVQ_STAMPS dump (developer tool): how full the 512 slots of the fused GEMM + tree kernel are, what
lies between one workgroup's end and the next one's start on a slot (the dispatcher's gap), and how even the eight XCD queues run."""
import sys, numpy as np
raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.bin", dtype=np.uint64).astype(np.int64)
SLOTS = int(sys.argv[2]) if len(sys.argv) > 2 else 512   # workgroup slots of the chip: 2 per CU (3 for blockdft_gemm_tree3)
n = len(raw) // 12
s8, s4 = raw[:n * 8].reshape(n, 8), raw[n * 8:n * 12].reshape(n, 4)
idx = np.nonzero(s4[:, 0] > 0)[0]          # (padding entries of the tile list leave no stamps)
s8, s4 = s8[idx], s4[idx]
t0 = s4[:, 0].min()
T = lambda a: (a - t0) * 0.01              # 100 MHz ticks -> us
entry, s0, s3, issued, done = T(s4[:, 0]), T(s8[:, 0]), T(s8[:, 3]), T(s4[:, 1]), T(s4[:, 2])
ok = s8[:, 7] > 0                          # (the range-checked tiles carry no prologue stamp)
kb, ke = np.where(ok, T(s8[:, 7]), 0.0), np.where(ok, T(s8[:, 1]), 0.0)
def pr(name, a): print(f"{name:44s} p10 {np.percentile(a, 10):6.2f}  p50 {np.percentile(a, 50):6.2f}  p90 {np.percentile(a, 90):6.2f}  mean {a.mean():6.2f} us")
print(f"workgroups {len(idx)}  span {done.max():.1f} us  sum of lives / {SLOTS} = {np.sum(done - entry) / SLOTS:.1f} us  in K loop / {SLOTS} = {np.sum(ke - kb) / SLOTS:.1f} us")
pr("first instruction -> first stamp (descriptor)", s0 - entry)
pr("wave 0's last store -> all waves' issued", issued - s3)
pr("all issued -> all acknowledged", done - issued)
pr("life (first instruction -> acknowledged)", done - entry)
gaps = []
for x in range(8):   # workgroup b runs on XCD b % 8: starts after the first 64 matched, in order, with the sorted ends
    m = idx % 8 == x
    bs, es = np.sort(entry[m]), np.sort(done[m])
    gaps.extend(bs[SLOTS // 8:] - es[:len(bs) - SLOTS // 8])
    print(f"  XCD {x}: {m.sum():4d} workgroups, first start {entry[m].min():6.1f}, last start {entry[m].max():6.1f}, end {done[m].max():6.1f} us")
pr("slot empty (start - matched end on the XCD)", np.array(gaps))
step = 10.0
for t in np.arange(0, done.max(), step):
    live = np.sum(((entry < t + step) & (done > t)) * (np.minimum(done, t + step) - np.maximum(entry, t)) / step)
    ink = np.sum(((kb < t + step) & (ke > t)) * (np.minimum(ke, t + step) - np.maximum(kb, t)) / step)
    print(f"{t:6.0f} us: live {live:6.1f}  in K loop {ink:6.1f}")
