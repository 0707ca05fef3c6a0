#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench lines (configs[1] and [2]), rocprofv3 kernel stats, HBM PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters of the dominant kernels.  Writes gpurun_out/<tag>_* and
# gpurun_out/traffic_latest.json (the captures bench.py attaches as roofline.traffic, stamped with the hash of the kernel sources
# they were measured on: bench.py refuses a capture whose hash is not the running library's).
# usage: scripts/collect_profiles.sh <tag>
set -e
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the program goes directly after `--`: no env / bash -c hop under rocprofv3)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-config > $OUT/${TAG}_stats.log 2>&1
for cfg in 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc${cfg}_$c -- python3 $ROOT/bench.py --steps 3 --warmup 1 --config $cfg --no-cpu-baseline --no-extra-config > $OUT/${TAG}_pmc${cfg}_$c.log 2>&1
  done
done
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_sq$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-config > $OUT/${TAG}_sq$i.log 2>&1 || echo "SQ pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, os
tag = "$TAG"
srchash = open("pitchvis_amd/lib/libpvq.so.srchash").read().strip()
SLOT = {"blockdft_gemm_tree": "blockdft_gemm", "blockdft_banddots": "blockdft_dots_db", "peaks_frames_lean": "peaks_frames"}
captures = []
for cfg, n_bins, fpl in ((1, 252, 65536), (2, 288, 131072)):
    rows = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/{tag}_pmc{cfg}_{c}/*/*counter_collection.csv")
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "pvq::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rows.setdefault(k, {})[c] = sum(v) / len(v)
    name = f"gpurun_out/{tag}_pmc_traffic{'' if cfg == 1 else '_config2'}.csv"
    with open(name, "w") as f:
        f.write("kernel,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_per_launch(2*FETCH+WRITE)*1024\n")
        for k, v in rows.items():
            fe, wr = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            f.write(f"\"{k}\",{fe:.1f},{wr:.1f},{(2*fe+wr)*1024:.0f}\n")
            for sub, slot in SLOT.items():
                if sub in k and "bf16x3" not in k:
                    captures.append({"kernel": slot, "kernel_name": k, "frames_per_launch": fpl, "n_bins": n_bins,
                                     "hbm_bytes_per_launch": round((2 * fe + wr) * 1024), "srchash": srchash,
                                     "file": f"profiles/{tag}_pmc_traffic{'' if cfg == 1 else '_config2'}.csv"})
    print(open(name).read())
# cycles, not microseconds: the dominant kernels' matrix-pipe busy fraction and active cycles per launch (clock-independent: rounds compare in these)
acc_sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pvq::" in r["Kernel_Name"] and "bf16x3" not in r["Kernel_Name"]:
            acc_sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for cap in captures:
    if cap["n_bins"] != 252:
        continue   # (the SQ passes run configs[1])
    d = acc_sq.get(cap["kernel_name"], {})
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("GRBM_GUI_ACTIVE"):
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
        cap["mfma_busy_cycles"] = round(busy)
        cap["gui_active_cycles"] = round(gui)            # summed over the 8 XCDs, as rocprofv3 reports it
        cap["mfma_busy_frac"] = round(busy / (1024.0 * gui / 8.0), 4)   # 1 024 SIMDs x the kernel's cycles
        if d.get("SQ_INSTS_MFMA"):
            cap["insts_mfma"] = round(sum(d["SQ_INSTS_MFMA"]) / len(d["SQ_INSTS_MFMA"]))
        cap["sq_file"] = f"profiles/{tag}_sq_counters.txt"
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB per launch, mean); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                   "on gfx950 FETCH_SIZE reports half the bytes of wide streaming reads (MI355X_MICROARCH.md, HBM); srchash = sha256 of the "
                   "kernel sources the measured library was built from (pitchvis_amd/lib/libpvq.so.srchash); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                   "gui_active_cycles = GRBM_GUI_ACTIVE per launch (sum over the 8 XCDs), both from separate --pmc passes of the same command",
           "captures": captures}, open("gpurun_out/traffic_latest.json", "w"), indent=1)
st = glob.glob(f"gpurun_out/{tag}_stats/*/*kernel_stats.csv")[0]
with open(f"gpurun_out/{tag}_kernel_stats.csv", "w") as f:
    for i, line in enumerate(open(st)):
        if i == 0 or "pvq::" in line:
            f.write(line)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pvq::" in r["Kernel_Name"] and "bf16x3" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/{tag}_sq_counters.txt", "w") as out:
    out.write(f"# rocprofv3 --pmc, mean per launch, bench.py configs[1]; kernel sources {srchash[:12]}\n")
    for k, d in acc.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open(f"gpurun_out/{tag}_kernel_stats.csv").read())
print(open(f"gpurun_out/{tag}_sq_counters.txt").read())
PY
# the bench lines LAST: they attach the traffic capture just taken on this build (profiles/traffic_latest.json on this box)
cp $OUT/traffic_latest.json $ROOT/profiles/traffic_latest.json
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 $ROOT/bench.py --steps 10 --warmup 3 --config 2 --no-cpu-baseline --no-extra-config > $OUT/${TAG}_bench_config2.json 2>> $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench.json
