#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats + vector-instruction counts of the FFT path's batch form, per-window kernels (PVQ_FFT_CT=1, the product's
# choice) against the walk (PVQ_FFT_CT=0), developer library.  usage: scripts/pmc_fft_path.sh <tag> [geometry index of scripts/dev_fft_path.py]
TAG=${1:-fftp}
GI=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PVQ_DEV_LIB=1
for ct in 0 1; do
  export PVQ_FFT_CT=$ct
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$ct -- python3 $ROOT/scripts/dev_fft_path.py - once $GI > $OUT/stats$ct.log 2>&1 || echo "stats pass $ct failed"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc$ct -- python3 $ROOT/scripts/dev_fft_path.py - once $GI > $OUT/pmc$ct.log 2>&1 || echo "pmc pass $ct failed"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmcb$ct -- python3 $ROOT/scripts/dev_fft_path.py - once $GI > $OUT/pmcb$ct.log 2>&1 || echo "pmc pass b $ct failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
with open("gpurun_out/$TAG/summary.txt", "w") as out:
    for ct in (0, 1):
        out.write(f"== PVQ_FFT_CT={ct} ({'per-window kernels vqt_fft_group' if ct else 'the walk vqt_fft_frames'}), geometry $GI of scripts/dev_fft_path.py, one batch call per launch, mean per launch\n")
        st = glob.glob(f"gpurun_out/$TAG/stats{ct}/*/*kernel_stats.csv")
        if st:
            for i, line in enumerate(open(st[0])):
                if i == 0 or "vqt_fft" in line or "db_rows" in line:
                    out.write("   " + line)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/$TAG/pmc{ct}/*/*counter_collection.csv") + glob.glob(f"gpurun_out/$TAG/pmcb{ct}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "vqt_fft" in r["Kernel_Name"] or "db_rows" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            out.write("   " + k + "\n")
            for c, v in sorted(d.items()):
                out.write(f"       {c:24s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open("gpurun_out/$TAG/summary.txt").read())
PY
