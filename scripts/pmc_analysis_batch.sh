#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats + SQ counters of the AnalysisBatch kernels (ab_recurrence, ab_frames, ab_tuning and the raw-frame peaks pre-pass; one 4096-stream x 128-frame call per pass).
# usage: scripts/pmc_analysis_batch.sh <tag> [bpo]     (bpo 36: 252 bins, 84: 588 bins; the program goes directly after `--`)
TAG=${1:-ab}
BPO=${2:-36}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/scripts/dev_analysis_batch.py - once $BPO > $OUT/stats.log 2>&1 || echo "stats pass failed"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/dev_analysis_batch.py - once $BPO > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$TAG/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pvq::ab_" in r["Kernel_Name"] or "peaks_frames" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/$TAG/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write(k + "   (one call: 4096 streams x 128 frames x 7 x $BPO bins, every output)\n")
        for c, v in sorted(d.items()):
            out.write(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
    st = glob.glob("gpurun_out/$TAG/stats/*/*kernel_stats.csv")
    if st:
        for i, line in enumerate(open(st[0])):
            if i == 0 or "pvq::ab_" in line or "peaks_frames" in line:
                out.write(line)
print(open("gpurun_out/$TAG/summary.txt").read())
PY
