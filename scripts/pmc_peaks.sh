#!/bin/bash
# Developer tool (GPU box): SQ counters of the peak kernels.  usage: scripts/pmc_peaks.sh <tag> [geom] [mask|full]
TAG=${1:-pmcp}
GEOM=${2:-bench_48k_252}
MODE=${3:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do   # (TA_* / TCP_* sets do not run on this pool)
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/dev_peaks.py $GEOM 65536 2 $MODE > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$TAG/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pvq::peaks" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/$TAG/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
print(open("gpurun_out/$TAG/summary.txt").read())
PY
