#!/bin/bash
# Developer tool (GPU box): memory-side counters of the block-DFT kernels.  usage: scripts/pmc_mem.sh <tag> <gemm_precision>
TAG=${1:-pmcm}
PREC=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# TCC_* (L2) sets only: TA_* / TCP_* counter sets do not run on this pool (configuration error or a hung GPU) and are NOT listed
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/dev_time.py 2 $PREC once > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$TAG/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pvq::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
