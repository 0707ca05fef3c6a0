#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats, HBM PMC passes.
# usage: scripts/collect_profiles.sh <tag>     -> files under gpurun_out/<tag>_*
set -e
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
tag = "$TAG"
rows = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pvq::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows.setdefault(k, {})[c] = sum(v) / len(v)
with open(f"gpurun_out/{tag}_pmc_traffic.csv", "w") as f:
    f.write("kernel,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_per_launch(2*FETCH+WRITE)*1024\n")
    for k, v in rows.items():
        fe, wr = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        f.write(f"\"{k}\",{fe:.1f},{wr:.1f},{(2*fe+wr)*1024:.0f}\n")
st = glob.glob(f"gpurun_out/{tag}_stats/*/*kernel_stats.csv")[0]
with open(f"gpurun_out/{tag}_kernel_stats.csv", "w") as f:
    for i, line in enumerate(open(st)):
        if i == 0 or "pvq::" in line:
            f.write(line)
print(open(f"gpurun_out/{tag}_pmc_traffic.csv").read())
print(open(f"gpurun_out/{tag}_kernel_stats.csv").read())
print(open(f"gpurun_out/{tag}_bench.json").read())
PY
