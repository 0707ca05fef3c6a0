#!/bin/bash
# usage: run2.sh tag  (env knobs inherited) — two bench ranks on one GPU over gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 16384 --backend gloo > gpurun_out/two_$1.txt 2>&1
rc=$?
echo "$1 rc=$rc $(grep -c 'Memory access fault' gpurun_out/two_$1.txt) faults"
exit $rc
