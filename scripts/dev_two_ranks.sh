#!/bin/bash
# Developer tool (GPU box): bench.py's N = 2 path as two ranks on ONE GPU over gloo (what tests/test_configs_gpu.py::test_bench_two_ranks_gloo_on_one_gpu runs),
# environment knobs inherited; output to gpurun_out/two_<tag>.txt.  usage: scripts/dev_two_ranks.sh <tag>
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 16384 --backend gloo > gpurun_out/two_$1.txt 2>&1
rc=$?
echo "$1 rc=$rc $(grep -c 'Memory access fault' gpurun_out/two_$1.txt) faults"
exit $rc
