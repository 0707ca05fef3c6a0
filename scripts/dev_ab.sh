#!/bin/bash
# Developer tool (GPU box): A/B of two prebuilt libraries on one box (boxes differ by 5-10 %).  Put them as
# pitchvis_amd/lib/ab/libpvq_head.so and libpvq_new.so (they travel with the snapshot), then:
#   gpurun -- scripts/dev_ab.sh "python scripts/dev_time.py 2 0 once" ["second command" ...]
for rep in 1 2 3; do
for v in head new; do
  cp pitchvis_amd/lib/ab/libpvq_$v.so pitchvis_amd/lib/libpvq.so
  echo "== $v"
  for c in "$@"; do $c 2>&1 | grep -v "amdgpu.ids\|^make\|hipcc\|mkdir"; done
done
done
cp pitchvis_amd/lib/ab/libpvq_new.so pitchvis_amd/lib/libpvq.so
