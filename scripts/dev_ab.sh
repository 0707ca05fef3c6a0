#!/bin/bash
# Developer tool (GPU box): A/B of two prebuilt libraries on one box (boxes differ by 5-10 %).  Put them as
# pitchvis_amd/lib/ab/libpvq_head.so and libpvq_new.so (they travel with the snapshot), then: gpurun -- scripts/dev_ab.sh
for rep in 1 2; do
for v in head new; do
  cp pitchvis_amd/lib/ab/libpvq_$v.so pitchvis_amd/lib/libpvq.so
  echo "== $v"
  python scripts/dev_peaks.py bench_48k_252 65536 20 full 2>&1 | grep peaks
  python scripts/dev_peaks.py bench_48k_252 65536 20 mask 2>&1 | grep peaks
  python scripts/dev_peaks.py default_22k_588 32768 20 full 2>&1 | grep peaks
  python scripts/dev_peaks.py hires_96k_840 32768 20 full 2>&1 | grep peaks
done
done
cp pitchvis_amd/lib/ab/libpvq_new.so pitchvis_amd/lib/libpvq.so
