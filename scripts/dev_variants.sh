#!/bin/bash
# Developer tool (GPU box): run commands against prebuilt library variants pitchvis_amd/lib/ab/libpvq_<v>.so (they travel with the
# snapshot).  usage: scripts/dev_variants.sh "<v1> <v2> ..." "<command>" ["<command>" ...]     (commands run with PVQ_SKIP_BUILD=1)
cp pitchvis_amd/lib/libpvq.so /tmp/libpvq_keep.so
for v in $1; do
  cp pitchvis_amd/lib/ab/libpvq_$v.so pitchvis_amd/lib/libpvq.so
  echo "== $v"
  for c in "${@:2}"; do PVQ_SKIP_BUILD=1 bash -c "$c" 2>&1 | grep -v "amdgpu.ids\|^make\|hipcc\|mkdir"; done
done
cp /tmp/libpvq_keep.so pitchvis_amd/lib/libpvq.so
