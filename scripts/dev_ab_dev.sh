#!/bin/bash
# Developer tool (GPU box): commands against developer-library variants pitchvis_amd/lib/ab/libpvq_<v>.so (copied over libpvq_dev.so in turn;
# PVQ_DEV_LIB=1 PVQ_SKIP_BUILD=1 are set).  usage: scripts/dev_ab_dev.sh "<v1> <v2> ..." "<command>" ...
export PVQ_DEV_LIB=1 PVQ_SKIP_BUILD=1
cp pitchvis_amd/lib/libpvq_dev.so /tmp/keepdev.so
for v in $1; do
  cp pitchvis_amd/lib/ab/libpvq_$v.so pitchvis_amd/lib/libpvq_dev.so
  echo "== $v"
  for c in "${@:2}"; do bash -c "$c" 2>&1 | grep -v "amdgpu.ids\|^make\|hipcc\|mkdir"; done
done
cp /tmp/keepdev.so pitchvis_amd/lib/libpvq_dev.so
