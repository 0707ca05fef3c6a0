#!/bin/bash
# Developer tool (GPU box): like dev_ab.sh, one pass, for commands that need environment knobs: scripts/dev_ab_env.sh "ENV=1 python x.py" ...
for v in head new; do
  cp pitchvis_amd/lib/ab/libpvq_$v.so pitchvis_amd/lib/libpvq.so
  echo "== $v"
  for c in "$@"; do bash -c "$c" 2>&1 | grep -v "amdgpu.ids\|^make\|hipcc\|mkdir"; done
done
cp pitchvis_amd/lib/ab/libpvq_new.so pitchvis_amd/lib/libpvq.so
