#!/bin/bash
# Run GPU steps one after another on the GPU box; a step that times out (or is killed) ends the call: no further GPU step
# is started behind a hung one.  usage: scripts/gpu_step.sh "<timeout s>|<log name>|<command>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for spec in "$@"; do
  IFS='|' read -r T LOG CMD <<< "$spec"
  echo "== step: $CMD (timeout $T s) -> gpurun_out/$LOG"
  timeout -k 10 $T bash -c "$CMD" > $ROOT/gpurun_out/$LOG 2>&1
  rc=$?
  echo "   exit $rc"
  tail -n 5 $ROOT/gpurun_out/$LOG
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   timed out: stopping here"; exit $rc; fi
done
exit 0
