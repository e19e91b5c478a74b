#!/bin/bash
# Dev helper (build container): gpurun with a retry while no slot / box is free (exit code 3 = nothing charged).
# usage: bash tools/gpurun_retry.sh <timeout-seconds> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
