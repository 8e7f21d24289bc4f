#!/bin/bash
# Run a command on the MI355X box, retrying while no slot is free (gpurun exit code 3).  usage: tools/gpu_retry.sh <timeout-s> '<command>'
T="$1"; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
